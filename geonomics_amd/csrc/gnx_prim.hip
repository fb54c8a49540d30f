// Plumbing only: stable radix sort of (cell key, slot) pairs and exclusive scans
// from rocPRIM.  Everything domain-specific is hand-written in the other files.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "gnx_internal.h"

int gnx_prim_sort_bytes(size_t n, int bits, size_t* bytes) {
  *bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                   (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, bits,
                                   (hipStream_t)0));
  return 0;
}

int gnx_prim_sort(void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout, const int32_t* vin,
                  int32_t* vout, size_t n, int bits, hipStream_t s) {
  HIPCHK(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, bits, s));
  return 0;
}

int gnx_prim_scan_bytes(size_t n, size_t* bytes) {
  *bytes = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, *bytes, (const int32_t*)nullptr, (int32_t*)nullptr, 0, n,
                                 rocprim::plus<int32_t>(), (hipStream_t)0));
  return 0;
}

int gnx_prim_scan(void* tmp, size_t bytes, const int32_t* in, int32_t* out, size_t n,
                  hipStream_t s) {
  HIPCHK(rocprim::exclusive_scan(tmp, bytes, in, out, 0, n, rocprim::plus<int32_t>(), s));
  return 0;
}

int gnx_prim_sort64_bytes(size_t n, size_t* bytes) {
  *bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                   (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, 64,
                                   (hipStream_t)0));
  return 0;
}

int gnx_prim_sort64(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                    const int32_t* vin, int32_t* vout, size_t n, hipStream_t s) {
  HIPCHK(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, 64, s));
  return 0;
}

// only the low `end_bit` bits of the keys are significant (fewer radix passes)
int gnx_prim_sort64_bits(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                         const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s) {
  HIPCHK(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, end_bit, s));
  return 0;
}
