// Plumbing only: stable radix sort of (cell key, slot) pairs and exclusive scans
// from rocPRIM.  Everything domain-specific is hand-written in the other files.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "gnx_internal.h"
#include "gnx_compact.h"

// rocPRIM's radix_sort switches to a merge sort below 2^20 items: ~22 launches of a few
// microseconds for the 2 x 10^5 pairs of a step, where Onesweep over the ~24 significant
// id bits takes 5.  Merge sort only for inputs that fit a few blocks.
using gnx_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                   rocprim::default_config, 8192>;

// wider digits, fewer passes: 10 bits per Onesweep pass (the rank table must fit a block of
// 1024 threads) sort the 40..50-bit (cell, id) keys in 4..5 passes instead of 5..7
template <int IPT>
using gnx_onesweep10_t = rocprim::radix_sort_onesweep_config<
    rocprim::kernel_config<1024, IPT>, rocprim::kernel_config<1024, IPT>, 10,
    rocprim::block_radix_rank_algorithm::match>;
template <int IPT>
using gnx_sort_config10_t = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                       gnx_onesweep10_t<IPT>, 8192>;
using gnx_sort_config10 = gnx_sort_config10_t<6>;

int gnx_prim_sort_bytes(size_t n, int bits, size_t* bytes) {
  *bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                   (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, bits,
                                   (hipStream_t)0));
  return 0;
}

int gnx_prim_sort(void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout, const int32_t* vin,
                  int32_t* vout, size_t n, int bits, hipStream_t s) {
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0, bits, s));
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
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                   (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, 64,
                                   (hipStream_t)0));
  size_t b10 = 0;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(nullptr, b10, (const uint64_t*)nullptr,
                                                           (uint64_t*)nullptr, (const int32_t*)nullptr,
                                                           (int32_t*)nullptr, n, 0, 64, (hipStream_t)0));
  if (b10 > *bytes) *bytes = b10;
  size_t b32 = 0;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(nullptr, b32, (const uint32_t*)nullptr,
                                                      (uint32_t*)nullptr, (const int32_t*)nullptr,
                                                      (int32_t*)nullptr, n, 0, 32, (hipStream_t)0));
  if (b32 > *bytes) *bytes = b32;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(nullptr, b32, (const uint32_t*)nullptr,
                                                    (uint32_t*)nullptr, (const int32_t*)nullptr,
                                                    (int32_t*)nullptr, n, 0, 32, (hipStream_t)0));
  if (b32 > *bytes) *bytes = b32;
  return 0;
}

int gnx_prim_sort64(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                    const int32_t* vin, int32_t* vout, size_t n, hipStream_t s) {
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0, 64, s));
  return 0;
}

// only the low `end_bit` bits of the keys are significant (fewer radix passes)
int gnx_prim_sort64_bits(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                         const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, bool alone) {
  // alone on the chip: 10-bit digits, 6 keys per thread: 0.138 ms for the 1.25 x 10^6 40-bit keys of the metric
  // workload (8-bit default config 0.187; 2 / 3 / 4 / 8 / 12 keys per thread 0.191 / 0.166 /
  // 0.154 / 0.151 / 0.167).  Beside a crossover (whole-step overlap) its 1024-thread blocks
  // wait longer for a CU than the default's and the step loses 5-9 %: default digits there.
  // GNX_SORT_BITS=8 / 10 forces one or the other.
  static const int forced = getenv("GNX_SORT_BITS") ? atoi(getenv("GNX_SORT_BITS")) : 0;
  const int digit = forced ? forced : (alone ? 10 : 8);
  if (digit == 10)
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                        end_bit, s));
  else
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                      end_bit, s));
  return 0;
}

// 32-bit keys (hash cells of the id-ordered index), same digit choice
int gnx_prim_sort32_bits(void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout,
                         const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, bool alone) {
  static const int forced = getenv("GNX_SORT_BITS") ? atoi(getenv("GNX_SORT_BITS")) : 0;
  const int digit = forced ? forced : (alone ? 10 : 8);
  if (digit == 10)
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                        end_bit, s));
  else
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                      end_bit, s));
  return 0;
}

// ---------------------------------------------------------------- the step's cell sort
// Stable LSD radix sort of (32-bit key, 32-bit value) pairs by the low `end_bit` key bits,
// built from rocPRIM's Onesweep device functions (histogram of every digit place in one pass
// over the keys, one decoupled-look-back pass per place) but launched here: rocPRIM's own
// driver clears its histogram, its look-back states and its block counter with a fill
// kernel each per place - six fills and four kernels for the two places of the step's
// 16-bit cell keys; here the scratch of all places is one contiguous region cleared by ONE
// fill (5 launches instead of 10; at 10^5 individuals the launches are the sort).
// (built on rocprim::detail: the Onesweep device functions and their scratch layout are not a
// public interface - checked against the rocPRIM of ROCm 7.x; another major version has to be
// looked at before this compiles)
#include <rocprim/rocprim_version.hpp>
#define GNX_ROCPRIM_MAJOR 4
static_assert(ROCPRIM_VERSION_MAJOR == GNX_ROCPRIM_MAJOR,
              "gnx_prim.hip: rocPRIM major version changed - re-check rocprim::detail::onesweep_* "
              "(signatures, look-back state, block-id usage) against gnx_os, then bump GNX_ROCPRIM_MAJOR");
namespace gnx_os {
using lookback_t = rocprim::detail::onesweep_lookback_state;
using obid_t = rocprim::detail::ordered_block_id<unsigned int>;

template <unsigned BS, unsigned IPT, unsigned RB>
__global__ void __launch_bounds__(BS)
k_hist(const uint32_t* keys, unsigned int* offs, unsigned int size, unsigned int full_blocks,
       unsigned int begin_bit, unsigned int end_bit) {
  rocprim::detail::onesweep_histograms<BS, IPT, RB, false>(keys, offs, size, full_blocks,
                                                          rocprim::identity_decomposer{}, begin_bit,
                                                          end_bit);
}

template <unsigned BS, unsigned RB>
__global__ void __launch_bounds__(BS) k_scan(unsigned int* offs) {
  rocprim::detail::onesweep_scan_histograms<BS, RB>(offs);
}

template <unsigned BS, unsigned IPT, unsigned RB>
__global__ void __launch_bounds__(BS)
k_iter(const uint32_t* kin, uint32_t* kout, const int32_t* vin, int32_t* vout, unsigned int size,
       unsigned int* offs_in, unsigned int* offs_out, lookback_t* lb, unsigned int bit,
       unsigned int cur_bits, unsigned int full_blocks, obid_t ob) {
  rocprim::detail::onesweep_iteration<BS, IPT, RB, false, rocprim::block_radix_rank_algorithm::match>(
      kin, kout, vin, vout, size, offs_in, offs_out, lb, rocprim::identity_decomposer{}, bit, cur_bits,
      full_blocks, ob);
}

template <unsigned BS, unsigned IPT, unsigned RB>
static size_t scratch_words(size_t n, int end_bit) {
  const size_t places = (end_bit + RB - 1) / RB, radix = 1u << RB;
  const size_t blocks = (n + BS * IPT - 1) / (BS * IPT);
  return places * radix + radix + places * blocks * radix + places + 16;
}

template <unsigned BS, unsigned IPT, unsigned RB>
static int sort(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin, uint32_t* kout,
                const int32_t* vin, int32_t* vout, size_t n, int end_bit, hipStream_t s) {
  const unsigned int places = (end_bit + RB - 1) / RB, radix = 1u << RB;
  const unsigned int items = BS * IPT;
  const unsigned int blocks = (unsigned int)((n + items - 1) / items);
  const unsigned int full_blocks = (unsigned int)(n / items);
  unsigned int* hist = (unsigned int*)scratch;                 // [places][radix]
  unsigned int* offs_tmp = hist + (size_t)places * radix;      // [radix]
  lookback_t* lb = (lookback_t*)(offs_tmp + radix);            // [places][blocks * radix]
  unsigned int* bid = (unsigned int*)(lb + (size_t)places * blocks * radix);   // [places]
  static_assert(sizeof(lookback_t) == sizeof(unsigned int), "look-back state is one word");
  HIPCHK(hipMemsetAsync(scratch, 0, scratch_words<BS, IPT, RB>(n, end_bit) * sizeof(unsigned int), s));
  hipLaunchKernelGGL((k_hist<BS, IPT, RB>), dim3(blocks), dim3(BS), 0, s, kin, hist, (unsigned int)n,
                     full_blocks, 0u, (unsigned int)end_bit);
  hipLaunchKernelGGL((k_scan<BS, RB>), dim3(places), dim3(BS), 0, s, hist);
  bool to_output = (places - 1) % 2 == 0;
  const uint32_t* ki = kin;
  const int32_t* vi = vin;
  for (unsigned int place = 0, bit = 0; place < places; ++place, bit += RB) {
    uint32_t* ko = to_output ? kout : ktmp;
    int32_t* vo = to_output ? vout : vtmp;
    const unsigned int cur = std::min<unsigned int>(RB, (unsigned int)end_bit - bit);
    hipLaunchKernelGGL((k_iter<BS, IPT, RB>), dim3(blocks), dim3(BS), 0, s, ki, ko, vi, vo,
                       (unsigned int)n, hist + (size_t)place * radix, offs_tmp,
                       lb + (size_t)place * blocks * radix, bit, cur, full_blocks,
                       obid_t::create(bid + place));
    ki = ko;
    vi = vo;
    to_output = !to_output;
  }
  HIPCHK(hipGetLastError());
  return 0;
}
// The same passes over 64-bit (cell << idbits | id) keys - what a tile with imports sorts, its
// id-ordered index gone (csrc/gnx_tile.hip).  `scratch` is zero on entry and the caller wipes
// what the sort dirtied (k_permute): no fills.  rocPRIM's own driver (radix_sort_pairs) puts a
// fill in front of the histograms and two in front of every pass - eleven of ~5 us each on the
// step's chain for 41-bit keys (profiles/r05_ab_runs.txt).
template <unsigned BS, unsigned IPT, unsigned RB>
__global__ void __launch_bounds__(BS)
k_hist64(const uint64_t* keys, unsigned int* offs, unsigned int size, unsigned int full_blocks,
         unsigned int begin_bit, unsigned int end_bit) {
  rocprim::detail::onesweep_histograms<BS, IPT, RB, false>(keys, offs, size, full_blocks,
                                                          rocprim::identity_decomposer{}, begin_bit,
                                                          end_bit);
}

template <unsigned BS, unsigned IPT, unsigned RB>
__global__ void __launch_bounds__(BS)
k_iter64(const uint64_t* kin, uint64_t* kout, const int32_t* vin, int32_t* vout, unsigned int size,
         unsigned int* offs_in, unsigned int* offs_out, lookback_t* lb, unsigned int bit,
         unsigned int cur_bits, unsigned int full_blocks, obid_t ob) {
  rocprim::detail::onesweep_iteration<BS, IPT, RB, false, rocprim::block_radix_rank_algorithm::match>(
      kin, kout, vin, vout, size, offs_in, offs_out, lb, rocprim::identity_decomposer{}, bit, cur_bits,
      full_blocks, ob);
}

template <unsigned BS, unsigned IPT, unsigned RB>
static int sort64_clean(void* scratch, uint64_t* ktmp, int32_t* vtmp, const uint64_t* kin,
                        uint64_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                        hipStream_t s) {
  const unsigned int places = (end_bit + RB - 1) / RB, radix = 1u << RB;
  const unsigned int items = BS * IPT;
  const unsigned int blocks = (unsigned int)((n + items - 1) / items);
  const unsigned int full_blocks = (unsigned int)(n / items);
  unsigned int* hist = (unsigned int*)scratch;                 // [places][radix]
  unsigned int* offs_tmp = hist + (size_t)places * radix;      // [radix]
  lookback_t* lb = (lookback_t*)(offs_tmp + radix);            // [places][blocks * radix]
  unsigned int* bid = (unsigned int*)(lb + (size_t)places * blocks * radix);   // [places]
  hipLaunchKernelGGL((k_hist64<BS, IPT, RB>), dim3(blocks), dim3(BS), 0, s, kin, hist,
                     (unsigned int)n, full_blocks, 0u, (unsigned int)end_bit);
  hipLaunchKernelGGL((k_scan<BS, RB>), dim3(places), dim3(BS), 0, s, hist);
  bool to_output = (places - 1) % 2 == 0;
  const uint64_t* ki = kin;
  const int32_t* vi = vin;
  for (unsigned int place = 0, bit = 0; place < places; ++place, bit += RB) {
    uint64_t* ko = to_output ? kout : ktmp;
    int32_t* vo = to_output ? vout : vtmp;
    const unsigned int cur = std::min<unsigned int>(RB, (unsigned int)end_bit - bit);
    hipLaunchKernelGGL((k_iter64<BS, IPT, RB>), dim3(blocks), dim3(BS), 0, s, ki, ko, vi, vo,
                       (unsigned int)n, hist + (size_t)place * radix, offs_tmp,
                       lb + (size_t)place * blocks * radix, bit, cur, full_blocks,
                       obid_t::create(bid + place));
    ki = ko;
    vi = vo;
    to_output = !to_output;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// Keys of the id-ordered sequence (entry k is slot ord[k] for k < ord_n, else slot k itself:
// offspring appended since the index was last compacted), their digit histograms and - in the
// workgroup that finishes last (ticket) - the exclusive scans of the histograms, i.e. what
// k_keys_ord + k_hist + k_scan left behind, in one launch.  hist: [places][2^RB], zero on
// entry (the caller keeps the scratch clean: k_permute wipes it after every sort).
// The counts travel as agent-scope atomics that have completed (s_waitcnt) before the ticket
// is taken; no release fence (gnx_compact.h: that is an L2 write-back).
template <unsigned RB, unsigned IPT>
__global__ void __launch_bounds__(1024)
k_keys_hist(long long N, long long ord_n, const int32_t* __restrict__ ord,
            const uint32_t* __restrict__ cell32, uint32_t* __restrict__ key,
            int32_t* __restrict__ val, unsigned int* __restrict__ hist, int places,
            unsigned int* __restrict__ ticket, long long n_fixed, unsigned int sentinel,
            const GnxDD* __restrict__ dd) {
  constexpr unsigned R = 1u << RB;
  // device-driven step: the population's size lives on the device and the sort runs over a
  // fixed n_fixed >= N entries; those behind the population carry the largest key, so the
  // stable sort leaves them behind everybody
  if (dd) {
    N = dd->N;
    ord_n = dd->ord_n;
  }
  static_assert(R == 1024, "one digit per thread");
  __shared__ unsigned int lh[3][R];
  __shared__ unsigned int wsum[16];
  __shared__ int last;
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int pl = 0; pl < places; ++pl) lh[pl][tid] = 0u;
  __syncthreads();
  const long long base = (long long)blockIdx.x * (1024 * IPT);
  int32_t slot[IPT];
  uint32_t c[IPT];
#pragma unroll
  for (unsigned r = 0; r < IPT; ++r) {
    const long long k = base + r * 1024 + tid;
    slot[r] = k < N ? (k < ord_n ? ord[k] : (int32_t)k) : -1;
  }
#pragma unroll
  for (unsigned r = 0; r < IPT; ++r) c[r] = slot[r] >= 0 ? cell32[slot[r]] : sentinel;
#pragma unroll
  for (unsigned r = 0; r < IPT; ++r) {
    const long long k = base + r * 1024 + tid;
    if (slot[r] < 0 && k >= n_fixed) continue;
    key[k] = c[r];
    val[k] = (int32_t)k;
    for (int pl = 0; pl < places; ++pl) atomicAdd(&lh[pl][(c[r] >> (pl * RB)) & (R - 1u)], 1u);
  }
  __syncthreads();
  for (int pl = 0; pl < places; ++pl) {
    const unsigned int v = lh[pl][tid];
    if (v) __hip_atomic_fetch_add(&hist[pl * R + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = t == gridDim.x - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last) return;
  for (int pl = 0; pl < places; ++pl) {
    const unsigned int v = __hip_atomic_load(&hist[pl * R + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned int y = __shfl_up(x, d);
      if (lane >= (unsigned)d) x += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned int woff = 0;
    for (unsigned w = 0; w < 16; ++w) woff += w < wave ? wsum[w] : 0u;
    hist[pl * R + tid] = woff + x - v;
  }
}

// the passes alone: histograms already counted and scanned in `scratch` (k_keys_hist), the
// rest of it zero
template <unsigned BS, unsigned IPT, unsigned RB>
static int sort_ranked(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin,
                       uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                       hipStream_t s) {
  const unsigned int places = (end_bit + RB - 1) / RB, radix = 1u << RB;
  const unsigned int items = BS * IPT;
  const unsigned int blocks = (unsigned int)((n + items - 1) / items);
  const unsigned int full_blocks = (unsigned int)(n / items);
  unsigned int* hist = (unsigned int*)scratch;
  unsigned int* offs_tmp = hist + (size_t)places * radix;
  lookback_t* lb = (lookback_t*)(offs_tmp + radix);
  unsigned int* bid = (unsigned int*)(lb + (size_t)places * blocks * radix);
  bool to_output = (places - 1) % 2 == 0;
  const uint32_t* ki = kin;
  const int32_t* vi = vin;
  for (unsigned int place = 0, bit = 0; place < places; ++place, bit += RB) {
    uint32_t* ko = to_output ? kout : ktmp;
    int32_t* vo = to_output ? vout : vtmp;
    const unsigned int cur = std::min<unsigned int>(RB, (unsigned int)end_bit - bit);
    hipLaunchKernelGGL((k_iter<BS, IPT, RB>), dim3(blocks), dim3(BS), 0, s, ki, ko, vi, vo,
                       (unsigned int)n, hist + (size_t)place * radix, offs_tmp,
                       lb + (size_t)place * blocks * radix, bit, cur, full_blocks,
                       obid_t::create(bid + place));
    ki = ko;
    vi = vo;
    to_output = !to_output;
  }
  HIPCHK(hipGetLastError());
  return 0;
}
}  // namespace gnx_os

// the fused front of the cell sort (variant 2's geometry: 1024 x 6 keys, 10-bit digits)
size_t gnx_os_words_used(size_t n, int end_bit, int geometry) {
  return geometry == 1 ? gnx_os::scratch_words<512, 4, 10>(n, end_bit)
                       : gnx_os::scratch_words<1024, 6, 10>(n, end_bit);
}
int gnx_os_keys_hist(void* scratch, unsigned int* ticket, int64_t N, int64_t ord_n,
                     const int32_t* ord, const uint32_t* cell32, uint32_t* key, int32_t* val,
                     int end_bit, hipStream_t s, const GnxDD* dd, int geometry) {
  const int places = (end_bit + 9) / 10;
  if (places > 3) return 1;
  if (geometry == 1) {
    // small populations: 2 048 keys per workgroup instead of 6 144 - three times the workgroups
    // for a sort that is all latency at 10^5 keys (43 workgroups on 256 CUs otherwise)
    const unsigned int blocks = (unsigned int)((N + 2047) / 2048);
    hipLaunchKernelGGL((gnx_os::k_keys_hist<10, 2>), dim3(blocks), dim3(1024), 0, s, (long long)N,
                       (long long)ord_n, ord, cell32, key, val, (unsigned int*)scratch, places, ticket,
                       (long long)N, (1u << end_bit) - 1u, dd);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // (2, 3, 4, 8 keys per thread instead of 6: the same step time, profiles/r03_ab_runs.txt)
  // dd: N is the fixed number of entries the sort runs over, the population's own size is read
  // on the device
  const unsigned int blocks = (unsigned int)((N + 6143) / 6144);
  hipLaunchKernelGGL((gnx_os::k_keys_hist<10, 6>), dim3(blocks), dim3(1024), 0, s, (long long)N,
                     (long long)ord_n, ord, cell32, key, val, (unsigned int*)scratch, places, ticket,
                     (long long)N, (1u << end_bit) - 1u, dd);
  HIPCHK(hipGetLastError());
  return 0;
}
int gnx_os_sort32_ranked(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin,
                         uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, int geometry) {
  if (geometry == 1)
    return gnx_os::sort_ranked<512, 4, 10>(scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s);
  return gnx_os::sort_ranked<1024, 6, 10>(scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s);
}

// 64-bit keys, 10-bit digits, 1024 x 6 keys; scratch zero on entry, *words = what the caller has
// to wipe afterwards; tmp (>= 12 n bytes) holds the keys and values between the passes
size_t gnx_os_words_used64(size_t n, int end_bit) {
  return gnx_os::scratch_words<1024, 6, 10>(n, end_bit);
}
int gnx_os_sort64_clean(void* scratch, void* tmp, const uint64_t* kin, uint64_t* kout,
                        const int32_t* vin, int32_t* vout, size_t n, int end_bit, hipStream_t s) {
  uint64_t* ktmp = (uint64_t*)tmp;
  int32_t* vtmp = (int32_t*)(ktmp + n);
  return gnx_os::sort64_clean<1024, 6, 10>(scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s);
}

// variant 0: 256 threads x 12 keys, 8-bit digits; 1: 512 x 8, 8 bits; 2: 1024 x 6, 10 bits
size_t gnx_os_scratch_bytes(size_t n, int end_bit) {
  size_t w = std::max({gnx_os::scratch_words<256, 12, 8>(n, end_bit),
                       gnx_os::scratch_words<512, 8, 8>(n, end_bit),
                       gnx_os::scratch_words<1024, 6, 10>(n, end_bit),
                       // (the small geometry: only ever used below GNX_DD_MAX_CAP slots)
                       gnx_os::scratch_words<512, 4, 10>(std::min<size_t>(n, 1u << 21), end_bit)});
  return w * sizeof(unsigned int);
}

int gnx_os_sort32(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin,
                  uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                  hipStream_t s, int variant) {
  switch (variant) {
    case 1: return gnx_os::sort<512, 8, 8>(scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s);
    case 2: return gnx_os::sort<1024, 6, 10>(scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s);
    default: return gnx_os::sort<256, 12, 8>(scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s);
  }
}

// ---------------------------------------------------------------- block-count scan
// Exclusive scan of K (<= 3) count arrays (cnt[k * stride + b], b < nb) into
// off[k * stride + b], off[k * stride + nb] = total.  One workgroup of 1024 threads.
// out[k] = total of array k; `host` (pinned, device-visible) receives the same numbers.
__global__ void __launch_bounds__(1024)
k_block_scan(int K, int nb, int stride, const int32_t* __restrict__ cnt, int32_t* __restrict__ off,
             int32_t* __restrict__ out, int64_t* __restrict__ host, long long seq,
             const int32_t* __restrict__ extra) {
  // all K (<= 3) arrays in one sweep: two barriers per 1024 blocks instead of three per array
  __shared__ int wsum[3][16];
  __shared__ int carry_s[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 3) carry_s[tid] = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + tid;
    int v[3], x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] = (k < K && b < nb) ? cnt[k * stride + b] : 0;
      x[k] = v[k];
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int y = __shfl_up(x[k], d);
        if (lane >= d) x[k] += y;
      }
    }
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 3; ++k) wsum[k][wave] = x[k];
    }
    __syncthreads();
    int tot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int woff = 0, t = 0;
      for (int w = 0; w < 16; ++w) {
        const int s = wsum[k][w];
        woff += w < wave ? s : 0;
        t += s;
      }
      tot[k] = t;
      if (k < K && b < nb) off[k * stride + b] = carry_s[k] + woff + x[k] - v[k];
    }
    __syncthreads();
    if (tid < 3) carry_s[tid] += tot[tid];
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = 0; k < 3; ++k) {
      const int total = k < K ? carry_s[k] : 0;
      if (k < K) off[k * stride + nb] = total;
      if (out) out[k] = total;
      if (host) host[k] = total;
    }
    if (host && extra) host[12] = (int64_t)*extra;    // one more word for the host
    // a host that polls host[3] for `seq` (gnx_wait_published) sees the totals first
    if (host && seq)
      __hip_atomic_store(&host[3], (int64_t)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int gnx_block_scan(gnx_state* h, int K, int64_t n_items, const int32_t* cnt, int32_t* off,
                   int32_t* out, int64_t* host, int64_t seq, hipStream_t st, const int32_t* extra) {
  const int nb = (int)((n_items + GNX_CB - 1) / GNX_CB);
  hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, st ? st : h->stream, K, nb, h->blk_stride, cnt, off,
                     out, host, (long long)seq, extra);
  HIPCHK(hipGetLastError());
  return 0;
}
