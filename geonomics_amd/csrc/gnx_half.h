// Physical half-rows of the genome table and who refers to them.
//
// An individual owns a LOGICAL genome row (GnxSoA.grow; handed out and taken back by
// rank from the free-row stack, counted exactly on the host as before).  Where the two
// homologues of a logical row actually live is a second table: hmap[2 * row + h] is the
// PHYSICAL half-row (W64 words at G + phys * W64) of homologue h.  Half-rows are
// immutable once written and reference-counted, so a gamete that carries no switch
// point - the child's homologue IS one of the parent's, bit for bit
// (ops/mating.py:165-167 with an all-0 or all-1 subsetter) - is not copied: the child's
// entry points at the parent's half-row and its count goes up.  With one expected
// crossover per gamete (r = 1/L) that is e^-1 = 37 % of all gametes, 12.5 KB each.
// A half-row returns to the free stack when its last referrer dies.
//
// Every referrer holds exactly one count, so the live half-rows never outnumber
// 2 x (logical rows in use): the stack cannot run dry while logical rows are left, and
// the host needs no count of it.  Pops happen only in the kernels that hand out rows,
// pushes only in the mortality compaction: never in the same kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GnxHalves {
  int32_t* hmap;    // [2 * row span]  logical half -> physical half
  int32_t* rc;      // [2 * row span]  references to a physical half
  int32_t* stack;   // free physical halves
  int32_t* top;     // how many
};

// wave-aggregated pop: every lane with want == true gets a free physical half-row
__device__ __forceinline__ int32_t gnx_half_pop(const GnxHalves& H, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return -1;
  const int lane = threadIdx.x & 63;
  const int n = __popcll(m);
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicSub(H.top, n);
  base = __shfl(base, leader);
  const int rank = __popcll(m & ((1ull << lane) - 1ull));
  return want ? H.stack[base - 1 - rank] : -1;
}

// wave-aggregated append to a list counter: returns this lane's index (or -1)
__device__ __forceinline__ int32_t gnx_wave_append(int32_t* counter, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return -1;
  const int lane = threadIdx.x & 63;
  const int n = __popcll(m);
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, n);
  base = __shfl(base, leader);
  return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// one reference less; the last one pushes the half-row on the free stack
__device__ __forceinline__ void gnx_half_release(const GnxHalves& H, int32_t phys) {
  if (phys < 0) return;
  if (atomicSub(&H.rc[phys], 1) == 1) H.stack[atomicAdd(H.top, 1)] = phys;
}

// a fresh physical half-row for logical half `lh`
__device__ __forceinline__ int32_t gnx_half_new(const GnxHalves& H, int64_t lh, bool want) {
  const int32_t p = gnx_half_pop(H, want);
  if (want) {
    H.rc[p] = 1;
    H.hmap[lh] = p;
  }
  return p;
}
