// Order-preserving stream compaction without look-back scans.
//
// A selection over N items is compacted in three launches that never wait on one
// another's workgroups (rocPRIM's decoupled look-back scan does, and crawls when its
// workgroups share the chip with a bandwidth-bound kernel):
//   1. the kernel that decides the flags also counts them per block of GNX_CB items
//      (gnx_block_ranks -> total) and stores the count in cnt[block];
//   2. k_block_scan (ONE workgroup) turns the counts into block offsets and totals;
//   3. the consumer recomputes the in-block ranks from the stored flags with the same
//      blocking and adds the block offset.
// Blocks are 256 threads x 4 rounds: item = block * GNX_CB + round * 256 + tid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The cross-workgroup hand-overs below (gnx_count_and_scan, and k_keys_hist / k_pair_compact
// which follow the same recipe) order their stores with `s_waitcnt vmcnt(0)` instead of a
// release fence: on gfx9 (CDNA) vmcnt counts stores too and agent / system-scope atomic stores
// write through, which is NOT what the HIP memory model promises on other targets.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "gnx_compact.h: the s_waitcnt-ordered hand-overs are written for gfx942 / gfx950 (CDNA3 / CDNA4) only"
#endif

#define GNX_CB 1024

// exclusive ranks of the flagged items of this block, in item order; lds: int[16]
__device__ __forceinline__ void gnx_block_ranks(const bool f[4], int rank[4], int& total,
                                                int* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long bal[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bal[r] = __ballot(f[r]);
    if (lane == 0) lds[r * 4 + wave] = __popcll(bal[r]);
  }
  __syncthreads();
  int run = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int off = run;
    for (int w = 0; w < wave; ++w) off += lds[r * 4 + w];
    rank[r] = off + __popcll(bal[r] & ((1ull << lane) - 1ull));
    run += lds[r * 4] + lds[r * 4 + 1] + lds[r * 4 + 2] + lds[r * 4 + 3];
  }
  total = run;
  __syncthreads();
}


// ---------------------------------------------------------------------------------------
// Step 2 without a launch of its own: the workgroup of the counting kernel that finishes
// LAST scans the block counts (a ticket counter tells it so).  Saves the k_block_scan
// launch - a one-workgroup kernel of 6-11 us on the step's critical path, 177 us when it
// crawls beside the crossover.  The counts travel between workgroups (and XCDs: every XCD
// has its own L2) through agent-scope atomics, ordered by a release fence before the ticket.
struct GnxScanOut {
  int32_t* off;           // [K][stride] exclusive block offsets, total at [nb]
  int32_t* out;           // device totals [K], or null
  int64_t* host;          // pinned host words: totals at [0 .. K), or null
  long long seq;          // != 0: host[3] = seq after the totals (gnx_wait_published)
  const int32_t* extra;   // one more device word for the host (host[12]), or null
  unsigned int* ticket;   // zero before the launch; the last workgroup zeroes it again
  int stride;
};

// call with the counts of this workgroup in v[0 .. K) (thread 0's copy is used), from ALL
// threads of a 256-thread block; lds: int[8].
// No release FENCE before the ticket: at agent scope that is a write-back of the whole L2
// (buffer_wbl2), i.e. of everything the kernel has written so far - it tripled k_pair_flags.
// The counts are the only data that travels, and they travel as agent-scope atomic stores
// (write-through) that have completed (s_waitcnt) before the ticket is taken; the scanning
// workgroup reads them with agent-scope atomic loads.
#define GNX_SCAN_CHUNKS 16     // block counts held in registers: 16 x 256 x 1024 = 4 M items
template <int K>
__device__ __forceinline__ void gnx_count_and_scan(const int v[K], int32_t* __restrict__ cnt,
                                                   const GnxScanOut& S, int* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = gridDim.x;
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k)
      __hip_atomic_store(&cnt[k * S.stride + blockIdx.x], v[k], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(S.ticket, 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
    lds[7] = (t == (unsigned int)nb - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!lds[7]) return;
  const int n_chunks = (nb + 255) >> 8;
  int totals[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int32_t* c = cnt + k * S.stride;
    int32_t* o = S.off + k * S.stride;
    int val[GNX_SCAN_CHUNKS];
#pragma unroll
    for (int q = 0; q < GNX_SCAN_CHUNKS; ++q) {
      const int b = q * 256 + tid;
      val[q] = (q < n_chunks && b < nb)
                   ? __hip_atomic_load(&c[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    int carry = 0;
    for (int q0 = 0; q0 < n_chunks; q0 += GNX_SCAN_CHUNKS) {
#pragma unroll
      for (int q = 0; q < GNX_SCAN_CHUNKS; ++q) {
        if (q0 + q >= n_chunks) break;
        const int b = (q0 + q) * 256 + tid;
        int vv = val[q];
        if (q0 > 0)        // (more than 4 M items: the rest comes straight from memory)
          vv = b < nb ? __hip_atomic_load(&c[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int x = vv;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int y = __shfl_up(x, d);
          if (lane >= d) x += y;
        }
        __syncthreads();
        if (lane == 63) lds[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += lds[w];
        if (b < nb) o[b] = carry + woff + x - vv;
        carry += lds[0] + lds[1] + lds[2] + lds[3];
      }
    }
    totals[k] = carry;
    if (tid == 0) o[nb] = carry;
  }
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (S.out) S.out[k] = totals[k];
    if (S.host) {
      // pinned host words, system-scope stores; the sequence number goes last, after the
      // others have completed (again no release fence: that would write the L2 back)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        __hip_atomic_store(&S.host[k], (int64_t)(k < K ? totals[k < K ? k : 0] : 0),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (S.extra)
        __hip_atomic_store(&S.host[12], (int64_t)*S.extra, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      if (S.seq) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&S.host[3], (int64_t)S.seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __hip_atomic_store(S.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
