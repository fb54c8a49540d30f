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

