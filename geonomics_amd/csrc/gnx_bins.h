// Half-window bin counts of the density estimate (utils/spatial.py:73-97),
// accumulated by the kernels that hold the positions anyway (k_permute: adults, k_offspring:
// newborns, k_pair_compact: pair midpoints) instead of by a pass of their own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GnxBinP {
  int32_t* bins;     // [nby * nbx] or null (not fused: gnx_l_bins counts later)
  double inv_hww;
  int nbx, nby;
};

// the bin of a point: the arithmetic of k_bins (f64, floor, clamped to the last bin)
__device__ __forceinline__ int gnx_bin_of(const GnxBinP& B, float x, float y) {
  const int hx = min(B.nbx - 1, (int)floor((double)x * B.inv_hww));
  const int hy = min(B.nby - 1, (int)floor((double)y * B.inv_hww));
  return hy * B.nbx + hx;
}

// one count per active lane.  Slots are cell-sorted, so the lanes of a wave fall into one
// or two bins: one atomic per distinct bin of the wave (integer counts: the result does not
// depend on the order).  Call from all lanes of the wave.
__device__ __forceinline__ void gnx_bin_add(int32_t* __restrict__ bins, int bin, bool act) {
  unsigned long long todo = __ballot(act);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lb = __shfl(bin, leader);
    const unsigned long long same = __ballot(act && bin == lb);
    if (lane == leader) atomicAdd(&bins[lb], (int)__popcll(same));
    todo &= ~same;
  }
}
