// Physical blocks of the genome table and who refers to them.
//
// An individual owns a LOGICAL genome row (GnxSoA.grow; handed out and taken back by
// rank from the free-row stack, counted exactly on the host as before).  A homologue
// (logical half lh = 2 * row + h) is NB blocks of BW words; where block b actually lives
// is a second table: hmap[lh * NB + b] is the PHYSICAL block (BW words at G + phys * BW).
// Blocks are immutable once written, so the part of a gamete that
// carries no switch point - the child's block IS the block of one of the parent's
// homologues, bit for bit (ops/mating.py:165-167: the subsetter is constant there) - is
// not copied: the child's entry points at the parent's block.  With one expected crossover
// per gamete (r = 1/L) e^-1 of all gametes carry none at all, and the others switch in one
// or two blocks.
//
// Nobody counts the references.  A block is live iff some living individual's table points
// at it; every few steps, when the stack of free blocks runs low, a mark-and-sweep pass
// (gnx_gc: mark the blocks of the living, put everything else back on the stack) finds the
// blocks the dead have left behind - about 0.15 ms every dozen steps instead of a million
// atomic count updates in every step.  The live blocks never outnumber 2 * NB x (logical
// rows in use), and the host makes sure before every kernel that pops blocks that the stack
// holds enough (gnx_half_reserve: it collects first if it may not), so the stack cannot
// run dry while logical rows are left.  The top bit of a table entry (GNX_OWN) says that the
// block was cut for this individual and never shared: only such a block may take a mutation
// in place.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GNX_MAX_NB 28

#define GNX_OWN 0x80000000u      // table entry: the block belongs to this logical block alone
#define GNX_BLK(e) ((int32_t)((uint32_t)(e) & 0x7fffffffu))

struct GnxHalves {
  int32_t* hmap;    // [2 * row span * NB]  logical block -> physical block | GNX_OWN
  int32_t* stack;   // free physical blocks
  int32_t* top;     // how many
  int NB;           // blocks per homologue
  int BW;           // u64 words per block (a multiple of 16: whole 128-byte lines)
};

// word w of logical half lh
__device__ __forceinline__ int64_t gnx_word_at(const GnxHalves& H, int64_t lh, int w) {
  const int b = w / H.BW;
  return (int64_t)GNX_BLK(H.hmap[lh * H.NB + b]) * H.BW + (w - b * H.BW);
}
// 16-byte chunk c of logical half lh (index into a u64x2 view of the table)
__device__ __forceinline__ int64_t gnx_chunk_at(const GnxHalves& H, int64_t lh, int c) {
  const int bw16 = H.BW >> 1;
  const int b = c / bw16;
  return (int64_t)GNX_BLK(H.hmap[lh * H.NB + b]) * bw16 + (c - b * bw16);
}

// wave-aggregated pop: every lane with want == true gets a free physical block
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

// a fresh physical block for logical block lb (= lh * NB + b)
__device__ __forceinline__ int32_t gnx_half_new(const GnxHalves& H, int64_t lb, bool want) {
  const int32_t p = gnx_half_pop(H, want);
  if (want) H.hmap[lb] = (int32_t)((uint32_t)p | GNX_OWN);
  return p;
}

// Which blocks of a gamete hold a switch point, and which homologue the others follow.
// bp[0 .. nbp) = the path's switch loci (ascending); start = start homologue.
// mixed bit b: block b holds a switch (it must be cut); sel bit b: homologue (0 / 1) the
// gamete is on at the first locus of block b.
__device__ __forceinline__ void gnx_block_masks(const int32_t* __restrict__ bp, int nbp, int start,
                                                int NB, int BW, unsigned int& mixed,
                                                unsigned int& sel) {
  const unsigned int all = (NB >= 32) ? ~0u : ((1u << NB) - 1u);
  unsigned int mx = 0u, sl = start ? all : 0u;
  const int loci_per_block = BW * 64;
  for (int q = 0; q < nbp; ++q) {
    const int blk = min(bp[q] / loci_per_block, NB - 1);
    mx |= 1u << blk;
    sl ^= all & ~((2u << blk) - 1u);        // every later block starts on the other homologue
  }
  mixed = mx;
  sel = sl;
}

// exclusive sums of four small counts per thread over a block of 256 threads, in item
// order (item = round * 256 + tid); lds: int[16]
__device__ __forceinline__ void gnx_block_sums(const int v[4], int off[4], int& total, int* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int inc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int x = v[r];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    inc[r] = x;
    if (lane == 63) lds[r * 4 + wave] = x;
  }
  __syncthreads();
  int run = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int o = run;
    for (int w = 0; w < wave; ++w) o += lds[r * 4 + w];
    off[r] = o + inc[r] - v[r];
    run += lds[r * 4] + lds[r * 4 + 1] + lds[r * 4 + 2] + lds[r * 4 + 3];
  }
  total = run;
  __syncthreads();
}
