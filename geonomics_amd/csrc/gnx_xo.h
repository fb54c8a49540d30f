// Bitmask crossover kernels (ops/mating.py:130-214) - shared by libgnxhip.so
// (gnx_kernels_genome.hip) and the kernel lab (tools/xo_lab.hip), so that what the lab
// times is the product kernel.
//
// For gamete p in {0,1} of an offspring:  gamete[l] = parent_p.g[l, path_k[l] XOR s]
// i.e. with m = path ^ (-s):  gamete = (hom0 & ~m) | (hom1 & m), 128 bits per lane per
// access.  Child homologue 0 <- pair[0]'s gamete, homologue 1 <- pair[1]'s (:169).
//
// Work arrives as a JOB LIST, one 16-byte record per gamete (GnxXoJob), built on the
// device after the step's death draws: only offspring that SURVIVE their first
// mortality round get a genome row (the others' 25-KB rows would be written and never
// read), and only gametes that carry a switch point get a job: the others alias the
// parent's blocks (gnx_half.h).  One WAVEFRONT owns one job at a time: the record is a scalar load,
// parent row / child row / path / start homologue live in SGPRs, and the 64 lanes stream
// the homologue in 16-byte chunks (1 KiB per wave-instruction, U chunks in flight per
// lane).  The grid is fixed (persistent-style, job-strided) and reads the job count from
// device memory, so the host never needs the count to launch.
//
// k_xo_sparse: masks rebuilt from the path's short breakpoint list (<= GNX_SPARSE_MAX_BP
//   switches, held in a VGPR and broadcast with v_readlane); a chunk loads only the ONE
//   homologue it copies (the rare chunks holding a switch are patched afterwards), which
//   halves the read traffic when crossovers are rare (r = 1/L): L/2 bytes per birth.
// k_xo_dense: masks read from the bit-packed path table (any recombination map): two
//   homologues + mask read, one written per gamete: L bytes per birth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gnx_half.h"

typedef unsigned long long u64;

struct alignas(16) u64x2 {
  u64 a, b;
};

struct alignas(16) GnxXoJob {
  int32_t ph0, ph1;   // the parent's two physical blocks at this block index (gnx_half.h)
  int32_t dst;        // physical block the gamete's block is written to
  int32_t ks;         // (recombination path * 2 + start homologue) | block index << 24
};

// One gamete of a child that has logical row `row`, block by block: a block without a
// switch point refers to the parent's block (neither may take a mutation in place from now
// on), a block with one gets a fresh physical block and a crossover job.  A
// ghost parent (prow < 0, tiled run: the gamete arrives from the tile that owns it) and
// dense masks (bp_off == null) get fresh blocks throughout, the former without jobs.
// Called by all lanes of a wave (act = this lane has a gamete).
__device__ __forceinline__ void gnx_xo_gamete(const GnxHalves& H, bool act, int32_t row, int p,
                                              int32_t prow, int key, int st,
                                              const int32_t* __restrict__ bp_off,
                                              const int32_t* __restrict__ bp_loci,
                                              GnxXoJob* __restrict__ jobs,
                                              int32_t* __restrict__ n_jobs) {
  const bool local = act && prow >= 0;
  unsigned int mixed = ~0u, sel = 0u;
  if (local && bp_off)
    gnx_block_masks(bp_loci + bp_off[key], bp_off[key + 1] - bp_off[key], st, H.NB, H.BW, mixed,
                    sel);
  for (int b = 0; b < H.NB; ++b) {
    const bool fresh = act && (((mixed >> b) & 1u) != 0u);
    const bool shared = act && !fresh;
    const int64_t lb = ((int64_t)row * 2 + p) * H.NB + b;
    const int32_t dst = gnx_half_new(H, lb, fresh);
    if (shared) {
      const int64_t plb = ((int64_t)prow * 2 + ((sel >> b) & 1u)) * H.NB + b;
      const int32_t e = H.hmap[plb];
      H.hmap[lb] = GNX_BLK(e);
      if (e < 0) H.hmap[plb] = GNX_BLK(e);        // the parent's block is shared from now on
    }
    const bool job = fresh && local;
    const int32_t idx = gnx_wave_append(n_jobs, job);
    if (job) {
      GnxXoJob j;
      j.ph0 = GNX_BLK(H.hmap[((int64_t)prow * 2) * H.NB + b]);
      j.ph1 = GNX_BLK(H.hmap[((int64_t)prow * 2 + 1) * H.NB + b]);
      j.dst = dst;
      j.ks = (key * 2 + st) | (b << 24);
      jobs[idx] = j;
    }
  }
}

template <bool NT>
__device__ __forceinline__ u64x2 xo_load(const u64x2* __restrict__ p) {
  if (NT) {
    u64x2 v;
    v.a = __builtin_nontemporal_load(&p->a);
    v.b = __builtin_nontemporal_load(&p->b);
    return v;
  }
  return *p;
}

// the child row is not read again this step: keep it out of L2 / MALL
__device__ __forceinline__ void xo_store(u64x2* __restrict__ p, u64x2 v) {
  __builtin_nontemporal_store(v.a, &p->a);
  __builtin_nontemporal_store(v.b, &p->b);
}

// mask of chunk c (loci [128c, 128c+128)) from the breakpoint list: lane q of `mybp`
// holds switch point q (ascending), broadcast with v_readlane - no memory touched
__device__ __forceinline__ u64x2 xo_mask_lanes(int c, u64 s, int mybp, int nbp) {
  const int lo = c * 128;
  u64 par = s;
  u64x2 m;
  m.a = 0;
  m.b = 0;
  for (int q = 0; q < nbp; ++q) {
    const int bpl = __builtin_amdgcn_readlane(mybp, q);
    const int d = bpl - lo;
    par ^= d < 0 ? ~0ull : 0ull;
    const u64 fa = ~0ull << (d & 63);
    m.a ^= (d >= 0 && d < 64) ? fa : 0ull;
    m.b ^= (d >= 0 && d < 64) ? ~0ull : ((d >= 64 && d < 128) ? fa : 0ull);
  }
  m.a ^= par;
  m.b ^= par;
  return m;
}

// The switch points that fall INSIDE a job's block, packed next to the job (round 3): loci
// relative to the block's first locus, 16 bits each (a block is at most 65 536 loci), the
// homologue the gamete is on at the block's start, and how many there are.  More than three
// in one block (GNX_BP_MORE): the kernel looks the path's list up as before.  With them the
// wave needs no look-up in bp_off / bp_loci before its first streaming load - two dependent
// loads less per 896-byte job - and the list is not fetched at all.
#define GNX_BP_MORE 0x8u
struct alignas(8) GnxJobBp {
  uint16_t bp[3];
  uint16_t meta;      // bits 0-1: how many (0 .. 3), bit 2: homologue at the block's start, bit 3: GNX_BP_MORE
};

// mask of chunk c of the BLOCK (loci [128c, 128c+128) relative to its start) from <= 3
// scalar switch points: the arithmetic of xo_mask_lanes
__device__ __forceinline__ u64x2 xo_mask_inline(int c, u64 s, int b0, int b1, int b2, int n) {
  const int lo = c * 128;
  u64 par = s;
  u64x2 m;
  m.a = 0;
  m.b = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int bpl = q == 0 ? b0 : (q == 1 ? b1 : b2);
    if (q < n) {
      const int d = bpl - lo;
      par ^= d < 0 ? ~0ull : 0ull;
      const u64 fa = ~0ull << (d & 63);
      m.a ^= (d >= 0 && d < 64) ? fa : 0ull;
      m.b ^= (d >= 0 && d < 64) ? ~0ull : ((d >= 64 && d < 128) ? fa : 0ull);
    }
  }
  m.a ^= par;
  m.b ^= par;
  return m;
}

// homologue (0/1) the gamete copies at locus l
__device__ __forceinline__ int xo_sel_sparse(int l, int start, int mybp, int nbp) {
  int sel = start;
  for (int q = 0; q < nbp; ++q) sel ^= (__builtin_amdgcn_readlane(mybp, q) <= l) ? 1 : 0;
  return sel;
}

// The streaming part is branch-free: every chunk issues exactly one load, from the
// homologue selected by a v_cndmask on the address, U loads back to back (divergent
// copy-h0 / copy-h1 / blend arms make the compiler drain vmcnt before each arm).
template <int U, bool NT_LD>
__global__ void __launch_bounds__(256)
k_xo_sparse(const int32_t* __restrict__ n_jobs_p, int W16, const u64x2* __restrict__ G,
            u64x2* __restrict__ Gout, const GnxXoJob* __restrict__ jobs,
            const int32_t* __restrict__ bp_off, const int32_t* __restrict__ bp_loci,
            int part_lo, int part_hi, unsigned long long* __restrict__ acc,
            const GnxJobBp* __restrict__ jobs_bp) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // this launch's share of the job list, in 1/1024ths of the (device-resident) count
  const int n_all = *n_jobs_p;
  const int j_lo = (int)(((long long)n_all * part_lo) >> 10);
  const int n_jobs = (int)(((long long)n_all * part_hi) >> 10);
  const int n_waves = (int)gridDim.x * 4;
  // gametes copied by the launches so far (the host prices them when it reads the timers)
  if (acc && blockIdx.x == 0 && threadIdx.x == 0 && n_jobs > j_lo)
    atomicAdd(acc, (unsigned long long)(n_jobs - j_lo));
  // software pipeline over the wave's jobs: the NEXT job's record (and its switch points) are
  // fetched before the current job streams, so a job's latency is the data's alone
  int j = j_lo + (int)blockIdx.x * 4 + wv;
  GnxXoJob jb_n;
  uint2 w_n = make_uint2(0u, GNX_BP_MORE << 16);
  if (j < n_jobs) {
    jb_n = jobs[j];
    if (jobs_bp) w_n = *(const uint2*)(jobs_bp + j);
  }
  for (; j < n_jobs; j += n_waves) {
    const GnxXoJob jb = jb_n;
    const uint2 w = w_n;
    if (j + n_waves < n_jobs) {
      jb_n = jobs[j + n_waves];
      if (jobs_bp) w_n = *(const uint2*)(jobs_bp + j + n_waves);
    }
    const int ph0 = __builtin_amdgcn_readfirstlane(jb.ph0);
    const int ph1 = __builtin_amdgcn_readfirstlane(jb.ph1);
    const int dsth = __builtin_amdgcn_readfirstlane(jb.dst);
    const int ks = __builtin_amdgcn_readfirstlane(jb.ks);
    const int key = (ks & 0xffffff) >> 1;
    const u64 s = (ks & 1) ? ~0ull : 0ull;
    const int cb = (ks >> 24) * W16;           // first chunk of this block in the homologue
    const u64x2* h0 = G + (int64_t)ph0 * W16;  // (W16 = chunks per BLOCK)
    const u64x2* h1 = G + (int64_t)ph1 * W16;
    u64x2* dst = Gout + (int64_t)dsth * W16;
    // the block's own switch points ride with the job (fused builder): no look-ups
    const unsigned int ib_lo = __builtin_amdgcn_readfirstlane(w.x);
    const unsigned int ib_hi = __builtin_amdgcn_readfirstlane(w.y);
    if (!((ib_hi >> 16) & GNX_BP_MORE)) {
      const int b0 = (int)(ib_lo & 0xffffu), b1 = (int)(ib_lo >> 16), b2 = (int)(ib_hi & 0xffffu);
      const int n = (int)((ib_hi >> 16) & 3u);
      const u64 sb = ((ib_hi >> 16) & 4u) ? ~0ull : 0ull;
      for (int c0 = lane; c0 < W16; c0 += 64 * U) {
        u64x2 v[U];
        bool mixed = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = min(c0 + u * 64, W16 - 1);
          const u64x2 m = xo_mask_inline(c, sb, b0, b1, b2, n);
          const bool one = (m.a & m.b) == ~0ull;
          mixed |= !one && (m.a | m.b) != 0ull;
          v[u] = xo_load<NT_LD>((one ? h1 : h0) + c);
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(mixed) != 0ull, 0)) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int c = min(c0 + u * 64, W16 - 1);
            const u64x2 m = xo_mask_inline(c, sb, b0, b1, b2, n);
            if ((m.a & m.b) != ~0ull && (m.a | m.b) != 0ull) {
              const u64x2 b = h1[c];
              v[u].a = (v[u].a & ~m.a) | (b.a & m.a);
              v[u].b = (v[u].b & ~m.b) | (b.b & m.b);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = c0 + u * 64;
          if (c < W16) xo_store(dst + c, v[u]);
        }
      }
      continue;
    }
    const int bp0 = __builtin_amdgcn_readfirstlane(bp_off[key]);
    const int nbp = __builtin_amdgcn_readfirstlane(bp_off[key + 1]) - bp0;
    const int mybp = lane < nbp ? bp_loci[bp0 + lane] : 0x7fffffff;
    for (int c0 = lane; c0 < W16; c0 += 64 * U) {
      // masks are not kept: the rare batch that holds a switch point rebuilds them (the
      // kernel's register footprint decides how many of the step's small kernels fit on
      // the CUs beside it)
      u64x2 v[U];
      bool mixed = false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = min(c0 + u * 64, W16 - 1);     // tail lanes re-read the last chunk
        const u64x2 m = xo_mask_lanes(cb + c, s, mybp, nbp);
        const bool one = (m.a & m.b) == ~0ull;
        mixed |= !one && (m.a | m.b) != 0ull;
        v[u] = xo_load<NT_LD>((one ? h1 : h0) + c);
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(mixed) != 0ull, 0)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = min(c0 + u * 64, W16 - 1);
          const u64x2 m = xo_mask_lanes(cb + c, s, mybp, nbp);
          if ((m.a & m.b) != ~0ull && (m.a | m.b) != 0ull) {
            const u64x2 b = h1[c];
            v[u].a = (v[u].a & ~m.a) | (b.a & m.a);
            v[u].b = (v[u].b & ~m.b) | (b.b & m.b);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64;
        if (c < W16) xo_store(dst + c, v[u]);
      }
    }
  }
}

// K jobs per wave and iteration (blocks of at most 64 chunks, switch points inline): a
// wave of k_xo_sparse<1> has ONE block's loads in flight (56 lanes x 16 B at L = 10^5), then
// stores, then the next job; here the loads of two blocks are issued before either is
// stored, and the records of the next two are fetched meanwhile.  A job whose block holds
// more than three switch points (GNX_BP_MORE) takes the list path on its own.
template <bool NT_LD>
__device__ __forceinline__ void xo_job_lists(const GnxXoJob& jb, int W16, const u64x2* __restrict__ G,
                                             u64x2* __restrict__ Gout,
                                             const int32_t* __restrict__ bp_off,
                                             const int32_t* __restrict__ bp_loci, int lane) {
  const int ph0 = __builtin_amdgcn_readfirstlane(jb.ph0);
  const int ph1 = __builtin_amdgcn_readfirstlane(jb.ph1);
  const int dsth = __builtin_amdgcn_readfirstlane(jb.dst);
  const int ks = __builtin_amdgcn_readfirstlane(jb.ks);
  const int key = (ks & 0xffffff) >> 1;
  const u64 s = (ks & 1) ? ~0ull : 0ull;
  const int cb = (ks >> 24) * W16;
  const u64x2* h0 = G + (int64_t)ph0 * W16;
  const u64x2* h1 = G + (int64_t)ph1 * W16;
  u64x2* dst = Gout + (int64_t)dsth * W16;
  const int bp0 = __builtin_amdgcn_readfirstlane(bp_off[key]);
  const int nbp = __builtin_amdgcn_readfirstlane(bp_off[key + 1]) - bp0;
  const int mybp = lane < nbp ? bp_loci[bp0 + lane] : 0x7fffffff;
  const int c = min(lane, W16 - 1);
  const u64x2 m = xo_mask_lanes(cb + c, s, mybp, nbp);
  const bool one = (m.a & m.b) == ~0ull;
  u64x2 v = xo_load<NT_LD>((one ? h1 : h0) + c);
  if (!one && (m.a | m.b) != 0ull) {
    const u64x2 b = h1[c];
    v.a = (v.a & ~m.a) | (b.a & m.a);
    v.b = (v.b & ~m.b) | (b.b & m.b);
  }
  if (lane < W16) xo_store(dst + lane, v);
}

// one job whose switch points ride with it (at most three inside the block)
template <bool NT_LD>
__device__ __forceinline__ void xo_job_inline(const GnxXoJob& jb, unsigned int lo, unsigned int hi,
                                              int W16, const u64x2* __restrict__ G,
                                              u64x2* __restrict__ Gout, int lane) {
  const u64x2* h0 = G + (int64_t)__builtin_amdgcn_readfirstlane(jb.ph0) * W16;
  const u64x2* h1 = G + (int64_t)__builtin_amdgcn_readfirstlane(jb.ph1) * W16;
  u64x2* dst = Gout + (int64_t)__builtin_amdgcn_readfirstlane(jb.dst) * W16;
  const int c = min(lane, W16 - 1);
  const u64x2 m = xo_mask_inline(c, ((hi >> 16) & 4u) ? ~0ull : 0ull, (int)(lo & 0xffffu),
                                 (int)(lo >> 16), (int)(hi & 0xffffu), (int)((hi >> 16) & 3u));
  const bool one = (m.a & m.b) == ~0ull;
  u64x2 v = xo_load<NT_LD>((one ? h1 : h0) + c);
  if (!one && (m.a | m.b) != 0ull) {
    const u64x2 b = h1[c];
    v.a = (v.a & ~m.a) | (b.a & m.a);
    v.b = (v.b & ~m.b) | (b.b & m.b);
  }
  if (lane < W16) xo_store(dst + lane, v);
}

template <bool NT_LD, int K>
__global__ void __launch_bounds__(256)
k_xo_sparse_pair(const int32_t* __restrict__ n_jobs_p, int W16, const u64x2* __restrict__ G,
                 u64x2* __restrict__ Gout, const GnxXoJob* __restrict__ jobs,
                 const int32_t* __restrict__ bp_off, const int32_t* __restrict__ bp_loci,
                 int part_lo, int part_hi, unsigned long long* __restrict__ acc,
                 const GnxJobBp* __restrict__ jobs_bp) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_all = *n_jobs_p;
  const int j_lo = (int)(((long long)n_all * part_lo) >> 10);
  const int n_jobs = (int)(((long long)n_all * part_hi) >> 10);
  const int n_waves = (int)gridDim.x * 4;
  if (acc && blockIdx.x == 0 && threadIdx.x == 0 && n_jobs > j_lo)
    atomicAdd(acc, (unsigned long long)(n_jobs - j_lo));
  int j = j_lo + (int)blockIdx.x * 4 + wv;
  // the records of this iteration's K jobs (fetched during the last one)
  GnxXoJob rec[K];
  uint2 wrd[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    wrd[k] = make_uint2(0u, 0u);
    if (j + k * n_waves < n_jobs) {
      rec[k] = jobs[j + k * n_waves];
      wrd[k] = *(const uint2*)(jobs_bp + j + k * n_waves);
    }
  }
  const int c = min(lane, W16 - 1);
  for (; j < n_jobs; j += K * n_waves) {
    GnxXoJob jb[K];
    unsigned int lo[K], hi[K];
    bool rare = j + (K - 1) * n_waves >= n_jobs;           // (uniform) fewer than K jobs left
#pragma unroll
    for (int k = 0; k < K; ++k) {
      jb[k] = rec[k];
      lo[k] = __builtin_amdgcn_readfirstlane(wrd[k].x);
      hi[k] = __builtin_amdgcn_readfirstlane(wrd[k].y);
      rare = rare || ((hi[k] >> 16) & GNX_BP_MORE) != 0u;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int jn = j + (K + k) * n_waves;
      if (jn < n_jobs) {
        rec[k] = jobs[jn];
        wrd[k] = *(const uint2*)(jobs_bp + jn);
      }
    }
    if (__builtin_expect(rare, 0)) {
      // the rare shapes, one job at a time
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (j + k * n_waves >= n_jobs) break;
        if ((hi[k] >> 16) & GNX_BP_MORE) xo_job_lists<NT_LD>(jb[k], W16, G, Gout, bp_off, bp_loci, lane);
        else xo_job_inline<NT_LD>(jb[k], lo[k], hi[k], W16, G, Gout, lane);
      }
      continue;
    }
    // straight-line from here: masks and addresses of all K jobs (pure ALU), all loads back to
    // back - the chunk that holds a switch point (one lane per job) fetches its second
    // homologue right away -, ONE wait, the blends, the stores.  (With a branch or a store
    // between them the compiler drains vmcnt - stores included - before the next store.)
    const u64x2 *h0[K], *h1[K];
    u64x2 m[K], v[K], x[K];
    bool mix[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      h0[k] = G + (int64_t)__builtin_amdgcn_readfirstlane(jb[k].ph0) * W16 + c;
      h1[k] = G + (int64_t)__builtin_amdgcn_readfirstlane(jb[k].ph1) * W16 + c;
      m[k] = xo_mask_inline(c, ((hi[k] >> 16) & 4u) ? ~0ull : 0ull, (int)(lo[k] & 0xffffu),
                            (int)(lo[k] >> 16), (int)(hi[k] & 0xffffu), (int)((hi[k] >> 16) & 3u));
      const bool one = (m[k].a & m[k].b) == ~0ull;
      mix[k] = !one && (m[k].a | m[k].b) != 0ull;
      if (one) h0[k] = h1[k];
      x[k].a = x[k].b = 0ull;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = xo_load<NT_LD>(h0[k]);
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (mix[k]) x[k] = *h1[k];
    __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const u64 keep = mix[k] ? ~0ull : 0ull;
      const u64 ma = m[k].a & keep, mb = m[k].b & keep;
      v[k].a = (v[k].a & ~ma) | (x[k].a & ma);
      v[k].b = (v[k].b & ~mb) | (x[k].b & mb);
    }
    if (lane < W16) {
#pragma unroll
      for (int k = 0; k < K; ++k)
        xo_store(Gout + (int64_t)__builtin_amdgcn_readfirstlane(jb[k].dst) * W16 + lane, v[k]);
    }
  }
}

template <int U, bool NT_LD>
__global__ void __launch_bounds__(256)
k_xo_dense(const int32_t* __restrict__ n_jobs_p, int W16, const u64x2* __restrict__ G,
           u64x2* __restrict__ Gout, const GnxXoJob* __restrict__ jobs,
           const u64x2* __restrict__ paths, int path_w16, int part_lo, int part_hi,
           unsigned long long* __restrict__ acc) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_all = *n_jobs_p;
  const int j_lo = (int)(((long long)n_all * part_lo) >> 10);
  const int n_jobs = (int)(((long long)n_all * part_hi) >> 10);
  const int n_waves = (int)gridDim.x * 4;
  // gametes copied by the launches so far (the host prices them when it reads the timers)
  if (acc && blockIdx.x == 0 && threadIdx.x == 0 && n_jobs > j_lo)
    atomicAdd(acc, (unsigned long long)(n_jobs - j_lo));
  for (int j = j_lo + (int)blockIdx.x * 4 + wv; j < n_jobs; j += n_waves) {
    const GnxXoJob jb = jobs[j];
    const int ph0 = __builtin_amdgcn_readfirstlane(jb.ph0);
    const int ph1 = __builtin_amdgcn_readfirstlane(jb.ph1);
    const int dsth = __builtin_amdgcn_readfirstlane(jb.dst);
    const int ks = __builtin_amdgcn_readfirstlane(jb.ks);
    const int key = (ks & 0xffffff) >> 1;
    const u64 s = (ks & 1) ? ~0ull : 0ull;
    const int cb = (ks >> 24) * W16;           // first chunk of this block in the homologue
    const u64x2* h0 = G + (int64_t)ph0 * W16;  // (W16 = chunks per BLOCK)
    const u64x2* h1 = G + (int64_t)ph1 * W16;
    u64x2* dst = Gout + (int64_t)dsth * W16;
    // the path table (n_recomb_sims x L/8 bytes) is re-read by every gamete that drew
    // the key: default cache policy, so it can stay in L2 / the Infinity Cache
    const u64x2* pm = paths + (int64_t)key * path_w16 + cb;
    for (int c0 = lane; c0 < W16; c0 += 64 * U) {
      u64x2 m[U], a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = min(c0 + u * 64, W16 - 1);
        m[u] = pm[c];
        a[u] = xo_load<NT_LD>(h0 + c);
        b[u] = xo_load<NT_LD>(h1 + c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64;
        const u64 ma = m[u].a ^ s, mb = m[u].b ^ s;
        u64x2 o;
        o.a = (a[u].a & ~ma) | (b[u].a & ma);
        o.b = (a[u].b & ~mb) | (b[u].b & mb);
        if (c < W16) xo_store(dst + c, o);
      }
    }
  }
}

// unroll that wastes the fewest wave-loads for a homologue of W16 chunks
static inline int gnx_xo_pick_unroll(int W16) {
  const int T = (W16 + 63) / 64;
  int best = 8, waste = 1 << 30;
  if (T <= 3) return T;                  // short blocks: exactly as many loads as chunks
  for (int U = 8; U >= 4; --U) {
    const int w = ((T + U - 1) / U) * U - T;
    if (w < waste) {
      waste = w;
      best = U;
    }
  }
  return best;
}
