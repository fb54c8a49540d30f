// Genome kernels: bitmask crossover (the dominant byte mover of the step),
// phenotype at birth, starting genomes, point mutations, genome gathers.
//
// Genome layout in HBM: G[row][hom][W64] u64, little-endian bits, bit l of
// homologue h == Individual.g[l, h] (structs/individual.py:103-104).  W64 is
// padded to a multiple of 16 words so every homologue starts on a 128-byte line
// and splits into whole 16-byte chunks.  Individuals reference rows through
// GnxSoA.grow; survivors' rows are never moved.
#include "gnx_internal.h"
#include "gnx_rng.h"

typedef unsigned long long u64;

struct alignas(16) u64x2 {
  u64 a, b;
};

// ---------------------------------------------------------------- crossover
// ops/mating.py:130-214.  For gamete p in {0,1} of an offspring:
//   gamete[l] = parent_p.g[l, path_{k_p}[l] XOR s_p]
// i.e. with m = path ^ (-s):  gamete = (hom0 & ~m) | (hom1 & m), 128 bits per
// lane per access.  child hom 0 <- parent pair[0], hom 1 <- pair[1] (:169).
//
// One WAVEFRONT owns one gamete at a time (grid-stride over the 2B gametes):
// parent row, child row, path key and start homologue are wave-uniform (scalar
// registers, no per-chunk metadata chain), and the 64 lanes stream the
// homologue in 16-byte chunks, 1 KiB per wave-instruction, U independent
// chunks in flight per lane.
//
// k_crossover        (dense) : masks are read from the bit-packed path table (any
//                    recombination map): 2 homologues + 1 mask read per gamete.
// k_crossover_stream (sparse): masks are rebuilt from the path's short breakpoint
//                    list (<= 24 switches) and each chunk loads only the one
//                    homologue it copies, which halves the read traffic when
//                    crossovers are rare (r = 1/L).
//
// Epilogue (fused phenotype input): lane e re-derives the gamete's allele at
// trait locus e from the just-read (L2-hot) parental chunk and stores it in the
// compact table tbits[gamete][e], so phenotypes never gather from the fresh
// 25-KB child rows.

__device__ __forceinline__ u64x2 xo_dense_chunk(int c, u64 s, const u64x2* __restrict__ path_row,
                                                const u64x2* __restrict__ h0,
                                                const u64x2* __restrict__ h1) {
  u64x2 m = path_row[c];
  m.a ^= s;
  m.b ^= s;
  const u64x2 a = h0[c];
  const u64x2 b = h1[c];
  u64x2 out;
  out.a = (a.a & ~m.a) | (b.a & m.a);
  out.b = (a.b & ~m.b) | (b.b & m.b);
  return out;
}

template <int XO_UNROLL>
__global__ void __launch_bounds__(256)
k_crossover(int64_t B, int W16, const u64x2* __restrict__ G, u64x2* __restrict__ Gout,
            const int32_t* __restrict__ grow, int64_t first_slot,
            const int32_t* __restrict__ off_parent, const int32_t* __restrict__ off_keys,
            const uint8_t* __restrict__ off_start, const u64x2* __restrict__ paths, int n_tl,
            const int32_t* __restrict__ tl_loci, uint8_t* __restrict__ tbits) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t gam = wave0; gam < 2 * B; gam += n_waves) {
    const int64_t k = gam >> 1;
    const int p = (int)(gam & 1);
    // wave-uniform metadata
    const int prow = __builtin_amdgcn_readfirstlane(grow[off_parent[2 * k + p]]);
    if (prow < 0) continue;      // ghost parent (tiled run): gamete arrives from its tile
    const int key = __builtin_amdgcn_readfirstlane(off_keys[2 * k + p]);
    const u64 s = __builtin_amdgcn_readfirstlane((int)off_start[2 * k + p]) ? ~0ull : 0ull;
    const int crow = __builtin_amdgcn_readfirstlane(grow[first_slot + k]);
    const u64x2* h0 = G + ((int64_t)prow * 2 + 0) * W16;
    const u64x2* h1 = G + ((int64_t)prow * 2 + 1) * W16;
    u64x2* dst = Gout + ((int64_t)crow * 2 + p) * W16;
    const u64x2* prow_mask = paths + (int64_t)key * W16;
    for (int c0 = lane; c0 < W16; c0 += 64 * XO_UNROLL) {
      u64x2 out[XO_UNROLL];
#pragma unroll
      for (int u = 0; u < XO_UNROLL; ++u) {
        const int c = c0 + u * 64;
        if (c < W16) out[u] = xo_dense_chunk(c, s, prow_mask, h0, h1);
      }
#pragma unroll
      for (int u = 0; u < XO_UNROLL; ++u) {
        const int c = c0 + u * 64;
        // the child row is not read again this step: keep it out of L2 / MALL
        if (c < W16) {
          __builtin_nontemporal_store(out[u].a, &dst[c].a);
          __builtin_nontemporal_store(out[u].b, &dst[c].b);
        }
      }
    }
    // alleles at the trait loci -> compact table for the phenotype kernel
    for (int e = lane; e < n_tl; e += 64) {
      const int l = tl_loci[e];
      const u64x2 v = xo_dense_chunk(l >> 7, s, prow_mask, h0, h1);
      const int bit = l & 127;
      tbits[gam * n_tl + e] = (uint8_t)(((bit < 64 ? v.a >> bit : v.b >> (bit - 64))) & 1ull);
    }
  }
}

// Sparse paths.  The streaming part is branch-free: every chunk issues exactly
// one load, from the homologue selected by a v_cndmask on the address, U loads
// back to back (classifying chunks with divergent copy-h0 / copy-h1 / blend
// branches makes the compiler drain vmcnt before every arm that reuses a
// destination register, i.e. one load in flight per lane); the few chunks that
// contain a switch point (<= 24 per gamete) are patched in a rare wave-level
// branch afterwards.  Breakpoints live in a VGPR (lane q holds switch q) and
// are broadcast with v_readlane, so building the mask touches no memory.
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

// the parental chunk that holds the gamete's allele at locus l (the mask bit at l
// selects the homologue), and the allele itself
__device__ __forceinline__ u64x2 xo_trait_chunk(int l, u64 s, int mybp, int nbp,
                                                const u64x2* __restrict__ h0,
                                                const u64x2* __restrict__ h1) {
  const int c = l >> 7, bit = l & 127;
  const u64x2 mm = xo_mask_lanes(c, s, mybp, nbp);
  const u64 sel = ((bit < 64 ? mm.a >> bit : mm.b >> (bit - 64))) & 1ull;
  return (sel ? h1 : h0)[c];
}

__device__ __forceinline__ uint8_t xo_bit(u64x2 w, int l) {
  const int bit = l & 127;
  return (uint8_t)(((bit < 64 ? w.a >> bit : w.b >> (bit - 64))) & 1ull);
}

template <int XO_UNROLL>
__global__ void __launch_bounds__(256)
k_crossover_stream(int64_t B, int W16, const u64x2* __restrict__ G, u64x2* __restrict__ Gout,
                   const int32_t* __restrict__ grow, int64_t first_slot,
                   const int32_t* __restrict__ off_parent, const int32_t* __restrict__ off_keys,
                   const uint8_t* __restrict__ off_start, const int32_t* __restrict__ bp_off,
                   const int32_t* __restrict__ bp_loci, int n_tl,
                   const int32_t* __restrict__ tl_loci, uint8_t* __restrict__ tbits) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t gam = wave0; gam < 2 * B; gam += n_waves) {
    const int64_t k = gam >> 1;
    const int p = (int)(gam & 1);
    const int prow = __builtin_amdgcn_readfirstlane(grow[off_parent[2 * k + p]]);
    if (prow < 0) continue;      // ghost parent (tiled run): gamete arrives from its tile
    const int key = __builtin_amdgcn_readfirstlane(off_keys[2 * k + p]);
    const u64 s = __builtin_amdgcn_readfirstlane((int)off_start[2 * k + p]) ? ~0ull : 0ull;
    const int crow = __builtin_amdgcn_readfirstlane(grow[first_slot + k]);
    const u64x2* h0 = G + ((int64_t)prow * 2 + 0) * W16;
    const u64x2* h1 = G + ((int64_t)prow * 2 + 1) * W16;
    u64x2* dst = Gout + ((int64_t)crow * 2 + p) * W16;
    const int bp0 = __builtin_amdgcn_readfirstlane(bp_off[key]);
    const int nbp = __builtin_amdgcn_readfirstlane(bp_off[key + 1]) - bp0;
    const int mybp = lane < nbp ? bp_loci[bp0 + lane] : 0x7fffffff;
    // trait alleles (fused phenotype input): lane e fetches the chunk of trait
    // locus e NOW, so that the 64-byte sectors it pulls in are the ones the stream
    // below reads microseconds later (L2 hits); fetched after the stream they
    // had been evicted again and cost ~10 % extra HBM reads (PMC, profiles/)
    const int tl = lane < n_tl ? tl_loci[lane] : 0;
    u64x2 tw;
    tw.a = 0;
    tw.b = 0;
    if (lane < n_tl) tw = xo_trait_chunk(tl, s, mybp, nbp, h0, h1);
    for (int c0 = lane; c0 < W16; c0 += 64 * XO_UNROLL) {
      u64x2 m[XO_UNROLL], v[XO_UNROLL];
      bool mixed = false;
#pragma unroll
      for (int u = 0; u < XO_UNROLL; ++u) {
        const int c = min(c0 + u * 64, W16 - 1);     // tail lanes re-read the last chunk
        m[u] = xo_mask_lanes(c, s, mybp, nbp);
        const bool one = (m[u].a & m[u].b) == ~0ull;
        mixed |= !one && (m[u].a | m[u].b) != 0ull;
        v[u] = (one ? h1 : h0)[c];
      }
      if (__builtin_expect(mixed, 0)) {
#pragma unroll
        for (int u = 0; u < XO_UNROLL; ++u) {
          const int c = min(c0 + u * 64, W16 - 1);
          if ((m[u].a & m[u].b) != ~0ull && (m[u].a | m[u].b) != 0ull) {
            const u64x2 b = h1[c];
            v[u].a = (v[u].a & ~m[u].a) | (b.a & m[u].a);
            v[u].b = (v[u].b & ~m[u].b) | (b.b & m[u].b);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < XO_UNROLL; ++u) {
        const int c = c0 + u * 64;
        if (c < W16) {
          __builtin_nontemporal_store(v[u].a, &dst[c].a);
          __builtin_nontemporal_store(v[u].b, &dst[c].b);
        }
      }
    }
    if (lane < n_tl) tbits[gam * n_tl + lane] = xo_bit(tw, tl);
    for (int e = lane + 64; e < n_tl; e += 64) {      // more than 64 trait loci: the rest
      const int l = tl_loci[e];
      tbits[gam * n_tl + e] = xo_bit(xo_trait_chunk(l, s, mybp, nbp, h0, h1), l);
    }
  }
}

template <int U>
static void xo_launch_stream(gnx_state* h, int grid, int64_t first_slot, int64_t B) {
  const int W16 = h->W64 / 2;
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL((k_crossover_stream<U>), dim3(grid), dim3(256), 0, h->stream, B, W16,
                     (const u64x2*)h->G, (u64x2*)h->G, s.grow, first_slot, h->off_parent,
                     h->off_keys, h->off_start, h->bp_off, h->bp_loci, h->n_tl, h->tl_loci,
                     h->tbits);
}

template <int U>
static void xo_launch_dense(gnx_state* h, int grid, int64_t first_slot, int64_t B) {
  const int W16 = h->W64 / 2;
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL((k_crossover<U>), dim3(grid), dim3(256), 0, h->stream, B, W16,
                     (const u64x2*)h->G, (u64x2*)h->G, s.grow, first_slot, h->off_parent,
                     h->off_keys, h->off_start, (const u64x2*)h->paths, h->n_tl, h->tl_loci,
                     h->tbits);
}

int gnx_l_crossover(gnx_state* h, int64_t first_slot, int64_t B) {
  if (B == 0) return 0;
  // one wave per gamete, 4 waves per block; cap the grid and stride beyond
  static const int blocks_per_cu = getenv("GNX_XO_BPC") ? atoi(getenv("GNX_XO_BPC")) : 32;
  static const int unroll = getenv("GNX_XO_UNROLL") ? atoi(getenv("GNX_XO_UNROLL")) : 0;
  int grid = gnx_grid(2 * B, 4, 256 * blocks_per_cu);
  gnx_time_begin(h);
  if (h->sparse_paths) {          // measured: 8 chunks in flight per lane is best (A/B in profiles/)
    if (unroll == 2) xo_launch_stream<2>(h, grid, first_slot, B);
    else if (unroll == 4) xo_launch_stream<4>(h, grid, first_slot, B);
    else xo_launch_stream<8>(h, grid, first_slot, B);
  } else {
    if (unroll == 2) xo_launch_dense<2>(h, grid, first_slot, B);
    else if (unroll == 8) xo_launch_dense<8>(h, grid, first_slot, B);
    else xo_launch_dense<4>(h, grid, first_slot, B);
  }
  // algorithmic bytes per birth.  Dense masks: 4 parental
  // homologues + 2 masks read, 2 homologues written = 8 * L/8 = L bytes.
  // Sparse paths: each gamete chunk copies ONE parental homologue (the other
  // is never needed, the mask comes from a handful of breakpoints), so the
  // kernel must move 2 reads + 2 writes = 4 * L/8 = L/2 bytes per birth.
  // (padded row width W64*8 is what actually moves)
  gnx_time_end(h, GNX_K_CROSSOVER,
               (double)B * (h->sparse_paths ? 4.0 : 8.0) * (double)h->W64 * 8.0);
  HIPCHK(hipGetLastError());
  return 0;
}

// phenotypes of newly born offspring from the compact allele table written by
// k_crossover (same arithmetic as k_phenotype)
__global__ void k_phenotype_tbits(int64_t first, int64_t n, int64_t cap, int n_tl,
                                  const uint8_t* tbits, GnxTraitTab T, const uint8_t* dom,
                                  float* z) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint8_t* t0 = tbits + (2 * k) * n_tl;
  const uint8_t* t1 = tbits + (2 * k + 1) * n_tl;
  int e = 0;
  for (int t = 0; t < T.n_traits; ++t) {
    const int nl = T.n_loci[t];
    double acc = 0.0, g0 = 0.0;
    for (int j = 0; j < nl; ++j, ++e) {
      double gt = 0.5 * (double)((int)t0[e] + (int)t1[e]);
      if (dom) gt = fmin(gt * (1.0 + (double)dom[T.loci[t][j]]), 1.0);
      if (j == 0) g0 = gt;
      acc = acc + gt * T.alpha[t][j];
    }
    z[(int64_t)t * cap + first + k] = (float)(nl > 1 ? 0.5 + acc : g0);
  }
}

int gnx_l_phenotype_births(gnx_state* h, int64_t first_slot, int64_t n) {
  if (n == 0 || h->cfg.n_traits == 0) return 0;
  gnx_time_begin(h);
  hipLaunchKernelGGL(k_phenotype_tbits, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream,
                     first_slot, n, h->cfg.cap_inds, h->n_tl, h->tbits, gnx_trait_tab(h), h->dom,
                     h->soa[h->cur].z);
  gnx_time_end(h, GNX_K_PHENOTYPE, (double)n * (2.0 * h->n_tl + 4.0 * h->cfg.n_traits));
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- phenotype
// ops/selection.py:22-48: gt_l = (g[l,0] + g[l,1]) / 2 at the trait's loci
// (x (1 + dom_l), capped at 1, if any dominance); z = 0.5 + sum gt_l alpha_l for
// polygenic traits, z = gt_0 for monogenic ones.  f64 accumulate, f32 store.
__global__ void k_phenotype(int64_t first, int64_t n, int64_t cap, int W64, const u64* G,
                            const int32_t* grow, GnxTraitTab T, const uint8_t* dom, float* z) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int64_t slot = first + k;
  const u64* r0 = G + (int64_t)grow[slot] * 2 * W64;
  const u64* r1 = r0 + W64;
  for (int t = 0; t < T.n_traits; ++t) {
    const int nl = T.n_loci[t];
    double acc = 0.0, g0 = 0.0;
    for (int j = 0; j < nl; ++j) {
      int l = T.loci[t][j];
      int a = (int)((r0[l >> 6] >> (l & 63)) & 1ull);
      int b = (int)((r1[l >> 6] >> (l & 63)) & 1ull);
      double gt = 0.5 * (double)(a + b);
      if (dom) gt = fmin(gt * (1.0 + (double)dom[l]), 1.0);
      if (j == 0) g0 = gt;
      acc = acc + gt * T.alpha[t][j];
    }
    z[(int64_t)t * cap + slot] = (float)(nl > 1 ? 0.5 + acc : g0);
  }
}

// phenotype of listed offspring (slot = first + list[q]) by gathering the trait
// loci from their genome rows (used for offspring that received a remote gamete)
__global__ void k_phenotype_list(int64_t first, int64_t n, const int32_t* list, int64_t cap,
                                 int W64, const u64* G, const int32_t* grow, GnxTraitTab T,
                                 const uint8_t* dom, float* z) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int64_t slot = first + list[q];
  const u64* r0 = G + (int64_t)grow[slot] * 2 * W64;
  const u64* r1 = r0 + W64;
  for (int t = 0; t < T.n_traits; ++t) {
    const int nl = T.n_loci[t];
    double acc = 0.0, g0 = 0.0;
    for (int j = 0; j < nl; ++j) {
      int l = T.loci[t][j];
      int a = (int)((r0[l >> 6] >> (l & 63)) & 1ull);
      int b = (int)((r1[l >> 6] >> (l & 63)) & 1ull);
      double gt = 0.5 * (double)(a + b);
      if (dom) gt = fmin(gt * (1.0 + (double)dom[l]), 1.0);
      if (j == 0) g0 = gt;
      acc = acc + gt * T.alpha[t][j];
    }
    z[(int64_t)t * cap + slot] = (float)(nl > 1 ? 0.5 + acc : g0);
  }
}

int gnx_l_phenotype_list(gnx_state* h, int64_t first_slot, int64_t n, const int32_t* d_list) {
  if (n == 0 || h->cfg.n_traits == 0) return 0;
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL(k_phenotype_list, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, first_slot,
                     n, d_list, h->cfg.cap_inds, h->W64, (const u64*)h->G, s.grow,
                     gnx_trait_tab(h), h->dom, s.z);
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_l_phenotype(gnx_state* h, int64_t first_slot, int64_t n) {
  if (n == 0 || h->cfg.n_traits == 0) return 0;
  GnxSoA s = h->soa[h->cur];
  gnx_time_begin(h);
  hipLaunchKernelGGL(k_phenotype, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, first_slot, n,
                     h->cfg.cap_inds, h->W64, (const u64*)h->G, s.grow, gnx_trait_tab(h), h->dom,
                     s.z);
  gnx_time_end(h, GNX_K_PHENOTYPE, (double)n * 4.0 * h->cfg.n_traits);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- starting genomes
// _make_starting_mutations (structs/genome.py:1108-1157): per site exactly
// n_l of the 2N homologues carry a 1.  Selection sampling (Knuth 3.4.2 S) over
// homologue index q = 2*ind + hom: take iff (u_q * (2N - q)) >> 32 < n_l - taken.
// One lane per site, one wavefront per 64-site word: __ballot of the 64 lanes'
// decisions IS the u64 genome word (row of individual q/2, homologue q%2).
__global__ void __launch_bounds__(256)
k_assign_genomes(int64_t N, int L, int W64, u64* G, const int32_t* grow,
                 const int32_t* n_per_site, unsigned long long site_seed) {
  const int lane = threadIdx.x & 63;
  const int64_t word = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (word >= W64) return;          // uniform per wave
  const int64_t site = word * 64 + lane;
  const bool live = site < L;
  int remaining = live ? n_per_site[site] : 0;
  const int64_t twoN = 2 * N;
  for (int64_t q = 0; q < twoN; ++q) {
    bool take = false;
    if (remaining > 0) {
      unsigned int u = gnx_site_hash(site_seed, (u64)site, (u64)q);
      take = (int64_t)(((u64)u * (u64)(twoN - q)) >> 32) < (int64_t)remaining;
    }
    remaining -= take ? 1 : 0;
    u64 w = __ballot(take);
    if (lane == 0) G[((int64_t)grow[q >> 1] * 2 + (q & 1)) * W64 + word] = w;
  }
}

__global__ void k_assign_rows(int64_t N, int64_t cap_rows, int32_t* grow, int32_t* free_rows) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) grow[i] = (int32_t)i;
  // free stack: rows N..cap_rows-1, popped from the top (highest index first)
  if (i < cap_rows - N) free_rows[i] = (int32_t)(cap_rows - 1 - i);
}

int gnx_l_assign_genomes(gnx_state* h, const int32_t* d_n_per_site) {
  const gnx_config& c = h->cfg;
  int64_t N = h->N;
  if (N > c.cap_rows) {
    gnx_set_error("cap_rows %lld < N %lld", (long long)c.cap_rows, (long long)N);
    return 2;
  }
  GnxSoA s = h->soa[h->cur];
  int64_t m = N > c.cap_rows - N ? N : c.cap_rows - N;
  hipLaunchKernelGGL(k_assign_rows, dim3(gnx_grid(m, 256)), dim3(256), 0, h->stream, N, c.cap_rows,
                     s.grow, h->free_rows);
  h->n_free = c.cap_rows - N;
  if (N > 0 && d_n_per_site) {
    int64_t threads = (int64_t)h->W64 * 64;
    hipLaunchKernelGGL(k_assign_genomes, dim3(gnx_grid(threads, 256)), dim3(256), 0, h->stream, N,
                       c.L, h->W64, (u64*)h->G, s.grow, d_n_per_site, gnx_site_seed(c.seed));
  }
  HIPCHK(hipGetLastError());
  h->genomes_assigned = true;
  return 0;
}

// ---------------------------------------------------------------- mutation
// ops/mutation.py:62-131: set allele 1 at (locus, homologue) of the chosen
// offspring.
__global__ void k_mutate(int n, int W64, u64* G, const int32_t* grow, const int64_t* slot,
                         const int32_t* locus, const uint8_t* hom) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int l = locus[i];
  u64* w = G + ((int64_t)grow[slot[i]] * 2 + hom[i]) * W64 + (l >> 6);
  atomicOr(w, 1ull << (l & 63));
}

int gnx_l_mutate(gnx_state* h, int n, const int64_t* d_slot, const int32_t* d_locus,
                 const uint8_t* d_hom) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_mutate, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, h->W64,
                     (u64*)h->G, h->soa[h->cur].grow, d_slot, d_locus, d_hom);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- genome gather
__global__ void k_gather_genomes(int64_t n, int W16, const u64x2* G, const int32_t* grow,
                                 const int64_t* slots, u64x2* out) {
  const int64_t total = n * 2 * (int64_t)W16;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    int64_t k = g / (2 * W16);
    int64_t c = g - k * 2 * W16;
    int64_t slot = slots ? slots[k] : k;
    out[g] = G[(int64_t)grow[slot] * 2 * W16 + c];
  }
}

int gnx_l_gather_genomes(gnx_state* h, int64_t n, const int64_t* d_slots, uint64_t* d_out) {
  if (n == 0) return 0;
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_gather_genomes, dim3(gnx_grid(n * 2 * W16, 256, 256 * 32)), dim3(256), 0,
                     h->stream, n, W16, (const u64x2*)h->G, h->soa[h->cur].grow, d_slots,
                     (u64x2*)d_out);
  HIPCHK(hipGetLastError());
  return 0;
}
