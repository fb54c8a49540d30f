// The compact table of alleles at the selected loci (GnxSoA.tb): device helpers shared by
// the stand-alone kernels (gnx_kernels_genome.hip) and the fused offspring kernel
// (gnx_kernels_pop.hip).
#pragma once
#include "gnx_internal.h"

// tb of one gamete: at selected locus e the gamete copies the parent's homologue
// path_sel[key][e] XOR start (ops/mating.py:165-168 at these loci only)
__device__ __forceinline__ void gnx_gamete_tb(int TW, const uint64_t* __restrict__ tb_par,
                                              const uint64_t* __restrict__ path_sel_row,
                                              bool start, uint64_t* __restrict__ tb_out) {
  const uint64_t s = start ? ~0ull : 0ull;
  for (int w = 0; w < TW; ++w) {
    const uint64_t mm = path_sel_row[w] ^ s;
    tb_out[w] = (tb_par[w] & ~mm) | (tb_par[TW + w] & mm);
  }
}

// ops/selection.py:22-48: gt_l = (g[l,0] + g[l,1]) / 2 at the trait's loci
// (x (1 + dom_l), capped at 1, if any dominance); z = 0.5 + sum gt_l alpha_l for
// polygenic traits, z = gt_0 for monogenic ones.  f64 accumulate, f32 store.  The
// alleles come from the compact table (trait loci are its first n_tl entries,
// trait-major): t0 / t1 = the individual's two homologues there.
__device__ __forceinline__ void gnx_phenotype_tb(const uint64_t* t0, const uint64_t* t1,
                                                 const GnxTraitTab& T, const uint8_t* dom,
                                                 int64_t cap, int64_t slot, float* z) {
  int e = 0;
  for (int t = 0; t < T.n_traits; ++t) {
    const int nl = T.n_loci[t];
    double acc = 0.0, g0 = 0.0;
    for (int j = 0; j < nl; ++j, ++e) {
      const int a = (int)((t0[e >> 6] >> (e & 63)) & 1ull);
      const int b = (int)((t1[e >> 6] >> (e & 63)) & 1ull);
      double gt = 0.5 * (double)(a + b);
      if (dom) gt = fmin(gt * (1.0 + (double)dom[T.loci[t][j]]), 1.0);
      if (j == 0) g0 = gt;
      acc = acc + gt * T.alpha[t][j];
    }
    z[(int64_t)t * cap + slot] = (float)(nl > 1 ? 0.5 + acc : g0);
  }
}

// the same from registers: the individual's homologues at the selected loci as at most two
// words each (TW <= 2: up to 128 selected loci); identical arithmetic, no loads of the alleles
__device__ __forceinline__ void gnx_phenotype_words(uint64_t t0a, uint64_t t0b, uint64_t t1a,
                                                    uint64_t t1b, const GnxTraitTab& T,
                                                    const uint8_t* dom, int64_t cap, int64_t slot,
                                                    float* z) {
  int e = 0;
  for (int t = 0; t < T.n_traits; ++t) {
    const int nl = T.n_loci[t];
    double acc = 0.0, g0 = 0.0;
    for (int j = 0; j < nl; ++j, ++e) {
      const uint64_t w0 = (e >> 6) ? t0b : t0a, w1 = (e >> 6) ? t1b : t1a;
      const int a = (int)((w0 >> (e & 63)) & 1ull);
      const int b = (int)((w1 >> (e & 63)) & 1ull);
      double gt = 0.5 * (double)(a + b);
      if (dom) gt = fmin(gt * (1.0 + (double)dom[T.loci[t][j]]), 1.0);
      if (j == 0) g0 = gt;
      acc = acc + gt * T.alpha[t][j];
    }
    z[(int64_t)t * cap + slot] = (float)(nl > 1 ? 0.5 + acc : g0);
  }
}
