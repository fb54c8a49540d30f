// On-device population-genetic statistics (reference
// sim/stats.py:359-435): per-locus allele-1 and heterozygote counts (-> het,
// MAF) and pairwise linkage disequilibrium r^2, computed as popcounts over the
// bit-packed genotype matrix without ever downloading N x L/4 bytes.
#include <algorithm>
#include "gnx_internal.h"

typedef unsigned long long u64;

// One wave = one 64-locus word of the genome, lane = locus; the 16 waves of a
// block read 16 adjacent words (one 128-byte line per row and homologue).
// blockIdx.y strides over the individuals.
__global__ void __launch_bounds__(1024)
k_locus_counts(int64_t N, int W64, int L, const u64* __restrict__ G,
               const int32_t* __restrict__ grow, GnxHalves H,
               int32_t* __restrict__ cnt1, int32_t* __restrict__ cnt_het) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 16 + (threadIdx.x >> 6);
  if (w >= W64) return;                      // wave-uniform
  int c1 = 0, ch = 0;
  for (int64_t i = blockIdx.y; i < N; i += gridDim.y) {
    const int64_t row = grow[i];
    const u64 v0 = G[gnx_word_at(H, row * 2 + 0, w)];
    const u64 v1 = G[gnx_word_at(H, row * 2 + 1, w)];
    const int a = (int)((v0 >> lane) & 1ull), b = (int)((v1 >> lane) & 1ull);
    c1 += a + b;
    ch += a ^ b;
  }
  const int l = w * 64 + lane;
  if (l < L) {
    atomicAdd(&cnt1[l], c1);
    atomicAdd(&cnt_het[l], ch);
  }
}

extern "C" int gnx_stats_locus_counts(gnx_state* h, int32_t* cnt1, int32_t* cnt_het) {
  if (h->cfg.L == 0 || !h->genomes_assigned) {
    gnx_set_error("gnx_stats_locus_counts: genomes not assigned");
    return 1;
  }
  GNXCHK(gnx_xo_join(h));
  const int L = h->cfg.L;
  int32_t *d1 = nullptr, *d2 = nullptr;
  HIPCHK(hipMalloc((void**)&d1, L * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&d2, L * sizeof(int32_t)));
  HIPCHK(hipMemsetAsync(d1, 0, L * sizeof(int32_t), h->stream));
  HIPCHK(hipMemsetAsync(d2, 0, L * sizeof(int32_t), h->stream));
  int64_t N = h->N;
  if (N > 0) {
    int gy = (int)std::min<int64_t>(N, 256);
    hipLaunchKernelGGL(k_locus_counts, dim3((h->W64 + 15) / 16, gy), dim3(1024), 0, h->stream, N,
                       h->W64, L, (const u64*)h->G, h->soa[h->cur].grow, gnx_halves(h), d1, d2);
  }
  int rc = gnx_d2h(h, cnt1, d1, L * sizeof(int32_t));
  if (!rc) rc = gnx_d2h(h, cnt_het, d2, L * sizeof(int32_t));
  (void)hipFree(d1);
  (void)hipFree(d2);
  HIPCHK(hipGetLastError());
  return rc;
}

// T[j][q] = bits of the 64 homologues 64q..64q+63 at locus loci[j]
// (homologue index = 2 * individual + hom)
__global__ void k_ld_transpose(int n_loci, int64_t n_hwords, int64_t N, int W64,
                               const int32_t* __restrict__ loci, const u64* __restrict__ G,
                               const int32_t* __restrict__ grow, GnxHalves H,
                               u64* __restrict__ T) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (q >= n_hwords) return;
  const int l = loci[j];
  u64 out = 0;
  for (int b = 0; b < 64; ++b) {
    const int64_t hidx = q * 64 + b;
    if (hidx >= 2 * N) break;
    const int64_t row = grow[hidx >> 1];
    const u64 v = G[gnx_word_at(H, row * 2 + (hidx & 1), l >> 6)];
    out |= ((v >> (l & 63)) & 1ull) << b;
  }
  T[(int64_t)j * n_hwords + q] = out;
}

// r^2 between loci i < j (reference sim/stats.py:376-390): f = allele-1
// frequencies over the 2N chromosomes, f11 = frequency of 1-1 chromosomes,
// D = f11 - f_i f_j, r2 = D^2 / (f_i (1-f_i) f_j (1-f_j)); diagonal NaN.
__global__ void k_ld_pairs(int n_loci, int64_t n_hwords, double two_N, const u64* __restrict__ T,
                           double* __restrict__ r2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n_loci) return;
  if (j <= i) {
    if (j == i) r2[(int64_t)i * n_loci + j] = __longlong_as_double(0x7ff8000000000000ll);
    return;
  }
  const u64* a = T + (int64_t)i * n_hwords;
  const u64* b = T + (int64_t)j * n_hwords;
  long long ci = 0, cj = 0, cij = 0;
  for (int64_t q = 0; q < n_hwords; ++q) {
    const u64 x = a[q], y = b[q];
    ci += __popcll(x);
    cj += __popcll(y);
    cij += __popcll(x & y);
  }
  const double fi = (double)ci / two_N, fj = (double)cj / two_N, f11 = (double)cij / two_N;
  const double D = f11 - (fi * fj);
  const double v = (D * D) / (fi * (1.0 - fi) * fj * (1.0 - fj));
  r2[(int64_t)i * n_loci + j] = v;
  r2[(int64_t)j * n_loci + i] = v;
}

// the counts behind r^2 (they add over tiles): c[i] = 1-alleles at locus i over the 2N
// chromosomes, cc[i][j] = chromosomes carrying 1 at both
__global__ void k_ld_counts(int n_loci, int64_t n_hwords, const u64* __restrict__ T,
                            long long* __restrict__ c, long long* __restrict__ cc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n_loci || j < i) return;
  const u64* a = T + (int64_t)i * n_hwords;
  const u64* b = T + (int64_t)j * n_hwords;
  long long cij = 0;
  for (int64_t q = 0; q < n_hwords; ++q) cij += __popcll(a[q] & b[q]);
  cc[(int64_t)i * n_loci + j] = cij;
  cc[(int64_t)j * n_loci + i] = cij;
  if (i == j) c[i] = cij;
}

extern "C" int gnx_stats_ld_counts(gnx_state* h, int32_t n_loci, const int32_t* loci,
                                   int64_t* c, int64_t* cc) {
  if (h->cfg.L == 0 || !h->genomes_assigned) {
    gnx_set_error("gnx_stats_ld_counts: genomes not assigned");
    return 1;
  }
  if (n_loci <= 0 || n_loci > 8192) {
    gnx_set_error("gnx_stats_ld_counts: 1..8192 loci per call (the matrix is n x n)");
    return 1;
  }
  for (int j = 0; j < n_loci; ++j)
    if (loci[j] < 0 || loci[j] >= h->cfg.L) {
      gnx_set_error("gnx_stats_ld_counts: locus out of range");
      return 1;
    }
  GNXCHK(gnx_xo_join(h));
  const int64_t N = h->N;
  const int64_t n_hwords = std::max<int64_t>(1, (2 * N + 63) / 64);
  int32_t* d_loci = nullptr;
  u64* T = nullptr;
  long long *d_c = nullptr, *d_cc = nullptr;
  HIPCHK(hipMalloc((void**)&d_loci, n_loci * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&T, (size_t)n_loci * n_hwords * 8));
  HIPCHK(hipMalloc((void**)&d_c, (size_t)n_loci * 8));
  HIPCHK(hipMalloc((void**)&d_cc, (size_t)n_loci * n_loci * 8));
  GNXCHK(gnx_h2d(h, d_loci, loci, n_loci * sizeof(int32_t)));
  hipLaunchKernelGGL(k_ld_transpose, dim3(gnx_grid(n_hwords, 128), n_loci), dim3(128), 0,
                     h->stream, n_loci, n_hwords, N, h->W64, d_loci, (const u64*)h->G,
                     h->soa[h->cur].grow, gnx_halves(h), T);
  hipLaunchKernelGGL(k_ld_counts, dim3(gnx_grid(n_loci, 128), n_loci), dim3(128), 0, h->stream,
                     n_loci, n_hwords, T, d_c, d_cc);
  int rc = gnx_d2h(h, c, d_c, (size_t)n_loci * 8);
  if (!rc) rc = gnx_d2h(h, cc, d_cc, (size_t)n_loci * n_loci * 8);
  (void)hipFree(d_loci);
  (void)hipFree(T);
  (void)hipFree(d_c);
  (void)hipFree(d_cc);
  HIPCHK(hipGetLastError());
  return rc;
}

extern "C" int gnx_stats_ld(gnx_state* h, int32_t n_loci, const int32_t* loci, double* r2) {
  if (h->cfg.L == 0 || !h->genomes_assigned) {
    gnx_set_error("gnx_stats_ld: genomes not assigned");
    return 1;
  }
  if (n_loci <= 0 || n_loci > 8192) {
    gnx_set_error("gnx_stats_ld: 1..8192 loci per call (the matrix is n x n)");
    return 1;
  }
  for (int j = 0; j < n_loci; ++j)
    if (loci[j] < 0 || loci[j] >= h->cfg.L) {
      gnx_set_error("gnx_stats_ld: locus out of range");
      return 1;
    }
  GNXCHK(gnx_xo_join(h));
  const int64_t N = h->N;
  const int64_t n_hwords = std::max<int64_t>(1, (2 * N + 63) / 64);
  int32_t* d_loci = nullptr;
  u64* T = nullptr;
  double* d_r2 = nullptr;
  HIPCHK(hipMalloc((void**)&d_loci, n_loci * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&T, (size_t)n_loci * n_hwords * 8));
  HIPCHK(hipMalloc((void**)&d_r2, (size_t)n_loci * n_loci * 8));
  GNXCHK(gnx_h2d(h, d_loci, loci, n_loci * sizeof(int32_t)));
  hipLaunchKernelGGL(k_ld_transpose, dim3(gnx_grid(n_hwords, 128), n_loci), dim3(128), 0,
                     h->stream, n_loci, n_hwords, N, h->W64, d_loci, (const u64*)h->G,
                     h->soa[h->cur].grow, gnx_halves(h), T);
  hipLaunchKernelGGL(k_ld_pairs, dim3(gnx_grid(n_loci, 128), n_loci), dim3(128), 0, h->stream,
                     n_loci, n_hwords, (double)(2 * N), T, d_r2);
  int rc = gnx_d2h(h, r2, d_r2, (size_t)n_loci * n_loci * 8);
  (void)hipFree(d_loci);
  (void)hipFree(T);
  (void)hipFree(d_r2);
  HIPCHK(hipGetLastError());
  return rc;
}

// ---------------------------------------------------------------- box copy rate (bench.py)
struct alignas(16) gnx_b16 {
  unsigned long long a, b;
};

template <int U>
__global__ void __launch_bounds__(256)
k_copy16(int64_t n16, const gnx_b16* __restrict__ in, gnx_b16* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i0 < n16; i0 += stride) {
    gnx_b16 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 256;
      if (i < n16) {
        v[u].a = __builtin_nontemporal_load(&in[i].a);
        v[u].b = __builtin_nontemporal_load(&in[i].b);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 256;
      if (i < n16) {
        __builtin_nontemporal_store(v[u].a, &out[i].a);
        __builtin_nontemporal_store(v[u].b, &out[i].b);
      }
    }
  }
}

extern "C" int gnx_measure_copy(int64_t bytes, int32_t reps, double* gbps) {
  *gbps = 0.0;
  if (bytes < 4096 || reps < 1) {
    gnx_set_error("gnx_measure_copy: bytes >= 4096 and reps >= 1");
    return 1;
  }
  gnx_b16 *a = nullptr, *b = nullptr;
  HIPCHK(hipMalloc((void**)&a, (size_t)bytes));
  if (hipMalloc((void**)&b, (size_t)bytes) != hipSuccess) {
    (void)hipFree(a);
    gnx_set_error("gnx_measure_copy: out of memory");
    return 1;
  }
  const int64_t n16 = bytes / 16;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipMemsetAsync(a, 1, (size_t)bytes, nullptr));
  const int grid = 256 * 16;          // 16 workgroups per CU, 4 chunks in flight per lane
  hipLaunchKernelGGL(k_copy16<4>, dim3(grid), dim3(256), 0, nullptr, n16, a, b);
  HIPCHK(hipEventRecord(e0, nullptr));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_copy16<4>, dim3(grid), dim3(256), 0, nullptr, n16, a, b);
  HIPCHK(hipEventRecord(e1, nullptr));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  HIPCHK(hipGetLastError());
  if (ms > 0.f) *gbps = 2.0 * (double)n16 * 16.0 * reps / (ms * 1e-3) / 1e9;
  return 0;
}
