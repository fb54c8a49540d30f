// Density estimation, density-dependent death probabilities, fitness,
// mortality and stream compaction.
//
// Density (utils/spatial.py:34-146,270-360): the union of the reference's four
// offset window grids is a regular lattice of spacing hww = ww/2; node j's
// window covers the two half-window bins j-1 and j.  So: count individuals into
// half-window bins (LDS histograms -> per-block partials, no global atomics),
// sum 2x2 bins per node, divide by the window area inside the landscape, then
// interpolate.  The reference interpolates with scipy's Clough-Tocher
// griddata(method='cubic'), which is not reproducible off qhull; the build uses
// the natural bicubic spline through the same nodes (tolerance: DESIGN.md).
// All density arithmetic is f64 so it matches the numpy oracle to ~1e-12.
#include "gnx_internal.h"
#include <chrono>
#include "gnx_rng.h"
#include "gnx_compact.h"
#include "gnx_half.h"
#include "gnx_xo.h"

#define BIN_BLOCKS 512

// ---------------------------------------------------------------- bins
__global__ void __launch_bounds__(256)
k_bins(int64_t n, const int32_t* n_dev, const float* x, const float* y, const uint8_t* ghost,
       double inv_hww, int nbx, int nby, int32_t* partials, GnxSetWords sw) {
  extern __shared__ int32_t lds_hist[];
  // (tiles: the four counter words behind the two density fields, on their way to the all-reduce)
  if (sw.dst && blockIdx.x == 0 && threadIdx.x < 4) sw.dst[threadIdx.x] = sw.v[threadIdx.x];
  // n_dev: the count is still on its way to the host (the compaction's scan kernel left it
  // in device memory too); the grid was sized for an upper bound
  if (n_dev) n = *n_dev;
  const int nb = nbx * nby;
  for (int k = threadIdx.x; k < nb; k += blockDim.x) lds_hist[k] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (ghost && ghost[i]) continue;
    int hx = min(nbx - 1, (int)floor((double)x[i] * inv_hww));
    int hy = min(nby - 1, (int)floor((double)y[i] * inv_hww));
    atomicAdd(&lds_hist[hy * nbx + hx], 1);
  }
  __syncthreads();
  // flush: one integer atomic per non-empty bin per block (counts are exact and
  // order-independent, so the result is deterministic)
  for (int k = threadIdx.x; k < nb; k += blockDim.x) {
    int v = lds_hist[k];
    if (v) atomicAdd(&partials[k], v);
  }
}

__global__ void k_set_words(GnxSetWords sw) {
  if (blockIdx.x == 0 && threadIdx.x < 4) sw.dst[threadIdx.x] = sw.v[threadIdx.x];
}

// variant for lattices too large for LDS: global atomics into partials[0]
__global__ void k_bins_global(int64_t n, const float* x, const float* y, const uint8_t* ghost,
                              double inv_hww, int nbx, int nby, int32_t* hist) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (ghost && ghost[i]) return;
  int hx = min(nbx - 1, (int)floor((double)x[i] * inv_hww));
  int hy = min(nby - 1, (int)floor((double)y[i] * inv_hww));
  atomicAdd(&hist[hy * nbx + hx], 1);
}

// node value = (sum of the 2x2 half-window bins around the node) / area
__global__ void k_nodes(int Jx, int Jy, int nbx, int n_part, const int32_t* partials,
                        const double* areas, double* V) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Jx * Jy) return;
  int i = idx / Jx, j = idx - i * Jx;
  const int nb = nbx * Jy;     // nby == Jy
  long long cnt = 0;
  for (int di = -1; di <= 0; ++di)
    for (int dj = -1; dj <= 0; ++dj) {
      int ii = i + di, jj = j + dj;
      if (ii < 0 || jj < 0) continue;
      int b = ii * nbx + jj;
      for (int p = 0; p < n_part; ++p) cnt += partials[(int64_t)p * nb + b];
    }
  V[idx] = (double)cnt / areas[idx];
}

// natural-cubic-spline second derivatives along one axis of a [Jy][Jx] field.
// One thread per line; Thomas algorithm with host-precomputed factors cp[].
// along_x: lines are rows (stride 1 inside a line); else columns (stride Jx).
__global__ void k_spline_m(int Jx, int Jy, int along_x, double h, const double* cp,
                           const double* V, double* M) {
  int line = blockIdx.x * blockDim.x + threadIdx.x;
  int n_lines = along_x ? Jy : Jx;
  if (line >= n_lines) return;
  int J = along_x ? Jx : Jy;
  int64_t stride = along_x ? 1 : Jx;
  int64_t base = along_x ? (int64_t)line * Jx : line;
  M[base] = 0.0;
  M[base + (int64_t)(J - 1) * stride] = 0.0;
  if (J <= 2) return;
  const double s = 6.0 / (h * h);
  // forward sweep on the (J-2) interior unknowns of  m[k-1] + 4 m[k] + m[k+1] = d[k];
  // cp[k] = c'_k, and d'_k is stored in M
  double dprev = 0.0;
  for (int k = 1; k <= J - 2; ++k) {
    double d = s * (V[base + (k - 1) * stride] - 2.0 * V[base + k * stride] +
                    V[base + (k + 1) * stride]);
    // (cp[k] = 1 / (4 - cp[k - 1]) is this row's pivot's reciprocal too, cp[1] = 1/4: a multiply
    // on the dependent chain where a f64 division - some 30 dependent instructions - was)
    double dp = (d - dprev) * cp[k];
    M[base + k * stride] = dp;
    dprev = dp;
  }
  // (the value above rides in a register: read back from LDS every step of the chain waited
  // for the step before's store)
  double mnext = dprev;                          // = M[J - 2]
  for (int k = J - 3; k >= 1; --k) {
    const double m = M[base + k * stride] - cp[k] * mnext;
    M[base + k * stride] = m;
    mnext = m;
  }
}

struct SplineC {
  const double* V;
  const double* Mx;
  const double* My;
  const double* Mxy;
  int Jx, Jy;
  double hww, inv_hww;
};

__device__ __forceinline__ double spl1(double v0, double v1, double m0, double m1, double t,
                                       double h2_6) {
  double a = 1.0 - t;
  return a * v0 + t * v1 + ((a * a * a - a) * m0 + (t * t * t - t) * m1) * h2_6;
}

__device__ __forceinline__ double spline_eval(const SplineC& S, double px, double py) {
  double fx = px * S.inv_hww, fy = py * S.inv_hww;
  int j = min(max((int)floor(fx), 0), S.Jx - 2);
  int i = min(max((int)floor(fy), 0), S.Jy - 2);
  double tx = fx - j, ty = fy - i;
  double h2_6 = S.hww * S.hww / 6.0;
  int64_t a = (int64_t)i * S.Jx + j, b = a + S.Jx;
  double v0 = spl1(S.V[a], S.V[a + 1], S.Mx[a], S.Mx[a + 1], tx, h2_6);
  double m0 = spl1(S.My[a], S.My[a + 1], S.Mxy[a], S.Mxy[a + 1], tx, h2_6);
  double v1 = spl1(S.V[b], S.V[b + 1], S.Mx[b], S.Mx[b + 1], tx, h2_6);
  double m1 = spl1(S.My[b], S.My[b + 1], S.Mxy[b], S.Mxy[b + 1], tx, h2_6);
  return spl1(v0, v1, m0, m1, ty, h2_6);
}

static SplineC make_splinec(const gnx_state* h, const GnxSpline& s) {
  SplineC c;
  int64_t n = (int64_t)h->lat.Jx * h->lat.Jy;
  c.V = s.c;
  c.Mx = s.c + n;
  c.My = s.c + 2 * n;
  c.Mxy = s.c + 3 * n;
  c.Jx = h->lat.Jx;
  c.Jy = h->lat.Jy;
  c.hww = h->lat.hww;
  c.inv_hww = 1.0 / h->lat.hww;
  return c;
}

// counts n points (ghosts skipped) into the half-window bins d_bins [nby*nbx]
int gnx_l_bins(gnx_state* h, int64_t n, const float* d_x, const float* d_y, const uint8_t* d_ghost,
               int32_t* d_bins, const int32_t* n_dev, const GnxSetWords* sw) {
  const GnxLattice& L = h->lat;
  const int nb = L.nbx * L.nby;
  const int field = (d_bins == h->bins_P) ? 1 : 0;
  if (!h->bins_zeroed[field])
    HIPCHK(hipMemsetAsync(d_bins, 0, (size_t)nb * sizeof(int32_t), h->stream));
  h->bins_zeroed[field] = false;
  if (n > 0) {
    if ((size_t)nb * sizeof(int32_t) <= 48 * 1024) {
      int blocks = (int)std::min<int64_t>(BIN_BLOCKS, std::max<int64_t>(1, (n + 255) / 256));
      hipLaunchKernelGGL(k_bins, dim3(blocks), dim3(256), (size_t)nb * sizeof(int32_t), h->stream,
                         n, n_dev, d_x, d_y, d_ghost, 1.0 / L.hww, L.nbx, L.nby, d_bins,
                         sw ? *sw : GnxSetWords{});
      sw = nullptr;
    } else {
      hipLaunchKernelGGL(k_bins_global, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, d_x,
                         d_y, d_ghost, 1.0 / L.hww, L.nbx, L.nby, d_bins);
    }
  }
  if (sw)                      // (nobody to count, or a lattice too fine for the LDS kernel)
    hipLaunchKernelGGL(k_set_words, dim3(1), dim3(64), 0, h->stream, *sw);
  HIPCHK(hipGetLastError());
  return 0;
}

// One workgroup does the whole lattice of one density field: node values from the bins,
// then the three families of natural-spline second derivatives (rows of V, columns of V,
// rows of My), with workgroup barriers in between - four launches of a few microseconds
// of work each otherwise (the lattice is ~(2 dim / ww)^2 = 41^2 nodes by default).
__device__ __forceinline__ void spline_line(int J, int64_t base, int64_t stride, double h,
                                            const double* cp, const double* V, double* M) {
  M[base] = 0.0;
  M[base + (int64_t)(J - 1) * stride] = 0.0;
  if (J <= 2) return;
  const double s = 6.0 / (h * h);
  double dprev = 0.0;
  for (int k = 1; k <= J - 2; ++k) {
    double d = s * (V[base + (k - 1) * stride] - 2.0 * V[base + k * stride] +
                    V[base + (k + 1) * stride]);
    // (cp[k] = 1 / (4 - cp[k - 1]) is this row's pivot's reciprocal too, cp[1] = 1/4: a multiply
    // on the dependent chain where a f64 division - some 30 dependent instructions - was)
    double dp = (d - dprev) * cp[k];
    M[base + k * stride] = dp;
    dprev = dp;
  }
  // (the value above rides in a register: read back from LDS every step of the chain waited
  // for the step before's store)
  double mnext = dprev;                          // = M[J - 2]
  for (int k = J - 3; k >= 1; --k) {
    const double m = M[base + k * stride] - cp[k] * mnext;
    M[base + k * stride] = m;
    mnext = m;
  }
}

// rows of V -> Mx on the first half of the workgroup, columns of V -> My on the second, at
// the same time (one after the other on the same threads: twice the serial sweeps)
__device__ __forceinline__ void spline_rows_and_cols(int Jx, int Jy, double hww, const double* cp,
                                                     const double* V, double* Mx, double* My) {
  const int half = blockDim.x / 2;
  if ((int)threadIdx.x < half) {
    for (int line = threadIdx.x; line < Jy; line += half)
      spline_line(Jx, (int64_t)line * Jx, 1, hww, cp, V, Mx);
  } else {
    for (int line = threadIdx.x - half; line < Jx; line += half)
      spline_line(Jy, line, Jx, hww, cp, V, My);
  }
}


// use_lds: the four coefficient planes and the Thomas factors live in LDS while the
// (dependent, one line per thread) sweeps run - a global round trip per element of a
// 43-long chain is what the kernel's time was - and are written out once at the end.
__global__ void __launch_bounds__(256)
k_lattice(int Jx, int Jy, int nbx, const int32_t* bins, const double* areas, double hww,
          const double* cp_g, double* C, int32_t* zero_bins, int n_zero,
          unsigned long long* zero_word, int use_lds, int zero_own) {
  extern __shared__ double lat_lds[];
  const int nn = Jx * Jy;
  // housekeeping that would otherwise be launches of their own: clear the OTHER density
  // field's bins for their next use, and the N.max() accumulator
  for (int k = threadIdx.x; k < n_zero; k += blockDim.x) zero_bins[k] = 0;
  if (zero_word && threadIdx.x == 0) *zero_word = 0ull;
  double* W = use_lds ? lat_lds : C;
  double* V = W;
  double* Mx = W + nn;
  double* My = W + 2 * (int64_t)nn;
  double* Mxy = W + 3 * (int64_t)nn;
  const double* cp = cp_g;
  if (use_lds) {
    double* cpl = lat_lds + 4 * (int64_t)nn;
    const int Jm = max(Jx, Jy);
    for (int k = threadIdx.x; k <= Jm; k += blockDim.x) cpl[k] = cp_g[k];
    cp = cpl;
    if (!bins)
      for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) V[idx] = C[idx];
  }
  if (bins) {
    for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
      const int i = idx / Jx, j = idx - i * Jx;
      long long cnt = 0;
      for (int di = -1; di <= 0; ++di)
        for (int dj = -1; dj <= 0; ++dj) {
          const int ii = i + di, jj = j + dj;
          if (ii >= 0 && jj >= 0) cnt += bins[ii * nbx + jj];
        }
      V[idx] = (double)cnt / areas[idx];
    }
  }
  __syncthreads();
  // (the bins counted by the step's own kernels, gnx_bins.h: read, now cleared for their
  // next use - this is the only workgroup that touches them)
  if (bins && zero_own)
    for (int k = threadIdx.x; k < nn; k += blockDim.x) const_cast<int32_t*>(bins)[k] = 0;
  spline_rows_and_cols(Jx, Jy, hww, cp, V, Mx, My);
  __syncthreads();
  for (int line = threadIdx.x; line < Jy; line += blockDim.x)
    spline_line(Jx, (int64_t)line * Jx, 1, hww, cp, My, Mxy);
  if (use_lds) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * nn; idx += blockDim.x) C[idx] = lat_lds[idx];
  }
}

// node densities (from bins, or given directly) and the spline coefficients
int gnx_l_spline(gnx_state* h, const int32_t* d_bins, GnxSpline* spl,
                 const double* d_nodes_override) {
  return gnx_l_spline_z(h, d_bins, spl, d_nodes_override, false);
}

int gnx_l_spline_z(gnx_state* h, const int32_t* d_bins, GnxSpline* spl,
                   const double* d_nodes_override, bool housekeeping) {
  const GnxLattice& L = h->lat;
  const int64_t nn = (int64_t)L.Jx * L.Jy;
  double* V = spl->c;
  if (d_nodes_override)
    HIPCHK(hipMemcpyAsync(V, d_nodes_override, nn * sizeof(double), hipMemcpyDeviceToDevice,
                          h->stream));
  if (nn <= 65536) {
    // single-GPU step: the lattice kernel of one field clears the other field's bins
    int32_t* zb = nullptr;
    int other = -1;
    if (housekeeping && !d_nodes_override) {
      other = (d_bins == h->bins_P) ? 0 : 1;
      zb = other ? h->bins_P : h->bin_partials;
    }
    const size_t lds_bytes = ((size_t)4 * nn + std::max(L.Jx, L.Jy) + 1) * sizeof(double);
    const int use_lds = lds_bytes <= 60 * 1024 ? 1 : 0;
    hipLaunchKernelGGL(k_lattice, dim3(1), dim3(256), use_lds ? lds_bytes : 0, h->stream, L.Jx,
                       L.Jy, L.nbx, d_nodes_override ? nullptr : d_bins, L.areas, L.hww, L.cprime,
                       V, zb, zb ? L.nbx * L.nby : 0,
                       (housekeeping && spl == &h->spl_N) ? h->nmax_bits : nullptr, use_lds, 0);
    if (zb) h->bins_zeroed[other] = true;
    if (housekeeping && spl == &h->spl_N) h->nmax_zeroed = true;
    HIPCHK(hipGetLastError());
    spl->valid = true;
    return 0;
  }
  // very fine lattices (tiny density_grid_window_width): one launch per stage
  if (!d_nodes_override)
    hipLaunchKernelGGL(k_nodes, dim3(gnx_grid(nn, 128)), dim3(128), 0, h->stream, L.Jx, L.Jy, L.nbx,
                       1, d_bins, L.areas, V);
  double* Mx = V + nn;
  double* My = V + 2 * nn;
  double* Mxy = V + 3 * nn;
  hipLaunchKernelGGL(k_spline_m, dim3(gnx_grid(L.Jy, 64)), dim3(64), 0, h->stream, L.Jx, L.Jy, 1,
                     L.hww, L.cprime, V, Mx);
  hipLaunchKernelGGL(k_spline_m, dim3(gnx_grid(L.Jx, 64)), dim3(64), 0, h->stream, L.Jx, L.Jy, 0,
                     L.hww, L.cprime, V, My);
  hipLaunchKernelGGL(k_spline_m, dim3(gnx_grid(L.Jy, 64)), dim3(64), 0, h->stream, L.Jx, L.Jy, 1,
                     L.hww, L.cprime, My, Mxy);
  HIPCHK(hipGetLastError());
  spl->valid = true;
  return 0;
}

int gnx_l_density(gnx_state* h, int64_t n, const float* d_x, const float* d_y, GnxSpline* spl,
                  const double* d_nodes_override, const int32_t* n_dev) {
  gnx_time_begin(h);
  int32_t* bins = (spl == &h->spl_P) ? h->bins_P : h->bin_partials;
  if (!d_nodes_override) GNXCHK(gnx_l_bins(h, n, d_x, d_y, nullptr, bins, n_dev));
  GNXCHK(gnx_l_spline_z(h, bins, spl, d_nodes_override, !d_nodes_override));
  gnx_time_end(h, GNX_K_DENSITY, (double)n * 8.0);
  return 0;
}

// ---------------------------------------------------------------- rasters
// max over all cells of N (needed by _calc_dNdt's clip, ops/demography.py:116).
// Row by row: along a raster row the lattice row i and the fraction ty are fixed, so the
// bicubic spline collapses to a 1-d spline along x whose node values A[j] and second
// derivatives Bm[j] are the y-interpolants of (V, My) and (Mx, Mxy) at column j (the
// interpolation operators commute: the same polynomial as spline_eval, rounded in another
// order).  Between two nodes that 1-d spline is ONE cubic in tx, and the largest of its samples
// at the cell centres is the first one, the last one, or one next to a root of its derivative
// (a quadratic): a handful of evaluations per (row, node interval) instead of one per cell -
// 2048 x 22 x <= 10 against 2048 x 2048 at the metric workload - of the same expression at
// the same tx, so the maximum is the one a scan of every cell finds (short intervals and
// non-finite coefficients are scanned).  Every workgroup of both kernels below calls this
// with the same arithmetic, so N.max() does not depend on which of them computed it.
// LDS: AB [R][2 * Jx] doubles (R rows at a time), seg [Jx] ints.
#define GNX_NMAX_R 8                 // raster rows per batch, at most
#define GNX_NMAX_SCAN 12             // node intervals of fewer cells than this are scanned

__host__ __device__ __forceinline__ int gnx_nmax_seg_ints(int Jx) { return 2 * ((Jx + 2) / 2); }

// rows per batch such that `base_doubles` + the N.max() workspace fit 64 KB (>= 1)
static int gnx_nmax_rows_fit(int Jx, size_t base_doubles) {
  int R = GNX_NMAX_R;
  while (R > 1 && (base_doubles + (size_t)R * 2 * Jx + 256 + gnx_nmax_seg_ints(Jx) / 2) * sizeof(double) >
                      64 * 1024)
    R >>= 1;
  return R;
}
static size_t gnx_nmax_lds_doubles(int Jx, int R) {
  return (size_t)R * 2 * Jx + 256 + gnx_nmax_seg_ints(Jx) / 2;
}
// GNX_NMAX_SCAN_ALL=1 (read per call; tests): every cell of every node interval is evaluated
static int gnx_nmax_scan_min() {
  const char* e = getenv("GNX_NMAX_SCAN_ALL");
  return e && atoi(e) ? 0x7fffffff : GNX_NMAX_SCAN;
}

__device__ __forceinline__ int nmax_seg_of(int cx, double inv_hww, int Jx) {
  return min(max((int)floor((cx + 0.5) * inv_hww), 0), Jx - 2);
}

__device__ __forceinline__ double nmax_rows(const double* __restrict__ C, int Jx, int Jy,
                                            double hww, int W, int H, int row0, int row_stride,
                                            double* AB, int* seg, int R, int scan_min) {
  const int nn = Jx * Jy;
  const double* V = C;
  const double* Mx = C + nn;
  const double* My = C + 2 * (int64_t)nn;
  const double* Mxy = C + 3 * (int64_t)nn;
  const double inv_hww = 1.0 / hww, h2_6 = hww * hww / 6.0;
  // seg[j]: the first cell of node interval j (the cells whose centre spline_eval puts there)
  for (int j = threadIdx.x; j < Jx; j += blockDim.x) {
    int g = j >= Jx - 1 ? W : 0;
    if (j > 0 && j < Jx - 1) {
      g = min(max((int)ceil(j * hww - 0.5), 0), W);
      while (g > 0 && nmax_seg_of(g - 1, inv_hww, Jx) >= j) --g;
      while (g < W && nmax_seg_of(g, inv_hww, Jx) < j) ++g;
    }
    seg[j] = g;
  }
  double m = 0.0;
  for (int base = row0; base < H; base += R * row_stride) {
    __syncthreads();
    for (int item = threadIdx.x; item < R * Jx; item += blockDim.x) {
      const int r = item / Jx, j = item - r * Jx;
      const int cy = base + r * row_stride;
      if (cy >= H) continue;
      const double fy = (cy + 0.5) * inv_hww;
      const int i = min(max((int)floor(fy), 0), Jy - 2);
      const double ty = fy - i;
      const int a = i * Jx + j, b = a + Jx;
      AB[r * 2 * Jx + j] = spl1(V[a], V[b], My[a], My[b], ty, h2_6);
      AB[r * 2 * Jx + Jx + j] = spl1(Mx[a], Mx[b], Mxy[a], Mxy[b], ty, h2_6);
    }
    __syncthreads();
    for (int item = threadIdx.x; item < R * (Jx - 1); item += blockDim.x) {
      const int r = item / (Jx - 1), j = item - r * (Jx - 1);
      if (base + r * row_stride >= H) continue;
      const int lo = seg[j], hi = seg[j + 1] - 1;
      if (lo > hi) continue;
      const double* ab = AB + r * 2 * Jx;
      const double y0 = ab[j], y1 = ab[j + 1], m0 = ab[Jx + j], m1 = ab[Jx + j + 1];
      auto at = [&](int cx) {
        const double tx = (cx + 0.5) * inv_hww - j;
        return spl1(y0, y1, m0, m1, tx, h2_6);
      };
      // d/dt of spl1 = qa t^2 + qb t + qc
      const double qa = 3.0 * h2_6 * (m1 - m0), qb = 6.0 * h2_6 * m0;
      const double qc = (y1 - y0) - h2_6 * (2.0 * m0 + m1);
      const double disc = qb * qb - 4.0 * qa * qc;
      if (hi - lo < scan_min || !isfinite(disc)) {
        for (int cx = lo; cx <= hi; ++cx) m = fmax(m, at(cx));
        continue;
      }
      m = fmax(m, fmax(at(lo), at(hi)));
      if (disc >= 0.0) {
        const double q = -0.5 * (qb + copysign(sqrt(disc), qb));
        const double root[2] = {q / qa, qc / q};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          // (a NaN or an infinity compares false / clamps to an end: nothing new to look at)
          const double cxr = (root[k] + j) * hww - 0.5;
          if (!(cxr > lo - 2.0 && cxr < hi + 2.0)) continue;
          const int k0 = (int)floor(cxr);
          for (int cx = max(k0 - 1, lo); cx <= min(k0 + 2, hi); ++cx) m = fmax(m, at(cx));
        }
      }
    }
  }
  return m;
}

__device__ __forceinline__ void nmax_publish(double m, unsigned long long* out_bits, double* red) {
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  // non-negative doubles order like their bit patterns
  if (threadIdx.x == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(red[0]));
}

__global__ void __launch_bounds__(256)
k_nmax(SplineC S, int W, int H, unsigned long long* out_bits, int R, int scan_min) {
  extern __shared__ double nmax_lds[];          // AB [R][2 * Jx] + red [256] + seg [Jx]
  double* red = nmax_lds + (size_t)R * 2 * S.Jx;
  const double m = nmax_rows(S.V, S.Jx, S.Jy, S.hww, W, H, blockIdx.x, gridDim.x, nmax_lds,
                             reinterpret_cast<int*>(red + 256), R, scan_min);
  nmax_publish(m, out_bits, red);
}

// Lattice + N.max() of the individuals' density in ONE launch (one GPU, bins counted by the
// step's kernels): every workgroup builds the whole lattice in its LDS (a few microseconds
// of serial sweeps whichever way - redundantly, but in parallel), workgroup 0 writes the
// coefficients out for the death probabilities, and all of them share the raster rows of
// N.max().  The bins these workgroups read cannot be cleared here; the OTHER step's buffer
// and N.max() word are (fb / nmax2 alternate).
__global__ void __launch_bounds__(256)
k_lattice_nmax(int Jx, int Jy, int nbx, const int32_t* __restrict__ bins,
               const double* __restrict__ areas, double hww, const double* __restrict__ cp_g,
               double* __restrict__ C, int W, int H, unsigned long long* __restrict__ out_bits,
               int32_t* __restrict__ zero_bins, unsigned long long* __restrict__ zero_word,
               int32_t* __restrict__ binsP, double* __restrict__ CP, GnxPubWords pub, int R, int scan_min) {
  extern __shared__ double lat_lds[];
  const int nn = Jx * Jy;
  const int Jm = max(Jx, Jy);
  // binsP (device-driven step): one workgroup more than the individuals' field needs builds
  // the PAIRS' lattice from binsP meanwhile (a launch of one workgroup otherwise, 11 us on
  // the step's chain), writes it to CP and clears binsP for their next use
  const bool isP = binsP != nullptr && blockIdx.x == gridDim.x - 1;
  const int nbN = binsP != nullptr ? (int)gridDim.x - 1 : (int)gridDim.x;
  if (isP) bins = binsP;
  double* V = lat_lds;
  double* Mx = V + nn;
  double* My = V + 2 * nn;
  double* Mxy = V + 3 * nn;
  double* cp = lat_lds + 4 * nn;                // [Jm + 1]
  double* AB = cp + Jm + 1;                     // [R][2 * Jx]
  double* red = AB + (size_t)R * 2 * Jx;        // [256]
  int* seg = reinterpret_cast<int*>(red + 256); // [Jx]
  if (blockIdx.x == 0) {
    if (zero_bins)
      for (int k = threadIdx.x; k < nn; k += blockDim.x) zero_bins[k] = 0;
    if (zero_word && threadIdx.x == 0) *zero_word = 0ull;
    // (tiles: the all-reduced counter words and the deferred checks go to the host from here;
    // the mortality's wait covers them)
    if ((int)threadIdx.x < pub.n1)
      __hip_atomic_store(&pub.host1[threadIdx.x], pub.src1[threadIdx.x], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < pub.n2)
      __hip_atomic_store(&pub.host2[threadIdx.x], pub.src2[threadIdx.x], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  for (int k = threadIdx.x; k <= Jm; k += blockDim.x) cp[k] = cp_g[k];
  for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
    const int i = idx / Jx, j = idx - i * Jx;
    long long cnt = 0;
    for (int di = -1; di <= 0; ++di)
      for (int dj = -1; dj <= 0; ++dj) {
        const int ii = i + di, jj = j + dj;
        if (ii >= 0 && jj >= 0) cnt += bins[ii * nbx + jj];
      }
    V[idx] = (double)cnt / areas[idx];
  }
  __syncthreads();
  spline_rows_and_cols(Jx, Jy, hww, cp, V, Mx, My);
  __syncthreads();
  for (int line = threadIdx.x; line < Jy; line += blockDim.x)
    spline_line(Jx, (int64_t)line * Jx, 1, hww, cp, My, Mxy);
  __syncthreads();
  if (isP) {
    for (int idx = threadIdx.x; idx < 4 * nn; idx += blockDim.x) CP[idx] = lat_lds[idx];
    for (int k = threadIdx.x; k < nn; k += blockDim.x) binsP[k] = 0;
    return;
  }
  if (blockIdx.x == 0)
    for (int idx = threadIdx.x; idx < 4 * nn; idx += blockDim.x) C[idx] = lat_lds[idx];
  const double m = nmax_rows(lat_lds, Jx, Jy, hww, W, H, blockIdx.x, nbN, AB, seg, R, scan_min);
  nmax_publish(m, out_bits, red);
}

struct DemP {
  double R, b, lam, d_min, d_max, K_factor;
  const double* K_over;    // explicit K raster (demographic change events) or null
  int K_layer, W, H;
  int have_pairs;
};

// carrying capacity of cell (cx, cy): the explicit raster if one was set
// (Species.K after a demographic change, ops/change.py:633-651), else
// rast[K_layer] * K_factor (structs/species.py:546)
__device__ __forceinline__ double k_at_cell(const DemP& P, const float* rast, int cx, int cy) {
  if (P.K_over) return P.K_over[(int64_t)cy * P.W + cx];
  return (double)rast[((int64_t)P.K_layer * P.H + cy) * P.W + cx] * P.K_factor;
}

// d at one cell (ops/demography.py:95-172, in the reference's order of
// operations): dNdt = R(1-N/K)N clipped to >= -Nmax, NaN/inf -> -Nmax;
// N_b = b*lambda*n_pairs; N_d = N_b - dNdt; d = N_d/N, NaN -> 0, clip.
__device__ __forceinline__ double d_at_cell(const DemP& P, double N, double npairs, double K,
                                            double nmax) {
  double dNdt = P.R * (1.0 - (N / K)) * N;
  dNdt = fmax(dNdt, -nmax);                    // np.clip(a_min=-N.max()); NaN propagates
  if (isnan(dNdt) || isinf(dNdt)) dNdt = -nmax;
  double N_b = P.b * P.lam * npairs;
  double N_d = N_b - dNdt;
  double d = N_d / N;
  if (isnan(d)) d = 0.0;
  return fmin(fmax(d, P.d_min), P.d_max);
}

__device__ __forceinline__ double clip_fmax_nan(double v, double lo) {
  // np.clip propagates NaN; fmax would drop it
  return isnan(v) ? v : fmax(v, lo);
}

__global__ void k_raster(int which, SplineC SN, SplineC SP, DemP P, const float* rast,
                         const unsigned long long* nmax_bits, double* out) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (int64_t)P.W * P.H) return;
  int cy = (int)(c / P.W), cx = (int)(c - (int64_t)cy * P.W);
  double px = cx + 0.5, py = cy + 0.5;
  double K = k_at_cell(P, rast, cx, cy);
  if (which == GNX_R_K) {
    out[c] = K;
    return;
  }
  double N = fmax(spline_eval(SN, px, py), 0.0);
  if (which == GNX_R_N) {
    out[c] = N;
    return;
  }
  double np_ = P.have_pairs ? fmax(spline_eval(SP, px, py), 0.0) : 0.0;
  if (which == GNX_R_NPAIRS) {
    out[c] = np_;
    return;
  }
  double nmax = __longlong_as_double((long long)*nmax_bits);
  out[c] = d_at_cell(P, N, np_, K, nmax);
}

static DemP make_demp(const gnx_state* h) {
  DemP P;
  P.R = h->sp.R;
  P.b = h->sp.b;
  P.lam = h->sp.n_births_lambda;
  P.d_min = h->sp.d_min;
  P.d_max = h->sp.d_max;
  P.K_factor = h->sp.K_factor;
  P.K_over = h->K_over;
  P.K_layer = h->sp.K_layer;
  P.W = h->cfg.W;
  P.H = h->cfg.H;
  P.have_pairs = h->spl_P.valid ? 1 : 0;
  return P;
}

int gnx_l_raster(gnx_state* h, int which, double* d_out) {
  if (which != GNX_R_K && !h->spl_N.valid) {
    gnx_set_error("density rasters are available after the first pop_dynamics call");
    return 3;
  }
  GNXCHK(gnx_wait_latP(h));
  SplineC SN = make_splinec(h, h->spl_N), SP = make_splinec(h, h->spl_P);
  int64_t cells = (int64_t)h->cfg.W * h->cfg.H;
  hipLaunchKernelGGL(k_raster, dim3(gnx_grid(cells, 256)), dim3(256), 0, h->stream, which, SN, SP,
                     make_demp(h), h->rast, h->nmax_cur ? h->nmax_cur : h->nmax_bits, d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- death probabilities
struct DeathP {
  int64_t N, cap;
  int n_layers, with_selection, max_age, n_delet, n_tl, TW;
  const GnxDD* dd;
};

// ops/demography.py:305-321 + ops/selection.py:51-125: d at the individual's
// cell; w = clip(prod_t 1 - phi_t |e^(not univ_adv) - z_t|^gamma_t, >= 0.001)
// x prod_del (1 - s_l (g_l0 + g_l1)); p = 1 - (1 - d) w; age > max_age => 1.
__device__ __forceinline__ double death_prob_one(const DeathP& Q, const DemP& P, const SplineC& SN,
                                                 const SplineC& SP, const GnxSoA& s,
                                                 const float* __restrict__ rast,
                                                 const GnxTraitTab& T,
                                                 const double* __restrict__ delet_s, double nmax,
                                                 int64_t i, double* __restrict__ d_cell) {
  int cx = (int)s.x[i], cy = (int)s.y[i];
  double px = cx + 0.5, py = cy + 0.5;
  double K = k_at_cell(P, rast, cx, cy);
  double N = fmax(spline_eval(SN, px, py), 0.0);
  double np_ = P.have_pairs ? fmax(spline_eval(SP, px, py), 0.0) : 0.0;
  double d = d_at_cell(P, N, np_, K, nmax);
  d_cell[i] = d;
  double p = d;
  if (Q.with_selection) {
    double w = 1.0;
    if (T.n_traits > 0) {
      for (int t = 0; t < T.n_traits; ++t) {
        double e = (double)s.e[(int64_t)T.layer[t] * Q.cap + i];
        if (T.univ_adv[t]) e = 1.0;                    // e ** 0
        double z = (double)s.z[(int64_t)t * Q.cap + i];
        double phi = T.phi_rast[t] ? (double)T.phi_rast[t][(int64_t)cy * P.W + cx] : T.phi[t];
        // (gamma = 1, the parameters-file default: x ** 1 is x, without the library call)
        const double dz = fabs(e - z);
        w *= 1.0 - phi * (T.gamma[t] == 1.0 ? dz : pow(dz, T.gamma[t]));
      }
      w = fmax(w, 0.001);
    }
    if (Q.n_delet > 0) {
      // deleterious loci follow the trait loci in the compact allele table
      const uint64_t* t0 = s.tb + (i * 2 + 0) * Q.TW;
      const uint64_t* t1 = t0 + Q.TW;
      for (int k = 0; k < Q.n_delet; ++k) {
        const int e = Q.n_tl + k;
        int cnt = (int)((t0[e >> 6] >> (e & 63)) & 1ull) + (int)((t1[e >> 6] >> (e & 63)) & 1ull);
        w *= 1.0 - (double)cnt * delet_s[k];
      }
    }
    s.fit[i] = (float)w;
    p = 1.0 - (1.0 - d) * w;
  }
  if (Q.max_age >= 0 && s.age[i] > Q.max_age) p = 1.0;
  return p;
}

__global__ void __launch_bounds__(256)
k_death_probs(DeathP Q, DemP P, SplineC SN, SplineC SP, GnxSoA s, const float* rast,
              GnxTraitTab T, const double* delet_s, const unsigned long long* nmax_bits,
              double* p_death, double* d_cell) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= gnx_dd_nb(Q.dd, Q.N)) return;
  const double nmax = __longlong_as_double((long long)*nmax_bits);
  p_death[i] = death_prob_one(Q, P, SN, SP, s, rast, T, delet_s, nmax, i, d_cell);
}

static DeathP make_deathp(const gnx_state* h, bool with_selection) {
  DeathP Q;
  Q.dd = nullptr;
  Q.N = h->N;
  Q.cap = h->cfg.cap_inds;
  Q.n_layers = h->cfg.n_layers;
  Q.with_selection = with_selection ? 1 : 0;
  Q.max_age = h->sp.max_age;
  Q.n_delet = (with_selection && h->genomes_assigned) ? h->n_delet : 0;
  Q.n_tl = h->n_tl;
  Q.TW = h->TW;
  return Q;
}

// N.max() over the cells (the clip of _calc_dNdt, ops/demography.py:116)
static int launch_nmax(gnx_state* h) {
  if (!h->nmax_zeroed)
    HIPCHK(hipMemsetAsync(h->nmax_bits, 0, sizeof(unsigned long long), h->stream));
  h->nmax_zeroed = false;
  static const int nmax_blocks = getenv("GNX_NMAX_BLOCKS") ? atoi(getenv("GNX_NMAX_BLOCKS")) : 512;
  const int R = gnx_nmax_rows_fit(h->lat.Jx, 0);
  hipLaunchKernelGGL(k_nmax, dim3(std::min(h->cfg.H, nmax_blocks)), dim3(256),
                     gnx_nmax_lds_doubles(h->lat.Jx, R) * sizeof(double), h->stream,
                     make_splinec(h, h->spl_N), h->cfg.W, h->cfg.H, h->nmax_bits, R, gnx_nmax_scan_min());
  h->nmax_cur = h->nmax_bits;
  return 0;
}

// ---- the density path without a counting pass (gnx_bins.h, gnx_internal.h: fb) ----------
bool gnx_fused_bins(const gnx_state* h) {
  static const bool on = !(getenv("GNX_FUSED_BINS") && atoi(getenv("GNX_FUSED_BINS")) == 0);
  const int64_t nn = (int64_t)h->lat.Jx * h->lat.Jy;
  const size_t lds = ((size_t)4 * nn + std::max(h->lat.Jx, h->lat.Jy) + 1 +
                      gnx_nmax_lds_doubles(h->lat.Jx, 1)) * sizeof(double);
  return on && !h->tiled && !h->tile2_mode && h->fb[0] != nullptr && h->stream3 != nullptr && lds <= 64 * 1024 &&
         (size_t)h->lat.nbx * h->lat.nby * sizeof(int32_t) <= 48 * 1024;
}

int gnx_wait_latP(gnx_state* h) {
  if (h->latP_inflight) {
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_latP, 0));
    h->latP_inflight = false;
  }
  return 0;
}

// the pairs' bins and lattice on stream3 (n_max bounds the grid, the pair count itself is
// read on the device): they run beside k_offspring, the death probabilities wait for them
int gnx_l_lattice_P_async(gnx_state* h, int64_t n_max) {
  const GnxLattice& L = h->lat;
  const int64_t nn = (int64_t)L.Jx * L.Jy;
  const int nb = L.nbx * L.nby;
  const size_t lds_bytes = ((size_t)4 * nn + std::max(L.Jx, L.Jy) + 1) * sizeof(double);
  GNXCHK(gnx_wait_latP(h));        // (a lattice of the last pair list nobody waited for)
  HIPCHK(hipEventRecord(h->ev_pairs, h->stream));
  HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_pairs, 0));
  // the adults' bins ride behind the same event (positions are final since the cell sort;
  // an event of their own right after k_permute cost the main stream a record)
  GNXCHK(gnx_bins_adults_launch(h));
  if (!h->fb_zero[2])
    HIPCHK(hipMemsetAsync(h->fb[2], 0, (size_t)nb * sizeof(int32_t), h->stream3));
  const int blocks = (int)std::min<int64_t>(BIN_BLOCKS, std::max<int64_t>(1, (n_max + 255) / 256));
  hipLaunchKernelGGL(k_bins, dim3(blocks), dim3(256), (size_t)nb * sizeof(int32_t), h->stream3,
                     n_max, (const int32_t*)h->cnt_dev, (const float*)h->mid_x,
                     (const float*)h->mid_y, (const uint8_t*)nullptr, 1.0 / L.hww, L.nbx, L.nby,
                     h->fb[2], GnxSetWords{});
  hipLaunchKernelGGL(k_lattice, dim3(1), dim3(256), lds_bytes, h->stream3, L.Jx, L.Jy, L.nbx,
                     (const int32_t*)h->fb[2], L.areas, L.hww, L.cprime, h->spl_P.c,
                     (int32_t*)nullptr, 0, (unsigned long long*)nullptr, 1, 1);
  HIPCHK(hipEventRecord(h->ev_latP, h->stream3));
  HIPCHK(hipGetLastError());
  h->latP_inflight = true;
  h->fb_zero[2] = true;
  h->spl_P.valid = true;
  return 0;
}

int gnx_l_bins_adults_async(gnx_state* h, const float* d_x, const float* d_y, int64_t N) {
  if (!(h->have_sp && gnx_fused_bins(h))) {
    h->fb_adults = false;
    return 0;
  }
  const GnxLattice& L = h->lat;
  const int nb = L.nbx * L.nby;
  const int cur = h->fb_cur;
  // (clear unless somebody counted and nobody consumed; on `stream`, ahead of k_offspring's
  // counts of the newborns)
  if (!h->fb_zero[cur])
    HIPCHK(hipMemsetAsync(h->fb[cur], 0, (size_t)nb * sizeof(int32_t), h->stream));
  // launched on stream3 by whoever orders stream3 behind this stream next: the pairs' density
  // (gnx_l_lattice_P_async), or gnx_l_density_N with an event of its own
  h->fbp_x = d_x;
  h->fbp_y = d_y;
  h->fbp_N = N;
  h->fb_pending = true;
  h->fb_zero[cur] = false;
  h->fb_adults = true;
  h->fb_count = N;
  return 0;
}

// stream3 is ordered behind the cell sort on `stream`: count the adults there
int gnx_bins_adults_launch(gnx_state* h) {
  if (!h->fb_pending) return 0;
  h->fb_pending = false;
  const GnxLattice& L = h->lat;
  const int nb = L.nbx * L.nby;
  const int64_t N = h->fbp_N;
  const int blocks = (int)std::min<int64_t>(BIN_BLOCKS, std::max<int64_t>(1, (N + 255) / 256));
  hipLaunchKernelGGL(k_bins, dim3(blocks), dim3(256), (size_t)nb * sizeof(int32_t), h->stream3, N,
                     (const int32_t*)nullptr, h->fbp_x, h->fbp_y, (const uint8_t*)nullptr,
                     1.0 / L.hww, L.nbx, L.nby, h->fb[h->fb_cur], GnxSetWords{});
  HIPCHK(hipEventRecord(h->ev_binsN, h->stream3));
  HIPCHK(hipGetLastError());
  h->binsN_inflight = true;
  return 0;
}

// positions or slots are about to change and nobody has counted: forget it (the density
// counts everybody itself then)
void gnx_bins_adults_drop(gnx_state* h) {
  if (h->fb_pending) {
    h->fb_pending = false;
    h->fb_adults = false;
  }
}

int gnx_l_density_N(gnx_state* h) {
  GnxSoA s = h->soa[h->cur];
  if (!(gnx_fused_bins(h) && h->fb_adults && h->fb_count == h->N)) {
    // nobody counted (tiles, operator calls, a population that changed since the sort)
    h->fb_adults = false;
    h->nmax_ready = false;
    h->last_N_fused = false;
    return gnx_l_density(h, h->N, s.x, s.y, &h->spl_N, nullptr);
  }
  const GnxLattice& L = h->lat;
  const int64_t nn = (int64_t)L.Jx * L.Jy;
  const size_t lat_doubles = (size_t)4 * nn + std::max(L.Jx, L.Jy) + 1;
  const int nmax_R = gnx_nmax_rows_fit(L.Jx, lat_doubles);
  const size_t lds_bytes = (lat_doubles + gnx_nmax_lds_doubles(L.Jx, nmax_R)) * sizeof(double);
  const int cur = h->fb_cur;
  if (h->fb_pending) {           // (no pairs' density this step: an event of their own)
    HIPCHK(hipEventRecord(h->ev_perm, h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_perm, 0));
    GNXCHK(gnx_bins_adults_launch(h));
  }
  if (h->binsN_inflight) {
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_binsN, 0));
    h->binsN_inflight = false;
  }
  static const int blocks_env = getenv("GNX_LATN_BLOCKS") ? atoi(getenv("GNX_LATN_BLOCKS")) : 256;
  gnx_time_begin(h);
  hipLaunchKernelGGL(k_lattice_nmax, dim3(std::max(1, std::min(h->cfg.H, blocks_env))), dim3(256),
                     lds_bytes, h->stream, L.Jx, L.Jy, L.nbx, (const int32_t*)h->fb[cur], L.areas,
                     L.hww, L.cprime, h->spl_N.c, h->cfg.W, h->cfg.H, h->nmax2 + cur,
                     h->fb[cur ^ 1], h->nmax2 + (cur ^ 1), (int32_t*)nullptr, (double*)nullptr,
                     GnxPubWords{}, nmax_R, gnx_nmax_scan_min());
  gnx_time_end(h, GNX_K_DENSITY, (double)h->N * 8.0);
  HIPCHK(hipGetLastError());
  h->spl_N.valid = true;
  h->nmax_cur = h->nmax2 + cur;
  h->nmax_ready = true;
  h->fb_zero[cur] = false;
  h->fb_zero[cur ^ 1] = true;
  h->fb_cur = cur ^ 1;
  h->fb_adults = false;
  h->last_N_fused = true;
  return 0;
}

// Tiles: both density fields' lattices (their all-reduced bins) and N.max() in ONE launch, the
// counter words and checks published from it - three launches of one field each and a
// publishing kernel otherwise.  false: the lattice is too fine for it, nothing was launched.
bool gnx_l_lattices_tiled(gnx_state* h, bool have_pairs, const GnxPubWords& pub) {
  static const bool on = !(getenv("GNX_TILE_LATN") && atoi(getenv("GNX_TILE_LATN")) == 0);
  const GnxLattice& L = h->lat;
  const int64_t nn = (int64_t)L.Jx * L.Jy;
  const size_t lat_doubles = (size_t)4 * nn + std::max(L.Jx, L.Jy) + 1;
  const int nmax_R = gnx_nmax_rows_fit(L.Jx, lat_doubles);
  const size_t lds_bytes = (lat_doubles + gnx_nmax_lds_doubles(L.Jx, nmax_R)) * sizeof(double);
  if (!on || lds_bytes > 64 * 1024 || !h->nmax_bits || !h->nmax_zeroed) return false;
  static const int blocks_env = getenv("GNX_LATN_BLOCKS") ? atoi(getenv("GNX_LATN_BLOCKS")) : 256;
  gnx_time_begin(h);
  hipLaunchKernelGGL(k_lattice_nmax,
                     dim3(std::max(1, std::min(h->cfg.H, blocks_env)) + (have_pairs ? 1 : 0)),
                     dim3(256), lds_bytes, h->stream, L.Jx, L.Jy, L.nbx,
                     (const int32_t*)h->bin_partials, L.areas, L.hww, L.cprime, h->spl_N.c, h->cfg.W,
                     h->cfg.H, h->nmax_bits, (int32_t*)nullptr, (unsigned long long*)nullptr,
                     have_pairs ? h->bins_P : (int32_t*)nullptr, h->spl_P.c, pub, nmax_R, gnx_nmax_scan_min());
  gnx_time_end(h, GNX_K_DENSITY, (double)h->N * 8.0);
  h->spl_N.valid = true;
  h->spl_P.valid = have_pairs;
  h->nmax_cur = h->nmax_bits;
  h->nmax_ready = true;
  h->nmax_zeroed = false;
  return true;
}

int gnx_l_death_probs(gnx_state* h, bool with_selection) {
  int64_t N = h->N;
  if (N == 0) return 0;
  SplineC SN = make_splinec(h, h->spl_N), SP = make_splinec(h, h->spl_P);
  GNXCHK(gnx_wait_latP(h));
  gnx_time_begin(h);
  if (!h->nmax_ready) GNXCHK(launch_nmax(h));
  h->nmax_ready = false;
  hipLaunchKernelGGL(k_death_probs, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream,
                     make_deathp(h, with_selection), make_demp(h), SN, SP, h->soa[h->cur], h->rast,
                     gnx_trait_tab(h), h->delet_s, h->nmax_cur, h->p_death, h->d_cell);
  gnx_time_end(h, GNX_K_DEATH, (double)N * (28.0 + 8.0 * h->cfg.n_traits));
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- mortality + compaction
// _do_mortality (ops/demography.py:175-180): dead ~ Bernoulli(p_death).  Ghosts
// (halo copies, tiled runs) are dropped here without counting as deaths.
// Compaction without look-back scans (gnx_compact.h): k_alive decides and counts per
// block, one workgroup scans the block counts, k_xo_jobs_surv / k_compact recompute
// the in-block ranks from the stored flags.
__global__ void __launch_bounds__(256)
k_alive(int64_t N, const double* p_death, const uint8_t* dead_in, const int64_t* id,
        const uint8_t* ghost, const int32_t* grow, long long step, unsigned long long seed,
        int32_t* alive, int32_t* dead_row, int32_t* cnt, int stride, int64_t xo_first,
        int32_t* zero_jobs, const GnxDD* __restrict__ dd, GnxScanOut scan) {
  __shared__ int lds[16];
  __shared__ int lds2[8];
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  if (dd) {
    // device-driven step: everybody incl. this step's offspring; the offspring (slots from
    // dd->N on) wait for their crossover iff xo_first >= 0
    N = (int64_t)dd->N + dd->B;
    step = dd->step;
    if (xo_first >= 0) xo_first = dd->N;
  }
  // the job list of the deferred crossover is appended to (k_xo_jobs_surv): empty it
  if (zero_jobs && blockIdx.x == 0 && threadIdx.x == 0) *zero_jobs = 0;
  bool fa[4], fd[4], fx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    fa[r] = fd[r] = fx[r] = false;
    if (i < N) {
      bool dead;
      const bool g = ghost[i] != 0;
      if (g) {
        dead = true;
      } else if (dead_in) {
        dead = dead_in[i] != 0;
      } else {
        uint4 rr = gnx_rand4(seed, (unsigned long long)id[i], step, OP_DEATH, 0);
        dead = (double)gnx_u01(rr.x) < p_death[i];
      }
      fa[r] = !dead;
      // rows to return to the free stack (offspring that die before their deferred
      // crossover never had one; ghosts own none)
      fd[r] = dead && !g && grow[i] >= 0;
      // surviving offspring of this step whose crossover was deferred: it gets its genome
      // row and its crossover now (xo_first < 0: nothing deferred).  alive[] bit 1 marks it.
      fx[r] = !dead && xo_first >= 0 && i >= xo_first && grow[i] < 0;
      alive[i] = (fa[r] ? 1 : 0) | (fx[r] ? 2 : 0);
      dead_row[i] = fd[r] ? 1 : 0;
    }
  }
  int rank[4], ta, td, tx;
  gnx_block_ranks(fa, rank, ta, lds);
  gnx_block_ranks(fd, rank, td, lds);
  gnx_block_ranks(fx, rank, tx, lds);
  if (scan.ticket) {
    // (device-driven step: the workgroup that finishes last turns the counts into offsets and
    // totals - no k_block_scan launch)
    const int v[3] = {ta, td, tx};
    gnx_count_and_scan<3>(v, cnt, scan, lds2);
    return;
  }
  if (threadIdx.x == 0) {
    cnt[blockIdx.x] = ta;
    cnt[stride + blockIdx.x] = td;
    cnt[2 * stride + blockIdx.x] = tx;
  }
}

struct GnxXoPlan {
  int32_t row;          // the offspring's logical row (-1: it gets none)
  int32_t pop, job;     // stack index of its first fresh block, its first slot in the job list
  int32_t prow[2];      // the parents' rows (-1: ghost)
  int32_t mixsel[2];    // blocks with a switch point | homologue at each block's start << 16
  int32_t ks[2];        // path * 2 + start homologue
  int32_t pad;
};

// Crossover jobs of the surviving offspring whose crossover was deferred (alive[] bit 1):
// the rank among them picks the row from the top of the free stack.  Offspring that died
// at age 0 never get a row; offspring that already have one (their mate was a ghost: the
// tile cut their genome at once, csrc/gnx_tile.hip) are left alone.  cnts[2] = how many.
__global__ void __launch_bounds__(256)
k_xo_jobs_surv(int64_t N, int64_t first, int32_t* __restrict__ grow,
               const int32_t* __restrict__ alive, const int32_t* __restrict__ blk_off3,
               const int32_t* __restrict__ off_parent, const int32_t* __restrict__ off_keys,
               const uint8_t* __restrict__ off_start, const int32_t* __restrict__ free_rows,
               int64_t n_free, GnxHalves H, const int32_t* __restrict__ bp_off,
               const int32_t* __restrict__ bp_loci, int32_t* __restrict__ n_jobs,
               GnxXoPlan* __restrict__ plan) {
  __shared__ int lds[16];
  __shared__ int s_pop, s_job;
  const int64_t b = first / GNX_CB + blockIdx.x;
  const int64_t base = b * GNX_CB;
  bool fx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    fx[r] = i < N && (alive[i] & 2) != 0;
  }
  int rank[4], tot;
  gnx_block_ranks(fx, rank, tot, lds);
  const int32_t boff = blk_off3[b];
  // Block by block (gnx_half.h), a gamete refers to the parent's block where its path has no
  // switch point and gets a block of its own and a job where it has one (parents are older:
  // slots < first).  The workgroup takes its blocks and its stretch of the job list with ONE
  // atomic each: those are single words, and a thousand waves taking turns on them cost more
  // than everything else in this kernel.
  // Every load of a stage is issued for all eight gametes of the thread before the first
  // result is used (unconditional loads from clamped indices: behind `fx ? load : 0` the
  // compiler waits for each one in turn).
  const int NB = H.NB;
  const unsigned int all = (1u << NB) - 1u;
  int32_t row[4], prow[4][2], ks[4][2];
  int32_t par[4][2], key[4][2], st[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    const int64_t k = fx[r] ? i - first : 0;
    row[r] = free_rows[n_free - 1 - (fx[r] ? boff + rank[r] : 0)];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      par[r][p] = off_parent[2 * k + p];
      key[r][p] = off_keys[2 * k + p];
      st[r][p] = off_start[2 * k + p];
    }
  }
  int32_t b0[4][2], b1[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (fx[r]) grow[i] = row[r];
    else row[r] = -1;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      prow[r][p] = grow[par[r][p]];
      b0[r][p] = bp_off ? bp_off[key[r][p]] : 0;
      b1[r][p] = bp_off ? bp_off[key[r][p] + 1] : 0;
    }
  }
  unsigned int mixed[4][2], sel[4][2];
  int cf[4], cj[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    cf[r] = cj[r] = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (!fx[r]) prow[r][p] = -1;
      ks[r][p] = key[r][p] * 2 + st[r][p];
      mixed[r][p] = all;                       // dense masks, ghost parent: cut everything
      sel[r][p] = 0u;
      if (fx[r] && prow[r][p] >= 0 && bp_off)
        gnx_block_masks(bp_loci + b0[r][p], b1[r][p] - b0[r][p], st[r][p], NB, H.BW, mixed[r][p],
                        sel[r][p]);
      if (fx[r]) {
        const int nf = __popc(mixed[r][p] & all);
        cf[r] += nf;
        cj[r] += prow[r][p] >= 0 ? nf : 0;
      }
    }
  }
  int of[4], oj[4], tf, tj;
  gnx_block_sums(cf, of, tf, lds);
  gnx_block_sums(cj, oj, tj, lds);
  if (threadIdx.x == 0) {
    s_pop = tf ? atomicSub(H.top, tf) : 0;
    s_job = tj ? atomicAdd(n_jobs, tj) : 0;
  }
  __syncthreads();
  const int pop0 = s_pop, job0 = s_job;
  // the plan of every surviving offspring; k_xo_jobs_write carries it out, one thread per
  // logical block
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < first || i >= N) continue;
    GnxXoPlan P;
    P.row = fx[r] ? row[r] : -1;
    P.pop = pop0 - 1 - of[r];          // stack index of its first fresh block (they go downwards)
    P.job = job0 + oj[r];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      P.prow[p] = prow[r][p];
      P.mixsel[p] = (mixed[r][p] & 0xffffu) | (sel[r][p] << 16);
      P.ks[p] = ks[r][p];
    }
    plan[i - first] = P;
  }
}

// One thread per logical block of every offspring that got a row: the block refers to the
// parent's block where the path has no switch point (neither may take a mutation in place
// from now on), or takes a fresh block and, with a local parent, a job.  Which fresh block
// and which job slot follow from the plan's offsets and the ranks of the block among the
// offspring's cut blocks - no thread waits for another.
__global__ void __launch_bounds__(256)
k_xo_jobs_write(int64_t B, GnxHalves H, const GnxXoPlan* __restrict__ plan,
                GnxXoJob* __restrict__ jobs) {
  const int NB = H.NB;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * 2 * NB) return;
  const int64_t k = t / (2 * NB);
  const int rem = (int)(t - k * 2 * NB);
  const int p = rem / NB, q = rem - p * NB;
  const GnxXoPlan P = plan[k];
  if (P.row < 0) return;
  const unsigned int all = (1u << NB) - 1u;
  const unsigned int m0 = P.mixsel[0] & all, m1 = P.mixsel[1] & all;
  const unsigned int mixed = p ? m1 : m0, sel = (unsigned int)P.mixsel[p] >> 16;
  const unsigned int below = (1u << q) - 1u;
  const bool local = P.prow[p] >= 0;
  const int64_t lb = ((int64_t)P.row * 2 + p) * NB + q;
  const int64_t ph0 = (int64_t)(local ? P.prow[p] : 0) * 2 * NB;
  if ((mixed >> q) & 1u) {
    // rank among the offspring's cut blocks (gamete 0 first), among its jobs likewise
    const int fr = (p ? __popc(m0) : 0) + __popc(mixed & below);
    const int jr = (p && P.prow[0] >= 0 ? __popc(m0) : 0) + __popc(mixed & below);
    const int32_t dst = H.stack[P.pop - fr];
    H.hmap[lb] = (int32_t)((uint32_t)dst | GNX_OWN);
    if (local) {
      GnxXoJob j;
      j.ph0 = GNX_BLK(H.hmap[ph0 + q]);
      j.ph1 = GNX_BLK(H.hmap[ph0 + NB + q]);
      j.dst = dst;
      j.ks = P.ks[p] | (q << 24);
      jobs[P.job + jr] = j;
    }
  } else {
    const int64_t plb = ph0 + ((sel >> q) & 1u) * NB + q;
    const int32_t e = H.hmap[plb];
    H.hmap[lb] = GNX_BLK(e);
    if (e < 0) H.hmap[plb] = GNX_BLK(e);     // (only the first child to share it writes)
  }
}

// The two kernels above in one (NB <= 16): one thread per offspring SLOT (256 per
// workgroup - four times the workgroups of k_xo_jobs_surv, which left a third of the CUs
// idle), the plan stays in registers, and the thread walks its own 2 x NB table entries
// with every load of a stage issued before the first result is used: the parent's 2 x NB
// entries are NB 8-byte loads, the child's NB 8-byte stores.  One thread per logical block
// (k_xo_jobs_write: 3.3 M threads, each a chain of three dependent loads) took 52 us,
// k_xo_jobs_surv 23.
#define GNX_JF_NB 28
__device__ __forceinline__ int32_t gnx_fresh_at(int fr, int32_t f0, int32_t f1, int32_t f2, int32_t f3,
                                                int32_t f4, int32_t f5,
                                                const int32_t* __restrict__ stack, int pop) {
  int32_t d = f0;
  d = fr == 1 ? f1 : d;
  d = fr == 2 ? f2 : d;
  d = fr == 3 ? f3 : d;
  d = fr == 4 ? f4 : d;
  d = fr == 5 ? f5 : d;
  if (fr >= 6) d = stack[pop - fr];
  return d;
}
// threads per workgroup: every workgroup takes its stretch of the free-block stack and of the
// job list with one atomic each, all on the same two words - 815 workgroups of 256 threads
// queue up there (64 us; 512 threads: 58 us; 1024 threads spill: 64 us)
#ifndef GNX_JF_TPB
#define GNX_JF_TPB 512
#endif
template <int NB, int TPB>
__global__ void __launch_bounds__(TPB)
k_xo_jobs_fused(int64_t N, int64_t first, int32_t* __restrict__ grow,
                const int32_t* __restrict__ alive, const int32_t* __restrict__ blk_off3,
                const int32_t* __restrict__ off_parent, const int32_t* __restrict__ off_keys,
                const uint8_t* __restrict__ off_start, const int32_t* __restrict__ free_rows,
                int64_t n_free, GnxHalves H, const int32_t* __restrict__ bp_off,
                const int32_t* __restrict__ bp_loci, int32_t* __restrict__ n_jobs,
                GnxXoJob* __restrict__ jobs, GnxJobBp* __restrict__ jobs_bp,
                const int32_t* __restrict__ cnt3, GnxDD* __restrict__ dd) {
  constexpr int WAVES = TPB / 64;
  if (dd) {
    N = (int64_t)dd->N + dd->B;
    first = dd->N;
    n_free = dd->n_free;
    if ((first / TPB + blockIdx.x) * TPB >= N) return;         // (block-uniform)
  }
  __shared__ int wsum[3][WAVES];
  __shared__ int prev_s[WAVES];
  __shared__ int psum[WAVES];
  __shared__ int s_pop, s_job;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (first / TPB + blockIdx.x) * TPB;
  const int64_t i = base + tid;
  const int64_t b = base / GNX_CB;                 // the compaction block of these TPB slots
  const int round = (int)((base - b * GNX_CB) / TPB);
  // stage 1: who is a surviving offspring without a row, here and in the earlier rounds of
  // the same compaction block (their number comes before this round's ranks)
  const bool fx = i < N && (alive[i] & 2) != 0;
  int prev = 0;
  for (int r = 0; r < round; ++r) {
    const int64_t j = b * GNX_CB + r * TPB + tid;
    prev += __popcll(__ballot(j < N && (alive[j] & 2) != 0));
  }
  // cnt3 != null: the block counts have not been scanned (the scan runs on the side stream,
  // for the compaction and the host) - this workgroup adds up the counts before its
  // compaction block itself: a few coalesced loads, and one launch less on the step's chain
  int part = 0;
  if (cnt3)
    for (int q = tid; q < (int)b; q += TPB) part += cnt3[q];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  const unsigned long long bal = __ballot(fx);
  if (lane == 0) {
    wsum[0][wave] = __popcll(bal);
    prev_s[wave] = prev;
    psum[wave] = part;
  }
  __syncthreads();
  int rank = __popcll(bal & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) rank += wsum[0][w];
  if (cnt3) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) rank += psum[w];
  } else {
    rank += blk_off3[b];
  }
#pragma unroll
  for (int w = 0; w < WAVES; ++w) rank += prev_s[w];
  // (device-driven step: nobody checked the rows on the host - running out of them is
  // reported, the indices stay inside the stacks, the run is invalid and ends with an error)
  if (dd && fx && rank >= n_free) {
    dd->err |= GNX_DD_ERR_ROWS;
    rank = 0;
  }
  // stage 2: row, parents, keys, start homologues (unconditional loads from clamped indices)
  const int64_t k = fx ? i - first : 0;
  int32_t row = free_rows[max((int64_t)0, n_free - 1 - (fx ? rank : 0))];
  int32_t par[2], key[2], st[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    par[p] = off_parent[2 * k + p];
    key[p] = off_keys[2 * k + p];
    st[p] = off_start[2 * k + p];
  }
  if (fx) grow[i] = row;
  else row = -1;
  // stage 3: the parents' rows, the paths' breakpoint ranges
  int32_t prow[2], b0[2], b1[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    prow[p] = grow[par[p]];
    b0[p] = bp_off ? bp_off[key[p]] : 0;
    b1[p] = bp_off ? bp_off[key[p] + 1] : 0;
  }
  // stage 4: which blocks hold a switch point.  The first four switch points of each path are
  // loaded once into registers (stage 7 looks them up again, block by block), and the block of
  // a locus comes from a float reciprocal with an exact correction (loci < 2^24) instead of an
  // integer division
  const unsigned int all = (1u << NB) - 1u;
  const int lpb = H.BW * 64;
  const float inv_lpb = 1.0f / (float)lpb;
  auto blk_of = [&](int l) __attribute__((always_inline)) {
    int q = (int)((float)l * inv_lpb);
    q -= (q * lpb > l) ? 1 : 0;
    q += ((q + 1) * lpb <= l) ? 1 : 0;
    return min(q, NB - 1);
  };
  unsigned int mixed[2], sel[2];
  int32_t bl[2][4];
  int nbp[2];
  int cf = 0, cj = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (!fx) prow[p] = -1;
    nbp[p] = (bp_off && prow[p] >= 0) ? b1[p] - b0[p] : 0;
#pragma unroll
    for (int z = 0; z < 4; ++z) bl[p][z] = z < nbp[p] ? bp_loci[b0[p] + z] : 0;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    mixed[p] = all;                              // dense masks, ghost parent: cut everything
    sel[p] = 0u;
    if (fx && prow[p] >= 0 && bp_off) {
      unsigned int mx = 0u, sl = st[p] ? all : 0u;
#pragma unroll
      for (int z = 0; z < 4; ++z)
        if (z < nbp[p]) {
          const int blk = blk_of(bl[p][z]);
          mx |= 1u << blk;
          sl ^= all & ~((2u << blk) - 1u);        // every later block starts on the other homologue
        }
      for (int z = 4; z < nbp[p]; ++z) {
        const int blk = blk_of(bp_loci[b0[p] + z]);
        mx |= 1u << blk;
        sl ^= all & ~((2u << blk) - 1u);
      }
      mixed[p] = mx;
      sel[p] = sl & all;
    }
    if (fx) {
      const int nf = __popc(mixed[p]);
      cf += nf;
      cj += prow[p] >= 0 ? nf : 0;
    }
  }
  // stage 5: this workgroup's stretch of the free-block stack and of the job list, with
  // ONE atomic each; exclusive sums of cf / cj over the TPB threads
  int xf = cf, xj = cj;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int yf = __shfl_up(xf, d), yj = __shfl_up(xj, d);
    if (lane >= d) {
      xf += yf;
      xj += yj;
    }
  }
  if (lane == 63) {
    wsum[1][wave] = xf;
    wsum[2][wave] = xj;
  }
  __syncthreads();
  int of = xf - cf, oj = xj - cj;
  for (int w = 0; w < wave; ++w) {
    of += wsum[1][w];
    oj += wsum[2][w];
  }
  if (tid == 0) {
    int tf = 0, tj = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      tf += wsum[1][w];
      tj += wsum[2][w];
    }
    // (without these two contended words - a timing experiment with wrong results - the kernel
    // takes the same time: profiles/r05_ab_runs.txt)
    s_pop = tf ? atomicSub(H.top, tf) : 0;
    s_job = tj ? atomicAdd(n_jobs, tj) : 0;
  }
  __syncthreads();
  if (dd && tid == 0) {
    int tf = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) tf += wsum[1][w];
    if (s_pop < tf) dd->err |= GNX_DD_ERR_BLOCKS;
  }
  if (!fx) return;
  // stack index of my first fresh block (dd: never below what this thread walks down)
  const int pop = dd ? max(s_pop - 1 - of, cf) : s_pop - 1 - of;
  const int job = s_job + oj;
  // stage 6: the parents' table entries (2 x NB each, 8-byte loads) and my first six fresh
  // blocks, all issued before anything is used.  (All 2 NB fresh blocks in registers cost a
  // (2 NB)^2 select chain - PMC: 4 300 vector instructions per wave at NB = 14; one load per
  // block inside stage 7 put a memory round trip into each of its 2 NB iterations.)
  int32_t pe[2][2 * NB];                           // [parent][hom * NB + q]
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int2* src = (const int2*)(H.hmap + (int64_t)max(prow[p], 0) * 2 * NB);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int2 v = src[q];
      pe[p][2 * q] = v.x;
      pe[p][2 * q + 1] = v.y;
    }
  }
  // (six scalars, not an array: the compiler turns a select chain over an array captured by
  // reference into an indexed load from scratch)
  const int32_t f0 = 0 < cf ? H.stack[pop] : 0, f1 = 1 < cf ? H.stack[pop - 1] : 0,
                f2 = 2 < cf ? H.stack[pop - 2] : 0, f3 = 3 < cf ? H.stack[pop - 3] : 0,
                f4 = 4 < cf ? H.stack[pop - 4] : 0, f5 = 5 < cf ? H.stack[pop - 5] : 0;
  const int32_t* stack = H.stack;
  // (a function of scalars, not a closure: with more than 20 blocks per homologue the closure
  // object stayed in scratch and the compiler chose between its address and the stack's)
#define fresh_at(fr_) gnx_fresh_at((fr_), f0, f1, f2, f3, f4, f5, stack, pop)
  // stage 7a: my table (NB 8-byte stores); the parents' blocks that are shared from now on
  // lose their never-shared flag (only the first child to share one writes).  Branch-free but
  // for the stores: every wave has some lane on either side of every block.
  int32_t ce[2 * NB];
  int fr = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const bool cut = ((mixed[p] >> q) & 1u) != 0u;
      const int hsel = (sel[p] >> q) & 1u;
      const int32_t pv = hsel ? pe[p][NB + q] : pe[p][q];
      const int32_t sv = GNX_BLK(pv);
      if (fx && !cut && pv < 0) H.hmap[((int64_t)prow[p] * 2 + hsel) * NB + q] = sv;
      const int32_t fv = (int32_t)((uint32_t)fresh_at(cut ? fr : 0) | GNX_OWN);
      ce[p * NB + q] = cut ? fv : sv;
      fr += cut ? 1 : 0;
    }
  }
  {
    int2* dstp = (int2*)(H.hmap + (int64_t)row * 2 * NB);
#pragma unroll
    for (int q = 0; q < NB; ++q) dstp[q] = make_int2(ce[2 * q], ce[2 * q + 1]);
  }
  // stage 7b: one job per cut block of a local parent, walking the set bits (about two per
  // homologue) rather than all 2 NB blocks; the switch points inside the block ride with the
  // job (gnx_xo.h: GnxJobBp), taken from the registers of stage 4
  int jr = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (prow[p] < 0) continue;
    const int fbase = p ? __popc(mixed[0]) : 0;
    unsigned int m = mixed[p];
    while (m) {
      const int q = __ffs(m) - 1;
      m &= m - 1u;
      const int32_t dst = fresh_at(fbase + __popc(mixed[p] & ((1u << q) - 1u)));
      int32_t e0 = pe[p][0], e1 = pe[p][NB];
      if constexpr (NB <= 16) {
#pragma unroll
        for (int q2 = 1; q2 < NB; ++q2) {
          e0 = q == q2 ? pe[p][q2] : e0;
          e1 = q == q2 ? pe[p][NB + q2] : e1;
        }
      } else {
        // (more blocks: the select chain over the registers outgrows the unroller and the table
        // in registers would move to scratch - the two entries are read again, from L2)
        const int32_t* prw = H.hmap + (int64_t)prow[p] * 2 * NB;
        e0 = prw[q];
        e1 = prw[NB + q];
      }
      GnxXoJob j;
      j.ph0 = GNX_BLK(e0);
      j.ph1 = GNX_BLK(e1);
      j.dst = dst;
      j.ks = (key[p] * 2 + st[p]) | (q << 24);
      jobs[job + jr] = j;
      unsigned int o0 = 0, o1 = 0, o2 = 0;
      int nin = 0;
      const int lo = q * lpb, hi = (q == NB - 1) ? 0x7fffffff : lo + lpb;
      auto take = [&](int l) __attribute__((always_inline)) {
        if (l >= lo && l < hi) {
          const unsigned int o = (unsigned int)(l - lo);
          o0 = nin == 0 ? o : o0;
          o1 = nin == 1 ? o : o1;
          o2 = nin == 2 ? o : o2;
          ++nin;
        }
      };
#pragma unroll
      for (int z = 0; z < 4; ++z)
        if (z < nbp[p]) take(bl[p][z]);
      for (int z = 4; z < nbp[p]; ++z) take(bp_loci[b0[p] + z]);
      // (offsets are 16 bits: a block of more than 65 536 loci looks its path up)
      const bool inl = bp_off != nullptr && nin <= 3 && lpb <= 65536;
      const unsigned int meta = (inl ? (unsigned int)nin : 0u) | (((sel[p] >> q) & 1u) << 2) |
                                (inl ? 0u : GNX_BP_MORE);
      *(uint2*)(jobs_bp + job + jr) = make_uint2((o0 & 0xffffu) | (o1 << 16),
                                                 (o2 & 0xffffu) | (meta << 16));
      ++jr;
    }
  }
}

#undef fresh_at

// ---- the job builder with a gamete's table laid across lanes (round 6) ------------------------
// k_xo_jobs_fused above gives every surviving offspring ONE thread that walks 2 x 2 NB parent
// entries and NB child entries by itself: 60 loads and stores of 8 bytes per thread at NB = 20,
// every one of them a request of its own (the lanes of a wave sit in 64 different rows) - ~3 200
// requests per wave for 30 KB, 49 % of the waves' cycles waiting for them (SQ counters,
// profiles/r05_pmc_job_builder.txt).  Here the per-offspring PLAN (row, parents' rows, which
// blocks are cut, which homologue the others follow, where its fresh blocks and jobs start) is
// worked out one thread per slot as before (stages 1-5, same arithmetic), left in LDS (64 bytes
// per slot), and the TABLES are then walked by the wave together: a gamete's NB entries lie in NB
// adjacent lanes - lane q loads the parent's two entries for block q (two coalesced 4 NB-byte
// rows per gamete), picks or takes a fresh block, stores the child's entry (one coalesced row) and
// writes the job of a cut block; 64 / NB gametes per wave-instruction, four instructions' loads
// in flight before the first store.  NB is a run-time value here: one kernel for every genome
// length.  Same plan, same stack / job indices as k_xo_jobs_fused: bit-identical tables and job
// lists (tests/test_gpu_product_path.py, test_gpu_halves.py, test_gpu_deferred.py).
struct alignas(16) GnxJobPlan {
  int32_t row, pop, job, m0cnt;      // m0cnt: cut blocks of gamete 0
  int32_t prow[2];
  uint32_t mixed[2], sel[2];
  int32_t ks[2];                     // path * 2 + start homologue
  int32_t nbp[2], b0[2];
};
static_assert(sizeof(GnxJobPlan) == 64, "one plan per slot, 64 bytes of LDS");

#ifdef GNX_JL_TRACE       // (tools/build_variant.sh ... -DGNX_JL_TRACE: when each workgroup ran, for tools/jl_trace.py)
__device__ unsigned long long g_jl_trace[4 * 4096];
extern "C" int gnx_debug_jl_trace(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jl_trace), sizeof(g_jl_trace));
}
#endif

// (six waves per SIMD = at most 80 vector registers: THREE 512-thread workgroups on a CU, so that
// the ~520 workgroups of a steady-state step are all resident at once - with the 82 registers the
// compiler took when left alone two fit, the last eight workgroups started when the first 512 had
// finished, and the kernel took 85 us instead of 55: profiles/r06_ab_runs.txt, tools/jl_trace.py)
template <int TPB>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_xo_jobs_lanes(int64_t N, int64_t first, int32_t* __restrict__ grow,
                const int32_t* __restrict__ alive, const int32_t* __restrict__ blk_off3,
                const int32_t* __restrict__ off_parent, const int32_t* __restrict__ off_keys,
                const uint8_t* __restrict__ off_start, const int32_t* __restrict__ free_rows,
                int64_t n_free, GnxHalves H, const int32_t* __restrict__ bp_off,
                const int32_t* __restrict__ bp_loci, int32_t* __restrict__ n_jobs,
                GnxXoJob* __restrict__ jobs, GnxJobBp* __restrict__ jobs_bp,
                const int32_t* __restrict__ cnt3, GnxDD* __restrict__ dd) {
  constexpr int WAVES = TPB / 64;
#ifdef GNX_JL_TRACE
  const unsigned long long jl_t0 = __builtin_readcyclecounter();
  const unsigned long long jl_w0 = wall_clock64();
#endif
  if (dd) {
    N = (int64_t)dd->N + dd->B;
    first = dd->N;
    n_free = dd->n_free;
    if ((first / TPB + blockIdx.x) * TPB >= N) return;         // (block-uniform)
  }
  __shared__ int wsum[3][WAVES];
  __shared__ int prev_s[WAVES];
  __shared__ int psum[WAVES];
  __shared__ int s_pop, s_job;
  __shared__ GnxJobPlan plan[TPB];
  __shared__ uint8_t fxlist[WAVES][64];
  __shared__ uint4 jq_a[WAVES][64];     // queued jobs: {parent block 0, 1, fresh block, path | block}
  __shared__ uint2 jq_b[WAVES][64];     // ... {job index, owner | gamete | homologue | block}
  const int NB = H.NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (first / TPB + blockIdx.x) * TPB;
  const int64_t i = base + tid;
  const int64_t b = base / GNX_CB;                 // the compaction block of these TPB slots
  const int round = (int)((base - b * GNX_CB) / TPB);
  // stages 1-5: as in k_xo_jobs_fused
  const bool fx = i < N && (alive[i] & 2) != 0;
  int prev = 0;
  for (int r = 0; r < round; ++r) {
    const int64_t j = b * GNX_CB + r * TPB + tid;
    prev += __popcll(__ballot(j < N && (alive[j] & 2) != 0));
  }
  int part = 0;
  if (cnt3)
    for (int q = tid; q < (int)b; q += TPB) part += cnt3[q];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  const unsigned long long bal = __ballot(fx);
  if (lane == 0) {
    wsum[0][wave] = __popcll(bal);
    prev_s[wave] = prev;
    psum[wave] = part;
  }
  __syncthreads();
  int rank = __popcll(bal & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) rank += wsum[0][w];
  if (cnt3) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) rank += psum[w];
  } else {
    rank += blk_off3[b];
  }
#pragma unroll
  for (int w = 0; w < WAVES; ++w) rank += prev_s[w];
  if (dd && fx && rank >= n_free) {
    dd->err |= GNX_DD_ERR_ROWS;
    rank = 0;
  }
  const int64_t k = fx ? i - first : 0;
  int32_t row = free_rows[max((int64_t)0, n_free - 1 - (fx ? rank : 0))];
  int32_t par[2], key[2], st[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    par[p] = off_parent[2 * k + p];
    key[p] = off_keys[2 * k + p];
    st[p] = off_start[2 * k + p];
  }
  if (fx) grow[i] = row;
  else row = -1;
  int32_t prow[2], b0[2], b1[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    prow[p] = grow[par[p]];
    b0[p] = bp_off ? bp_off[key[p]] : 0;
    b1[p] = bp_off ? bp_off[key[p] + 1] : 0;
  }
  const unsigned int all = (1u << NB) - 1u;
  const int lpb = H.BW * 64;
  const float inv_lpb = 1.0f / (float)lpb;
  auto blk_of = [&](int l) __attribute__((always_inline)) {
    int q = (int)((float)l * inv_lpb);
    q -= (q * lpb > l) ? 1 : 0;
    q += ((q + 1) * lpb <= l) ? 1 : 0;
    return min(q, NB - 1);
  };
  unsigned int mixed[2], sel[2];
  int nbp[2];
  int cf = 0, cj = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (!fx) prow[p] = -1;
    nbp[p] = (bp_off && prow[p] >= 0) ? b1[p] - b0[p] : 0;
    mixed[p] = all;                              // dense masks, ghost parent: cut everything
    sel[p] = 0u;
    if (fx && prow[p] >= 0 && bp_off) {
      unsigned int mx = 0u, sl = st[p] ? all : 0u;
      for (int z = 0; z < nbp[p]; ++z) {
        const int blk = blk_of(bp_loci[b0[p] + z]);
        mx |= 1u << blk;
        sl ^= all & ~((2u << blk) - 1u);        // every later block starts on the other homologue
      }
      mixed[p] = mx;
      sel[p] = sl & all;
    }
    if (fx) {
      const int nf = __popc(mixed[p]);
      cf += nf;
      cj += prow[p] >= 0 ? nf : 0;
    }
  }
  int xf = cf, xj = cj;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int yf = __shfl_up(xf, d), yj = __shfl_up(xj, d);
    if (lane >= d) {
      xf += yf;
      xj += yj;
    }
  }
  if (lane == 63) {
    wsum[1][wave] = xf;
    wsum[2][wave] = xj;
  }
  __syncthreads();
  int of = xf - cf, oj = xj - cj;
  for (int w = 0; w < wave; ++w) {
    of += wsum[1][w];
    oj += wsum[2][w];
  }
  if (tid == 0) {
    int tf = 0, tj = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      tf += wsum[1][w];
      tj += wsum[2][w];
    }
    s_pop = tf ? atomicSub(H.top, tf) : 0;
    s_job = tj ? atomicAdd(n_jobs, tj) : 0;
    if (dd && s_pop < tf) dd->err |= GNX_DD_ERR_BLOCKS;
  }
  // the plan of this slot, and the wave's list of the lanes that have one
  if (fx) fxlist[wave][__popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)lane;
  __syncthreads();
  {
    GnxJobPlan P;
    P.row = row;
    // stack index of my first fresh block (dd: never below what the walk goes down to)
    P.pop = dd ? max(s_pop - 1 - of, cf) : s_pop - 1 - of;
    P.job = s_job + oj;
    P.m0cnt = __popc(mixed[0]);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      P.prow[p] = prow[p];
      P.mixed[p] = mixed[p];
      P.sel[p] = sel[p];
      P.ks[p] = key[p] * 2 + st[p];
      P.nbp[p] = nbp[p];
      P.b0[p] = b0[p];
    }
    plan[tid] = P;
  }
  // (the wave reads only its own lanes' plans: LDS operations of a wave complete in order)
  __builtin_amdgcn_wave_barrier();
  // stages 6-7: the tables, NB adjacent lanes per gamete.  The jobs of the cut blocks (two or
  // three lanes of an instruction's 60) are not written where they are found - ~80 instructions
  // for three lanes, 36 times per wave: the kernel was bound by instruction issue - but queued
  // in LDS, 64 to a wave, and written by all lanes together when the queue is full.
#ifdef GNX_JL_TIMING_SKIP_WALK            // (timing experiment only: wrong results)
  const int n_g = 0;
#else
  const int n_g = 2 * __popcll(bal);               // gametes of this wave
#endif
  const int GPL = 64 / NB;                         // gametes per wave-instruction
  const int sub = lane / NB, q = lane - sub * NB;
  const bool lane_on = sub < GPL;
  const unsigned int below = (1u << q) - 1u;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int qn = 0;                                      // queued jobs (wave-uniform)
  auto flush = [&]() __attribute__((always_inline)) {
    if (lane < qn) {
      const uint4 a = jq_a[wave][lane];
      const uint2 m = jq_b[wave][lane];
      const int idx = (int)m.x;
      const int owner = (int)(m.y & 0xffffu), p = (int)((m.y >> 16) & 1u);
      const unsigned int hsel = (m.y >> 17) & 1u;
      const int q2 = (int)(m.y >> 24);
      const int nb_p = plan[owner].nbp[p], bb = plan[owner].b0[p];
      unsigned int o0 = 0, o1 = 0, o2 = 0;
      int nin = 0;
      const int lo = q2 * lpb, hi = (q2 == NB - 1) ? 0x7fffffff : lo + lpb;
      for (int z = 0; z < nb_p; ++z) {
        const int l = bp_loci[bb + z];
        if (l >= lo && l < hi) {
          const unsigned int o = (unsigned int)(l - lo);
          o0 = nin == 0 ? o : o0;
          o1 = nin == 1 ? o : o1;
          o2 = nin == 2 ? o : o2;
          ++nin;
        }
      }
      // (offsets are 16 bits: a block of more than 65 536 loci looks its path up)
      const bool inl = bp_off != nullptr && nin <= 3 && lpb <= 65536;
      const unsigned int meta = (inl ? (unsigned int)nin : 0u) | (hsel << 2) |
                                (inl ? 0u : GNX_BP_MORE);
      GnxXoJob j;
      j.ph0 = (int32_t)a.x;
      j.ph1 = (int32_t)a.y;
      j.dst = (int32_t)a.z;
      j.ks = (int32_t)a.w;
      jobs[idx] = j;
      *(uint2*)(jobs_bp + idx) = make_uint2((o0 & 0xffffu) | (o1 << 16),
                                            (o2 & 0xffffu) | (meta << 16));
    }
    qn = 0;
  };
#ifndef GNX_JL_U
#define GNX_JL_U 4
#endif
  constexpr int U = GNX_JL_U;                      // instructions' loads in flight
  for (int g0 = 0; g0 < n_g; g0 += GPL * U) {
    bool on[U], cut[U];
    int32_t e0[U], e1[U], fresh[U], pr[U], rowu[U], jidx[U];
    unsigned int hs[U];
    int own[U], pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = g0 + u * GPL + sub;
      on[u] = lane_on && g < n_g;
      own[u] = wave * 64 + (on[u] ? (int)fxlist[wave][g >> 1] : 0);
      pp[u] = g & 1;
      const GnxJobPlan& P = plan[own[u]];
      pr[u] = on[u] ? P.prow[pp[u]] : -1;
      rowu[u] = P.row;
      const unsigned int mx = P.mixed[pp[u]];
      hs[u] = (P.sel[pp[u]] >> q) & 1u;
      const int32_t* src = H.hmap + (int64_t)max(pr[u], 0) * 2 * NB;
      e0[u] = src[q];
      e1[u] = src[NB + q];
      cut[u] = on[u] && ((mx >> q) & 1u) != 0u;
      const int before = __popc(mx & below);
      fresh[u] = cut[u] ? H.stack[P.pop - ((pp[u] ? P.m0cnt : 0) + before)] : 0;
      // one job per cut block of a local parent: gamete 0's jobs first
      jidx[u] = P.job + (pp[u] && P.prow[0] >= 0 ? P.m0cnt : 0) + before;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t pv = hs[u] ? e1[u] : e0[u];
      const int32_t sv = GNX_BLK(pv);
      if (on[u]) {
        // the parent's block is shared from now on: it loses its never-shared flag (only the
        // first child to share it writes)
        if (!cut[u] && pv < 0) H.hmap[((int64_t)pr[u] * 2 + hs[u]) * NB + q] = sv;
        H.hmap[((int64_t)rowu[u] * 2 + pp[u]) * NB + q] =
            cut[u] ? (int32_t)((uint32_t)fresh[u] | GNX_OWN) : sv;
      }
      const bool jb = cut[u] && pr[u] >= 0;
      const unsigned long long jm = __ballot(jb);
      if (jm == 0ull) continue;                    // (wave-uniform)
      const int nj = __popcll(jm);
      if (qn + nj > 64) flush();
      if (jb) {
        const int slot = qn + __popcll(jm & lt);
        jq_a[wave][slot] = make_uint4((unsigned int)GNX_BLK(e0[u]), (unsigned int)GNX_BLK(e1[u]),
                                      (unsigned int)fresh[u],
                                      (unsigned int)(plan[own[u]].ks[pp[u]] | (q << 24)));
        jq_b[wave][slot] = make_uint2((unsigned int)jidx[u],
                                      (unsigned int)own[u] | ((unsigned int)pp[u] << 16) |
                                          (hs[u] << 17) | ((unsigned int)q << 24));
      }
      qn += nj;
    }
  }
  flush();
#ifdef GNX_JL_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    unsigned int xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned int hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    g_jl_trace[4 * blockIdx.x + 0] = jl_w0;
    g_jl_trace[4 * blockIdx.x + 1] = wall_clock64();
    g_jl_trace[4 * blockIdx.x + 2] = __builtin_readcyclecounter() - jl_t0;
    g_jl_trace[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
}

template <int NB>
static void launch_jobs_fused(gnx_state* h, int64_t first_slot, const int32_t* d_alive,
                              const int32_t* d_blk_off, int buf) {
  const int64_t N = h->N;
  // (device-driven step: first slot and N come from the device, the grid covers what a step's
  // births can take - half the capacity - and the workgroups behind them leave at once)
  const bool ddm = h->dd_active;
  // GNX_JF_LANES=0: one thread per offspring walks its tables alone (k_xo_jobs_fused)
  static const bool lanes = !(getenv("GNX_JF_LANES") && atoi(getenv("GNX_JF_LANES")) == 0);
  if (lanes) {
    // small populations (the device-driven step's sizes): 256-thread workgroups - at 10^5 individuals
    // 26 000 births are 51 workgroups of 512 on 256 CUs (C3 steady 0.2023 against 0.2054 ms/step)
    const bool small = h->cfg.cap_inds <= 600000 && GNX_JF_TPB > 256;      // (GNX_DD_MAX_CAP's default)
    const int tpb = small ? 256 : GNX_JF_TPB;
    const int nbl = ddm ? (int)(h->cfg.cap_inds / 2 / tpb + 2)
                        : (int)((N - 1) / tpb - first_slot / tpb + 1);
#define GNX_JL_LAUNCH(TPB_)                                                                         \
    hipLaunchKernelGGL((k_xo_jobs_lanes<TPB_>), dim3(nbl), dim3(TPB_), 0, h->stream, N, first_slot,  \
                       h->soa[h->cur].grow, d_alive, d_blk_off + 2 * h->blk_stride, h->off_parent,  \
                       h->off_keys, h->off_start, h->free_rows, h->n_free, gnx_halves(h),           \
                       gnx_alias_bp(h), gnx_alias_loci(h), h->n_jobs_dev[buf],                      \
                       (GnxXoJob*)h->jobs[buf], (GnxJobBp*)h->jobs_bp[buf],                         \
                       h->jobs_self_scan ? (const int32_t*)(h->blk_cnt + 2 * h->blk_stride)         \
                                         : (const int32_t*)nullptr, ddm ? h->dd : (GnxDD*)nullptr)
    if (small) GNX_JL_LAUNCH(256);
    else GNX_JL_LAUNCH(GNX_JF_TPB);
#undef GNX_JL_LAUNCH
    h->jobs_inline[buf] = true;
    return;
  }
  const int nbf = ddm ? (int)(h->cfg.cap_inds / 2 / GNX_JF_TPB + 2)
                      : (int)((N - 1) / GNX_JF_TPB - first_slot / GNX_JF_TPB + 1);
  hipLaunchKernelGGL((k_xo_jobs_fused<NB, GNX_JF_TPB>), dim3(nbf), dim3(GNX_JF_TPB), 0, h->stream, N, first_slot,
                     h->soa[h->cur].grow, d_alive, d_blk_off + 2 * h->blk_stride, h->off_parent,
                     h->off_keys, h->off_start, h->free_rows, h->n_free, gnx_halves(h),
                     gnx_alias_bp(h), gnx_alias_loci(h), h->n_jobs_dev[buf],
                     (GnxXoJob*)h->jobs[buf], (GnxJobBp*)h->jobs_bp[buf],
                     h->jobs_self_scan ? (const int32_t*)(h->blk_cnt + 2 * h->blk_stride)
                                       : (const int32_t*)nullptr, ddm ? h->dd : (GnxDD*)nullptr);
  h->jobs_inline[buf] = true;
}

// Stable compaction of the SoA (survivors keep their relative order); genome
// rows are NOT moved: the dead's rows are pushed on the free stack.
__global__ void __launch_bounds__(256)
k_compact(int64_t N, int64_t cap, const int32_t* alive, const int32_t* dead_row,
          const int32_t* blk_off, int stride, const int32_t* cnts, GnxSoA a, GnxSoA b,
          int n_layers, int n_traits, int tbw, int32_t* free_rows, int64_t n_free, int has_rows,
          int xo, int32_t* __restrict__ newslot) {
  __shared__ int lds[16];
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool fa[4], fd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    fa[r] = i < N && (alive[i] & 1) != 0;
    fd[r] = i < N && dead_row[i] != 0;
  }
  int ra[4], rd[4], ta, td;
  gnx_block_ranks(fa, ra, ta, lds);
  gnx_block_ranks(fd, rd, td, lds);
  // deferred crossover: the surviving offspring have just popped one row each from the
  // top of the free stack (k_xo_jobs_surv)
  if (xo) n_free -= cnts[2];
  const int32_t oa = blk_off[blockIdx.x], od = blk_off[stride + blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (newslot && i < N) newslot[i] = fa[r] ? oa + ra[r] : -1;   // for the id-ordered index
    if (fa[r]) {
      const int64_t k = oa + ra[r];            // survivors before i
      GnxRec rec = gnx_rec_load(a, i, cap, n_layers, n_traits, tbw);
      rec.ghost = 0;
      gnx_rec_store(b, k, cap, n_layers, n_traits, tbw, rec);
      gnx_rec_rest(a, i, b, k, cap, n_layers, n_traits, tbw);
    } else if (has_rows && fd[r]) {
      free_rows[n_free + od + rd[r]] = a.grow[i];
    }
  }
}

// Which entries of the id-ordered index survive (k_ord_flags; also the tail of k_fill_lists in
// the device-driven step): entry k names slot ord[k] (or slot k itself behind the index); it
// stays iff that slot's individual is alive - newslot >= 0 once the compaction has run, or the
// death draws' own flags before it has.
struct GnxOrdF {
  const int32_t* ord;
  const int32_t* newslot;      // or null: use alive
  const int32_t* alive;
  int32_t* cnt;
  GnxScanOut S;
  int64_t ord_n;               // entries of the index (host-driven step; the device block's else)
  // device-driven step with tile-major offspring ids: k_offspring filed this step's newborns
  // behind the index itself (their ids do not ascend with their slots) - every one of the
  // N + B entries is explicit
  int tail_explicit;
};
__device__ __forceinline__ void gnx_ord_flags_body(int64_t N, int64_t ord_n, const GnxOrdF& F,
                                                   int* lds, int* lds2) {
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool f[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t k = base + r * 256 + threadIdx.x;
    if (k < N) {
      const int64_t slot = k < ord_n ? F.ord[k] : k;
      f[r] = F.newslot ? F.newslot[slot] >= 0 : (F.alive[slot] & 1) != 0;
    } else {
      f[r] = false;
    }
  }
  int rank[4], tot[1];
  gnx_block_ranks(f, rank, tot[0], lds);
  gnx_count_and_scan<1>(tot, F.cnt, F.S, lds2);
}

// In-place compaction (one device, no tiles): the survivors of the tail [S, N) - S = the number
// of survivors; the tail is mostly this step's offspring - move into the holes the dead left in
// [0, S); everybody else stays where it is.  About 2 x deaths records move instead of every
// survivor (k_compact copies the whole population to the other buffer: 262 MB a step on the
// metric workload, 53 us on the step's critical chain).  Slot order means nothing between a
// compaction and the next cell sort: every draw is keyed by id, the sort runs over the
// id-ordered index (which follows through newslot), offspring ids follow the pair keys.
//   k_fill_lists (stream3, beside the crossover's job builder): holes and movers in slot order
//     (hole r takes mover r), newslot of everybody who stays or dies, the dead's genome rows
//   k_fill: the moves, the movers' newslot, the rows onto the free stack (the job builder
//     has popped its rows from the same stretch by then)
__global__ void __launch_bounds__(256)
k_fill_lists(int64_t N, const int32_t* __restrict__ alive, const int32_t* __restrict__ dead_row,
             const int32_t* __restrict__ blk_off, int stride, const int32_t* __restrict__ cnts,
             const int32_t* __restrict__ grow, int has_rows, int32_t* __restrict__ holes,
             int32_t* __restrict__ movers, int32_t* __restrict__ rows_tmp,
             int32_t* __restrict__ newslot, int32_t* __restrict__ n_move,
             const GnxDD* __restrict__ dd, GnxOrdF ordf) {
  __shared__ int lds[16];
  __shared__ int sb_s[4];
  __shared__ int lds2[8];
  N = gnx_dd_nb(dd, N);
  const int64_t S = cnts[0];
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool fa[4], fd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    fa[r] = i < N && (alive[i] & 1) != 0;
    fd[r] = i < N && dead_row[i] != 0;
  }
  int ra[4], rd[4], ta, td;
  gnx_block_ranks(fa, ra, ta, lds);
  gnx_block_ranks(fd, rd, td, lds);
  const int32_t oa = blk_off[blockIdx.x], od = blk_off[stride + blockIdx.x];
  // the survivors before slot S (blocks that reach into the tail; uniform per block)
  int sb = 0;
  if (base + GNX_CB > S && S < N) {
    const int64_t bS = S / GNX_CB;
    const int within = (int)(S - bS * GNX_CB);
    int c = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = r * 256 + threadIdx.x;
      c += __popcll(__ballot(q < within && (alive[bS * GNX_CB + q] & 1) != 0));
    }
    if (lane == 0) sb_s[wave] = c;
    __syncthreads();
    sb = blk_off[bS] + sb_s[0] + sb_s[1] + sb_s[2] + sb_s[3];
  }
  if (threadIdx.x == 0 && blockIdx.x == (unsigned int)min((long long)(S / GNX_CB), (long long)gridDim.x - 1))
    *n_move = S < N ? (int32_t)(S - sb) : 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i >= N) continue;
    const int k = oa + ra[r];                       // survivors before i
    if (fa[r]) {
      if (i < S) {
        if (newslot) newslot[i] = (int32_t)i;
      } else {
        movers[k - sb] = (int32_t)i;                // (newslot: k_fill)
      }
    } else {
      if (newslot) newslot[i] = -1;
      if (i < S) holes[i - k] = (int32_t)i;
    }
    if (has_rows && fd[r]) rows_tmp[od + rd[r]] = grow[i];
  }
  // (device-driven step: the index's flags and block counts in the same launch - they only
  // need the death draws)
  if (ordf.ord) {
    __syncthreads();
    gnx_ord_flags_body(N, dd ? (ordf.tail_explicit ? N : (int64_t)dd->ord_n) : ordf.ord_n, ordf,
                       lds, lds2);
  }
}

__global__ void __launch_bounds__(256)
k_fill(int64_t cap, const int32_t* __restrict__ n_move, const int32_t* __restrict__ holes,
       const int32_t* __restrict__ movers, const int32_t* __restrict__ cnts, GnxSoA a, int n_layers,
       int n_traits, int tbw, const int32_t* __restrict__ rows_tmp, int32_t* __restrict__ free_rows,
       int64_t n_free, int has_rows, int xo, int32_t* __restrict__ newslot,
       const GnxDD* __restrict__ dd, uint32_t* __restrict__ cell32) {
  if (dd) n_free = dd->n_free;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t H = *n_move;
  for (int64_t r = t0; r < H; r += stride) {
    const int64_t i = movers[r], k = holes[r];
    GnxRec rec = gnx_rec_load(a, i, cap, n_layers, n_traits, tbw);
    rec.ghost = 0;
    // (the next step's movement has run already - gnx_l_move_ahead: its sort key moves too)
    if (cell32) cell32[k] = cell32[i];
    if (newslot) newslot[i] = (int32_t)k;
    gnx_rec_store(a, k, cap, n_layers, n_traits, tbw, rec);
    gnx_rec_rest(a, i, a, k, cap, n_layers, n_traits, tbw);
  }
  if (has_rows) {
    // deferred crossover: the surviving offspring have just popped one row each from the top
    if (xo) n_free -= cnts[2];
    const int64_t D = cnts[1];
    for (int64_t r = t0; r < D; r += stride) free_rows[n_free + r] = rows_tmp[r];
  }
}

static bool jobs_fused_ok(const gnx_state* h) {
  static const bool fused_env = !(getenv("GNX_JOBS_FUSED") && atoi(getenv("GNX_JOBS_FUSED")) == 0);
  // (the two-kernel builder's plan packs the block masks into 16 bits each)
  return (fused_env || h->NB > 16) && h->NB <= GNX_JF_NB;
}

void gnx_launch_xo_jobs_surv(gnx_state* h, int64_t first_slot, const int32_t* d_alive,
                             const int32_t* d_blk_off, int buf) {
  const int64_t N = h->N;
  const int nbj = (int)((N - 1) / GNX_CB - first_slot / GNX_CB + 1);
  if (jobs_fused_ok(h)) {
    switch (h->NB) {
      case 1: launch_jobs_fused<1>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 2: launch_jobs_fused<2>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 3: launch_jobs_fused<3>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 4: launch_jobs_fused<4>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 5: launch_jobs_fused<5>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 6: launch_jobs_fused<6>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 7: launch_jobs_fused<7>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 8: launch_jobs_fused<8>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 9: launch_jobs_fused<9>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 10: launch_jobs_fused<10>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 11: launch_jobs_fused<11>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 12: launch_jobs_fused<12>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 13: launch_jobs_fused<13>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 14: launch_jobs_fused<14>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 15: launch_jobs_fused<15>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 16: launch_jobs_fused<16>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 17: launch_jobs_fused<17>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 20: launch_jobs_fused<20>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 18: launch_jobs_fused<18>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 19: launch_jobs_fused<19>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 21: launch_jobs_fused<21>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 22: launch_jobs_fused<22>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 23: launch_jobs_fused<23>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 24: launch_jobs_fused<24>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 25: launch_jobs_fused<25>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 26: launch_jobs_fused<26>(h, first_slot, d_alive, d_blk_off, buf); break;
      case 27: launch_jobs_fused<27>(h, first_slot, d_alive, d_blk_off, buf); break;
      default: launch_jobs_fused<28>(h, first_slot, d_alive, d_blk_off, buf); break;
    }
    return;
  }
  h->jobs_inline[buf] = false;
  hipLaunchKernelGGL(k_xo_jobs_surv, dim3(nbj), dim3(256), 0, h->stream, N, first_slot,
                     h->soa[h->cur].grow, d_alive, d_blk_off + 2 * h->blk_stride, h->off_parent,
                     h->off_keys, h->off_start, h->free_rows, h->n_free, gnx_halves(h),
                     gnx_alias_bp(h), gnx_alias_loci(h), h->n_jobs_dev[buf], (GnxXoPlan*)h->xo_plan);
  const int64_t B = N - first_slot;
  hipLaunchKernelGGL(k_xo_jobs_write, dim3(gnx_grid(B * 2 * h->NB, 256)), dim3(256), 0, h->stream,
                     B, gnx_halves(h), (const GnxXoPlan*)h->xo_plan, (GnxXoJob*)h->jobs[buf]);
}

// The end of a device-driven step (the last workgroup of its last kernel): the device block
// moves on to the next step and a record of this one goes to pinned host memory (the host
// reads it when it likes - nothing waits for it).
struct GnxDDEnd {
  GnxDD* dd;
  const int32_t* cnts;
  const int32_t* half_top;
  GnxDDRec* ring;
  int has_rows, xo;
  unsigned int* ticket;
};
__device__ __forceinline__ void gnx_dd_end_step(const GnxDDEnd& E) {
  GnxDD* dd = E.dd;
  const int32_t S = E.cnts[0], freed = E.cnts[1], X = E.xo ? E.cnts[2] : 0;
  GnxDDRec r;
  r.N0 = dd->N;
  r.P = dd->P;
  r.B = dd->B;
  r.S = S;
  r.xo = X;
  const int32_t n_free = dd->n_free - X + (E.has_rows ? freed : 0);
  r.n_free = n_free;
  r.half_top = E.half_top ? *E.half_top : 0;
  r.err = dd->err;
  r.max_id = dd->max_id + dd->B;
  r.step = dd->step;
  const int32_t seq = dd->seq + 1;
  dd->N = S;
  dd->ord_n = S;
  dd->n_free = n_free;
  dd->max_id = r.max_id;
  dd->step = dd->step + 1;
  dd->seq = seq;
  dd->P = 0;
  dd->B = 0;
  GnxDDRec* slot = E.ring + ((seq - 1) % GNX_DD_RING);
  __hip_atomic_store(&slot->seq, (int64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  int64_t* w = (int64_t*)slot;
  const int64_t* v = (const int64_t*)&r;
  for (int k = 1; k < (int)(sizeof(GnxDDRec) / 8); ++k)
    __hip_atomic_store(&w[k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&slot->seq, (int64_t)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The id-ordered index follows the compaction: entry k (slot ord[k], or slot k itself for the
// entries appended since the last sort) stays iff its slot survived, and then names the
// slot's new place.  Same three-launch compaction as the population's (gnx_compact.h), on
// the side stream: nothing needs the index before the next cell sort.
__global__ void __launch_bounds__(256)
k_ord_flags(int64_t N, int64_t ord_n, const int32_t* __restrict__ ord,
            const int32_t* __restrict__ newslot, int32_t* __restrict__ cnt, GnxScanOut S,
            const GnxDD* __restrict__ dd, const int32_t* __restrict__ alive) {
  __shared__ int lds[16];
  __shared__ int lds2[8];
  if (dd) {
    N = (int64_t)dd->N + dd->B;
    ord_n = dd->ord_n;
  }
  // (alive: nobody moved - the lazy mortality of gnx_walk - the death draws' own flags decide)
  gnx_ord_flags_body(N, ord_n, GnxOrdF{ord, alive ? nullptr : newslot, alive, cnt, S, ord_n, 0},
                     lds, lds2);
}

__global__ void __launch_bounds__(256)
k_ord_write(int64_t N, int64_t ord_n, const int32_t* __restrict__ ord,
            const int32_t* __restrict__ newslot, const int32_t* __restrict__ off,
            int32_t* __restrict__ ord_new, const GnxDD* __restrict__ dd, GnxDDEnd E,
            int tail_explicit, const int32_t* __restrict__ alive) {
  __shared__ int lds[16];
  __shared__ int last_s;
  if (dd) {
    N = (int64_t)dd->N + dd->B;
    ord_n = tail_explicit ? N : (int64_t)dd->ord_n;
  }
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool f[4];
  int32_t ns[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t k = base + r * 256 + threadIdx.x;
    if (alive) {                  // nobody moved: a living individual keeps its slot
      const int32_t slot = k < N ? (int32_t)(k < ord_n ? ord[k] : k) : -1;
      ns[r] = (slot >= 0 && (alive[slot] & 1) != 0) ? slot : -1;
    } else {
      ns[r] = k < N ? newslot[k < ord_n ? ord[k] : k] : -1;
    }
    f[r] = ns[r] >= 0;
  }
  int rank[4], tot;
  gnx_block_ranks(f, rank, tot, lds);
  const int32_t o = off[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (f[r]) ord_new[o + rank[r]] = ns[r];
  if (!E.dd) return;
  // device-driven step: this is its last kernel, and the workgroup that finishes last moves
  // the device block on to the next step (every workgroup has read it by then)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(E.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = t == gridDim.x - 1u;
    if (last_s) __hip_atomic_store(E.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (last_s && threadIdx.x == 0) gnx_dd_end_step(E);
}

// The lazy mortality of gnx_walk (gnx_internal.h: holes): nobody moves, so all that is left of
// the compaction is the genome rows of the dead going back on the free stack - behind the rows
// the surviving offspring have just popped from its top (k_xo_jobs_*).
__global__ void __launch_bounds__(256)
k_dead_rows(int64_t N, const int32_t* __restrict__ dead_row, const int32_t* __restrict__ blk_off_d,
            int32_t* __restrict__ grow, const int32_t* __restrict__ cnts, int xo,
            int32_t* __restrict__ free_rows, int64_t n_free) {
  __shared__ int lds[16];
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool fd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    fd[r] = i < N && dead_row[i] != 0;
  }
  int rd[4], td;
  gnx_block_ranks(fd, rd, td, lds);
  if (xo) n_free -= cnts[2];
  const int32_t od = blk_off_d[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (fd[r]) {
      free_rows[n_free + od + rd[r]] = grow[i];
      grow[i] = -1;               // (the slot stays until the next cell sort: it owns no row any more)
    }
  }
}

// k_ord_flags + k_ord_write in ONE launch: the stable compaction of the id-ordered index with a
// decoupled look-back over the workgroups' counts (one status | count word per workgroup, 64
// predecessors per round trip: the lanes of wave 0 read them together) instead of a counting
// kernel, a scan by its last workgroup and a writing kernel - one gather of the flags through
// the index instead of two.  Tickets give the workgroups their order; the last one to finish
// clears the words for the next launch.  state: [nb] zero on entry, tick: [2] zero on entry.
__global__ void __launch_bounds__(256)
k_ord_compact(int64_t N, int64_t ord_n, const int32_t* __restrict__ ord,
              const int32_t* __restrict__ newslot, const int32_t* __restrict__ alive,
              int32_t* __restrict__ ord_new, uint32_t* __restrict__ state,
              uint32_t* __restrict__ tick) {
  constexpr uint32_t PART = 1u << 30, INCL = 2u << 30, VAL = (1u << 30) - 1u;
  __shared__ int lds[16];
  __shared__ uint32_t s_bid, s_off, s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_bid = __hip_atomic_fetch_add(&tick[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint32_t bid = s_bid;
  const int64_t base = (int64_t)bid * GNX_CB;
  bool f[4];
  int32_t ns[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t k = base + r * 256 + tid;
    const int32_t slot = k < N ? (int32_t)(k < ord_n ? ord[k] : k) : -1;
    if (alive) ns[r] = (slot >= 0 && (alive[slot] & 1) != 0) ? slot : -1;
    else ns[r] = slot >= 0 ? newslot[slot] : -1;
    f[r] = ns[r] >= 0;
  }
  int rank[4], tot;
  gnx_block_ranks(f, rank, tot, lds);
  if (wave == 0) {
    if (lane == 0)
      __hip_atomic_store(&state[bid], (bid == 0 ? INCL : PART) | (uint32_t)tot, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prefix = 0;
    int b = (int)bid - 1;
    while (b >= 0) {
      const int idx = b - lane;
      const uint32_t v = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : INCL;
      const unsigned long long notready = __ballot(v == 0u);
      const unsigned long long incl = __ballot(v != 0u && (v & INCL) != 0u);
      const int first_nr = notready ? __ffsll((long long)notready) - 1 : 64;
      const int first_in = incl ? __ffsll((long long)incl) - 1 : 64;
      if (first_nr == 0) {                     // the nearest predecessor has not published yet
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      const bool done = first_in < first_nr;
      const int limit = done ? first_in : first_nr - 1;
      uint32_t c = lane <= limit ? (v & VAL) : 0u;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
      prefix += c;
      if (done) break;
      b -= limit + 1;
    }
    if (lane == 0) {
      if (bid > 0)
        __hip_atomic_store(&state[bid], INCL | (prefix + (uint32_t)tot), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_off = prefix;
    }
  }
  __syncthreads();
  const uint32_t o = s_off;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (f[r]) ord_new[o + rank[r]] = ns[r];
  // the last workgroup to get here (everybody has finished looking back by then) leaves the
  // words zero for the next launch
  if (tid == 0) {
    const uint32_t d = __hip_atomic_fetch_add(&tick[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = d == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  for (unsigned int q = tid; q < gridDim.x; q += 256)
    __hip_atomic_store(&state[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    __hip_atomic_store(&tick[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&tick[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int gnx_l_mortality(gnx_state* h, const uint8_t* d_dead_inject, int64_t* deaths_out) {
  *deaths_out = 0;
  GNXCHK(gnx_l_mortality_enqueue(h, d_dead_inject));
  return gnx_l_mortality_finish(h, deaths_out);
}

int gnx_l_mortality_enqueue(gnx_state* h, const uint8_t* d_dead_inject) {
  int64_t N = h->N;
  h->mort_wait = false;
  gnx_bins_adults_drop(h);
  if (N == 0) return 0;
  const gnx_config& c = h->cfg;
  GnxSoA a = h->soa[h->cur], b = h->soa[h->cur ^ 1];
  const int nb = (int)((N + GNX_CB - 1) / GNX_CB);
  const bool xo = h->xo_deferred;
  const int64_t xo_first = h->xo_first, xo_B = xo ? h->xo_B : 0;
  const bool ord_keep = h->ord_mode && h->ord_valid && !h->tiled;   // the id-ordered index follows
  // the job buffer the deferred crossover is about to fill: nobody reads it any more,
  // and its counter starts at zero
  int32_t* zero_jobs = nullptr;
  if (xo) {
    GNXCHK(gnx_xo_prepare_jobs(h, &zero_jobs));
    // the survivors' gametes will pop at most this many blocks (the dead's blocks come back
    // through the collector, gnx_gc, when the stack runs low)
    GNXCHK(gnx_half_reserve(h, 2 * (int64_t)h->NB * xo_B));
  }
  gnx_time_begin(h);
  hipLaunchKernelGGL(k_alive, dim3(nb), dim3(256), 0, h->stream, N, h->p_death, d_dead_inject,
                     a.id, a.ghost, a.grow, h->step, c.seed, h->flag, h->flag2, h->blk_cnt,
                     h->blk_stride, xo ? xo_first : (int64_t)-1, zero_jobs, (const GnxDD*)nullptr,
                     GnxScanOut{});
  // survivors, rows freed and (deferred crossover) the surviving offspring that need a
  // row: block offsets on the device, totals also straight into pinned host memory.
  // (Measured and dropped: death probabilities + death draws + the scan by the last
  // workgroup in ONE kernel took 73 us against 34 + 13 + 10 apart, 0.820 against 0.808
  // ms/step - four individuals per thread starve the f64 spline gathers of parallelism;
  // with one individual per thread of a 1024-thread workgroup and the scan left apart the
  // step takes the same 0.766 ms either way: the launches are not what the chain costs.)
  // in-place compaction (k_fill_lists above): its lists are made on stream3 while this stream
  // builds the crossover's jobs
  // (a caller that names the dead by position - gnx_op_mortality - gets the survivors back in
  // their order: the stable compaction)
  // (tiles too since round 4: the ghosts are dead to the compaction like anybody the draws took,
  // and nothing on a tile depends on the survivors' order either - GNX_TILE_FILL=0: the stable
  // copy there)
  static const bool tile_fill = !(getenv("GNX_TILE_FILL") && atoi(getenv("GNX_TILE_FILL")) == 0);
  const bool fill = h->compact_fill && (tile_fill || (!h->tiled && h->n_ghost == 0)) &&
                    h->stream3 != nullptr && d_dead_inject == nullptr;
  // ... and then the scan of the block counts moves there too: the job builder adds up the
  // counts it needs itself (k_xo_jobs_fused: cnt3), the lists and the host get theirs from
  // stream3, and this stream goes from the death draws straight to the job builder
  static const bool side_scan_env = !(getenv("GNX_SIDE_SCAN") && atoi(getenv("GNX_SIDE_SCAN")) == 0);
  const bool side_scan = side_scan_env && fill && xo && xo_B > 0 && jobs_fused_ok(h);
  // gnx_walk: the NEXT step's age + movement now, on the crossover's stream (idle until the jobs
  // are built), beside the job builder - over slots still in this step's cell order, the dead
  // skipped; the compaction below moves the moved records (gnx_l_move_ahead)
  // (measured, profiles/r05_ab_runs.txt: 0.590 ms/step on the crossover's stream - the movement
  // then holds the crossover back - and 0.70 on a stream of its own against 0.568 without: the
  // movement is bound by its own arithmetic, ~1 200 vector instructions per individual, wherever
  // it runs.  Off unless GNX_MOVE_AHEAD=1 / 2.)
  static const bool ahead_env = getenv("GNX_MOVE_AHEAD") && atoi(getenv("GNX_MOVE_AHEAD")) != 0;
  const bool ahead = ahead_env && h->eager_move && fill && ord_keep && h->sp.move && h->stream2 != nullptr &&
                     h->key_bits <= 24 && h->n_ghost == 0 && !h->tile2_mode;
  // gnx_walk, every step but the last: NO compaction (gnx_internal.h: holes) - GNX_LAZY_COMPACT=0: off
  static const bool lazy_env = !(getenv("GNX_LAZY_COMPACT") && atoi(getenv("GNX_LAZY_COMPACT")) == 0);
  // (tiles: inside gnx_tile_walk, without an index - gnx_internal.h)
  // (measured with two tiles sharing one GPU, profiles/r06_ab_runs.txt: 1.64 ms/step with, 1.59-1.64
  // without - what the compaction cost comes back as 20 % more slots for the movement, the two
  // routing passes and the 64-bit sort to look at; parity-green, GNX_TILE_LAZY=1 turns it on)
  const bool tile_lazy_env = getenv("GNX_TILE_LAZY") && atoi(getenv("GNX_TILE_LAZY")) != 0;   // (read per call: the tests switch it)
  const bool lazy_tile = lazy_env && tile_lazy_env && h->tile_lazy_ok && h->tile2_mode &&
                         h->tile_R * h->tile_C > 1 &&
                         fill && !ord_keep &&
                         h->sp.move && h->sp.mating_radius >= 0 && !ahead;
  const bool lazy = lazy_tile ||
                    (lazy_env && h->eager_move && !ahead && fill && ord_keep && h->sp.move &&
                     h->sp.mating_radius >= 0 && h->key_bits <= 24 && h->n_ghost == 0 && !h->tile2_mode);
  if (side_scan || ahead) HIPCHK(hipEventRecord(h->ev_alive, h->stream));
  if (ahead) {
    if (!h->ev_move)
      HIPCHK(hipEventCreateWithFlags(&h->ev_move, hipEventDisableTiming | hipEventDisableSystemFence));
    // (a stream of its own: on the crossover's it would hold the crossover back - the movement
    // takes as long as the job builder it runs beside; GNX_MOVE_AHEAD=2: on the crossover's)
    static const int ahead_mode = getenv("GNX_MOVE_AHEAD") ? atoi(getenv("GNX_MOVE_AHEAD")) : 1;
    if (!h->stream4) HIPCHK(hipStreamCreateWithFlags(&h->stream4, hipStreamNonBlocking));
    hipStream_t sm = ahead_mode == 2 ? h->stream2 : h->stream4;
    HIPCHK(hipStreamWaitEvent(sm, h->ev_alive, 0));
    GNXCHK(gnx_l_move_ahead(h, N, h->flag, sm));
    HIPCHK(hipEventRecord(h->ev_move, sm));
    h->moved_ahead = true;
  }
  if (side_scan) {
    HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_alive, 0));
    GNXCHK(gnx_block_scan(h, 3, N, h->blk_cnt, h->blk_off, h->cnt_dev, h->h_pin_dev, 0, h->stream3));
    gnx_time_end(h, GNX_K_COMPACT, 0.0);
    HIPCHK(hipEventRecord(h->ev_counts, h->stream3));
  } else {
    GNXCHK(gnx_block_scan(h, 3, N, h->blk_cnt, h->blk_off, h->cnt_dev, h->h_pin_dev));
    gnx_time_end(h, GNX_K_COMPACT, 0.0);
    // the host only needs the counts: it waits for the scan, not for the compaction, and
    // enqueues the next step's first kernels while the compaction still runs
    HIPCHK(hipEventRecord(h->ev_counts, h->stream));
  }
  h->jobs_self_scan = side_scan;
  int has_rows = (h->genomes_assigned && c.L > 0) ? 1 : 0;
  // GNX_ORD_FUSED=1: the index's flags and block offsets with the compaction's lists (they need
  // the death draws, not the compaction) - what the device-driven step does; here it lengthens
  // the lists the compaction and with it the next movement wait for: 0.587 against 0.582 ms/step
  // (profiles/r04_ab_runs.txt), so k_ord_flags stays a launch of its own beside the crossover
  static const bool ord_fused_env = getenv("GNX_ORD_FUSED") && atoi(getenv("GNX_ORD_FUSED")) != 0;
  const bool ord_fused = ord_fused_env && fill && ord_keep && !lazy;
  if (fill && !lazy) {
    if (!side_scan) HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_counts, 0));
    hipLaunchKernelGGL(k_fill_lists, dim3(nb), dim3(256), 0, h->stream3, N, h->flag, h->flag2,
                       h->blk_off, h->blk_stride, h->cnt_dev, a.grow, has_rows,
                       (int32_t*)h->os_ktmp, (int32_t*)h->os_ktmp + c.cap_inds / 2,
                       (int32_t*)h->os_vtmp, ord_keep ? h->newslot : nullptr, h->fill_cnt,
                       (const GnxDD*)nullptr,
                       ord_fused ? GnxOrdF{h->ord[h->ord_cur], nullptr, h->flag, h->ord_cnt,
                                           GnxScanOut{h->ord_off, nullptr, nullptr, 0, nullptr,
                                                      h->tickets + 2, h->blk_stride}, h->ord_n, 0}
                                 : GnxOrdF{});
    HIPCHK(hipEventRecord(h->ev_fill, h->stream3));
  }
  // deferred crossover of this step's births: rows and jobs for the survivors, the kernel
  // itself on stream2 (it runs on under the compaction and the next step's movement)
  if (xo) {
    h->xo_deferred = false;
    GNXCHK(gnx_l_crossover_survivors(h, xo_first, xo_B, h->flag, h->blk_off));
    h->jobs_self_scan = false;
    if (h->xo_sort_waits && h->xo_wait_at == 2) GNXCHK(gnx_xo_wait_inflight(h));
  }
  // the index's compaction of the PREVIOUS mortality round (stream3) still reads newslot when
  // no cell sort has waited for it in between (gnx_op_mortality right after a step)
  if (ord_keep && h->ord_inflight) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ord, 0));
  gnx_time_begin(h);
  const double rec_bytes = 34.0 + 4.0 * c.n_layers + 4.0 * c.n_traits + 16.0 * h->TW;
  if (lazy) {
    // the dead's rows (their block offsets: the scan's second array - from stream3 when the
    // scan ran there)
    if (has_rows) {
      if (side_scan) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_counts, 0));
      hipLaunchKernelGGL(k_dead_rows, dim3(nb), dim3(256), 0, h->stream, N,
                         (const int32_t*)h->flag2, (const int32_t*)(h->blk_off + h->blk_stride),
                         a.grow, (const int32_t*)h->cnt_dev, xo ? 1 : 0,
                         h->free_rows, h->n_free);
    }
    gnx_time_end(h, GNX_K_COMPACT, (double)N * 8.0);
    h->holes = true;
    h->holes_N = N;
    h->holes_flagged = N;
  } else if (fill) {
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_fill, 0));
    if (ahead) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_move, 0));
    const int64_t guess = std::max<int64_t>(h->fill_guess * 2, 4096);      // (grid-stride: any count)
    const int fb_ = (int)std::min<int64_t>(gnx_grid(std::min<int64_t>(guess, N), 256), 8192);
    hipLaunchKernelGGL(k_fill, dim3(fb_), dim3(256), 0, h->stream, c.cap_inds, h->fill_cnt,
                       (const int32_t*)h->os_ktmp, (const int32_t*)h->os_ktmp + c.cap_inds / 2,
                       h->cnt_dev, a, c.n_layers, c.n_traits, a.tb ? 2 * h->TW : 0,
                       (const int32_t*)h->os_vtmp, h->free_rows, h->n_free, has_rows, xo ? 1 : 0,
                       ord_keep ? h->newslot : nullptr, (const GnxDD*)nullptr,
                       ahead ? h->cell32 : (uint32_t*)nullptr);
    // flags and offsets of everybody, the records of about as many movers as the last round had deaths
    gnx_time_end(h, GNX_K_COMPACT, (double)N * 16.0 + (double)h->fill_guess * (16.0 + 2.0 * rec_bytes));
  } else {
  hipLaunchKernelGGL(k_compact, dim3(nb), dim3(256), 0, h->stream, N, c.cap_inds, h->flag,
                     h->flag2, h->blk_off, h->blk_stride, h->cnt_dev, a, b, c.n_layers, c.n_traits,
                     a.tb ? 2 * h->TW : 0, h->free_rows, h->n_free, has_rows, xo ? 1 : 0,
                     ord_keep ? h->newslot : nullptr);
  gnx_time_end(h, GNX_K_COMPACT, (double)N * (24.0 + 2.0 * rec_bytes));
  }
  if (ord_keep) {
    HIPCHK(hipEventRecord(h->ev_compact, h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_compact, 0));
    if (h->ord_inflight) h->ord_inflight = false;       // (stream3 runs them in order)
    GnxScanOut So{h->ord_off, nullptr, nullptr, 0, nullptr, h->tickets + 2, h->blk_stride};
    // GNX_ORD_ONE=1: the one-launch compaction (k_ord_compact) instead of counting kernel + scan by
    // its last workgroup + writing kernel - measured 0.526 against 0.519 ms/step beside the
    // crossover (its look-back waits where the two-kernel form just streams): off by default
    static const bool ord_one = getenv("GNX_ORD_ONE") && atoi(getenv("GNX_ORD_ONE")) != 0;
    if (ord_one && !ord_fused && h->ord_state) {
      hipLaunchKernelGGL(k_ord_compact, dim3(nb), dim3(256), 0, h->stream3, N, h->ord_n,
                         (const int32_t*)h->ord[h->ord_cur], (const int32_t*)h->newslot,
                         lazy ? (const int32_t*)h->flag : (const int32_t*)nullptr,
                         h->ord[h->ord_cur ^ 1], h->ord_state, h->ord_state + h->blk_stride);
    } else {
    if (!ord_fused)
      hipLaunchKernelGGL(k_ord_flags, dim3(nb), dim3(256), 0, h->stream3, N, h->ord_n,
                         h->ord[h->ord_cur], h->newslot, h->ord_cnt, So, (const GnxDD*)nullptr,
                         lazy ? (const int32_t*)h->flag : (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_ord_write, dim3(nb), dim3(256), 0, h->stream3, N, h->ord_n,
                       h->ord[h->ord_cur], h->newslot, h->ord_off, h->ord[h->ord_cur ^ 1],
                       (const GnxDD*)nullptr, GnxDDEnd{}, 0,
                       lazy ? (const int32_t*)h->flag : (const int32_t*)nullptr);
    }
    // the cell sort waits for the crossover AND for this: stream3 waits for the crossover here,
    // where nothing waits for stream3, and the sort's stream waits for one event instead of two
    h->ord_covers_xo = false;
    if (h->xo_sort_waits && !h->xo_last_split && h->xo_ready_buf < 0) {
      for (int k = 0; k < 2; ++k)
        if (h->xo_inflight[k]) HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_xo_done[k], 0));
      h->ord_covers_xo = true;
    }
    HIPCHK(hipEventRecord(h->ev_ord, h->stream3));
    h->ord_inflight = true;
    h->ord_cur ^= 1;
  } else {
    h->ord_valid = false;
  }
  HIPCHK(hipGetLastError());
  if (xo && h->xo_sort_waits && h->xo_wait_at == 3) GNXCHK(gnx_xo_wait_inflight(h));
  h->mort_wait = true;
  h->mort_xo = xo;
  h->mort_fill = fill;
  h->mort_ord_keep = ord_keep;
  h->mort_has_rows = has_rows;
  h->mort_N = N;
  return 0;
}

int gnx_l_mortality_finish(gnx_state* h, int64_t* deaths_out) {
  *deaths_out = 0;
  if (!h->mort_wait) return 0;
  h->mort_wait = false;
  const bool xo = h->mort_xo, fill = h->mort_fill, ord_keep = h->mort_ord_keep;
  const int has_rows = h->mort_has_rows;
  const int64_t N = h->mort_N;
  {
    const bool ht = gnx_host_times();
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventSynchronize(h->ev_counts));
    if (ht) g_host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    gnx_host_mark(2);
  }
  const int64_t survivors = h->h_pin[0];
  const int64_t rows_freed = h->h_pin[1];
  if (xo) {
    const int64_t S = h->h_pin[2];
    h->n_free -= S;
    h->last_xo_births = S;
  }
  if (has_rows) h->n_free += rows_freed;
  *deaths_out = N - h->n_ghost - survivors;       // ghosts are dropped, not counted
  h->fill_guess = N - survivors;
  h->N = survivors;
  h->n_ghost = 0;
  if (survivors == 0) h->holes = false;        // (nobody left to move or sort)
  if (!fill) h->cur ^= 1;
  if (ord_keep) h->ord_n = survivors;
  return 0;
}

// ---------------------------------------------------------------- device-driven step
// (gnx_dd.hip) the density, death and mortality kernels with capacity-sized grids and their
// counts read from h->dd; `par` = the step's parity (which of the alternating buffers)
int gnx_dd_l_bins_adults(gnx_state* h, int par, hipStream_t st) {
  const GnxLattice& L = h->lat;
  const int nb = L.nbx * L.nby;
  GnxSoA s = h->soa[h->cur];
  const int64_t cap = h->cfg.cap_inds;
  const int blocks = (int)std::min<int64_t>(BIN_BLOCKS, std::max<int64_t>(1, (cap + 255) / 256));
  hipLaunchKernelGGL(k_bins, dim3(blocks), dim3(256), (size_t)nb * sizeof(int32_t), st, cap,
                     (const int32_t*)&h->dd->N, (const float*)s.x, (const float*)s.y,
                     (const uint8_t*)nullptr, 1.0 / L.hww, L.nbx, L.nby, h->fb[par], GnxSetWords{});
  HIPCHK(hipGetLastError());
  return 0;
}

// the pair midpoints' bins (fb[2]); their lattice is built by an extra workgroup of
// k_lattice_nmax (gnx_dd_l_density_N), which clears fb[2] again
int gnx_dd_l_density_pairs(gnx_state* h, hipStream_t st) {
  const GnxLattice& L = h->lat;
  const int nb = L.nbx * L.nby;
  const int64_t cap = h->cfg.cap_inds;
  const int blocks = (int)std::min<int64_t>(BIN_BLOCKS, std::max<int64_t>(1, (cap / 2 + 255) / 256));
  hipLaunchKernelGGL(k_bins, dim3(blocks), dim3(256), (size_t)nb * sizeof(int32_t), st, cap,
                     (const int32_t*)&h->dd->P, (const float*)h->mid_x, (const float*)h->mid_y,
                     (const uint8_t*)nullptr, 1.0 / L.hww, L.nbx, L.nby, h->fb[2], GnxSetWords{});
  HIPCHK(hipGetLastError());
  return 0;
}

// lattice + N.max() of everybody (adults + newborns in fb[par]) and, in one more workgroup,
// the pairs' lattice; clears the other parity's bins and N.max() word for the next step
int gnx_dd_l_density_N(gnx_state* h, int par, hipStream_t st) {
  const GnxLattice& L = h->lat;
  const int64_t nn = (int64_t)L.Jx * L.Jy;
  const size_t lat_doubles = (size_t)4 * nn + std::max(L.Jx, L.Jy) + 1;
  const int nmax_R = gnx_nmax_rows_fit(L.Jx, lat_doubles);
  const size_t lds_bytes = (lat_doubles + gnx_nmax_lds_doubles(L.Jx, nmax_R)) * sizeof(double);
  static const int blocks_env = getenv("GNX_LATN_BLOCKS") ? atoi(getenv("GNX_LATN_BLOCKS")) : 256;
  hipLaunchKernelGGL(k_lattice_nmax, dim3(std::max(1, std::min(h->cfg.H, blocks_env)) + 1), dim3(256),
                     lds_bytes, st, L.Jx, L.Jy, L.nbx, (const int32_t*)h->fb[par], L.areas, L.hww,
                     L.cprime, h->spl_N.c, h->cfg.W, h->cfg.H, h->nmax2 + par, h->fb[par ^ 1],
                     h->nmax2 + (par ^ 1), h->fb[2], h->spl_P.c, GnxPubWords{}, nmax_R, gnx_nmax_scan_min());
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_dd_l_death_probs(gnx_state* h, bool with_selection, int par, hipStream_t st) {
  h->spl_N.valid = h->spl_P.valid = true;
  SplineC SN = make_splinec(h, h->spl_N), SP = make_splinec(h, h->spl_P);
  DeathP Q = make_deathp(h, with_selection);
  Q.dd = h->dd;
  hipLaunchKernelGGL(k_death_probs, dim3(gnx_grid(h->cfg.cap_inds, 256)), dim3(256), 0, st, Q,
                     make_demp(h), SN, SP, h->soa[h->cur], h->rast, gnx_trait_tab(h), h->delet_s,
                     (const unsigned long long*)(h->nmax2 + par), h->p_death, h->d_cell);
  HIPCHK(hipGetLastError());
  return 0;
}

// death draws and block counts (xo: the surviving offspring wait for their genome rows;
// the job list of buffer `buf` is emptied), then the scan of the block counts
int gnx_dd_l_alive(gnx_state* h, bool xo, int buf, hipStream_t st) {
  const gnx_config& c = h->cfg;
  GnxSoA a = h->soa[h->cur];
  const int nb = (int)((c.cap_inds + GNX_CB - 1) / GNX_CB);
  hipLaunchKernelGGL(k_alive, dim3(nb), dim3(256), 0, st, (int64_t)c.cap_inds, h->p_death,
                     (const uint8_t*)nullptr, a.id, a.ghost, a.grow, 0ll, c.seed, h->flag, h->flag2,
                     h->blk_cnt, h->blk_stride, xo ? (int64_t)0 : (int64_t)-1,
                     xo ? h->n_jobs_dev[buf] : (int32_t*)nullptr, (const GnxDD*)h->dd,
                     GnxScanOut{h->blk_off, h->cnt_dev, nullptr, 0, nullptr, h->tickets + 0,
                                h->blk_stride});
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_dd_l_jobs(gnx_state* h, int buf, hipStream_t st) {
  hipStream_t keep = h->stream;
  h->stream = st;
  h->jobs_self_scan = false;
  gnx_launch_xo_jobs_surv(h, 0, h->flag, h->blk_off, buf);
  h->stream = keep;
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_dd_l_fill_lists(gnx_state* h, int has_rows, hipStream_t st) {
  const gnx_config& c = h->cfg;
  const int nb = (int)((c.cap_inds + GNX_CB - 1) / GNX_CB);
  hipLaunchKernelGGL(k_fill_lists, dim3(nb), dim3(256), 0, st, (int64_t)c.cap_inds, h->flag, h->flag2,
                     h->blk_off, h->blk_stride, h->cnt_dev, h->soa[h->cur].grow, has_rows,
                     (int32_t*)h->os_ktmp, (int32_t*)h->os_ktmp + c.cap_inds / 2,
                     (int32_t*)h->os_vtmp, h->newslot, h->fill_cnt, (const GnxDD*)h->dd,
                     GnxOrdF{h->ord[h->ord_cur], nullptr, h->flag, h->ord_cnt,
                             GnxScanOut{h->ord_off, nullptr, nullptr, 0, nullptr, h->tickets + 2,
                                        h->blk_stride}, 0, (h->vt_fused && h->id_order == 1) ? 1 : 0});
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_dd_l_fill(gnx_state* h, int has_rows, bool xo, hipStream_t st) {
  const gnx_config& c = h->cfg;
  GnxSoA a = h->soa[h->cur];
  const int grid = (int)std::min<int64_t>(gnx_grid(std::max<int64_t>(c.cap_inds / 8, 4096), 256), 8192);
  hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, st, (int64_t)c.cap_inds, h->fill_cnt,
                     (const int32_t*)h->os_ktmp, (const int32_t*)h->os_ktmp + c.cap_inds / 2,
                     h->cnt_dev, a, c.n_layers, c.n_traits, a.tb ? 2 * h->TW : 0,
                     (const int32_t*)h->os_vtmp, h->free_rows, (int64_t)0, has_rows, xo ? 1 : 0,
                     h->newslot, (const GnxDD*)h->dd, (uint32_t*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

// the id-ordered index follows the compaction (flips ord_cur); the last workgroup of the second
// kernel ends the step (gnx_dd_end_step)
int gnx_dd_l_ord_end(gnx_state* h, int has_rows, bool xo, hipStream_t st) {
  const gnx_config& c = h->cfg;
  const int nb = (int)((c.cap_inds + GNX_CB - 1) / GNX_CB);
  // (the flags and block offsets of the index were made by k_fill_lists)
  hipLaunchKernelGGL(k_ord_write, dim3(nb), dim3(256), 0, st, (int64_t)c.cap_inds, (int64_t)0,
                     h->ord[h->ord_cur], h->newslot, h->ord_off, h->ord[h->ord_cur ^ 1],
                     (const GnxDD*)h->dd,
                     GnxDDEnd{h->dd, h->cnt_dev, h->half_top, h->dd_ring_dev, has_rows, xo ? 1 : 0,
                              h->tickets + 4}, (h->vt_fused && h->id_order == 1) ? 1 : 0,
                     (const int32_t*)nullptr);
  HIPCHK(hipGetLastError());
  h->ord_cur ^= 1;
  return 0;
}

// ---------------------------------------------------------------- burn-in spatial tester
// SpatialTester.update (sim/burnin.py:44-59): per-cell counts, diff to the
// previous counts, mean and std (population std, np.std) of the diff raster.
__global__ void k_cell_counts(int64_t N, const float* x, const float* y, int W, int32_t* counts) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  atomicAdd(&counts[(int64_t)((int)y[i]) * W + (int)x[i]], 1);
}

__global__ void __launch_bounds__(256)
k_diff_stats(int64_t cells, const int32_t* now, const int32_t* prev, double* red) {
  __shared__ double s1[256], s2[256];
  double a = 0.0, b = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += stride) {
    double d = (double)(now[c] - prev[c]);
    a += d;
    b += d * d;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      s1[threadIdx.x] += s1[threadIdx.x + s];
      s2[threadIdx.x] += s2[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&red[0], s1[0]);
    atomicAdd(&red[1], s2[0]);
  }
}

int gnx_l_spatial_diff(gnx_state* h, double* mean, double* sd, double* sums) {
  int64_t cells = (int64_t)h->cfg.W * h->cfg.H;
  int nowi = h->counts_cur ^ 1, prev = h->counts_cur;
  if (!h->counts_init) {
    HIPCHK(hipMemsetAsync(h->counts_rast[prev], 0, cells * sizeof(int32_t), h->stream));
    h->counts_init = true;
  }
  HIPCHK(hipMemsetAsync(h->counts_rast[nowi], 0, cells * sizeof(int32_t), h->stream));
  HIPCHK(hipMemsetAsync(h->red, 0, 2 * sizeof(double), h->stream));
  if (h->N > 0)
    hipLaunchKernelGGL(k_cell_counts, dim3(gnx_grid(h->N, 256)), dim3(256), 0, h->stream, h->N,
                       h->soa[h->cur].x, h->soa[h->cur].y, h->cfg.W, h->counts_rast[nowi]);
  hipLaunchKernelGGL(k_diff_stats, dim3(gnx_grid(cells, 256, 1024)), dim3(256), 0, h->stream, cells,
                     h->counts_rast[nowi], h->counts_rast[prev], h->red);
  double r[2];
  HIPCHK(hipMemcpyAsync(r, h->red, sizeof(r), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  double m = r[0] / (double)cells;
  double var = r[1] / (double)cells - m * m;
  if (mean) *mean = m;
  if (sd) *sd = var > 0 ? sqrt(var) : 0.0;
  if (sums) {                 // integers (sums of count differences and of their squares)
    sums[0] = r[0];
    sums[1] = r[1];
  }
  h->counts_cur = nowi;
  return 0;
}

#ifdef GNX_JF_TRY
// (compile test of one more instantiation of the fused job builder: hipcc -DGNX_JF_TRY=25 -c ...)
template __global__ void k_xo_jobs_fused<GNX_JF_TRY, GNX_JF_TPB>(
    int64_t, int64_t, int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
    const uint8_t*, const int32_t*, int64_t, GnxHalves, const int32_t*, const int32_t*, int32_t*,
    GnxXoJob*, GnxJobBp*, const int32_t*, GnxDD*);
#endif
