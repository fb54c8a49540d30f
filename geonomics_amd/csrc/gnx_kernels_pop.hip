// Population kernels: initial placement, ageing, movement (+ environment
// gather), spatial sort, mate search over a cell list, pair filtering and
// offspring records.  gfx950 only (64-wide wavefronts are assumed).
#include "gnx_internal.h"
#include "gnx_rng.h"
#include "gnx_compact.h"
#include "gnx_tb.h"
#include "gnx_bins.h"

// ---------------------------------------------------------------- helpers
// bits that hold every resident id (ids are handed out upwards from max_id)
static int gnx_id_bits(const gnx_state* h) {
  int b = 1;
  while (b < 40 && (h->max_id >> b) != 0) ++b;
  return b;
}
// the coming cell sort can run over the id-ordered index with the cell as its only key
static bool gnx_ord_sort(const gnx_state* h) {
  return h->ord_mode && h->ord_valid && !h->tiled;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ long long readlane_ll(long long v, int lane) {
  int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), lane);
  int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// ---------------------------------------------------------------- init
// _make_individual (structs/individual.py:213-219): x,y ~ U(0,dim), clipped to
// dim-0.001; sex ~ Bernoulli(0.5); age 0; ids 0..N-1.
__global__ void k_init_population(int64_t N, GnxSoA s, int cap, float Wf, float Hf, float xmax,
                                  float ymax, unsigned long long seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  uint4 r = gnx_rand4(seed, (unsigned long long)i, 0, OP_INIT, 0);
  float x = fminf(gnx_u01(r.x) * Wf, xmax);
  float y = fminf(gnx_u01(r.y) * Hf, ymax);
  s.x[i] = x;
  s.y[i] = y;
  s.age[i] = 0;
  s.sex[i] = gnx_u01(r.z) < 0.5f ? 1 : 0;
  s.id[i] = i;
  s.fit[i] = 1.0f;
  s.grow[i] = -1;
  s.ghost[i] = 0;
}

__global__ void k_gather_e(int64_t first, int64_t n, GnxSoA s, int64_t cap, const float* rast,
                           int n_layers, int W, int H) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int64_t i = first + k;
  int cx = (int)s.x[i], cy = (int)s.y[i];
  for (int l = 0; l < n_layers; ++l)
    s.e[(int64_t)l * cap + i] = rast[((int64_t)l * H + cy) * W + cx];
}

int gnx_l_gather_e(gnx_state* h, int64_t first, int64_t n) {
  if (n == 0) return 0;
  const gnx_config& c = h->cfg;
  hipLaunchKernelGGL(k_gather_e, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, first, n,
                     h->soa[h->cur], c.cap_inds, h->rast, c.n_layers, c.W, c.H);
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_l_init_population(gnx_state* h, int64_t N) {
  const gnx_config& c = h->cfg;
  GnxSoA s = h->soa[h->cur];
  float xmax = (float)(c.W - 0.001), ymax = (float)(c.H - 0.001);
  hipLaunchKernelGGL(k_init_population, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s,
                     (int)c.cap_inds, (float)c.W, (float)c.H, xmax, ymax, c.seed);
  hipLaunchKernelGGL(k_gather_e, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, (int64_t)0, N, s,
                     c.cap_inds, h->rast, c.n_layers, c.W, c.H);
  HIPCHK(hipGetLastError());
  h->N = N;
  h->max_id = N - 1;
  h->ord_valid = true;            // ids 0 .. N-1 in slot order: the index is the identity
  h->ord_n = 0;
  return 0;
}

// ---------------------------------------------------------------- age
__global__ void k_age(int64_t N, int32_t* age) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) age[i] += 1;
}

int gnx_l_age(gnx_state* h) {
  if (h->N == 0) return 0;
  hipLaunchKernelGGL(k_age, dim3(gnx_grid(h->N, 256)), dim3(256), 0, h->stream, h->N,
                     h->soa[h->cur].age);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- movement
struct MoveP {
  int64_t N, cap;
  int W, H, n_layers;
  float xmax, ymax;          // dim - 0.001 (ops/movement.py:90-92)
  float rrx, rry;            // Landscape._res_ratio
  int distr;
  float p1, p2, mu, kappa;
  int surf, surf_layer;
  float surf_kappa;
  int inc_age, apply;
  long long step;
  unsigned long long seed;
  // sort keys of the step's cell sort (gnx_step: the sort follows the movement at once)
  uint64_t* key;
  int32_t* idx;
  double inv_cs;
  int ncx, ncy, idbits;
  uint32_t* cell32;
  int tile_floats;
  const GnxDD* dd;           // device-driven step: N and the step index live on the device
  // the NEXT step's movement, run right after this step's death draws on the uncompacted
  // population (gnx_l_move_ahead): the dead are skipped
  const int32_t* alive;
  // the cell sort's global digit counts (gnx_prim.hip), counted here when the movement writes
  // the sort's keys anyway: hist[place * 2^hist_rb + digit of cell32 at that place]
  uint32_t* hist;
  int hist_rb, hist_places;
  // tiles: the routing's counting pass (csrc/gnx_tile.hip: k_route<false>) on the new positions
  // while they are in registers: rcnt[0 .. T) migrants per tile, [T .. 2T) ghosts per tile
  const RouteGeo* rg;
  int32_t* rcnt;
};
#define GNX_MOVE_HIST_WORDS 1024      // LDS words of k_move's digit table: 2 places of <= 9 bits

__constant__ float c_queen_dirs[8] = {-2.35619449019234492885f, -1.57079632679489661923f,
                                      -0.78539816339744830962f, 3.14159265358979323846f,
                                      0.0f, 2.35619449019234492885f,
                                      1.57079632679489661923f, 0.78539816339744830962f};
__constant__ int c_queen_dy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__constant__ int c_queen_dx[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

// direction from a conductance surface, sampled on the fly from the generating
// process of the reference's per-cell LUT (utils/spatial.py:365-461): 8 queen
// neighbours on the zero-embedded raster, weights = value / sum (1/8 each if the
// sum is 0); mixture: pick a bearing ~ weights then von Mises(kappa) about it;
// unimodal: von Mises about the arithmetic mean of the arg-max bearings.
__device__ __forceinline__ void surf_neighbours(const float* rast, int W, int H, int cx, int cy,
                                                float n[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int yy = cy + c_queen_dy[k], xx = cx + c_queen_dx[k];
    n[k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? rast[(int64_t)yy * W + xx] : 0.f;
  }
}

__device__ __forceinline__ float surf_sample(const float n[8], int mode, float kappa,
                                             GnxStream& s) {
  float sum = 0.f, mx = -1.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sum += n[k];
    mx = fmaxf(mx, n[k]);
  }
  float loc;
  if (mode == GNX_SURF_MIXTURE) {
    float u = gnx_u01(s.next());
    int pick = 7;
    if (sum > 0.f) {
      float t = u * sum, c = 0.f;
      pick = -1;
      int last = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c += n[k];
        if (n[k] > 0.f) last = k;
        if (pick < 0 && t < c) pick = k;
      }
      if (pick < 0) pick = last;
    } else {
      pick = min(7, (int)(u * 8.0f));
    }
    // (a select chain: a __constant__ table indexed per lane is read by a scalar loop over the
    // wave's distinct indices)
    loc = c_queen_dirs[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) loc = (pick == k) ? c_queen_dirs[k] : loc;
  } else {
    float acc = 0.f;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (n[k] == mx) {
        acc += c_queen_dirs[k];
        cnt++;
      }
    loc = acc / (float)cnt;
  }
  return loc + gnx_vonmises(s, 0.0f, kappa);
}

__device__ __forceinline__ float surf_direction(const float* rast, int W, int H, int cx, int cy,
                                                int mode, float kappa, GnxStream& s) {
  float n[8];
  surf_neighbours(rast, W, H, cx, cy, n);
  return surf_sample(n, mode, kappa, s);
}

// LDS-tiled 3x3 gather for a workgroup of (cell-sorted) individuals: the block
// takes the bounding box of its individuals' cells, widened by one cell, stages
// that window of the conductance raster in LDS with coalesced row reads (zeros
// outside the landscape = the reference's zero-embedding) and every lane reads
// its 8 neighbours from LDS.  A block whose box does not fit (e.g. the unsorted
// tail of newborns) falls back to global gathers.  Returns false on fallback.
// IPT individuals per thread (act / cx / cy / n per individual).
#define SURF_TILE_FLOATS 8192
template <int IPT>
__device__ __forceinline__ bool surf_neighbours_lds(const float* rast, int W, int H,
                                                    const bool act[IPT], const int cx[IPT],
                                                    const int cy[IPT], float* tile, int tile_floats,
                                                    int* box, float n[IPT][8]) {
  // box = {xmin, xmax, ymin, ymax} in LDS
  if (threadIdx.x == 0) {
    box[0] = 0x7fffffff;
    box[1] = -1;
    box[2] = 0x7fffffff;
    box[3] = -1;
  }
  __syncthreads();
  {
    // the wave's own bounding box by shuffles, then ONE lane per wave touches the LDS words
    // (256 lanes on four LDS words serialise: that, not the tile, was the surface's cost)
    int lx = 0x7fffffff, hx = -1, ly = 0x7fffffff, hy = -1;
#pragma unroll
    for (int u = 0; u < IPT; ++u)
      if (act[u]) {
        lx = min(lx, cx[u]);
        hx = max(hx, cx[u]);
        ly = min(ly, cy[u]);
        hy = max(hy, cy[u]);
      }
    lx = wave_min_i(lx);
    hx = wave_max_i(hx);
    ly = wave_min_i(ly);
    hy = wave_max_i(hy);
    if ((threadIdx.x & 63) == 0 && hx >= 0) {
      atomicMin(&box[0], lx);
      atomicMax(&box[1], hx);
      atomicMin(&box[2], ly);
      atomicMax(&box[3], hy);
    }
  }
  __syncthreads();
  const int x0 = box[0] - 1, x1 = box[1] + 1, y0 = box[2] - 1, y1 = box[3] + 1;
  const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
  if (box[1] < 0 || (int64_t)bw * bh > tile_floats) return false;   // block-uniform
  // a wave per window row, lanes along it: coalesced row reads and no index division (the
  // flat t / bw form cost ~300 vector instructions per wave, PMC SQ_INSTS_VALU)
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (int ty = wave; ty < bh; ty += n_waves) {
      const int yy = y0 + ty;
      const bool row_in = yy >= 0 && yy < H;
      const float* src = rast + (int64_t)(row_in ? yy : 0) * W;
      for (int tx = lane; tx < bw; tx += 64) {
        const int xx = x0 + tx;
        tile[ty * bw + tx] = (row_in && xx >= 0 && xx < W) ? src[xx] : 0.f;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < IPT; ++u)
    if (act[u]) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        n[u][k] = tile[(cy[u] + c_queen_dy[k] - y0) * bw + (cx[u] + c_queen_dx[k] - x0)];
    }
  return true;
}

// ops/movement.py:34-95 + Species._set_e (structs/species.py:913-922).
// Optionally increments age first (Species._set_age_stage, :567-569).
// IPT individuals per thread, interleaved (thread t of a workgroup takes the workgroup's
// individuals t, t + 256, ...: every access stays coalesced): with two, a wave has twice the
// loads in flight at every step of its chain of dependent round trips - beside the crossover
// (the step's normal state) the kernel is bound by those round trips.
template <int IPT>
__global__ void __launch_bounds__(256)
k_move(MoveP P, GnxSoA s, const float* rast, const float* inj_theta, const float* inj_dist,
       float* out_theta, float* out_dist) {
  extern __shared__ float surf_tile[];       // P.tile_floats floats
  __shared__ int surf_box[4];
  __shared__ uint32_t mhist[GNX_MOVE_HIST_WORDS];
  if (P.hist) {                              // (block-uniform; the conductance tile's barriers publish it)
    for (int q = threadIdx.x; q < P.hist_places << P.hist_rb; q += 256) mhist[q] = 0u;
    __syncthreads();
  }
  P.N = gnx_dd_n(P.dd, P.N);
  P.step = gnx_dd_step(P.dd, P.step);
  const int64_t base = (int64_t)blockIdx.x * (256 * IPT) + threadIdx.x;
  int64_t i[IPT];
  bool act[IPT];
  // every load that does not depend on the draw goes out with the position: beside the
  // crossover a memory round trip costs several microseconds, and id and age each had one of
  // their own further down the thread's chain
  float x[IPT], y[IPT];
  unsigned long long id[IPT];
  int32_t age0[IPT];
  int cx[IPT], cy[IPT];
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    i[u] = base + u * 256;
    act[u] = i[u] < P.N && (!P.alive || (P.alive[i[u]] & 1) != 0);
    x[u] = y[u] = 0.f;
    id[u] = 0ull;
    age0[u] = 0;
    if (act[u]) {
      x[u] = s.x[i[u]];
      y[u] = s.y[i[u]];
      id[u] = (unsigned long long)s.id[i[u]];
      if (P.apply && P.inc_age) age0[u] = s.age[i[u]];
    }
  }
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    cx[u] = (int)x[u];
    cy[u] = (int)y[u];
  }
  float nb[IPT][8];
  if (P.surf != GNX_SURF_NONE && !inj_theta) {     // block-uniform condition
    const float* cond = rast + (int64_t)P.surf_layer * P.H * P.W;
    const bool have_nb = surf_neighbours_lds<IPT>(cond, P.W, P.H, act, cx, cy, surf_tile,
                                                  P.tile_floats, surf_box, nb);
    if (!have_nb) {
#pragma unroll
      for (int u = 0; u < IPT; ++u)
        if (act[u]) surf_neighbours(cond, P.W, P.H, cx[u], cy[u], nb[u]);
    }
  }
  float nx[IPT], ny[IPT];
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    nx[u] = ny[u] = 0.f;
    if (!act[u]) continue;
    float theta, dist;
    if (inj_theta) {
      theta = inj_theta[i[u]];
      dist = inj_dist[i[u]];
    } else {
      if (P.surf != GNX_SURF_NONE) {
        GnxStream st(P.seed, id[u], P.step, OP_MOVE_SURF);
        theta = surf_sample(nb[u], P.surf, P.surf_kappa, st);
      } else {
        GnxStream st(P.seed, id[u], P.step, OP_MOVE_DIR);
        theta = gnx_vonmises(st, P.mu, P.kappa);
      }
      dist = gnx_distance(P.distr, P.p1, P.p2, gnx_rand4(P.seed, id[u], P.step, OP_MOVE_DIST, 0));
    }
    if (out_theta) {
      out_theta[i[u]] = theta;
      out_dist[i[u]] = dist;
    }
    float dx = cosf(theta) * dist;
    float dy = sinf(theta) * dist;
    if (P.rrx != 1.0f) dx *= P.rrx;
    if (P.rry != 1.0f) dy *= P.rry;
    nx[u] = fminf(fmaxf(x[u] + dx, 0.0f), P.xmax);
    ny[u] = fminf(fmaxf(y[u] + dy, 0.0f), P.ymax);
  }
  if (!P.apply) return;
  // the environment at the new cells: all gathers of the thread before its first store
  float e[IPT][4];
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    const int ncx = (int)nx[u], ncy = (int)ny[u];
#pragma unroll
    for (int l = 0; l < 4; ++l)
      e[u][l] = (act[u] && l < P.n_layers) ? rast[((int64_t)l * P.H + ncy) * P.W + ncx] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    if (!act[u]) continue;
    const int64_t k = i[u];
    s.x[k] = nx[u];
    s.y[k] = ny[u];
    if (P.inc_age) s.age[k] = age0[u] + 1;
    const int ncx = (int)nx[u], ncy = (int)ny[u];
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < P.n_layers) s.e[(int64_t)l * P.cap + k] = e[u][l];
    for (int l = 4; l < P.n_layers; ++l)
      s.e[(int64_t)l * P.cap + k] = rast[((int64_t)l * P.H + ncy) * P.W + ncx];
    if (P.key || P.cell32) {
      const int hx = min(P.ncx - 1, (int)((double)nx[u] * P.inv_cs));
      const int hy = min(P.ncy - 1, (int)((double)ny[u] * P.inv_cs));
      if (P.cell32) {               // the sort runs over the id-ordered index: the cell is all it needs
        const uint32_t cell = (uint32_t)(hy * P.ncx + hx);
        P.cell32[k] = cell;
        if (P.hist) {
          // (slots are nearly in cell order: a wave's lanes mostly hit one or two counters of the
          // high place - a few serialised LDS atomics against ~1 200 vector instructions)
          const uint32_t dm = (1u << P.hist_rb) - 1u;
          for (int pl = 0; pl < P.hist_places; ++pl)
            atomicAdd(&mhist[(pl << P.hist_rb) + ((cell >> (pl * P.hist_rb)) & dm)], 1u);
        }
      } else {
        P.key[k] = ((uint64_t)(hy * P.ncx + hx) << P.idbits) | (uint64_t)id[u];
        P.idx[k] = (int32_t)k;
      }
    }
  }
  if (P.rg) {
    // (every lane of the wave takes part: the group appends are wave-wide)
    const RouteGeo& g = *P.rg;
    const int T = g.R * g.C;
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      int oc = 0, orow = 0, hx = 0, hy = 0;
      if (act[u]) {
        oc = gnx_tile_index(nx[u], g.tw, g.C);
        orow = gnx_tile_index(ny[u], g.th, g.R);
        hx = min(g.ncx - 1, (int)((double)nx[u] * g.inv_cs));
        hy = min(g.ncy - 1, (int)((double)ny[u] * g.inv_cs));
      }
      for (int k = 0; k < 9; ++k) {
        const int dest = act[u] ? gnx_route_dest(g, k, orow, oc, hx, hy) : -1;
        if (__ballot(dest >= 0) == 0ull) continue;
        (void)gnx_route_append(P.rcnt + (k == 4 ? 0 : T), dest);
      }
    }
  }
  if (P.hist) {
    __syncthreads();
    for (int q = threadIdx.x; q < P.hist_places << P.hist_rb; q += 256) {
      const uint32_t v = mhist[q];
      if (v) atomicAdd(&P.hist[q], v);
    }
  }
}

// the digit counts k_move left at the head of os_scratch will not be used (another sort path, a
// second movement, the device-driven step): that stretch is zero again
int gnx_os_hist_discard(gnx_state* h, hipStream_t st) {
  if (!h->hist_fresh) return 0;
  h->hist_fresh = false;
  HIPCHK(hipMemsetAsync(h->os_scratch, 0, GNX_MOVE_HIST_WORDS * sizeof(uint32_t), st ? st : h->stream));
  return 0;
}

// k_move counts the sort's digits when the keys fit its LDS table (2 places of <= 9 bits: up to
// 2^18 hash cells - GNX_OS_MOVE_HIST=0: never)
static bool gnx_move_hist(const gnx_state* h, int* places, int* rb) {
  static const bool on = !(getenv("GNX_OS_MOVE_HIST") && atoi(getenv("GNX_OS_MOVE_HIST")) == 0);
  gnx_os_digits(h->key_bits, places, rb);
  return on && h->os_scratch != nullptr && (*places << *rb) <= GNX_MOVE_HIST_WORDS;
}

int gnx_l_move(gnx_state* h, bool inc_age, const float* inj_theta, const float* inj_dist,
               float* out_theta, float* out_dist, bool apply) {
  // device-driven step (gnx_dd.hip): the grid covers the capacity, the kernel reads N itself
  const bool ddm = h->dd_active;
  if (!ddm && h->N == 0) return 0;
  if (apply) {                            // bins counted before a movement are stale
    h->fb_adults = false;
    h->fb_pending = false;
  }
  const gnx_config& c = h->cfg;
  const gnx_species_params& sp = h->sp;
  MoveP P;
  P.dd = ddm ? h->dd : nullptr;
  P.alive = nullptr;
  P.N = h->N;
  // (gnx_walk: the last mortality left its dead in place - gnx_internal.h: holes - the
  // movement looks at every slot of that step and skips them)
  const bool holes = h->holes && apply && !ddm && !inj_theta;
  if (h->holes && !holes) {
    gnx_set_error("internal: a movement other than the step's own over an uncompacted population");
    return 1;
  }
  if (holes) {
    P.alive = h->flag;
    P.N = h->holes_N;
  }
  P.cap = c.cap_inds;
  P.W = c.W;
  P.H = c.H;
  P.n_layers = c.n_layers;
  P.xmax = (float)(c.W - 0.001);
  P.ymax = (float)(c.H - 0.001);
  P.rrx = (float)sp.res_ratio[0];
  P.rry = (float)sp.res_ratio[1];
  P.distr = sp.move_distr;
  P.p1 = (float)sp.move_p1;
  P.p2 = (float)sp.move_p2;
  P.mu = (float)sp.dir_mu;
  P.kappa = (float)sp.dir_kappa;
  P.surf = sp.move_surf;
  P.surf_layer = sp.move_surf_layer;
  P.surf_kappa = (float)sp.move_surf_kappa;
  P.inc_age = inc_age ? 1 : 0;
  P.apply = apply ? 1 : 0;
  P.step = h->step;
  P.seed = c.seed;
  // inside gnx_step the cell sort comes next: write its keys here (k_keys otherwise)
  const bool with_keys = h->move_writes_keys && apply && !inj_theta;
  const bool ordm = with_keys && gnx_ord_sort(h);
  P.key = (with_keys && !ordm) ? h->key64[0] : nullptr;
  P.cell32 = ordm ? h->cell32 : nullptr;
  P.idx = h->perm[0];
  P.inv_cs = h->inv_cs;
  P.ncx = h->ncx;
  P.ncy = h->ncy;
  P.idbits = gnx_id_bits(h);
  GNXCHK(gnx_os_hist_discard(h));          // (a movement whose keys nobody sorted)
  P.hist = nullptr;
  P.hist_rb = P.hist_places = 0;
  if (ordm && !ddm && gnx_move_hist(h, &P.hist_places, &P.hist_rb)) {
    P.hist = (uint32_t*)h->os_scratch;
    h->hist_fresh = true;
  }
  P.rg = nullptr;
  P.rcnt = nullptr;
  if (h->move_counts_routes && apply && !ddm && !inj_theta && h->route_geo_dev && h->route_cnt) {
    P.rg = h->route_geo_dev;
    P.rcnt = h->route_cnt;
    h->move_counted_routes = true;
  }
  h->keys_fresh = with_keys;
  if (with_keys) h->keys_ordmode = ordm;
  // LDS window of the conductance raster per workgroup (GNX_MOVE_TILE floats): smaller
  // windows let more workgroups share a CU while the kernel crawls beside the crossover
  // (2048 floats by default: 0.795 against 0.805-0.84 ms/step with 8192, no further gain below)
  static const int tile_env = getenv("GNX_MOVE_TILE") ? atoi(getenv("GNX_MOVE_TILE")) : 2048;
  // (per 256 individuals of the workgroup: two per thread cover twice the stretch of cells)
  static const int ipt_env = getenv("GNX_MOVE_IPT") ? atoi(getenv("GNX_MOVE_IPT")) : 2;
  P.tile_floats = sp.move_surf != GNX_SURF_NONE
                      ? std::max(256, std::min(tile_env * (ipt_env == 2 ? 2 : 1), 12288)) : 0;
  gnx_time_begin(h);
  // individuals per thread (GNX_MOVE_IPT; 2 by default: profiles/r04_ab_runs.txt)
  static const int ipt = getenv("GNX_MOVE_IPT") ? atoi(getenv("GNX_MOVE_IPT")) : 2;
  const int64_t n_grid = ddm ? (int64_t)c.cap_inds : (holes ? h->holes_N : h->N);
  if (ipt == 2)
    hipLaunchKernelGGL(k_move<2>, dim3(gnx_grid(n_grid, 512)), dim3(256),
                       (size_t)P.tile_floats * sizeof(float), h->stream, P,
                       h->soa[h->cur], h->rast, inj_theta, inj_dist, out_theta, out_dist);
  else
    hipLaunchKernelGGL(k_move<1>, dim3(gnx_grid(n_grid, 256)), dim3(256),
                       (size_t)P.tile_floats * sizeof(float), h->stream, P,
                       h->soa[h->cur], h->rast, inj_theta, inj_dist, out_theta, out_dist);
  // per individual: x,y rw 16 + id 8 + age rw 8 + e store 4*n_lyr + raster gathers 4*n_lyr
  // (+36 for the 3x3 conductance neighbourhood)
  gnx_time_end(h, GNX_K_MOVE,
               (double)h->N * (32.0 + 8.0 * c.n_layers + (sp.move_surf ? 36.0 : 0.0)));
  HIPCHK(hipGetLastError());
  return 0;
}

// The NEXT step's age + movement, launched by THIS step's mortality right after the death draws,
// on stream `st`, over the still uncompacted population (everybody incl. this step's offspring;
// the dead are skipped) - gnx_walk only, where nobody looks at the population between two
// steps.  Where it runs today - beside the crossover, which it slows and which slows it: 0.10-0.13
// ms against 0.05 alone, the last kernel of that phase to finish - it is latency-bound on HBM
// that the crossover saturates; here it runs beside the crossover's job builder, a chain of
// dependent table look-ups, and over slots that are still in this step's (hash cell, id) order,
// offspring in their parents' order: every workgroup's conductance window fits its LDS tile
// (after the in-place compaction a sixth of every workgroup's slots hold newborns from anywhere).
// Same draws (keyed by id and step), same positions: the compaction then moves the moved
// records - and their sort keys (cell32).  Reference: Species._set_age_stage + _do_movement of
// step t + 1 (structs/species.py:567-585) after _do_pop_dynamics of step t - nothing in between.
int gnx_l_move_ahead(gnx_state* h, int64_t N_all, const int32_t* d_alive, hipStream_t st) {
  const gnx_config& c = h->cfg;
  const gnx_species_params& sp = h->sp;
  MoveP P;
  P.dd = nullptr;
  P.alive = d_alive;
  P.N = N_all;
  P.cap = c.cap_inds;
  P.W = c.W;
  P.H = c.H;
  P.n_layers = c.n_layers;
  P.xmax = (float)(c.W - 0.001);
  P.ymax = (float)(c.H - 0.001);
  P.rrx = (float)sp.res_ratio[0];
  P.rry = (float)sp.res_ratio[1];
  P.distr = sp.move_distr;
  P.p1 = (float)sp.move_p1;
  P.p2 = (float)sp.move_p2;
  P.mu = (float)sp.dir_mu;
  P.kappa = (float)sp.dir_kappa;
  P.surf = sp.move_surf;
  P.surf_layer = sp.move_surf_layer;
  P.surf_kappa = (float)sp.move_surf_kappa;
  P.inc_age = 1;
  P.apply = 1;
  P.step = h->step + 1;
  P.seed = c.seed;
  P.key = nullptr;
  P.cell32 = h->cell32;          // (the cell sort runs over the id-ordered index: gnx_l_mortality checks)
  P.idx = h->perm[0];
  P.inv_cs = h->inv_cs;
  P.ncx = h->ncx;
  P.ncy = h->ncy;
  P.idbits = 0;
  // (the digit counts of the coming step's cell sort: the dead are skipped, so the counts are
  // those of the population that sort will see; os_scratch was wiped by this step's k_permute)
  GNXCHK(gnx_os_hist_discard(h, st));
  P.hist = nullptr;
  P.hist_rb = P.hist_places = 0;
  if (gnx_move_hist(h, &P.hist_places, &P.hist_rb)) {
    P.hist = (uint32_t*)h->os_scratch;
    h->hist_fresh = true;
  }
  P.rg = nullptr;
  P.rcnt = nullptr;
  static const int tile_env = getenv("GNX_MOVE_TILE") ? atoi(getenv("GNX_MOVE_TILE")) : 2048;
  P.tile_floats = sp.move_surf != GNX_SURF_NONE ? std::max(256, std::min(tile_env * 2, 12288)) : 0;
  hipLaunchKernelGGL(k_move<2>, dim3(gnx_grid(N_all, 512)), dim3(256),
                     (size_t)P.tile_floats * sizeof(float), st, P, h->soa[h->cur], h->rast,
                     (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- spatial sort
__device__ __forceinline__ int gnx_cell_of(float x, float y, double inv_cs, int ncx, int ncy) {
  int cx = min(ncx - 1, (int)((double)x * inv_cs));
  int cy = min(ncy - 1, (int)((double)y * inv_cs));
  return cy * ncx + cx;
}

// sort key: hash cell, then individual id - the order of the sorted arrays is
// CANONICAL (it does not depend on the order the individuals were stored in, nor
// on how the landscape is tiled), which is what lets the mate search address
// candidates by index
// (the id takes the low idbits bits - every resident id is <= max_id < 2^idbits - and the
// cell the bits above, so that the radix sort passes over idbits + cell bits only)
__global__ void k_keys(int64_t N, const float* x, const float* y, const int64_t* id,
                       double inv_cs, int ncx, int ncy, int idbits, uint64_t* key, int32_t* idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  key[i] = ((uint64_t)gnx_cell_of(x[i], y[i], inv_cs, ncx, ncy) << idbits) | (uint64_t)id[i];
  idx[i] = (int32_t)i;
}

// tile2: an individual that left the tile (and is no ghost) sorts behind every cell - the
// sort that orders the population also removes the emigrants (no compaction of its own)
// (alive: an uncompacted population, gnx_tile_walk - the first n_flagged slots carry the last death
// draws' flags, and the dead sort behind the emigrants: one more cell value)
__global__ void k_keys_evict(int64_t N, const float* x, const float* y, const int64_t* id,
                             const uint8_t* ghost, float x0, float x1, float y0, float y1,
                             double inv_cs, int ncx, int ncy, int idbits, uint64_t* key,
                             int32_t* idx, const int32_t* __restrict__ alive, int64_t n_flagged) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float xi = x[i], yi = y[i];
  const bool dead = alive != nullptr && i < n_flagged && (alive[i] & 1) == 0;
  const bool out = !ghost[i] && (xi < x0 || xi >= x1 || yi < y0 || yi >= y1);
  const uint64_t cell = dead ? (uint64_t)(ncx * ncy) + 1ull
                             : (out ? (uint64_t)(ncx * ncy) : (uint64_t)gnx_cell_of(xi, yi, inv_cs, ncx, ncy));
  key[i] = (cell << idbits) | (uint64_t)id[i];
  idx[i] = (int32_t)i;
}

// the evicted individuals' genome rows go back on the free stack (their blocks return
// through the collector); they sit in the slots [N, N + n) behind the population
__global__ void k_free_tail(int64_t N, int64_t n, const int32_t* grow, int32_t* free_rows,
                            int64_t n_free) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) free_rows[n_free + k] = grow[N + k];
}

__global__ void k_permute(int64_t N, int64_t cap, const int32_t* perm, GnxSoA a, GnxSoA b,
                          int n_layers, int n_traits, int tbw, unsigned long long pair_seed,
                          uint32_t* tag, uint4* cand, uint64_t* key, int idbits,
                          int32_t* cell_start, int ncells, const uint32_t* __restrict__ cellk,
                          const int32_t* __restrict__ ord, int64_t ord_n,
                          int32_t* __restrict__ ord_new, int32_t* __restrict__ perm_out,
                          uint4* __restrict__ wipe, int64_t wipe_n, int hot_only,
                          const GnxDD* __restrict__ dd, GnxBinP bins, int32_t* __restrict__ vt_zero) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (tile-major offspring ids: the virtual tiles' pair counts of this step start at zero -
  // k_pair_flags adds to them)
  if (vt_zero && i < 64) vt_zero[i] = 0;
  if (dd) {
    // device-driven step: the sort ran over the handle's capacity, the entries behind the
    // population carry the largest key and came out behind it
    N = dd->N;
    ord_n = dd->ord_n;
    pair_seed = gnx_pair_seed(pair_seed, dd->step);      // (pair_seed = the seed here)
  }
  // the radix sort's scratch (histograms, look-back states) is zero again for the next sort:
  // no fill kernels in front of it (two, 6 us each, on the step's critical path)
  for (int64_t w = i; w < wipe_n; w += (int64_t)gridDim.x * blockDim.x)
    wipe[w] = make_uint4(0u, 0u, 0u, 0u);
  if (i >= N) return;
  // sorted (cell << idbits | id) keys with the slots as values, or - the sort ran over the
  // id-ordered index - sorted cells (cellk) with id ranks as values
  {
    // cell_start[c] = first sorted slot whose cell >= c; cell_start[ncells] = N
    const int k = cellk ? (int)cellk[i] : (int)(key[i] >> idbits);
    const int prev = (i == 0) ? -1 : (cellk ? (int)cellk[i - 1] : (int)(key[i - 1] >> idbits));
    for (int c = prev + 1; c <= k; ++c) cell_start[c] = (int32_t)i;
    if (i == N - 1)
      for (int c = k + 1; c <= ncells; ++c) cell_start[c] = (int32_t)N;
  }
  int64_t j = perm[i];
  if (cellk) {
    const int64_t kk = j;                       // id rank
    j = kk < ord_n ? ord[kk] : kk;              // the slot it was in
    ord_new[kk] = (int32_t)i;                   // and the slot it is in now
    perm_out[i] = (int32_t)j;                   // (the sort permutation, as the other path leaves it)
  }
  const uint32_t ck = cellk ? cellk[i] : 0u;
  GnxRec r;
  if (hot_only) {
    // what the mate search and the pair list read: position, age, sex, id, ghost flag; the
    // other columns follow on the side stream (k_permute_rest) while those run
    r.x = a.x[j];
    r.y = a.y[j];
    r.age = a.age[j];
    r.sex = a.sex[j];
    r.id = a.id[j];
    r.ghost = a.ghost[j];
    b.x[i] = r.x;
    b.y[i] = r.y;
    b.age[i] = r.age;
    b.sex[i] = r.sex;
    b.id[i] = r.id;
    b.ghost[i] = r.ghost;
  } else {
    // the whole record in registers before the first store (GnxRec, gnx_internal.h)
    r = gnx_rec_load(a, j, cap, n_layers, n_traits, tbw);       // tbw = 2 * TW
    gnx_rec_store(b, i, cap, n_layers, n_traits, tbw, r);
  }
  if (cellk) key[i] = ((uint64_t)ck << idbits) | (uint64_t)r.id;   // what the pair list reads
  const uint32_t tg = gnx_ind_tag(pair_seed, (unsigned long long)r.id);   // (rides in the candidate record)
  // packed candidate record for the mate search: one 16-byte load per candidate
  cand[i] = make_uint4(__float_as_uint(r.x), __float_as_uint(r.y), tg, (uint32_t)r.id);
  if (!hot_only) gnx_rec_rest(a, j, b, i, cap, n_layers, n_traits, tbw);
  // (device-driven step at small sizes: the adults' density bins here instead of a launch of
  // their own - the wave's lanes are neighbours in space: one atomic per distinct bin)
  if (bins.bins) gnx_bin_add(bins.bins, gnx_bin_of(bins, r.x, r.y), true);
}

// the columns k_permute(hot_only) left behind: fitness, genome row, environment,
// phenotypes, alleles at the selected loci; perm[i] = the slot sorted position i came from
__global__ void k_permute_rest(int64_t N, int64_t cap, const int32_t* __restrict__ perm, GnxSoA a,
                               GnxSoA b, int n_layers, int n_traits, int tbw) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t j = perm[i];
  const float fit = a.fit[j];
  const int32_t grow = a.grow[j];
  float e[4], z[4];
  uint64_t tb[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) e[l] = l < n_layers ? a.e[(int64_t)l * cap + j] : 0.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t) z[t] = t < n_traits ? a.z[(int64_t)t * cap + j] : 0.0f;
#pragma unroll
  for (int w = 0; w < 4; ++w) tb[w] = w < tbw ? a.tb[j * tbw + w] : 0ull;
  b.fit[i] = fit;
  b.grow[i] = grow;
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (l < n_layers) b.e[(int64_t)l * cap + i] = e[l];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < n_traits) b.z[(int64_t)t * cap + i] = z[t];
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < tbw) b.tb[i * tbw + w] = tb[w];
  gnx_rec_rest(a, j, b, i, cap, n_layers, n_traits, tbw);
}

__global__ void k_cells(int64_t N, const float* x, const float* y, double inv_cs, int ncx, int ncy,
                        uint32_t* cell32) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) cell32[i] = (uint32_t)gnx_cell_of(x[i], y[i], inv_cs, ncx, ncy);
}

// keys of the id-ordered sequence: entry k is slot ord[k] (k < ord_n), or slot k itself
// (appended since the index was last compacted: offspring, ascending ids)
__global__ void k_keys_ord(int64_t N, int64_t ord_n, const int32_t* __restrict__ ord,
                           const uint32_t* __restrict__ cell32, uint32_t* __restrict__ key,
                           int32_t* __restrict__ val) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const int64_t slot = k < ord_n ? ord[k] : k;
  key[k] = cell32[slot];
  val[k] = (int32_t)k;
}

__global__ void k_iota(int64_t N, int32_t* v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) v[i] = (int32_t)i;
}

// stream3 takes the columns k_permute(hot_only) left behind, from here on in `stream`'s order
int gnx_permute_rest_launch(gnx_state* h) {
  if (!h->perm_rest_pending) return 0;
  h->perm_rest_pending = false;
  const gnx_config& c = h->cfg;
  HIPCHK(hipEventRecord(h->ev_perm, h->stream));
  HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_perm, 0));
  hipLaunchKernelGGL(k_permute_rest, dim3(gnx_grid(h->perm_rest_N, 256)), dim3(256), 0, h->stream3,
                     h->perm_rest_N, c.cap_inds, h->perm[1], h->perm_rest_a, h->perm_rest_b,
                     c.n_layers, c.n_traits, h->perm_rest_a.tb ? 2 * h->TW : 0);
  HIPCHK(hipEventRecord(h->ev_perm_rest, h->stream3));
  HIPCHK(hipGetLastError());
  h->perm_rest_inflight = true;
  return 0;
}

// the side stream's share of the last cell sort's permutation has arrived
int gnx_wait_permute_rest(gnx_state* h, bool late_ok) {
  GNXCHK(gnx_permute_rest_launch(h));
  // (late mode: the step's own hand-over point lets it run on - gnx_l_death_probs waits)
  if (late_ok && h->perm_rest_late) return 0;
  if (h->perm_rest_inflight) {
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_perm_rest, 0));
    h->perm_rest_inflight = false;
  }
  h->perm_rest_late = false;
  return 0;
}

// Sort of the whole SoA by (hash cell, id); cell size >= mating radius.
int gnx_l_sort_by_cell(gnx_state* h, bool split_rest) {
  int64_t N = h->N;
  gnx_bins_adults_drop(h);
  GNXCHK(gnx_wait_permute_rest(h));
  if (N == 0) {
    h->holes = false;                 // (nobody left: nothing to gather)
    return 0;
  }
  GNXCHK(gnx_xo_flush_deferred(h));   // slots move: offspring still waiting for their crossover get it now
  // the radix sort runs alone, or beside a narrow tail.  (The index's compaction on stream3
  // has already waited for the crossover it was launched with - gnx_l_mortality - and this
  // stream is about to wait for that compaction: one event covers both.)
  const bool xo_covered = h->xo_sort_waits && h->ord_covers_xo && h->ord_inflight && h->tile_evict == 0 &&
                          (h->keys_fresh ? h->keys_ordmode : gnx_ord_sort(h));
  if (xo_covered) {
    for (int k = 0; k < 2; ++k) h->xo_inflight[k] = h->xo_wide_inflight[k] = false;
    h->ord_covers_xo = false;
  } else if (h->xo_sort_waits) {
    GNXCHK(gnx_xo_wait_wide(h));
  }
  const gnx_config& c = h->cfg;
  GnxSoA a = h->soa[h->cur], b = h->soa[h->cur ^ 1];
  const int idbits = gnx_id_bits(h);
  const bool alone = h->xo_sort_waits || !h->xo_running;
  // an uncompacted tile (gnx_tile_walk): the sort runs over every slot of the stretch, the dead
  // carry a key behind the emigrants' and leave with them
  const bool tile_holes = h->holes && h->tile2_mode;
  const int64_t n_sort = tile_holes ? h->holes_N : N;
  const bool ordm = (h->tile_evict > 0 || tile_holes) ? false
                                                      : (h->keys_fresh ? h->keys_ordmode : gnx_ord_sort(h));
  if (h->holes && !tile_holes && !(ordm && h->keys_fresh)) {
    gnx_set_error("internal: the cell sort of an uncompacted population needs the id-ordered index "
                  "and the movement's keys");
    return 1;
  }
  h->holes = false;          // (k_permute below gathers the living: the population is compact again)
  gnx_time_begin(h);
  int64_t wipe_words = 0;       // of os_scratch, dirtied by this sort
  if (ordm) {
    // stable sort of the id-ordered index by cell alone (gnx_internal.h)
    if (!h->keys_fresh)
      hipLaunchKernelGGL(k_cells, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, a.x, a.y,
                         h->inv_cs, h->ncx, h->ncy, h->cell32);
    if (h->ord_inflight) {        // the index's own compaction (stream3) has finished
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ord, 0));
      h->ord_inflight = false;
    }
    static const int os_variant = getenv("GNX_OS_SORT") ? atoi(getenv("GNX_OS_SORT")) : 2;
    static const bool os_fused = !getenv("GNX_OS_FUSED") || atoi(getenv("GNX_OS_FUSED")) != 0;
    static const bool os_gather = !(getenv("GNX_OS_GATHER") && atoi(getenv("GNX_OS_GATHER")) == 0);
    if (h->keys_fresh && h->hist_fresh && os_gather && os_variant == 2 && os_fused) {
      // k_move wrote the cells AND counted their digits: the passes alone, the first one
      // gathering its keys through the id-ordered index
      h->hist_fresh = false;
      GNXCHK(gnx_os_sort32_gather(h->os_scratch, h->os_ktmp, h->os_vtmp, h->keyk[1], h->valk[1],
                                  (size_t)N, h->key_bits, h->ord[h->ord_cur], h->ord_n, h->cell32,
                                  h->stream));
      wipe_words = (int64_t)gnx_os_words_used((size_t)N, h->key_bits);
    } else if (os_variant == 2 && os_fused && h->key_bits <= 24) {
      GNXCHK(gnx_os_hist_discard(h));
      // keys, histograms and their scans in one launch, then the two or three passes; the
      // scratch is zero on entry (allocation, k_permute below)
      GNXCHK(gnx_os_keys_hist(h->os_scratch, h->tickets + 3, N, h->ord_n, h->ord[h->ord_cur],
                              h->cell32, h->keyk[0], h->valk[0], h->key_bits, h->stream));
      GNXCHK(gnx_os_sort32_ranked(h->os_scratch, h->os_ktmp, h->os_vtmp, h->keyk[0], h->keyk[1],
                                  h->valk[0], h->valk[1], (size_t)N, h->key_bits, h->stream));
      wipe_words = (int64_t)gnx_os_words_used((size_t)N, h->key_bits);
    } else {
    GNXCHK(gnx_os_hist_discard(h));
    hipLaunchKernelGGL(k_keys_ord, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, h->ord_n,
                       h->ord[h->ord_cur], h->cell32, h->keyk[0], h->valk[0]);
    if (os_variant >= 0 && h->key_bits <= 24) {
      wipe_words = (int64_t)(gnx_os_scratch_bytes((size_t)N, h->key_bits) / 4);
      GNXCHK(gnx_os_sort32(h->os_scratch, h->os_ktmp, h->os_vtmp, h->keyk[0], h->keyk[1],
                           h->valk[0], h->valk[1], (size_t)N, h->key_bits, h->stream, os_variant));
    } else {
      GNXCHK(gnx_prim_sort32_bits(h->sort64_tmp, h->sort64_tmp_bytes, h->keyk[0], h->keyk[1],
                                  h->valk[0], h->valk[1], (size_t)N, h->key_bits, h->stream, alone));
    }
    }
  } else {
    GNXCHK(gnx_os_hist_discard(h));
    int cell_bits = h->key_bits;
    if (h->tile_evict > 0 || tile_holes) {
      // (one more cell value: the emigrants'; two: the dead's behind it)
      while ((1ll << cell_bits) <= (int64_t)h->ncx * h->ncy + (tile_holes ? 1 : 0)) ++cell_bits;
      if (h->tile_evict == 0) {            // (nobody left this step: the box takes everybody)
        h->evict_box[0] = h->evict_box[2] = -1.0f;
        h->evict_box[1] = (float)c.W + 1.0f;
        h->evict_box[3] = (float)c.H + 1.0f;
      }
      hipLaunchKernelGGL(k_keys_evict, dim3(gnx_grid(n_sort, 256)), dim3(256), 0, h->stream, n_sort,
                         a.x, a.y, a.id, a.ghost, h->evict_box[0], h->evict_box[1], h->evict_box[2],
                         h->evict_box[3], h->inv_cs, h->ncx, h->ncy, idbits, h->key64[0],
                         h->perm[0], tile_holes ? (const int32_t*)h->flag : (const int32_t*)nullptr,
                         h->holes_flagged);
    } else if (!h->keys_fresh) {
      hipLaunchKernelGGL(k_keys, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, a.x, a.y, a.id,
                         h->inv_cs, h->ncx, h->ncy, idbits, h->key64[0], h->perm[0]);
    }
    // GNX_TILE_OS64=0: rocPRIM's own driver (a fill before the histograms, two before every pass)
    static const bool os64 = !(getenv("GNX_TILE_OS64") && atoi(getenv("GNX_TILE_OS64")) == 0);
    if (os64 && h->sort64_tmp_bytes >= (size_t)n_sort * 12 && h->os_scratch) {
      GNXCHK(gnx_os_sort64_clean(h->os_scratch, h->sort64_tmp, h->key64[0], h->key64[1], h->perm[0],
                                 h->perm[1], (size_t)n_sort, idbits + cell_bits, h->stream));
      wipe_words = (int64_t)gnx_os_words_used64((size_t)n_sort, idbits + cell_bits);
    } else {
      GNXCHK(gnx_prim_sort64_bits(h->sort64_tmp, h->sort64_tmp_bytes, h->key64[0], h->key64[1],
                                  h->perm[0], h->perm[1], (size_t)n_sort, idbits + cell_bits,
                                  h->stream, alone));
    }
  }
  h->keys_fresh = false;
  gnx_time_end(h, GNX_K_SORT, (double)N * 40.0);
  gnx_time_begin(h);
  // (tiles with imports sort (cell, id) keys: the sort's values ARE the slots the records come
  // from - all k_permute_rest needs; the evicted tail's genome rows, a cold column, are then
  // freed on the side stream behind it.  Measured with two tiles on one GPU: 1.72-1.94 ms/step
  // against 1.58-1.69 with one kernel for every column - parity-green, GNX_TILE_SPLIT=1 turns it on)
  static const bool tile_split = getenv("GNX_TILE_SPLIT") && atoi(getenv("GNX_TILE_SPLIT")) != 0;
  const bool split = split_rest && h->permute_split && (ordm ? !h->tiled : (tile_split && h->tile2_mode)) &&
                     h->stream3 != nullptr;
  hipLaunchKernelGGL(k_permute, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, c.cap_inds,
                     ordm ? h->valk[1] : h->perm[1], a, b, c.n_layers, c.n_traits,
                     a.tb ? 2 * h->TW : 0, gnx_pair_seed(c.seed, h->step), h->tag, (uint4*)h->cand,
                     h->key64[1], idbits, h->cell_start, h->ncx * h->ncy,
                     ordm ? h->keyk[1] : nullptr, h->ord[h->ord_cur], h->ord_n,
                     h->ord[h->ord_cur ^ 1], h->perm[1], (uint4*)h->os_scratch, (wipe_words + 3) / 4,
                     split ? 1 : 0, (const GnxDD*)nullptr, GnxBinP{nullptr, 0.0, 0, 0},
                     h->id_order == 1 ? h->vt_count : (int32_t*)nullptr);
  if (split) {
    // the columns nobody reads before the births follow on stream3, beside the mate search and
    // the pair list; whoever asked for the split waits (gnx_wait_permute_rest).
    // (GNX_PERMUTE_REST_AT=1: only behind the mate search, whose random 16-byte loads they
    // slow - 0.596 against 0.592 ms/step: they are then late for the births)
    h->perm_rest_a = a;
    h->perm_rest_b = b;
    h->perm_rest_N = N;
    h->perm_rest_pending = true;
    // GNX_PERMUTE_REST_AT=2 (round 5): not beside the mate search and the pair list either - random
    // 16-byte record loads and chains of dependent filters that take 43 + 34 us beside it against
    // 25 + 26 alone - but beside the births and the densities (gnx_l_find_pairs_enqueue launches
    // it behind the pair list): k_offspring takes the parents' alleles from the buffer the column
    // is permuted FROM, and only the death probabilities wait for the permutation.  Only where
    // nothing else reads those columns in between: gnx_step (h->perm_rest_late_ok).
    static const int rest_at = getenv("GNX_PERMUTE_REST_AT") ? atoi(getenv("GNX_PERMUTE_REST_AT")) : 0;
    h->perm_rest_late = rest_at == 2 && h->perm_rest_late_ok && h->defer_xo && !h->tile2_mode;
    if (rest_at == 0 || (rest_at == 2 && !h->perm_rest_late)) GNXCHK(gnx_permute_rest_launch(h));
  }
  gnx_time_end(h, GNX_K_PERMUTE,
               (double)N * 2.0 * (33.0 + 4.0 * c.n_layers + 4.0 * c.n_traits + 16.0 * h->TW));
  // the individuals' density bins (positions are final for this step): counted on stream3,
  // beside the mate search, when the step's density path is the fused one (gnx_internal.h: fb)
  GNXCHK(gnx_l_bins_adults_async(h, b.x, b.y, N));
  if (ordm) {
    h->ord_cur ^= 1;
    h->ord_n = N;
  } else if (h->ord_mode && !h->tiled) {
    // no index (an upload in another order, a tile that became a single device): one sort of
    // (id, sorted slot) rebuilds it
    if (h->ord_inflight) {
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ord, 0));
      h->ord_inflight = false;
    }
    hipLaunchKernelGGL(k_iota, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, h->perm[0]);
    GNXCHK(gnx_prim_sort64_bits(h->sort64_tmp, h->sort64_tmp_bytes, (const uint64_t*)b.id,
                                h->key64[0], h->perm[0], h->ord[h->ord_cur], (size_t)N, idbits,
                                h->stream, alone));
    h->ord_valid = true;
    h->ord_n = N;
  } else {
    h->ord_valid = false;         // the slots moved and nobody followed them
  }
  HIPCHK(hipGetLastError());
  h->cur ^= 1;
  if (h->tile_evict > 0) {
    // the emigrants are the last tile_evict slots of the sorted population: gone
    const int64_t n = h->tile_evict;
    h->tile_evict = 0;
    h->N -= n;
    if (h->genomes_assigned && c.L > 0) {
      // (split permutation: `grow` arrives on stream3 - the rows are freed there, behind it; every
      // user of the free-row stack comes after gnx_wait_permute_rest)
      hipStream_t fs = h->stream;
      if (split) {
        GNXCHK(gnx_permute_rest_launch(h));
        fs = h->stream3;
      }
      hipLaunchKernelGGL(k_free_tail, dim3(gnx_grid(n, 256)), dim3(256), 0, fs, h->N, n,
                         h->soa[h->cur].grow, h->free_rows, h->n_free);
      if (split) HIPCHK(hipEventRecord(h->ev_perm_rest, h->stream3));
      h->n_free += n;
    }
    h->ord_valid = false;
    h->fb_adults = false;
  }
  if (h->xo_launch_policy == 1) GNXCHK(gnx_xo_launch_pending(h));
  return 0;
}

// ---------------------------------------------------------------- mate search
// Species._get_mating_pairs / _KDTree._get_mating_pairs (structs/species.py:
// 2157-2215; utils/spatial.py:191-245) on a cell list instead of a KD-tree.
//
// Who needs a mate at all?  A pair (i, mate_i) survives only if i's own
// Bernoulli(b) draw keeps it (structs/species.py:2210-2214) and i passes the
// focal side of the sex / reproductive-age filters (ops/mating.py:41-104);
// the reciprocity test of the de-duplication only looks at mates of kept
// individuals.  The draw is keyed by i's id, not by the mate, so the kernel
// evaluates it FIRST and the search runs for the kept ~b*N individuals only
// (b = 0.2 -> five times less work); everybody else gets mate = -1.  The
// compaction is block-local (each block lists the kept ones among its 1024
// consecutive individuals in LDS, in order): no global atomics, no extra pass.
//
// Candidates of a focal individual: the 3x3 block of hash cells around its own, i.e.
// per cell row ONE contiguous range [cell_start(ry, cx-1), cell_start(ry, cx+2)) of the
// sorted arrays, one 16-byte record {x, y, tag, id_lo} per candidate.  Cells are >=
// mating_radius wide, so every neighbour within the radius lies in the three ranges;
// the distance test is dx*dx + dy*dy <= r*r in f32, the expression the oracle evaluates.
//
// UNIFORM choice (the reference's default: np.random.choice over the neighbours within
// the radius, utils/spatial.py:236-241) by REJECTION SAMPLING IN INDEX SPACE: the three
// ranges are one canonical list of M candidates (cells in grid order, ids ascending
// inside a cell); the focal draws indices from its own Philox stream
// (seed, id, step, OP_MATE_PICK) until the candidate drawn is within the radius (and is
// not itself) - a uniform pick from the in-radius set in ~M/m expected draws, whatever M
// is.  That matters: conductance-surface movement piles individuals onto ridges (the
// metric model at equilibrium holds up to 6,700 individuals in one cell, a mean of
// 4,600 candidates per focal against 270 for a uniform spread) and a scan of all
// candidates per focal then costs as much as the crossover.  After about M/2 rejected
// draws (32 .. 8192, a multiple of 4 fixed by M) the lane falls back to an exact scan:
// count the m in-radius candidates, take the (u*m)-th.  The result depends on ids, positions and the seed only: slot order and
// tiling do not matter (tiles import whole cells, csrc/gnx_tile.hip).
//
// INVERSE-DISTANCE choice (P(j) ~ r - d_ij over the neighbours with d > 0,
// utils/spatial.py:209-229) is the same index sampling with an acceptance draw: a
// candidate drawn is taken iff u * r < r - d (two stream words per try); the exact
// fallback walks the candidates once for the total weight and once for the pick.
// NEAREST: FM_LANES adjacent lanes share one focal's cells (64 contiguous bytes per group
// and step) and combine their branch-free composite-key minima (d2, id) with DPP
// shuffles, own cell first; a neighbour cell is read only if its rectangle comes closer
// than the best candidate so far.
// (A wave-broadcast variant - the wave loads the union of its lanes' ranges and
// v_readlane's each candidate - was 5x slower: 3.4x more vector instructions, PMC
// SQ_INSTS_VALU, round-1 profiles.)
#define FM_PER_BLOCK 1024      // individuals per 256-thread block (~b * 1024 are searched)
#define FM_LANES 4             // lanes that share one focal individual's candidate ranges
#define FM_MIN_BLOCKS 8         // index draws before the exact fallback scan: 4 per Philox
#define FM_MAX_BLOCKS 2048      // block, between 32 and 8192, about half the candidate count

struct FocalP {
  int64_t N;
  const uint8_t* keep_in;
  float b;
  int sexed, ra_f;
  long long step;
  unsigned long long seed;
  const GnxDD* dd;
};

// the three row ranges of a focal individual's 3x3 block of cells (an absent row has
// length 0) and the slot of index j of their concatenation
#define FM_ROWS 5              // cell rows of a focal individual's block: 2 * ref + 1, ref <= 2
struct FmRanges {
  int st[FM_ROWS], len[FM_ROWS];
  unsigned int M;
  __device__ __forceinline__ int slot(int j) const {
    int s = st[0] + j;
    int acc = len[0];
#pragma unroll
    for (int q = 1; q < FM_ROWS; ++q) {
      s = j >= acc ? st[q] + (j - acc) : s;
      acc += len[q];
    }
    return s;
  }
};

// ref: cells that cover the mating radius (1: cells of a radius, the 3 x 3 block; 2: cells of half
// a radius, the 5 x 5 block); rows beyond 2 * ref + 1 have length 0
__device__ __forceinline__ FmRanges fm_ranges(float fx, float fy, double inv_cs, int ncx, int ncy,
                                              const int32_t* __restrict__ cell_start, int ref) {
  FmRanges R;
  const int k = gnx_cell_of(fx, fy, inv_cs, ncx, ncy);
  const int cy = k / ncx;
  const int cx = k - cy * ncx;
  const int lo = max(cx - ref, 0), hi = min(cx + ref, ncx - 1);
  unsigned int M = 0;
#pragma unroll
  for (int q = 0; q < FM_ROWS; ++q) {
    const int ry = cy - ref + q;
    const bool in = q <= 2 * ref && ry >= 0 && ry < ncy;
    R.st[q] = in ? cell_start[ry * ncx + lo] : 0;
    R.len[q] = in ? cell_start[ry * ncx + hi + 1] - R.st[q] : 0;
    M += (unsigned int)R.len[q];
  }
  R.M = M;
  return R;
}

// Philox blocks before the exact scan (a scan costs 2M loads): about M/8 blocks of 4 index
// draws (32 .. 8192 draws), or M/4 blocks of 2 weighted tries (32 .. 8192)
template <bool WT>
__device__ __forceinline__ int fm_blocks(unsigned int M) {
  return WT ? min(max((int)(M >> 2), 2 * FM_MIN_BLOCKS), 2 * FM_MAX_BLOCKS)
            : min(max((int)(M >> 3), FM_MIN_BLOCKS), FM_MAX_BLOCKS);
}

template <int MODE>
__global__ void __launch_bounds__(256)
k_find_mates(FocalP fp, GnxSoA s, const uint4* __restrict__ cand, double inv_cs,
             const int32_t* __restrict__ cell_start, int ncx, int ncy, int ref, float r, float r2,
             int easy_rounds, int32_t* __restrict__ mate) {
  __shared__ int32_t list[FM_PER_BLOCK];
  __shared__ int32_t hard[FM_PER_BLOCK];
  __shared__ int32_t wcnt[FM_PER_BLOCK / 256][4];
  __shared__ int32_t n_hard_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * FM_PER_BLOCK;
  fp.N = gnx_dd_n(fp.dd, fp.N);
  fp.step = gnx_dd_step(fp.dd, fp.step);
  if (base >= fp.N) return;                     // (block-uniform)
  // phase 1: block-local, order-preserving list of the individuals that need a mate
  // (all four rounds' loads and draws first, then one exchange of the wave counts)
  constexpr int ROUNDS = FM_PER_BLOCK / 256;
  bool okr[ROUNDS];
  unsigned long long balr[ROUNDS];
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    const int64_t i = base + round * 256 + tid;
    bool ok = false;
    if (i < fp.N) {
      if (fp.keep_in) {
        ok = fp.keep_in[i] != 0;
      } else {
        uint4 rr = gnx_rand4(fp.seed, (unsigned long long)s.id[i], fp.step, OP_PAIR_KEEP, 0);
        ok = gnx_u01(rr.x) < fp.b;
      }
      if (fp.sexed) ok = ok && s.sex[i] == 0;
      ok = ok && s.age[i] >= fp.ra_f;
      if (!ok) mate[i] = -1;
    }
    okr[round] = ok;
    balr[round] = __ballot(ok);
    if (lane == 0) wcnt[round][wave] = __popcll(balr[round]);
  }
  if (tid == 0) n_hard_s = 0;
  __syncthreads();
  int n_list = 0;
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    int off = n_list;
    for (int w = 0; w < wave; ++w) off += wcnt[round][w];
    if (okr[round])
      list[off + __popcll(balr[round] & ((1ull << lane) - 1ull))] =
          (int32_t)(base + round * 256 + tid);
    n_list += wcnt[round][0] + wcnt[round][1] + wcnt[round][2] + wcnt[round][3];
  }
  __syncthreads();
  // phase 2
  if (MODE != GNX_MATE_NEAREST) {
    // UNIFORM: one stream word per try (the index).  INVERSE-DISTANCE (P(j) ~ r - d_ij
    // over the neighbours with d > 0, utils/spatial.py:209-229): two words per try, the
    // index and an acceptance draw u: the candidate is taken iff u * r < r - d.
    // The tries are examined in stream order, so the first accepted try is the one a
    // sequential walk of the stream would take.  Most focal individuals are served by
    // their first few tries (2a: one lane each, a round's candidates fetched TOGETHER -
    // the loads are independent, the round trip to L2 / HBM is what a try costs); whoever
    // is not (the edge of a clump: one in-radius neighbour among hundreds of candidates)
    // would hold its whole wave for tens of rounds, so the rest go on a list that the
    // block's waves work off one focal individual at a time (2b), a Philox block per lane:
    // 256 (128 weighted) tries in flight per round.
    constexpr bool WT = MODE == GNX_MATE_INVERSE;
    constexpr int STEP = WT ? 2 : 1;
    for (int t = tid; t < n_list; t += 256) {
      const int i = list[t];
      const uint4 me = cand[i];
      const float fx = __uint_as_float(me.x), fy = __uint_as_float(me.y);
      const FmRanges R = fm_ranges(fx, fy, inv_cs, ncx, ncy, cell_start, min(ref, 2));
      const unsigned int M = R.M;
      const unsigned long long fid = (unsigned long long)s.id[i];
      int found = -1;
      bool open = false;
      if (M > 1) {
        const int early = min(fm_blocks<WT>(M), easy_rounds * STEP);
        for (int blk = 0; blk < early && found < 0; blk += STEP) {
          // four tries per round: one Philox block of index words, or two blocks of
          // (index, acceptance) words
          const uint4 w = gnx_rand4(fp.seed, fid, fp.step, OP_MATE_PICK, blk);
          const uint4 w2 = WT ? gnx_rand4(fp.seed, fid, fp.step, OP_MATE_PICK, blk + 1) : w;
          const unsigned int wi[4] = {w.x, WT ? w.z : w.y, WT ? w2.x : w.z, WT ? w2.z : w.w};
          const unsigned int wa[4] = {w.y, w.w, w2.y, w2.w};
          int slot[4];
          uint4 c[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            slot[q] = R.slot((int)__umulhi(wi[q], M));
            c[q] = cand[slot[q]];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float dx = __uint_as_float(c[q].x) - fx, dy = __uint_as_float(c[q].y) - fy;
            const float d2 = dx * dx + dy * dy;
            bool ok = slot[q] != i && d2 <= r2;
            if (WT) ok = ok && d2 > 0.f && gnx_u01(wa[q]) * r < r - sqrtf(d2);
            if (ok && found < 0) found = slot[q];
          }
        }
        open = found < 0;
      }
      if (open)
        hard[atomicAdd(&n_hard_s, 1)] = i;
      else
        mate[i] = found;
    }
    __syncthreads();
    const int n_hard = n_hard_s;
    for (int hh = wave; hh < n_hard; hh += 4) {
      const int i = __builtin_amdgcn_readfirstlane(hard[hh]);
      const uint4 me = cand[i];
      const float fx = __uint_as_float(me.x), fy = __uint_as_float(me.y);
      const FmRanges R = fm_ranges(fx, fy, inv_cs, ncx, ncy, cell_start, min(ref, 2));
      const unsigned int M = R.M;
      const unsigned long long fid = (unsigned long long)s.id[i];
      const int blocks = fm_blocks<WT>(M);
      int found = -1;
      for (int b0 = easy_rounds * STEP; b0 < blocks && found < 0; b0 += 64) {
        const int blk = b0 + lane;
        const uint4 w = gnx_rand4(fp.seed, fid, fp.step, OP_MATE_PICK, blk);
        constexpr int NT = WT ? 2 : 4;
        const unsigned int wi[4] = {w.x, WT ? w.z : w.y, w.z, w.w};
        const unsigned int wa[2] = {w.y, w.w};
        int slot[NT];
        uint4 c[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          slot[q] = R.slot((int)__umulhi(wi[q], M));
          c[q] = cand[slot[q]];
        }
        int mine = -1;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          const float dx = __uint_as_float(c[q].x) - fx, dy = __uint_as_float(c[q].y) - fy;
          const float d2 = dx * dx + dy * dy;
          bool ok = slot[q] != i && d2 <= r2;
          if (WT) ok = ok && d2 > 0.f && gnx_u01(wa[q]) * r < r - sqrtf(d2);
          if (ok && mine < 0) mine = slot[q];
        }
        // lanes hold ascending blocks: the lowest lane with an accepted try has the first
        const unsigned long long bal = __ballot(blk < blocks && mine >= 0);
        if (bal)
          found = __builtin_amdgcn_readlane(
              mine, __builtin_amdgcn_readfirstlane(__ffsll((long long)bal) - 1));
      }
      if (found < 0) {
        // exact fallback in canonical order, 256 candidates per step (four independent
        // loads per lane): the (u * m)-th of the m in-radius candidates, or the first whose
        // running weight passes u * (total weight).  The f32 weights are added one by one in
        // canonical order (v_readlane of the in-radius lanes only: skipping a zero changes
        // nothing), so the sums are the sequential ones the oracle forms.
        unsigned int m = 0;
        float wsum = 0.f;
        for (unsigned int j0 = 0; j0 < M; j0 += 256) {
          uint4 c[4];
          int sl[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned int j = j0 + q * 64 + lane;
            sl[q] = j < M ? R.slot((int)j) : -1;
            if (sl[q] >= 0) c[q] = cand[sl[q]];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float wj = 0.f;
            bool in = false;
            if (sl[q] >= 0) {
              const float dx = __uint_as_float(c[q].x) - fx, dy = __uint_as_float(c[q].y) - fy;
              const float d2 = dx * dx + dy * dy;
              in = sl[q] != i && d2 <= r2 && (!WT || d2 > 0.f);
              if (WT && in) wj = r - sqrtf(d2);
            }
            unsigned long long bal = __ballot(in);
            m += (unsigned int)__popcll(bal);
            if (WT)
              while (bal) {
                const int k = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                wsum = wsum + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wj), k));
              }
          }
        }
        if (m > 0) {
          const uint4 w0 = gnx_rand4(fp.seed, fid, fp.step, OP_MATE_PICK, blocks);
          unsigned int want = __umulhi(w0.x, m);
          const float target = gnx_u01(w0.x) * wsum;
          float run = 0.f;
          int last = -1;
          for (unsigned int j0 = 0; j0 < M && found < 0; j0 += 256) {
            uint4 c[4];
            int sl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned int j = j0 + q * 64 + lane;
              sl[q] = j < M ? R.slot((int)j) : -1;
              if (sl[q] >= 0) c[q] = cand[sl[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float wj = 0.f;
              bool in = false;
              if (sl[q] >= 0) {
                const float dx = __uint_as_float(c[q].x) - fx, dy = __uint_as_float(c[q].y) - fy;
                const float d2 = dx * dx + dy * dy;
                in = sl[q] != i && d2 <= r2 && (!WT || d2 > 0.f);
                if (WT && in) wj = r - sqrtf(d2);
              }
              unsigned long long bal = __ballot(in);
              if (found >= 0) continue;
              if (WT) {
                while (bal && found < 0) {
                  const int k = __ffsll((long long)bal) - 1;
                  bal &= bal - 1;
                  run = run + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wj), k));
                  last = __builtin_amdgcn_readlane(sl[q], k);
                  if (run > target) found = last;
                }
              } else {
                const unsigned int n_in = (unsigned int)__popcll(bal);
                if (want < n_in) {
                  const unsigned long long pick = __ballot(
                      in && (unsigned int)__popcll(bal & ((1ull << lane) - 1ull)) == want);
                  found = __builtin_amdgcn_readlane(
                      sl[q], __builtin_amdgcn_readfirstlane(__ffsll((long long)pick) - 1));
                } else {
                  want -= n_in;
                }
              }
            }
          }
          if (WT && found < 0) found = last;
        }
      }
      if (lane == 0) mate[i] = found;
    }
    return;
  }
  // NEAREST: FM_LANES adjacent lanes per listed focal individual; cells are `ref` to a
  // mating radius (fine cells), searched ring by ring around the focal's own cell: ring k
  // is the square frame at Chebyshev distance k - its top and bottom rows are one
  // contiguous range of the sorted records each, its sides two single cells per row.  A
  // ring whose nearest point is farther than the best candidate so far ends the search; a
  // row or cell of the ring whose rectangle is farther is skipped.
  const double cs = 1.0 / inv_cs;
  for (int t = tid; t < n_list * FM_LANES; t += 256) {
    const int i = list[t / FM_LANES];
    const int sub = t % FM_LANES;
    const uint4 me = cand[i];
    const float fx = __uint_as_float(me.x), fy = __uint_as_float(me.y);
    const int k0 = gnx_cell_of(fx, fy, inv_cs, ncx, ncy);
    const int cy = k0 / ncx;
    const int cx = k0 - cy * ncx;
    unsigned long long best = ~0ull;
    int best_slot = -1;
    for (int k = 0; k <= ref; ++k) {
      float lim = best_slot >= 0 ? __uint_as_float((unsigned int)(best >> 32)) : r2;
      if (k >= 2) {
        // every cell of ring k is at least (k - 1) cells away (shrunk a little so that
        // rounding can never hide a candidate at equal distance)
        const double g = (k - 1) * cs;
        if ((float)(g * g * 0.99999) > lim) break;
      }
      for (int dy = -k; dy <= k; ++dy) {
        const int ry = cy + dy;
        if (ry < 0 || ry >= ncy) continue;
        const double gy = fmax(fmax(ry * cs - (double)fy, (double)fy - (ry + 1) * cs), 0.0);
        const bool full_row = dy == -k || dy == k;
        // full rows: one range [cx - k, cx + k]; side rows: the two cells cx - k and cx + k
        for (int part = 0; part < (full_row || k == 0 ? 1 : 2); ++part) {
          int x0 = full_row ? cx - k : (part == 0 ? cx - k : cx + k);
          int x1 = full_row ? cx + k : x0;
          x0 = max(x0, 0);
          x1 = min(x1, ncx - 1);
          if (x0 > x1) continue;
          const double gx = fmax(fmax(x0 * cs - (double)fx, (double)fx - (x1 + 1) * cs), 0.0);
          if ((float)((gx * gx + gy * gy) * 0.99999) > lim) continue;   // uniform over the lanes
          const int st = cell_start[ry * ncx + x0];
          const int e = cell_start[ry * ncx + x1 + 1];
          for (int j = st + sub; j < e; j += FM_LANES) {
            const uint4 c = cand[j];
            const float dx = __uint_as_float(c.x) - fx, dyy = __uint_as_float(c.y) - fy;
            const float d2 = dx * dx + dyy * dyy;
            const bool ok = (d2 <= r2) & (j != i);
            const unsigned long long comp =
                ok ? (((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)c.w)
                   : ~0ull;
            const bool better = comp < best;
            best = better ? comp : best;
            best_slot = better ? j : best_slot;
          }
          // the lanes of the group agree on the best so far before the next bound test
#pragma unroll
          for (int m = 1; m < FM_LANES; m <<= 1) {
            const unsigned long long ob = __shfl_xor(best, m);
            const int os = __shfl_xor(best_slot, m);
            const bool take = ob < best;
            best = take ? ob : best;
            best_slot = take ? os : best_slot;
          }
          lim = best_slot >= 0 ? __uint_as_float((unsigned int)(best >> 32)) : r2;
        }
      }
    }
    if (sub == 0) mate[i] = best_slot;
  }
}

// ---------------------------------------------------------------- offspring ids, tile-major
// gnx_set_id_order(h, 1): offspring ids are handed out virtual tile by virtual tile - a fixed
// 8 x 8 blocking of the landscape that every tile grid dividing 8 x 8 is a union of - and
// inside a virtual tile in the canonical (hash cell, focal id) order of the pairs.  A tile of a
// tiled run owns whole virtual tiles, so the rank of a pair inside its virtual tile is the
// same number on the tile and on one device, and the only thing the tiles have to tell each
// other is how many births each virtual tile has: 64 counts that ride on the count exchange,
// instead of an all-gather of every pair's order key (gnx_comm.hip; VERDICT r3 #1b).
// Round 5: the Model API numbers its offspring this way on ONE device too (whenever the
// landscape's dimensions are divisible by 8), so that a model run over several ranks equals the
// one-process run id by id; to make that cheap the classification rides inside k_pair_compact
// (class and in-block rank of every pair, per-block counts per class: blocks of 1024 SLOTS), one
// small kernel scans the blocks' counts (k_cls_scan) and k_offspring adds the three numbers up
// itself - round 4 ran three kernels of their own on the step's chain (k_pair_cls, k_cls_scan,
// k_goff_vt).  Poisson births (ops/mating.py:120-126): the ranks are in BIRTHS, not in pairs -
// k_pair_cls below, weighted by the pairs' birth counts, once those are drawn.
#define GNX_VT 8
#define GNX_VTN (GNX_VT * GNX_VT)

struct GnxVtP {
  int vw, vh;                 // raster cells per virtual tile: W / 8, H / 8 (exact: gnx_set_id_order)
  float inv_w, inv_h;         // 1 / vw, 1 / vh: the estimate the exact boundaries correct
};
// (exact on the integer boundaries, like tile ownership - gnx_tile_index -, without its division:
// the reciprocal's estimate is off by one at most, two comparisons with the exact boundaries
// k * vw settle it)
__device__ __forceinline__ int gnx_vt_axis(float v, int vw, float inv) {
  int c = min(GNX_VT - 1, max(0, (int)(v * inv)));
  c -= (c > 0 && (float)(c * vw) > v) ? 1 : 0;
  c += (c + 1 < GNX_VT && (float)((c + 1) * vw) <= v) ? 1 : 0;
  return c;
}
__device__ __forceinline__ int gnx_vt_of(const GnxVtP& V, float x, float y) {
  return gnx_vt_axis(y, V.vh, V.inv_h) * GNX_VT + gnx_vt_axis(x, V.vw, V.inv_w);
}

// Tile-major ids with a fixed number of births per pair, in the two kernels of the pair list
// (cls == null: off):
//   k_pair_flags   - the virtual tile of every kept pair (by its focal individual's position:
//                    scls, one byte per slot), the block's pairs per virtual tile (blk_cnt) and,
//                    one atomic per block and virtual tile it touches, the device's totals
//                    (total[64]: zeroed by k_permute; on tiles the counts that travel);
//   k_pair_compact - per pair its virtual tile (cls) and its rank among ALL pairs of that
//                    virtual tile before it on this device (rank): the in-block rank plus the
//                    counts of the blocks before - a workgroup adds those up for the one to three
//                    virtual tiles it touches itself, a few coalesced loads from L2 (blocks are
//                    1024 SLOTS, neighbours in space); workgroup 0 also turns the totals into the
//                    virtual tiles' base offsets when nobody else has pairs (local).
// k_offspring: id offset = base[cls] + rank * lambda + ordinal.  No kernel of their own on the
// step's chain (the first cut of round 5 scanned the blocks' counts in one: 23 us).
struct GnxVtOut {
  uint8_t* scls;              // [slots] virtual tile of a kept pair's focal individual
  int32_t* blk_cnt;           // [64][stride] pairs per virtual tile and block, class-major
  unsigned long long* blk_nz; // [blocks] the virtual tiles the block touches
  int stride;
  int32_t* total;             // [64] pairs per virtual tile on this device
  uint8_t* cls;               // [pairs]
  int32_t* rank;              // [pairs]
  int64_t* base;              // [64] written by k_pair_compact when local
  int local;
  int64_t lam;
  GnxVtP V;
};

// One round of 256 items: the class ranks inside the wave and the wave's count per class.
// The lanes of a wave are neighbours in the canonical order: one or two virtual tiles per wave.
// WEIGHTED: ranks and counts in units of w (births), else in items (pairs).
template <bool WEIGHTED>
__device__ __forceinline__ void gnx_vt_wave_ranks(bool act, int c, int w, int lane, int* cnt_row,
                                                  int& rw) {
  unsigned long long todo = __ballot(act);
  rw = 0;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lc = __shfl(c, leader);
    const bool mine = act && c == lc;
    const unsigned long long same = __ballot(mine);
    if (WEIGHTED) {
      int x = mine ? w : 0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
      }
      const int tot = __shfl(x, 63);
      if (mine) rw = x - w;
      if (lane == leader) cnt_row[lc] = tot;
    } else {
      if (mine) rw = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == leader) cnt_row[lc] = __popcll(same);
    }
    todo &= ~same;
  }
}

// Poisson births: the virtual tile of every pair, its rank IN BIRTHS among the pairs of the same
// virtual tile inside its block of 1024 pairs, and the block's births per virtual tile.
// P: the pair count, or read from P_dev.
__global__ void __launch_bounds__(256)
k_pair_cls(int64_t P, const int32_t* __restrict__ P_dev, const int32_t* __restrict__ pairs,
           const float* __restrict__ x, const float* __restrict__ y,
           const int32_t* __restrict__ nbirths, GnxVtP V, uint8_t* __restrict__ cls,
           int32_t* __restrict__ rank, int32_t* __restrict__ pblk, int32_t* __restrict__ blk_cnt) {
  __shared__ int cnt[16][GNX_VTN];
  if (P_dev) P = *P_dev;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < 16 * GNX_VTN; k += 256) (&cnt[0][0])[k] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  int c[4], rw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t p = base + r * 256 + tid;
    const bool act = p < P;
    c[r] = GNX_VTN;
    int w = 0;
    if (act) {
      const int fo = pairs[2 * p];
      c[r] = gnx_vt_of(V, x[fo], y[fo]);
      w = nbirths[p];
    }
    gnx_vt_wave_ranks<true>(act, c[r], w, lane, cnt[r * 4 + wave], rw[r]);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t p = base + r * 256 + tid;
    if (p >= P) continue;
    int before = 0;
    for (int j = 0; j < r * 4 + wave; ++j) before += cnt[j][c[r]];
    cls[p] = (uint8_t)c[r];
    rank[p] = before + rw[r];
    pblk[p] = (int32_t)blockIdx.x;
  }
  if (tid < GNX_VTN) {
    int t = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += cnt[j][tid];
    blk_cnt[(int64_t)blockIdx.x * GNX_VTN + tid] = t;
  }
}

// per virtual tile (one lane each) the exclusive offsets of the blocks' counts and the total:
// the 16 waves of the workgroup take a stretch of the blocks each, the stretches' totals meet in
// LDS; with `local` the virtual tiles' base offsets too (one device: nobody else has pairs), in
// births.  n_dev / dd: the item count (pairs, or slots: blocks of 1024 of them) on the device.
__global__ void __launch_bounds__(1024)
k_cls_scan(int nb, const int32_t* __restrict__ n_dev, const GnxDD* __restrict__ dd,
           const int32_t* __restrict__ blk_cnt, int32_t* __restrict__ blk_off,
           int32_t* __restrict__ vt_count, int64_t* __restrict__ vt_base, int local, int64_t lam) {
  __shared__ int seg_tot[16][GNX_VTN];
  const int c = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (n_dev) nb = (int)(((int64_t)*n_dev + GNX_CB - 1) / GNX_CB);
  if (dd) nb = (int)(((int64_t)dd->N + GNX_CB - 1) / GNX_CB);
  const int seg = (nb + 15) / 16;
  const int b0 = wave * seg, b1 = min(nb, b0 + seg);
  int run = 0;
  for (int b = b0; b < b1; ++b) run += blk_cnt[(int64_t)b * GNX_VTN + c];
  seg_tot[wave][c] = run;
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    const int t = seg_tot[w][c];
    before += w < wave ? t : 0;
    total += t;
  }
  run = before;
  for (int b = b0; b < b1; ++b) {
    const int v = blk_cnt[(int64_t)b * GNX_VTN + c];
    blk_off[(int64_t)b * GNX_VTN + c] = run;
    run += v;
  }
  if (wave != 0) return;
  vt_count[c] = total;
  if (local) {
    int xs = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int yv = __shfl_up(xs, d);
      if (c >= d) xs += yv;
    }
    vt_base[c] = (int64_t)(xs - total) * lam;
  }
}

// how k_offspring finds a pair's global offspring offset: handed in pair by pair (the
// Python-driven tile protocol with (hash cell, focal id) ids), or tile-major from the pieces
// above: vt_base[class] + (blk_off[block][class] + rank) * mul, or - all null - the pair's own
// offset on this device
struct GnxGoff {
  const int64_t* goff;
  const uint8_t* cls;
  const int32_t* rank;
  const int32_t* pblk;        // Poisson births: the pair's block and the blocks' offsets (else null)
  const int32_t* blk_off;
  const int64_t* vt_base;
  int64_t mul;                // births per unit of rank: lambda (fixed births), 1 (Poisson)
};

int gnx_vt_buffers(gnx_state* h) {
  if (h->vt_cls) return 0;
  const size_t cap = (size_t)h->cfg.cap_inds;
  const size_t nb = cap / GNX_CB + 2;
  HIPCHK(hipMalloc((void**)&h->vt_cls, cap));
  HIPCHK(hipMalloc((void**)&h->vt_scls, cap));
  HIPCHK(hipMalloc((void**)&h->vt_blk_nz, nb * sizeof(unsigned long long)));
  HIPCHK(hipMalloc((void**)&h->vt_rank, cap * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&h->vt_pblk, cap * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&h->vt_blk_cnt, nb * GNX_VTN * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&h->vt_blk_off, nb * GNX_VTN * sizeof(int32_t)));
  // (+ room behind the 64 counts for a tile's gamete-request counts: one device vector for the
  // count exchange of gnx_tile_step)
  HIPCHK(hipMalloc((void**)&h->vt_count, (GNX_VTN + GNX_MAX_TILES + 8) * sizeof(int32_t)));
  HIPCHK(hipMemset(h->vt_count, 0, (GNX_VTN + GNX_MAX_TILES + 8) * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&h->vt_base, GNX_VTN * sizeof(int64_t)));
  return 0;
}

static GnxVtP gnx_vtp(const gnx_state* h) {
  const int vw = h->cfg.W / GNX_VT, vh = h->cfg.H / GNX_VT;
  return GnxVtP{vw, vh, 1.0f / (float)vw, 1.0f / (float)vh};
}

// what k_pair_compact is handed when the ids are tile-major and every pair has the same
// number of births (else nothing: the weighted classification follows the birth draws)
static int gnx_vt_out(gnx_state* h, GnxVtOut* out) {
  *out = GnxVtOut{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 1,
                  GnxVtP{1, 1, 1.f, 1.f}};
  h->vt_fused = false;
  if (h->id_order != 1 || !h->sp.n_births_fixed || h->sp.mating_radius < 0) return 0;
  GNXCHK(gnx_vt_buffers(h));
  // (several tiles: the bases come from every tile's totals, gnx_tile_step)
  *out = GnxVtOut{h->vt_scls, h->vt_blk_cnt, h->vt_blk_nz, (int)(h->cfg.cap_inds / GNX_CB + 2),
                  h->vt_count, h->vt_cls, h->vt_rank, h->vt_base,
                  h->tiled ? 0 : 1, (int64_t)h->sp.n_births_lambda, gnx_vtp(h)};
  h->vt_fused = true;
  return 0;
}

// The virtual tiles' counts (h->vt_count, device: in births / vt_mul) and the blocks' offsets
// behind the classification - k_pair_compact's (fixed births; blocks of slots) or, Poisson
// births, k_pair_cls's here (blocks of pairs; needs h->nbirths).  local: the virtual tiles' base
// offsets too, from this device's own counts; else the caller sets h->vt_base (the tiles' sums).
int gnx_l_pair_cls(gnx_state* h, int64_t P, bool local) {
  GNXCHK(gnx_vt_buffers(h));
  GnxSoA s = h->soa[h->cur];
  if (h->sp.n_births_fixed) {
    // the two kernels of the pair list did it all (k_pair_flags, k_pair_compact): nothing to
    // launch.  (P < 0: not known on the host - gnx_tile_step; an empty tile made no pair list)
    if (!h->vt_fused && h->N > 0 && P != 0) {
      gnx_set_error("tile-major offspring ids: the pair list was made without its classification");
      return 1;
    }
    if (local != !h->tiled) {
      gnx_set_error("tile-major offspring ids: the bases' owner does not match the tile grid");
      return 1;
    }
    h->vt_mul = (int64_t)h->sp.n_births_lambda;
    h->vt_weighted = false;
    if (h->N == 0 || P == 0)          // (no pair list was made: clean zeros for whoever reads them)
      HIPCHK(hipMemsetAsync(h->vt_count, 0, GNX_VTN * sizeof(int32_t), h->stream));
    h->pair_goff_local_base = local;
    h->pair_goff_ready = true;
    return 0;
  }
  int nb;
  {
    h->vt_weighted = true;
    nb = (int)((P + GNX_CB - 1) / GNX_CB);
    h->vt_mul = 1;
    if (P > 0)
      hipLaunchKernelGGL(k_pair_cls, dim3(nb), dim3(256), 0, h->stream, P, (const int32_t*)nullptr,
                         (const int32_t*)h->pairs, (const float*)s.x, (const float*)s.y,
                         (const int32_t*)h->nbirths, gnx_vtp(h), h->vt_cls, h->vt_rank, h->vt_pblk,
                         h->vt_blk_cnt);
  }
  if (P == 0) nb = 0;             // (no pair list was written: counts of zero)
  hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(1024), 0, h->stream, nb, (const int32_t*)nullptr,
                     (const GnxDD*)nullptr, (const int32_t*)h->vt_blk_cnt, h->vt_blk_off,
                     h->vt_count, h->vt_base, local ? 1 : 0, h->vt_mul);
  HIPCHK(hipGetLastError());
  h->pair_goff_local_base = local;
  h->pair_goff_ready = true;
  return 0;
}

static GnxGoff gnx_goff_vt(const gnx_state* h) {
  if (!h->vt_weighted)
    return GnxGoff{nullptr, h->vt_cls, h->vt_rank, nullptr, nullptr, h->vt_base, h->vt_mul};
  return GnxGoff{nullptr, h->vt_cls, h->vt_rank, h->vt_pblk, h->vt_blk_off, h->vt_base, h->vt_mul};
}

// Bernoulli(b) thinning (structs/species.py:2210-2214), sex filter
// (ops/mating.py:41-55), reproductive-age filter (:79-104) of the pair (i, mate[i]).
struct PairP {
  int64_t N;
  const int32_t* focal;
  const int32_t* mate;
  const uint8_t* keep_in;
  float b;
  int sexed, ra_f, ra_m, dedup;
  long long step;
  unsigned long long seed;
  const GnxDD* dd;
};

__device__ __forceinline__ bool pair_ok(const PairP& P, const GnxSoA& s, int64_t i) {
  const int m = P.mate[i];
  if (m < 0) return false;
  const int fo = P.focal ? P.focal[i] : (int)i;
  bool keep;
  if (P.keep_in) {
    keep = P.keep_in[i] != 0;
  } else {
    uint4 r = gnx_rand4(P.seed, (unsigned long long)s.id[i], P.step, OP_PAIR_KEEP, 0);
    keep = gnx_u01(r.x) < P.b;
  }
  bool ok = keep;
  if (P.sexed) ok = ok && (s.sex[fo] == 0) && (s.sex[m] == 1);
  return ok && (s.age[fo] >= P.ra_f) && (s.age[m] >= P.ra_m);
}

// + unordered-pair de-duplication (ops/mating.py:62-65): the reference keeps one of
// (i,m),(m,i); drop (i,m) iff (m,i) is also present and id_m < id_i (ids, not slots, so
// every tile of a tiled run takes the same decision).  A pair belongs to the tile that
// owns its focal individual: ghost focals only serve the reciprocity test.  Flags and
// their per-block counts (gnx_compact.h).
__global__ void __launch_bounds__(256)
k_pair_flags(PairP P, GnxSoA s, int32_t* flag2, int32_t* cnt, GnxVtOut vt) {
  __shared__ int lds[16];
  __shared__ int vtot[GNX_VTN];
  if (vt.scls && threadIdx.x < GNX_VTN) vtot[threadIdx.x] = 0;
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  P.N = gnx_dd_n(P.dd, P.N);
  P.step = gnx_dd_step(P.dd, P.step);
  bool f[4];
  // (tile-major ids: everybody's position goes out with the first loads - coalesced, and not one
  // more round trip behind the filters' chain of dependent loads)
  float px[4] = {0.f, 0.f, 0.f, 0.f}, py[4] = {0.f, 0.f, 0.f, 0.f};
  if (vt.scls) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = base + r * 256 + threadIdx.x;
      if (i < P.N) {
        px[r] = s.x[i];
        py[r] = s.y[i];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    f[r] = false;
    if (i < P.N) {
      bool ok = pair_ok(P, s, i);
      if (ok && P.dedup) {
        const int m = P.mate[i];
        if (P.mate[m] == (int32_t)i && s.id[m] < s.id[i] && pair_ok(P, s, m)) ok = false;
      }
      // (a pair belongs to the tile that owns its focal individual: slot i itself, or - panmixia -
      // the individual the trial of slot i drew)
      if (s.ghost[P.focal ? P.focal[i] : (int)i]) ok = false;
      f[r] = ok;
      flag2[i] = ok ? 1 : 0;
    }
  }
  int rank[4], tot;
  gnx_block_ranks(f, rank, tot, lds);
  // (the scan of the block counts stays a launch of its own: done by the last workgroup of
  // this kernel - gnx_count_and_scan - it cost 10 us more than k_block_scan's 6)
  if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
  if (!vt.scls) return;
  // tile-major offspring ids: the kept pairs' virtual tiles and how many of each this block has
  // (the lanes of a wave are neighbours in space: one or two virtual tiles per wave)
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    int c = GNX_VTN;
    if (f[r]) {
      c = gnx_vt_of(vt.V, px[r], py[r]);
      vt.scls[i] = (uint8_t)c;
    }
    unsigned long long todo = __ballot(f[r]);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lc = __shfl(c, leader);
      const unsigned long long same = __ballot(f[r] && c == lc);
      if (lane == leader) atomicAdd(&vtot[lc], __popcll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  if (threadIdx.x < GNX_VTN) {
    // (class-major: the counts of one virtual tile over the blocks lie side by side - what
    // k_pair_compact adds up)
    const int t = vtot[threadIdx.x];
    vt.blk_cnt[(int64_t)threadIdx.x * vt.stride + blockIdx.x] = t;
    if (t) atomicAdd(&vt.total[threadIdx.x], t);
    const unsigned long long nz = __ballot(t != 0);
    if (threadIdx.x == 0) vt.blk_nz[blockIdx.x] = nz;
  }
}

// Pairs in slot order, i.e. in the CANONICAL (hash cell, id) order of their focal individual
// - the order offspring ids are handed out in: it depends on positions and ids only, not
// on storage order or on how the landscape is tiled (the reference's own order is that of
// a Python set, i.e. unspecified: ops/mating.py:63).  Also their midpoints (n_pairs
// density, ops/demography.py:69-70) and, for the tiles' merge of their pair lists, the
// order key (cell << 40 | id) of every pair.
__global__ void __launch_bounds__(256)
k_pair_compact(int64_t N, const int32_t* focal, const int32_t* mate, const int32_t* flag2,
               const int32_t* blk_off, const float* x, const float* y, const uint64_t* popkey,
               int idbits, int32_t* pairs, float* mid_x, float* mid_y, uint64_t* key,
               const int32_t* __restrict__ cnt, int32_t* __restrict__ total_dev,
               int64_t* __restrict__ host, long long seq, const int32_t* __restrict__ extra,
               GnxDD* __restrict__ dd, int dd_births, int64_t dd_cap, GnxBinP bins, GnxVtOut vt) {
  __shared__ int lds[16];
  __shared__ int psum[4];
  __shared__ int vcnt[16][GNX_VTN];
  __shared__ int vpre[GNX_VTN];
  if (dd) N = dd->N;
  if (vt.cls) {
    for (int k = threadIdx.x; k < 16 * GNX_VTN; k += 256) (&vcnt[0][0])[k] = 0;
    // the pairs of the blocks before this one, for the (one to three) virtual tiles this block
    // touches: wave w takes every fourth of them and adds up a stretch of the class's row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long nz = vt.blk_nz[blockIdx.x];
    for (int k = 0; nz; ++k) {
      const int c = __ffsll((long long)nz) - 1;
      nz &= nz - 1;
      if ((k & 3) != wave) continue;
      const int32_t* row = vt.blk_cnt + (int64_t)c * vt.stride;
      int part = 0;
      for (int b2 = lane; b2 < (int)blockIdx.x; b2 += 64) part += row[b2];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
      if (lane == 0) vpre[c] = part;
    }
    // (one device: the virtual tiles' base offsets, in births, from the totals k_pair_flags left)
    if (vt.local && blockIdx.x == 0 && wave == 3) {
      const int tot = vt.total[lane];
      int xs = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int yv = __shfl_up(xs, d);
        if (lane >= d) xs += yv;
      }
      vt.base[lane] = (int64_t)(xs - tot) * vt.lam;
    }
  }
  // cnt != null: no scan kernel ran - every workgroup adds up the block counts before its own
  // (a few coalesced loads from L2: 1 210 counts at the metric size), and workgroup 0, the
  // first to start, adds up all of them and hands the total to the host and to the device
  // (k_block_scan as a launch of its own: 8 us + two dispatch gaps on the step's chain)
  int self_off = 0;
  if (cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int upto = blockIdx.x == 0 ? (int)gridDim.x : (int)blockIdx.x;
    int part = 0;
    for (int i = threadIdx.x; i < upto; i += 256) part += cnt[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
    if (lane == 0) psum[wave] = part;
    __syncthreads();
    const int sum = psum[0] + psum[1] + psum[2] + psum[3];
    self_off = blockIdx.x == 0 ? 0 : sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *total_dev = sum;
      if (dd) {
        // device-driven step: pairs and births of the step (a fixed number per pair), and the
        // slots they need - a step whose offspring do not fit appends none and says so
        int64_t births = (int64_t)sum * dd_births;
        if ((int64_t)dd->N + births > dd_cap) {
          dd->err |= GNX_DD_ERR_SLOTS;
          births = 0;
        }
        dd->P = sum;
        dd->B = (int32_t)births;
      }
      if (host) {
        // (system-scope stores that have completed before the sequence number goes out: no
        // release fence, which would write this XCD's L2 back - gnx_compact.h)
        __hip_atomic_store(&host[0], (int64_t)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (extra)
          __hip_atomic_store(&host[12], (int64_t)*extra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (seq)
          __hip_atomic_store(&host[3], (int64_t)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool f[4];
  int sc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    f[r] = i < N && flag2[i] != 0;
    if (vt.cls && i < N) sc[r] = vt.scls[i];     // (with the flags: no round trip of its own)
  }
  int rank[4], tot;
  gnx_block_ranks(f, rank, tot, lds);            // (barriers: vcnt is zero, vpre written)
  // tile-major offspring ids: the pair's virtual tile (by its focal individual's position), its
  // rank among the block's pairs of the same virtual tile, the block's count per virtual tile
  int vc[4] = {0, 0, 0, 0}, vrw[4] = {0, 0, 0, 0};
  float fx[4], fy[4];
  int fo4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    fo4[r] = f[r] ? (focal ? focal[i] : (int)i) : 0;
    fx[r] = f[r] ? x[fo4[r]] : 0.f;
    fy[r] = f[r] ? y[fo4[r]] : 0.f;
  }
  if (vt.cls) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vc[r] = f[r] ? sc[r] : GNX_VTN;
      gnx_vt_wave_ranks<false>(f[r], vc[r], 1, lane, vcnt[r * 4 + wave], vrw[r]);
    }
    __syncthreads();
  }
  const int32_t bo = cnt ? self_off : blk_off[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    int bin = 0;
    if (f[r]) {
      const int p = bo + rank[r];
      const int m = mate[i];
      const int fo = fo4[r];
      pairs[2 * p] = fo;
      pairs[2 * p + 1] = m;
      const float mx = (fx[r] + x[m]) / 2.0f, my = (fy[r] + y[m]) / 2.0f;
      mid_x[p] = mx;
      mid_y[p] = my;
      const uint64_t k = popkey[i];
      key[p] = ((k >> idbits) << 40) | (k & ((1ull << idbits) - 1ull));
      if (bins.bins) bin = gnx_bin_of(bins, mx, my);
      if (vt.cls) {
        const int wave = threadIdx.x >> 6;
        int before = vpre[vc[r]];
        for (int j = 0; j < r * 4 + wave; ++j) before += vcnt[j][vc[r]];
        vt.cls[p] = (uint8_t)vc[r];
        vt.rank[p] = before + vrw[r];
      }
    }
    // (device-driven step at small sizes: the pair midpoints' density bins in the same launch)
    if (bins.bins) gnx_bin_add(bins.bins, bin, f[r]);
  }
}

// panmixia (structs/species.py:2178-2194): n ~ Binomial(N, b) pairs, both
// members drawn uniformly with replacement, selfing pairs dropped, no
// de-duplication (ops/mating.py:59-65).  Slot i stands for the i-th Bernoulli
// trial of the binomial (kept w.p. b by k_pair_flags) and carries two draws.
__global__ void k_panmixia(int64_t N, const int64_t* id, long long step, unsigned long long seed,
                           int32_t* focal, int32_t* mate) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  uint4 r = gnx_rand4(seed, (unsigned long long)id[i], step, OP_PAIR_KEEP, 1);
  int f = (int)(((unsigned long long)r.x * (unsigned long long)N) >> 32);
  int m = (int)(((unsigned long long)r.y * (unsigned long long)N) >> 32);
  focal[i] = f;
  mate[i] = (m == f) ? -1 : m;
}

int gnx_l_find_pairs(gnx_state* h, const uint8_t* d_keep, int64_t* n_pairs_out, bool with_density) {
  int rc = gnx_l_find_pairs_enqueue(h, d_keep, with_density);
  // (the caller of the whole thing orders the side stream's columns before it looks at the
  // error: keep that order)
  if (rc) {
    *n_pairs_out = 0;
    return rc;
  }
  return gnx_l_find_pairs_finish(h, n_pairs_out);
}

int gnx_l_find_pairs_enqueue(gnx_state* h, const uint8_t* d_keep, bool with_density) {
  int64_t N = h->N;
  h->n_pairs = 0;
  h->pairs_wait = false;
  if (N == 0) return 0;
  const gnx_species_params& sp = h->sp;
  GnxSoA s = h->soa[h->cur];
  int sexed = sp.sexed;
  const int32_t* focal = nullptr;
  gnx_time_begin(h);
  if (sp.mating_radius < 0) {
    focal = h->off_pair;   // scratch, free until births are expanded
    hipLaunchKernelGGL(k_panmixia, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s.id,
                       h->step, h->cfg.seed, h->off_pair, h->mate);
  } else {
    float r = (float)sp.mating_radius;
    float r2 = r * r;
    FocalP fp{N, d_keep, (float)sp.b, sexed, sp.repro_age[0], h->step, h->cfg.seed};
    dim3 grid(gnx_grid(N, FM_PER_BLOCK)), blk(256);
    const uint4* cd = (const uint4*)h->cand;
    static const int easy = getenv("GNX_FM_EASY") ? std::max(1, atoi(getenv("GNX_FM_EASY"))) : 2;
    if (sp.mate_mode == GNX_MATE_NEAREST)
      hipLaunchKernelGGL(k_find_mates<GNX_MATE_NEAREST>, grid, blk, 0, h->stream, fp, s, cd,
                         h->inv_cs, h->cell_start, h->ncx, h->ncy, h->cell_ref, r, r2, easy, h->mate);
    else if (sp.mate_mode == GNX_MATE_INVERSE)
      hipLaunchKernelGGL(k_find_mates<GNX_MATE_INVERSE>, grid, blk, 0, h->stream, fp, s, cd,
                         h->inv_cs, h->cell_start, h->ncx, h->ncy, h->cell_ref, r, r2, easy, h->mate);
    else
      hipLaunchKernelGGL(k_find_mates<GNX_MATE_UNIFORM>, grid, blk, 0, h->stream, fp, s, cd,
                         h->inv_cs, h->cell_start, h->ncx, h->ncy, h->cell_ref, r, r2, easy, h->mate);
  }
  gnx_time_end(h, GNX_K_FIND_MATES, (double)N * 16.0);
  if (!h->perm_rest_late)
    GNXCHK(gnx_permute_rest_launch(h));     // (GNX_PERMUTE_REST_AT=1: not before the mate search)
  gnx_time_begin(h);
  const int nb = (int)((N + GNX_CB - 1) / GNX_CB);
  PairP pp{N, focal, h->mate, d_keep, (float)sp.b, sexed, sp.repro_age[0], sp.repro_age[1],
           (sexed || sp.mating_radius < 0) ? 0 : 1,      // no dedup for sexed / panmictic
           h->step, h->cfg.seed};
  GnxVtOut vto;
  GNXCHK(gnx_vt_out(h, &vto));
  hipLaunchKernelGGL(k_pair_flags, dim3(nb), dim3(256), 0, h->stream, pp, s, h->flag2, h->blk_cnt, vto);
  const int64_t seq = ++h->pin_seq;
  // (the height of the free-block stack rides along: the host's count of it is exact again)
  const bool with_top = h->half_top && h->genomes_assigned;
  static const bool self_scan = !(getenv("GNX_PAIR_SELFSCAN") && atoi(getenv("GNX_PAIR_SELFSCAN")) == 0);
  if (!self_scan)
    GNXCHK(gnx_block_scan(h, 1, N, h->blk_cnt, h->blk_off, h->cnt_dev, h->h_pin_dev + 4, seq, nullptr,
                          with_top ? h->half_top : nullptr));
  // (the population was sorted by gnx_l_sort_by_cell with this idbits: max_id has not moved)
  hipLaunchKernelGGL(k_pair_compact, dim3(nb), dim3(256), 0, h->stream, N, focal, h->mate,
                     h->flag2, h->blk_off, s.x, s.y, h->key64[1], gnx_id_bits(h), h->pairs,
                     h->mid_x, h->mid_y, h->key64[0],
                     self_scan ? (const int32_t*)h->blk_cnt : (const int32_t*)nullptr, h->cnt_dev,
                     h->h_pin_dev + 4, (long long)seq,
                     with_top ? (const int32_t*)h->half_top : (const int32_t*)nullptr,
                     (GnxDD*)nullptr, 0, (int64_t)0, GnxBinP{nullptr, 0.0, 0, 0}, vto);
  gnx_time_end(h, GNX_K_PAIRS, (double)N * 40.0);
  HIPCHK(hipGetLastError());
  if (h->perm_rest_late_ok && !h->tiled) {     // (gnx_step: the births right behind the pair list)
    const int rc_ahead = gnx_l_offspring_ahead(h, h->step_burn, true);
    h->pairs_wait = false;
    GNXCHK(rc_ahead);
  }
  if (h->perm_rest_late) GNXCHK(gnx_permute_rest_launch(h));
  // the pair midpoints' density (ops/demography.py:60-91): on one GPU bins + lattice run on
  // stream3 beside k_offspring and the death probabilities wait for them
  if (with_density && gnx_fused_bins(h)) {
    GNXCHK(gnx_l_lattice_P_async(h, N));
  } else if (with_density) {
    // the density of the pair midpoints reads the pair count on the device (at most N
    // pairs: the grid's bound) while the host waits for its copy in pinned memory
    const bool lds_bins = (size_t)h->lat.nbx * h->lat.nby * sizeof(int32_t) <= 48 * 1024;
    if (lds_bins)
      GNXCHK(gnx_l_density(h, N, h->mid_x, h->mid_y, &h->spl_P, nullptr, h->cnt_dev));
    else
      with_density = false;
  }
  h->pairs_wait = true;
  h->pairs_seq = seq;
  h->pairs_with_top = with_top;
  h->pairs_with_density = with_density;
  return 0;
}

int gnx_l_find_pairs_finish(gnx_state* h, int64_t* n_pairs_out) {
  *n_pairs_out = 0;
  if (!h->pairs_wait) return 0;
  h->pairs_wait = false;
  // the pair count comes from the scan kernel through pinned memory; the kernels queued
  // behind it (pair list, midpoint density) keep the GPU busy while the host goes on
  GNXCHK(gnx_wait_published(h, 4, h->pairs_seq));
  h->n_pairs = h->h_pin[4];
  if (h->pairs_with_top) h->half_free_est = h->h_pin[16];
  *n_pairs_out = h->n_pairs;
  if (!h->pairs_with_density) h->spl_P.valid = false;
  else if (h->n_pairs == 0) h->spl_P.valid = false;
  if (h->xo_launch_policy == 2) GNXCHK(gnx_xo_launch_pending(h));
  return 0;
}

// ---------------------------------------------------------------- births / offspring
// Poisson by multiplication of uniforms (oracle: poisson_knuth), clipped to >= 1
// (ops/mating.py:124-125); keyed by the focal parent's id.
__global__ void k_births(int64_t P, const int32_t* pairs, const int64_t* id, float thr,
                         long long step, unsigned long long seed, int32_t* nb) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  GnxStream st(seed, (unsigned long long)id[pairs[2 * p]], step, OP_BIRTHS);
  int k = 0;
  float prod = 1.0f;
  for (int j = 0; j < 64; ++j) {
    prod = prod * gnx_u01(st.next());
    if (!(prod > thr)) break;
    k++;
  }
  nb[p] = max(k, 1);
}

__global__ void k_expand_births(int64_t P, const int32_t* nb, const int32_t* boff,
                                int32_t* off_pair) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int o = boff[p];
  for (int q = 0; q < nb[p]; ++q) off_pair[o + q] = (int32_t)p;
}

struct OffP {
  int64_t N, B, cap;
  int W, H, n_layers;
  float xmax, ymax, rrx, rry;
  int distr;
  float p1, p2;
  int surf, surf_layer;
  float surf_kappa;
  int sexed;
  float p_male;
  int fixed_nb;          // > 0: every pair has this many births
  int genomes;           // draw recombination keys and start homologues
  int n_paths;
  int64_t id_base, n_free;
  long long step;
  unsigned long long seed;
  // one GPU: the offspring's alleles at the selected loci and its phenotype in the same
  // kernel (tiles wait for the gametes of ghost mates first)
  int fuse_tb, TW;
  const uint64_t* path_sel;
  const uint8_t* dom;
  GnxBinP bins;          // the individuals' density bins (the newborns join the adults)
  const GnxDD* dd;       // device-driven step: N, B, first id and step index from the device
  // the parents' alleles at the selected loci: s.tb, or - the cell sort's permutation of that
  // column is still on its way (GNX_PERMUTE_REST_AT=2) - the buffer it is permuted FROM, through
  // the sort's permutation (pmap[sorted slot] = the slot before the sort)
  const uint64_t* ptb;
  const int32_t* pmap;
  // the births launched AHEAD of the host's read-back of the pair count (gnx_l_offspring_ahead):
  // the grid covers what the population could bear, the pair count comes from the device
  const int32_t* P_dev;
};

// gamete requests of a tiled run: the mate is a ghost (it lives on a neighbour
// tile), so its gamete is computed there and shipped back
struct GnxReq {
  int64_t* pid;      // parent (ghost) id
  int32_t* k;        // local offspring index
  int32_t* key;      // recombination path
  uint8_t* start;    // start homologue
  float* px;         // ghost position (-> owner tile)
  float* py;
  int32_t* count;
};

// one dispersal attempt (ops/movement.py:98-141): returns true when accepted
__device__ __forceinline__ bool disperse_once(float mx, float my, float theta, float dist,
                                              float rrx, float rry, float xmax, float ymax,
                                              float& ox, float& oy) {
  float dx = cosf(theta) * dist;
  float dy = sinf(theta) * dist;
  if (rrx != 1.0f) dx *= rrx;
  if (rry != 1.0f) dy *= rry;
  ox = fminf(fmaxf(mx + dx, 0.0f), xmax);
  oy = fminf(fmaxf(my + dy, 0.0f), ymax);
  // after clipping the only way to fail 0 < x < dim is x == 0
  return ox > 0.0f && oy > 0.0f;
}

// Offspring records (structs/species.py:613-688): ids id_base.. in (pair,
// birth) order with pairs ordered by focal id, parents' midpoint, dispersal,
// age 0, sex, environment, genome row, recombination keys, start homologues.
__global__ void __launch_bounds__(256)
k_offspring(OffP P, GnxSoA s, const float* rast, const int32_t* pairs, const int32_t* off_pair,
            const int32_t* boff, GnxGoff gf,
            int32_t* off_parent, int32_t* off_keys, uint8_t* off_start, GnxReq rq, GnxTraitTab T,
            int32_t* __restrict__ ord_tail) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (P.dd) {
    P.N = P.dd->N;
    P.B = P.dd->B;
    P.id_base = P.dd->max_id + 1;
    P.step = P.dd->step;
  }
  if (P.P_dev) {
    // (a step whose births do not fit: nothing is written, the host raises once it has the count)
    P.B = (int64_t)*P.P_dev * P.fixed_nb;
    if (P.N + P.B > P.cap || (P.genomes && P.B > P.n_free)) return;
  }
  if (k >= P.B) return;
  int64_t p, ord;
  if (P.fixed_nb > 0) {
    p = k / P.fixed_nb;
    ord = k - p * P.fixed_nb;
  } else {
    p = off_pair[k];
    ord = k - boff[p];
  }
  // global offspring index: local k on one GPU, the pair's global offset on tiles; tile-major
  // ids: the virtual tile's base, the blocks before the pair's, the pair's rank in its block
  int64_t gk = k;
  if (gf.cls) {
    const int c = gf.cls[p];
    int64_t rk = gf.rank[p];
    if (gf.pblk) rk += gf.blk_off[(int64_t)gf.pblk[p] * GNX_VTN + c];      // (Poisson: k_pair_cls)
    gk = gf.vt_base[c] + rk * gf.mul + ord;
  } else if (gf.goff) {
    gk = gf.goff[p] + ord;
  }
  int i = pairs[2 * p], m = pairs[2 * p + 1];
  int64_t slot = P.N + k;
  unsigned long long oid = (unsigned long long)(P.id_base + gk);
  // (tile-major ids on one device: the newborns' ids do not ascend with their slots - the
  // id-ordered index takes its tail from here: the gk-th smallest new id sits in this slot;
  // device-driven step: the tail starts behind the dd->N entries of the index)
  if (ord_tail) ord_tail[(P.dd ? P.N : 0) + gk] = (int32_t)slot;
  float mx = (s.x[i] + s.x[m]) / 2.0f;
  float my = (s.y[i] + s.y[m]) / 2.0f;
  // the draws that do not depend on the position
  const uint4 r = gnx_rand4(P.seed, oid, P.step, OP_OFFSPRING, 0);
  const uint8_t st0 = (uint8_t)(r.x & 1u), st1 = (uint8_t)((r.x >> 1) & 1u);
  const int32_t k0 = (int32_t)(((unsigned long long)r.y * (unsigned long long)P.n_paths) >> 32);
  const int32_t k1 = (int32_t)(((unsigned long long)r.z * (unsigned long long)P.n_paths) >> 32);
  const bool tb_regs = P.genomes && P.fuse_tb && P.TW > 0 && P.TW <= 2;
  uint64_t pi[4] = {0, 0, 0, 0}, pm[4] = {0, 0, 0, 0}, ps0[2] = {0, 0}, ps1[2] = {0, 0};
  const int64_t ip = P.pmap ? (int64_t)P.pmap[i] : (int64_t)i;
  const int64_t mp = P.pmap ? (int64_t)P.pmap[m] : (int64_t)m;
  float ox = mx, oy = my;
  for (int a = 0; a < GNX_DISP_ATTEMPTS; ++a) {
    float theta = 0.f;
    if (P.surf != GNX_SURF_NONE) {
      GnxStream st(P.seed, oid, P.step, OP_DISP_SURF, a * 8);
      theta = surf_direction(rast + (int64_t)P.surf_layer * P.H * P.W, P.W, P.H, (int)mx, (int)my,
                             P.surf, P.surf_kappa, st);
    }
    uint4 r = gnx_rand4(P.seed, oid, P.step, OP_DISPERSAL, a);
    if (P.surf == GNX_SURF_NONE) theta = GNX_PI_F * (2.0f * gnx_u01(r.w) - 1.0f);
    float dist = gnx_distance(P.distr, P.p1, P.p2, r);
    if (disperse_once(mx, my, theta, dist, P.rrx, P.rry, P.xmax, P.ymax, ox, oy)) break;
  }
  // sex (structs/species.py:659-662 then structs/individual.py:110-115): a
  // drawn 0 is falsy in `if sex:` and is replaced by a Bernoulli(0.5) draw -
  // kept as is; unsexed species carry a Bernoulli(0.5) sex.
  uint8_t sx;
  uint4 r2 = gnx_rand4(P.seed, oid, P.step, OP_OFFSPRING, 1);
  if (P.sexed && gnx_u01(r2.x) < P.p_male)
    sx = 1;
  else
    sx = gnx_u01(r2.y) < 0.5f ? 1 : 0;
  s.x[slot] = ox;
  s.y[slot] = oy;
  s.age[slot] = 0;
  s.sex[slot] = sx;
  s.id[slot] = (int64_t)oid;
  s.fit[slot] = 1.0f;
  s.ghost[slot] = 0;
  int cx = (int)ox, cy = (int)oy;
  for (int l = 0; l < P.n_layers; ++l)
    s.e[(int64_t)l * P.cap + slot] = rast[((int64_t)l * P.H + cy) * P.W + cx];
  off_parent[2 * k] = i;
  off_parent[2 * k + 1] = m;
  // the genome row comes with the crossover (k_xo_jobs_*): at once for every birth, or
  // after the death draws for the survivors only
  s.grow[slot] = -1;
  if (P.genomes) {
    // start homologues ~ Bernoulli(.5) x2 (ops/mating.py:133); keys ~
    // randint(0, n_paths) x2 (structs/species.py:625)
    off_start[2 * k] = st0;
    off_start[2 * k + 1] = st1;
    off_keys[2 * k] = k0;
    off_keys[2 * k + 1] = k1;
    if (s.ghost[m] && rq.count) {
      int q = atomicAdd(rq.count, 1);
      rq.pid[q] = s.id[m];
      rq.k[q] = (int32_t)k;
      rq.key[q] = k1;
      rq.start[q] = st1;
      rq.px[q] = s.x[m];
      rq.py[q] = s.y[m];
    }
    if (tb_regs) {
      // the parents' words are loaded HERE, not ahead of the dispersal's chain of dependent loads
      // (round 5 had them ride along: 24 registers held across the loop, 120 in all, four waves
      // per SIMD = 262 144 threads on the chip, fewer than a steady-state step's births; 92 now,
      // five waves: profiles/r06_ab_runs.txt)
      {
        const uint64_t* ti = P.ptb + ip * 2 * P.TW;
        const uint64_t* tm = P.ptb + mp * 2 * P.TW;
        for (int w = 0; w < 2; ++w) {
          if (w < P.TW) {
            pi[w] = ti[w];
            pi[2 + w] = ti[P.TW + w];
            pm[w] = tm[w];
            pm[2 + w] = tm[P.TW + w];
            ps0[w] = P.path_sel[(int64_t)k0 * P.TW + w];
            ps1[w] = P.path_sel[(int64_t)k1 * P.TW + w];
          }
        }
      }
      // (gnx_gamete_tb / gnx_phenotype_tb on those words)
      uint64_t* t0 = s.tb + slot * 2 * P.TW;
      uint64_t g0[2], g1[2];
      const uint64_t s0 = st0 ? ~0ull : 0ull, s1 = st1 ? ~0ull : 0ull;
      for (int w = 0; w < 2; ++w) {
        const uint64_t m0 = ps0[w] ^ s0, m1 = ps1[w] ^ s1;
        g0[w] = (pi[w] & ~m0) | (pi[2 + w] & m0);
        g1[w] = (pm[w] & ~m1) | (pm[2 + w] & m1);
        if (w < P.TW) {
          t0[w] = g0[w];
          t0[P.TW + w] = g1[w];
        }
      }
      if (T.n_traits > 0) gnx_phenotype_words(g0[0], g0[1], g1[0], g1[1], T, P.dom, P.cap, slot, s.z);
    } else if (P.fuse_tb) {
      uint64_t* t0 = s.tb + slot * 2 * P.TW;
      if (P.TW > 0) {
        gnx_gamete_tb(P.TW, P.ptb + ip * 2 * P.TW, P.path_sel + (int64_t)k0 * P.TW, st0 != 0, t0);
        gnx_gamete_tb(P.TW, P.ptb + mp * 2 * P.TW, P.path_sel + (int64_t)k1 * P.TW, st1 != 0,
                      t0 + P.TW);
      }
      if (T.n_traits > 0) gnx_phenotype_tb(t0, t0 + P.TW, T, P.dom, P.cap, slot, s.z);
    }
  }
  if (P.bins.bins) gnx_bin_add(P.bins.bins, gnx_bin_of(P.bins, ox, oy), true);
}

// appends B offspring whose parents/keys/starts were uploaded to
// off_parent/off_keys/off_start (operator-level test entry); no positions drawn
__global__ void k_offspring_inject(int64_t N, int64_t B, int64_t cap, GnxSoA s, const float* rast,
                                   int n_layers, int W, int H, const int32_t* off_parent,
                                   int64_t max_id) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= B) return;
  int i = off_parent[2 * k], m = off_parent[2 * k + 1];
  int64_t slot = N + k;
  float ox = (s.x[i] + s.x[m]) / 2.0f, oy = (s.y[i] + s.y[m]) / 2.0f;
  s.x[slot] = ox;
  s.y[slot] = oy;
  s.age[slot] = 0;
  s.sex[slot] = 0;
  s.id[slot] = max_id + 1 + k;
  s.fit[slot] = 1.0f;
  s.ghost[slot] = 0;
  s.grow[slot] = -1;
  int cx = (int)ox, cy = (int)oy;
  for (int l = 0; l < n_layers; ++l)
    s.e[(int64_t)l * cap + slot] = rast[((int64_t)l * H + cy) * W + cx];
}

// number of births of the current pair list (fixed lambda or max(Poisson, 1))
int gnx_l_births(gnx_state* h, int64_t* births_out) {
  const gnx_species_params& sp = h->sp;
  GnxSoA s = h->soa[h->cur];
  int64_t P = h->n_pairs, B = 0;
  if (P > 0) {
    if (sp.n_births_fixed) {
      B = P * (int64_t)sp.n_births_lambda;
    } else {
      float thr = (float)exp(-sp.n_births_lambda);
      hipLaunchKernelGGL(k_births, dim3(gnx_grid(P, 256)), dim3(256), 0, h->stream, P, h->pairs,
                         s.id, thr, h->step, h->cfg.seed, h->nbirths);
      HIPCHK(hipMemsetAsync(h->nbirths + P, 0, sizeof(int32_t), h->stream));
      GNXCHK(gnx_prim_scan(h->scan_tmp, h->scan_tmp_bytes, h->nbirths, h->boff, (size_t)P + 1,
                           h->stream));
      GNXCHK(gnx_publish(h, 0, h->boff + P));
      HIPCHK(hipStreamSynchronize(h->stream));
      B = h->h_pin[0];
      if (B > h->cfg.cap_inds) {
        gnx_set_error("capacity exceeded: births %lld > cap_inds %lld", (long long)B,
                      (long long)h->cfg.cap_inds);
        return 2;
      }
      hipLaunchKernelGGL(k_expand_births, dim3(gnx_grid(P, 256)), dim3(256), 0, h->stream, P,
                         h->nbirths, h->boff, h->off_pair);
      HIPCHK(hipGetLastError());
    }
  }
  h->n_births_pending = B;
  *births_out = B;
  return 0;
}

static OffP gnx_make_offp(gnx_state* h, bool genomes, bool tiled, int64_t id_base, int64_t B) {
  const gnx_config& c = h->cfg;
  const gnx_species_params& sp = h->sp;
  OffP Q;
  Q.dd = nullptr;
  Q.N = h->N;
  Q.B = B;
  Q.cap = c.cap_inds;
  Q.W = c.W;
  Q.H = c.H;
  Q.n_layers = c.n_layers;
  Q.xmax = (float)(c.W - 0.001);
  Q.ymax = (float)(c.H - 0.001);
  Q.rrx = (float)sp.res_ratio[0];
  Q.rry = (float)sp.res_ratio[1];
  Q.distr = sp.disp_distr;
  Q.p1 = (float)sp.disp_p1;
  Q.p2 = (float)sp.disp_p2;
  Q.surf = sp.disp_surf;
  Q.surf_layer = sp.disp_surf_layer;
  Q.surf_kappa = (float)sp.disp_surf_kappa;
  Q.sexed = sp.sexed;
  Q.p_male = (float)sp.p_male;
  Q.fixed_nb = sp.n_births_fixed ? (int)sp.n_births_lambda : 0;
  Q.genomes = genomes ? 1 : 0;
  Q.n_paths = h->n_paths;
  Q.id_base = id_base >= 0 ? id_base : h->max_id + 1;
  Q.n_free = h->n_free;
  Q.step = h->step;
  Q.seed = c.seed;
  // (tiles too since round 4: an offspring whose mate is a ghost gets its tables and phenotype
  // from the mate's resident slot - garbage - and again, from its finished row, once the remote
  // gamete is in: gnx_tile_finish_births)
  static const bool tile_fuse = !(getenv("GNX_TILE_FUSE_TB") && atoi(getenv("GNX_TILE_FUSE_TB")) == 0);
  Q.fuse_tb = (genomes && (!tiled || tile_fuse)) ? 1 : 0;
  Q.TW = h->TW;
  Q.path_sel = h->path_sel;
  Q.dom = h->dom;
  Q.bins = GnxBinP{nullptr, 1.0 / h->lat.hww, h->lat.nbx, h->lat.nby};
  Q.P_dev = nullptr;
  Q.ptb = h->soa[h->cur].tb;
  Q.pmap = nullptr;
  if (h->perm_rest_late && (h->perm_rest_inflight || h->perm_rest_pending)) {
    Q.ptb = h->perm_rest_a.tb;
    Q.pmap = h->perm[1];
  }
  return Q;
}

// gnx_step, one device, a fixed number of births per pair: the births kernel goes on the stream
// BEFORE the host has read the pair count back (gnx_l_find_pairs_finish).  The host's wait for
// the count and its enqueueing of everything behind the births then run while k_offspring
// does, instead of leaving the chip idle between the pair list and the births (~15 us of a
// 0.55-ms step, profiles/r06_timeline.txt) and again behind them.  The kernel takes the pair
// count from the device word k_pair_compact leaves (cnt_dev[0]); its grid covers the births the
// population could have at most (every individual the focal one of a pair); the host's checks
// (capacity, free genome rows) follow in gnx_l_mate once it has the count - a step that does not
// fit has written nothing by then (the kernel checks the same bounds itself).
// Same kernel, same arguments as gnx_l_mate's own launch: the reference's order of events
// (structs/species.py:595-805: pairs, then births) is the stream's.
int gnx_l_offspring_ahead(gnx_state* h, bool burn, bool inside_enqueue) {
  // (measured, profiles/r06_ab_runs.txt: 0.544 against 0.542 ms/step - the gap in front of the
  // births is not the host's wait but the event hand-over to the side stream that precedes them;
  // parity-green, kept behind GNX_BIRTHS_AHEAD=1.  =2: right behind the pair list, in front of
  // that hand-over - gnx_l_find_pairs_enqueue)
  static const int mode = getenv("GNX_BIRTHS_AHEAD") ? atoi(getenv("GNX_BIRTHS_AHEAD")) : 0;
  if (h->births_ahead) return 0;               // (already on the stream)
  const bool on = inside_enqueue ? mode == 2 : mode == 1;
  if (inside_enqueue) h->pairs_wait = true;    // (the enqueue sets it at its end: the test below)
  h->births_ahead = false;
  const gnx_config& c = h->cfg;
  const gnx_species_params& sp = h->sp;
  if (!on || !h->pairs_wait || h->tiled || h->tile2_mode || !sp.n_births_fixed ||
      sp.n_births_lambda < 1 || sp.mating_radius < 0 || h->N == 0 || h->dd_active ||
      (h->profiling && h->profile_only < 0))        // (every family timed: the births by themselves)
    return 0;
  const int64_t lam = (int64_t)sp.n_births_lambda;
  const int64_t room = c.cap_inds - h->N;
  if (room <= 0) return 0;
  GNXCHK(gnx_xo_flush_deferred(h));
  const bool genomes = !burn && c.L > 0 && h->genomes_assigned;
  GnxSoA s = h->soa[h->cur];
  // (tile-major offspring ids: k_pair_compact's classification, flags only - gnx_l_mate repeats it)
  if (h->id_order == 1) GNXCHK(gnx_l_pair_cls(h, -1, true));
  OffP Q = gnx_make_offp(h, genomes, false, -1, 0);
  Q.P_dev = h->cnt_dev;
  if (gnx_fused_bins(h) && h->fb_adults && h->fb_count == h->N) Q.bins.bins = h->fb[h->fb_cur];
  int32_t* ord_tail = nullptr;
  if (h->pair_goff_ready && h->pair_goff_local_base && h->ord_mode && h->ord_valid && h->ord_n == h->N)
    ord_tail = h->ord[h->ord_cur] + h->ord_n;
  GnxGoff gf{};
  if (h->pair_goff_ready) gf = gnx_goff_vt(h);
  const int64_t bound = std::min(h->N * lam, room);
  hipLaunchKernelGGL(k_offspring, dim3(gnx_grid(bound, 256)), dim3(256), 0, h->stream, Q, s, h->rast,
                     h->pairs, h->off_pair, h->boff, gf, h->off_parent, h->off_keys, h->off_start,
                     GnxReq{}, gnx_trait_tab(h), ord_tail);
  HIPCHK(hipGetLastError());
  h->births_ahead = true;
  return 0;
}

// Appends the offspring of the current pair list (or, inject: of the uploaded
// parent list).  tiled: births were counted by gnx_l_births, offspring ids are
// id_base + pair_goff[pair] + ordinal, gametes of ghost mates are requested,
// and phenotypes are left to the caller (they need the remote gametes).
int gnx_l_mate(gnx_state* h, bool burn, bool inject, int64_t B_inject, int64_t* births_out,
               int64_t id_base, bool tiled) {
  const gnx_config& c = h->cfg;
  GnxSoA s = h->soa[h->cur];
  *births_out = 0;
  int64_t B = 0;
  bool ord_tail_used = false;
  const bool ahead = h->births_ahead;       // k_offspring is already on the stream (gnx_l_offspring_ahead)
  h->births_ahead = false;
  if (ahead && (tiled || inject)) {
    gnx_set_error("births launched ahead of a tiled / injected mating");
    return 1;
  }
  GNXCHK(gnx_xo_flush_deferred(h));
  bool genomes = !burn && c.L > 0 && h->genomes_assigned;
  if (inject) {
    B = B_inject;
    if (B == 0) return 0;
    if (h->N + B > c.cap_inds || B > h->n_free) {
      gnx_set_error("capacity exceeded: N=%lld + B=%lld > cap_inds=%lld or free rows %lld",
                    (long long)h->N, (long long)B, (long long)c.cap_inds, (long long)h->n_free);
      return 2;
    }
    hipLaunchKernelGGL(k_offspring_inject, dim3(gnx_grid(B, 256)), dim3(256), 0, h->stream, h->N,
                       B, c.cap_inds, s, h->rast, c.n_layers, c.W, c.H, h->off_parent, h->max_id);
  } else {
    if (!tiled) {
      GNXCHK(gnx_l_births(h, &B));
      // tile-major offspring ids on one device (gnx_set_id_order): the blocks' offsets behind
      // k_pair_compact's classification, or - Poisson births - the classification itself, now
      // that the pairs' birth counts are drawn; the virtual tiles' bases from this device alone
      if (h->id_order == 1 && h->n_pairs > 0 && h->sp.mating_radius >= 0)
        GNXCHK(gnx_l_pair_cls(h, h->n_pairs, true));
    }
    B = h->n_births_pending;
    if (B == 0) {
      h->pair_goff_ready = false;
      return 0;
    }
    if (h->N + B > c.cap_inds || (genomes && B > h->n_free)) {
      gnx_set_error("capacity exceeded: N=%lld + births=%lld > cap_inds=%lld (free rows %lld)",
                    (long long)h->N, (long long)B, (long long)c.cap_inds, (long long)h->n_free);
      return 2;
    }
    OffP Q = gnx_make_offp(h, genomes, tiled, id_base, B);
    // (the adults were counted by k_permute of this very population)
    if (gnx_fused_bins(h) && h->fb_adults && h->fb_count == h->N) {
      Q.bins.bins = h->fb[h->fb_cur];
      h->fb_count += B;
    } else {
      h->fb_adults = false;
    }
    GnxReq rq{};
    if (tiled && genomes) {
      rq.pid = h->req_pid;
      rq.k = h->req_k;
      rq.key = h->req_key;
      rq.start = h->req_start;
      rq.px = h->req_px;
      rq.py = h->req_py;
      rq.count = h->req_count;
      if (!h->req_zeroed) HIPCHK(hipMemsetAsync(h->req_count, 0, sizeof(int32_t), h->stream));
      h->req_zeroed = false;
    }
    // tile-major ids while the id-ordered index is alive (one device, or a single tile): the
    // index gets the newborns in id order from the kernel itself
    int32_t* ord_tail = nullptr;
    if (h->pair_goff_ready && h->pair_goff_local_base && h->ord_mode && h->ord_valid && !h->tiled &&
        h->ord_n == h->N)
      ord_tail = h->ord[h->ord_cur] + h->ord_n;
    ord_tail_used = ord_tail != nullptr;
    GnxGoff gf{};
    if (h->pair_goff_ready) gf = gnx_goff_vt(h);
    else if (tiled && !h->pair_goff_local) gf.goff = h->pair_goff;
    if (!ahead) {
      gnx_time_begin(h);
      hipLaunchKernelGGL(k_offspring, dim3(gnx_grid(B, 256)), dim3(256), 0, h->stream, Q, s, h->rast,
                         h->pairs, h->off_pair, h->boff, gf,
                         h->off_parent, h->off_keys, h->off_start, rq, gnx_trait_tab(h), ord_tail);
      gnx_time_end(h, GNX_K_OFFSPRING, (double)B * (60.0 + 8.0 * c.n_layers + 48.0 * h->TW +
                                                    4.0 * c.n_traits));
    } else if (h->profiling && h->profile_only < 0) {
      h->timers[GNX_K_OFFSPRING].bytes += (double)B * (60.0 + 8.0 * c.n_layers + 48.0 * h->TW +
                                                       4.0 * c.n_traits);
    }
    if (tiled && genomes) {
      // the number of gamete requests is known before the crossover is launched: the host
      // layer serves the neighbour tiles while the crossover runs.  (tile2: it was counted
      // with the pair list - no read-back here)
      if (h->n_req_known >= 0) {
        h->n_req = h->n_req_known;
      } else {
        GNXCHK(gnx_publish(h, 0, h->req_count));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->n_req = h->h_pin[0];
      }
    }
  }
  HIPCHK(hipGetLastError());
  if (genomes || inject) {
    // alleles at the selected loci and the phenotype come from the parents' compact
    // tables; the 25-KB rows are only needed by the NEXT generation's crossover, so on one
    // GPU they are cut after this step's death draws, for the survivors only
    // (gnx_l_mortality), unless somebody asks for them earlier (gnx_xo_join)
    static const bool tile_fuse = !(getenv("GNX_TILE_FUSE_TB") && atoi(getenv("GNX_TILE_FUSE_TB")) == 0);
    const bool fused = !inject && (!tiled || tile_fuse);        // k_offspring did both already
    if (!fused) GNXCHK(gnx_l_newborn_tb(h, h->N, B));
    const bool defer = h->defer_xo && !inject && h->stream2 != nullptr;
    if (defer) {
      // tiles: the few offspring whose mate is a ghost cannot wait (their alleles at the
      // selected loci need the remote gamete): row and local gamete now
      if (tiled) GNXCHK(gnx_l_crossover_requests(h, h->N, h->n_req));
      h->xo_deferred = true;
      h->xo_first = h->N;
      h->xo_B = B;
    } else {
      GNXCHK(gnx_l_crossover_all(h, h->N, B));
    }
    if (c.n_traits > 0 && !tiled && !fused) GNXCHK(gnx_l_phenotype(h, h->N, B));
  }
  h->N += B;
  if (!tiled) h->max_id += B;
  if (h->pair_goff_ready) {
    // tile-major ids: the newborns' ids do not ascend with their slots, which is what the
    // id-ordered index takes its unsorted tail to do: either k_offspring has just filed them
    // (ord_tail), or the next cell sort takes (cell, id) keys
    h->pair_goff_ready = false;
    if (ord_tail_used)
      h->ord_n += B;
    else
      h->ord_valid = false;
  }
  *births_out = B;
  return 0;
}

// ---------------------------------------------------------------- device-driven step
// The launchers of gnx_dd.hip's step: the kernels above with their grids sized by the
// handle's capacity and their counts read from h->dd on the device; no host state moves.
int gnx_dd_l_sort(gnx_state* h, int32_t* d_bins, hipStream_t st) {
  const gnx_config& c = h->cfg;
  const int64_t n_fixed = c.cap_inds;
  GnxSoA a = h->soa[h->cur], b = h->soa[h->cur ^ 1];
  // stable sort of the id-ordered index by cell alone, over the capacity: entries behind the
  // population carry the largest key (k_keys_hist)
  // GNX_DD_SORT_GEO: 1 (default) the front (keys + histograms) in workgroups of 2 048 keys - three
  // times the workgroups, 15.8 -> 9.0 us at 10^5 individuals - and the passes in rocPRIM's own
  // 1024 x 6 tiles (512 x 4 tiles take 16.9 us a pass against 13.7: `2`); 0: both 1024 x 6
  static const int geo = getenv("GNX_DD_SORT_GEO") ? atoi(getenv("GNX_DD_SORT_GEO")) : 1;
  const bool small = n_fixed <= (1 << 21);
  const int geometry = (geo == 2 && small) ? 1 : 0;
  GNXCHK(gnx_os_keys_hist(h->os_scratch, h->tickets + 3, n_fixed, 0, h->ord[h->ord_cur], h->cell32,
                          h->keyk[0], h->valk[0], h->key_bits, st, h->dd,
                          (geo != 0 && small) ? 1 : 0));
  GNXCHK(gnx_os_sort32_ranked(h->os_scratch, h->os_ktmp, h->os_vtmp, h->keyk[0], h->keyk[1],
                              h->valk[0], h->valk[1], (size_t)n_fixed, h->key_bits, st, geometry));
  const int64_t wipe_words = (int64_t)gnx_os_words_used((size_t)n_fixed, h->key_bits, geometry);
  hipLaunchKernelGGL(k_permute, dim3(gnx_grid(n_fixed, 256)), dim3(256), 0, st, n_fixed, c.cap_inds,
                     h->valk[1], a, b, c.n_layers, c.n_traits, a.tb ? 2 * h->TW : 0,
                     (unsigned long long)c.seed, h->tag, (uint4*)h->cand, h->key64[1], 40,
                     h->cell_start, h->ncx * h->ncy, h->keyk[1], h->ord[h->ord_cur], (int64_t)0,
                     h->ord[h->ord_cur ^ 1], h->perm[1], (uint4*)h->os_scratch, (wipe_words + 3) / 4,
                     0, (const GnxDD*)h->dd,
                     GnxBinP{d_bins, 1.0 / h->lat.hww, h->lat.nbx, h->lat.nby},
                     h->id_order == 1 ? h->vt_count : (int32_t*)nullptr);
  HIPCHK(hipGetLastError());
  h->ord_cur ^= 1;
  h->cur ^= 1;
  return 0;
}

int gnx_dd_l_pairs(gnx_state* h, int32_t* d_bins, hipStream_t st) {
  const gnx_species_params& sp = h->sp;
  const int64_t cap = h->cfg.cap_inds;
  GnxSoA s = h->soa[h->cur];
  const float r = (float)sp.mating_radius, r2 = r * r;
  FocalP fp{cap, nullptr, (float)sp.b, sp.sexed, sp.repro_age[0], 0, h->cfg.seed, h->dd};
  dim3 grid(gnx_grid(cap, FM_PER_BLOCK)), blk(256);
  const uint4* cd = (const uint4*)h->cand;
  static const int easy = getenv("GNX_FM_EASY") ? std::max(1, atoi(getenv("GNX_FM_EASY"))) : 2;
  if (sp.mate_mode == GNX_MATE_NEAREST)
    hipLaunchKernelGGL(k_find_mates<GNX_MATE_NEAREST>, grid, blk, 0, st, fp, s, cd, h->inv_cs,
                       h->cell_start, h->ncx, h->ncy, h->cell_ref, r, r2, easy, h->mate);
  else if (sp.mate_mode == GNX_MATE_INVERSE)
    hipLaunchKernelGGL(k_find_mates<GNX_MATE_INVERSE>, grid, blk, 0, st, fp, s, cd, h->inv_cs,
                       h->cell_start, h->ncx, h->ncy, h->cell_ref, r, r2, easy, h->mate);
  else
    hipLaunchKernelGGL(k_find_mates<GNX_MATE_UNIFORM>, grid, blk, 0, st, fp, s, cd, h->inv_cs,
                       h->cell_start, h->ncx, h->ncy, h->cell_ref, r, r2, easy, h->mate);
  const int nb = (int)((cap + GNX_CB - 1) / GNX_CB);
  PairP pp{cap, nullptr, h->mate, nullptr, (float)sp.b, sp.sexed, sp.repro_age[0], sp.repro_age[1],
           sp.sexed ? 0 : 1, 0, h->cfg.seed, h->dd};
  GnxVtOut vto;
  GNXCHK(gnx_vt_out(h, &vto));
  hipLaunchKernelGGL(k_pair_flags, dim3(nb), dim3(256), 0, st, pp, s, h->flag2, h->blk_cnt, vto);
  // the pair list, in slot order; workgroup 0 adds up the block counts and leaves the pair
  // count, the births (a fixed number per pair) and the capacity check in the device block
  hipLaunchKernelGGL(k_pair_compact, dim3(nb), dim3(256), 0, st, cap, (const int32_t*)nullptr,
                     h->mate, h->flag2, h->blk_off, s.x, s.y, h->key64[1], 40, h->pairs, h->mid_x,
                     h->mid_y, h->key64[0], (const int32_t*)h->blk_cnt, h->cnt_dev,
                     (int64_t*)nullptr, 0ll, (const int32_t*)nullptr, h->dd,
                     (int)sp.n_births_lambda, (int64_t)cap,
                     GnxBinP{d_bins, 1.0 / h->lat.hww, h->lat.nbx, h->lat.nby}, vto);
  if (vto.cls) {
    h->vt_mul = (int64_t)sp.n_births_lambda;
    h->vt_weighted = false;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_dd_l_offspring(gnx_state* h, bool genomes, int32_t* d_bins, hipStream_t st) {
  OffP Q = gnx_make_offp(h, genomes, false, 0, 0);
  Q.dd = h->dd;
  Q.bins.bins = d_bins;
  // (a pair has a fixed number of births here: at most every second individual is a focal one
  // of a kept pair, so the capacity bounds the grid generously; the kernel reads B itself)
  // (tile-major ids: the offsets from k_pair_compact's classification; the newborns' ids do not
  // ascend with their slots then - the kernel files them behind the index's dd->N entries)
  const bool vt = h->vt_fused && h->id_order == 1;
  hipLaunchKernelGGL(k_offspring, dim3(gnx_grid(h->cfg.cap_inds, 256)), dim3(256), 0, st, Q,
                     h->soa[h->cur], h->rast, h->pairs, h->off_pair, h->boff,
                     vt ? gnx_goff_vt(h) : GnxGoff{}, h->off_parent, h->off_keys, h->off_start,
                     GnxReq{}, gnx_trait_tab(h), vt ? h->ord[h->ord_cur] : (int32_t*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

// ops/movement.py:98-141 with injected attempts (parity tests)
__global__ void k_dispersal_inject(int64_t B, int A, const float* mx, const float* my,
                                   const float* theta, const float* dist, float rrx, float rry,
                                   float xmax, float ymax, float* ox, float* oy, int32_t* used) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= B) return;
  float x = 0.f, y = 0.f;
  int u = A - 1;
  for (int a = 0; a < A; ++a) {
    if (disperse_once(mx[k], my[k], theta[(int64_t)a * B + k], dist[(int64_t)a * B + k], rrx, rry,
                      xmax, ymax, x, y)) {
      u = a;
      break;
    }
  }
  ox[k] = x;
  oy[k] = y;
  used[k] = u;
}

int gnx_l_dispersal_inject(gnx_state* h, int64_t B, int A, const float* d_mx, const float* d_my,
                           const float* d_theta, const float* d_dist, float* d_ox, float* d_oy,
                           int32_t* d_used) {
  const gnx_config& c = h->cfg;
  hipLaunchKernelGGL(k_dispersal_inject, dim3(gnx_grid(B, 256)), dim3(256), 0, h->stream, B, A,
                     d_mx, d_my, d_theta, d_dist, (float)h->sp.res_ratio[0],
                     (float)h->sp.res_ratio[1], (float)(c.W - 0.001), (float)(c.H - 0.001), d_ox,
                     d_oy, d_used);
  HIPCHK(hipGetLastError());
  return 0;
}
