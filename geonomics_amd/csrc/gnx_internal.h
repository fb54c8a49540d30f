// Internal state and launcher declarations of libgnxhip.so (not part of the ABI).
#pragma once
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <utility>
#include <hip/hip_runtime.h>
#include "../../include/gnx_hip.h"
#include "gnx_half.h"

#define GNX_MAX_TRAITS 16
#define GNX_MAX_LAYERS 16
#define GNX_DISP_ATTEMPTS 8
#define GNX_SPARSE_MAX_BP 24     // paths with more switches than this use bit masks

void gnx_set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                          \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      gnx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                    __FILE__, __LINE__);                                      \
      return 1;                                                               \
    }                                                                         \
  } while (0)

#define GNXCHK(expr)                                                          \
  do {                                                                        \
    int _r = (expr);                                                          \
    if (_r) return _r;                                                        \
  } while (0)

struct GnxTrait {
  int n_loci = 0;
  int32_t* loci = nullptr;     // device [n_loci]
  double* alpha = nullptr;     // device [n_loci]
  int layer = 0;
  double phi = 0;
  float* phi_rast = nullptr;   // device [H][W] or null
  double gamma = 1;
  int univ_adv = 0;
};

// Per-trait table handed to kernels by value
struct GnxTraitTab {
  int n_traits;
  int n_loci[GNX_MAX_TRAITS];
  const int32_t* loci[GNX_MAX_TRAITS];
  const double* alpha[GNX_MAX_TRAITS];
  int layer[GNX_MAX_TRAITS];
  double phi[GNX_MAX_TRAITS];
  const float* phi_rast[GNX_MAX_TRAITS];
  double gamma[GNX_MAX_TRAITS];
  int univ_adv[GNX_MAX_TRAITS];
};

// Struct-of-arrays view of the population (one of the two ping-pong copies)
struct GnxSoA {
  float* x;
  float* y;
  int32_t* age;
  uint8_t* sex;
  int64_t* id;
  float* e;        // [n_layers][cap]
  float* z;        // [n_traits][cap]
  float* fit;
  int32_t* grow;   // genome row, -1 before genomes are assigned
  uint8_t* ghost;  // 1 = halo copy of a neighbour tile's individual (tiled runs)
  // alleles at the SELECTED loci (trait loci trait-major, then deleterious loci), bit e of
  // tb[(i*2 + hom)*TW + e/64]: phenotype, fitness and the newborns' alleles at these loci
  // never touch the 25-KB genome rows
  uint64_t* tb;
};

// One individual's record in registers.  The kernels that move records (k_permute,
// k_compact, k_fill) load ALL of it before they store any of it: written column by column
// (b.x[k] = a.x[i]; b.y[k] = a.y[i]; ...) every load waits for the store before it - the
// compiler cannot know that the two views never overlap - and a thread makes a dozen
// dependent memory round trips instead of one.  Up to 4 layers, 4 traits and 4 words of the
// selected-locus table ride in registers; more than that is copied by gnx_rec_rest.
struct GnxRec {
  float x, y, fit;
  int32_t age, grow;
  int64_t id;
  uint8_t sex, ghost;
  float e[4], z[4];
  uint64_t tb[4];
};
#ifdef __HIPCC__
__device__ __forceinline__ GnxRec gnx_rec_load(const GnxSoA& a, int64_t i, int64_t cap, int nl,
                                               int nt, int tbw) {
  GnxRec r;
  r.x = a.x[i];
  r.y = a.y[i];
  r.age = a.age[i];
  r.sex = a.sex[i];
  r.id = a.id[i];
  r.fit = a.fit[i];
  r.grow = a.grow[i];
  r.ghost = a.ghost[i];
#pragma unroll
  for (int l = 0; l < 4; ++l) r.e[l] = l < nl ? a.e[(int64_t)l * cap + i] : 0.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t) r.z[t] = t < nt ? a.z[(int64_t)t * cap + i] : 0.0f;
#pragma unroll
  for (int w = 0; w < 4; ++w) r.tb[w] = w < tbw ? a.tb[i * tbw + w] : 0ull;
  return r;
}
__device__ __forceinline__ void gnx_rec_store(const GnxSoA& b, int64_t k, int64_t cap, int nl, int nt,
                                              int tbw, const GnxRec& r) {
  b.x[k] = r.x;
  b.y[k] = r.y;
  b.age[k] = r.age;
  b.sex[k] = r.sex;
  b.id[k] = r.id;
  b.fit[k] = r.fit;
  b.grow[k] = r.grow;
  b.ghost[k] = r.ghost;
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (l < nl) b.e[(int64_t)l * cap + k] = r.e[l];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nt) b.z[(int64_t)t * cap + k] = r.z[t];
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < tbw) b.tb[k * tbw + w] = r.tb[w];
}
// layers, traits and table words beyond the four that GnxRec holds
__device__ __forceinline__ void gnx_rec_rest(const GnxSoA& a, int64_t i, const GnxSoA& b, int64_t k,
                                             int64_t cap, int nl, int nt, int tbw) {
  for (int l = 4; l < nl; ++l) b.e[(int64_t)l * cap + k] = a.e[(int64_t)l * cap + i];
  for (int t = 4; t < nt; ++t) b.z[(int64_t)t * cap + k] = a.z[(int64_t)t * cap + i];
  for (int w = 4; w < tbw; ++w) b.tb[k * tbw + w] = a.tb[i * tbw + w];
}
#endif

// Density lattice (utils/spatial.py _DensityGridStack restated)
struct GnxLattice {
  int Jx = 0, Jy = 0;     // nodes per axis
  int nbx = 0, nby = 0;   // half-window bins per axis (= J)
  double hww = 0;         // half window width = node spacing
  double* areas = nullptr;     // device [Jy][Jx]
  double* cprime = nullptr;    // device [max(Jx,Jy)] Thomas factors
};

// spline coefficient set for one density field: V, Mx, My, Mxy, each [Jy][Jx]
struct GnxSpline {
  double* c = nullptr;   // device [4][Jy*Jx]
  bool valid = false;
};

struct GnxKernelTimer {
  double ms = 0;
  int64_t launches = 0;
  double bytes = 0;
};

// ---- the device-driven step (gnx_dd.hip) ------------------------------------------------
// gnx_step keeps the population's counts on the HOST: it reads the pair count and the
// survivor count back in every step and sizes the next kernels with them - two round trips
// and ~45 runtime calls per step, which at 10^5 individuals cost more host time than the
// kernels take on the GPU (GNX_HOST_TIMES: 185 us of enqueueing against ~150 us of kernels).
// gnx_walk keeps them on the DEVICE: every kernel of the step reads what it needs from this
// block, its grid is sized by the handle's capacity, nothing is read back - so a whole step
// is one HIP graph, replayed.  The host looks at a copy the step's last kernel leaves in
// pinned memory (GnxDDRec, one step or more late) for its totals, the collector and errors.
struct GnxDD {
  int32_t N;          // occupied slots at the start of the step (after the last compaction)
  int32_t P;          // pairs of this step
  int32_t B;          // births of this step
  int32_t n_free;     // free genome rows on the stack at the start of the step
  int32_t ord_n;      // entries of the id-ordered index
  int32_t err;        // sticky: GNX_DD_ERR_*
  int32_t seq;        // steps carried out since gnx_dd entered
  int32_t pad;
  int64_t max_id;
  int64_t step;       // the step index of the random streams
};
#define GNX_DD_ERR_SLOTS 1     // N + births > cap_inds
#define GNX_DD_ERR_ROWS 2      // surviving offspring > free genome rows
#define GNX_DD_ERR_BLOCKS 4    // the free-block stack ran dry
// what a step leaves for the host (pinned memory, ring of GNX_DD_RING records)
struct GnxDDRec {
  int64_t seq;        // written last: 1-based number of the step since gnx_dd entered
  int32_t N0, P, B, S;        // population at the start, pairs, births, survivors
  int32_t xo, n_free, half_top, err;   // births that got a genome, free rows / blocks after the step
  int64_t max_id, step;
  int64_t pad[2];
};
#define GNX_DD_RING 4096
#ifdef __HIPCC__
// N / step of a kernel of either step: from the device block when there is one
__device__ __forceinline__ int64_t gnx_dd_n(const GnxDD* dd, int64_t n) { return dd ? (int64_t)dd->N : n; }
__device__ __forceinline__ int64_t gnx_dd_nb(const GnxDD* dd, int64_t n) {
  return dd ? (int64_t)dd->N + dd->B : n;                      // ... with this step's offspring
}
__device__ __forceinline__ long long gnx_dd_step(const GnxDD* dd, long long s) {
  return dd ? (long long)dd->step : s;
}
// Which of n equal stretches of tw raster cells coordinate v falls in.  EXACT on the integer
// boundaries k * tw: a reciprocal (v * (n / W)) puts the float just below a boundary on the
// wrong side whenever n / W is not a power of two (W = 1000: x = 499.99997 -> stretch 4 of 8) -
// tile ownership, the routing and the virtual tiles of the offspring ids all use this one.
__device__ __forceinline__ int gnx_tile_index(float v, int tw, int n) {
  int c = (int)(v / (float)tw);
  c = max(0, min(n - 1, c));
  while (c + 1 < n && (float)((c + 1) * tw) <= v) ++c;     // exact: boundaries are integers
  while (c > 0 && (float)(c * tw) > v) --c;
  return c;
}
#endif

#define GNX_MAX_TILES 4096
// the tile grid as the routing sees it (gnx_tile.hip: route_geo): tile sizes, this tile, the hash
// grid, and the hash-cell span of every tile column / row widened by the halo's two rings
#define GNX_TILE_DIM 64          // tiles per axis the routing kernels hold spans for
struct RouteGeo {
  int R, C, tw, th, me, ncx, ncy;
  double inv_cs;
  int cx0[GNX_TILE_DIM], cx1[GNX_TILE_DIM], cy0[GNX_TILE_DIM], cy1[GNX_TILE_DIM];
};
#ifdef __HIPCC__
// index inside the group of rank `dest` for every lane with dest >= 0: one atomic per wave
// and distinct destination (lanes of a wave mostly share theirs).  Called by the whole wave.
__device__ __forceinline__ int gnx_route_append(int32_t* __restrict__ counts, int dest) {
  unsigned long long todo = __ballot(dest >= 0);
  const int lane = threadIdx.x & 63;
  int idx = -1;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int d = __shfl(dest, leader);
    const unsigned long long same = __ballot(dest == d);
    int base = 0;
    if (lane == leader) base = atomicAdd(&counts[d], (int)__popcll(same));
    base = __shfl(base, leader);
    if (dest == d) idx = base + (int)__popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  return idx;
}
// the (up to nine) destinations of an individual at (x, y): slot 4 = the tile that owns it if that
// is another one (a migrant), the others = the tiles around its owner whose widened cell span
// holds its hash cell (a ghost there); -1 = none
__device__ __forceinline__ int gnx_route_dest(const RouteGeo& g, int k, int orow, int oc, int cx, int cy) {
  if (k == 4) return (orow * g.C + oc) != g.me ? orow * g.C + oc : -1;
  const int rr = orow + k / 3 - 1, cc = oc + k % 3 - 1;
  if (rr >= 0 && rr < g.R && cc >= 0 && cc < g.C && cx >= g.cx0[cc] && cx <= g.cx1[cc] &&
      cy >= g.cy0[rr] && cy <= g.cy1[rr])
    return rr * g.C + cc;
  return -1;
}
#endif
struct gnx_state {
  gnx_config cfg{};
  gnx_species_params sp{};
  bool have_sp = false;
  hipStream_t stream = nullptr;
  // tiled runs: while this step's crossover is in flight on `stream`, the gamete service
  // for the neighbour tiles (requests, lookups, cuts, puts) runs on `stream2`
  hipStream_t stream2 = nullptr;
  bool xo_pending = false;
  bool own_stream = false;
  int W64 = 0;                 // u64 words per homologue
  int64_t N = 0;
  int64_t max_id = -1;
  int64_t step = 0;            // global step counter (RNG addressing)
  int64_t last_births = 0, last_deaths = 0;
  int64_t tot[6]{};            // gnx_totals: steps, sum of N at step start, births, deaths, crossover births
  int64_t n_ghost = 0;         // ghosts currently resident (counted in N)

  GnxSoA soa[2]{};
  int cur = 0;

  // genomes
  // [cap_rows * row_spread][2][W64]: logical row r lives at physical row r * row_spread.
  // The crossover's achieved bandwidth depends on how much of the HBM address space its
  // rows cover (profiles/r02_xo_lab_footprint.txt: 5.6 TB/s on a 41-GB table, 6.1 TB/s on the
  // same rows spread over 82 GB), so the table is spread when memory allows; `grow`, the
  // free stack and the crossover jobs hold PHYSICAL row numbers.
  uint64_t* G = nullptr;
  int row_spread = 1;
  int32_t* free_rows = nullptr;
  int64_t n_free = 0;
  // physical blocks (gnx_half.h): hmap / half_mark over 2 * cap_rows * row_spread * NB blocks,
  // the stack of free ones and its height on the device
  int32_t* hmap = nullptr;
  uint8_t* half_mark = nullptr;        // physical block referred to by somebody alive (gnx_gc)
  int32_t* half_free = nullptr;
  int32_t* half_top = nullptr;
  int64_t half_free_est = 0;           // free blocks the host can count on (a lower bound)
  int64_t gc_runs = 0;
  void* xo_plan = nullptr;             // [cap] GnxXoPlan: what the job builder decided per offspring
  int32_t* gc_cnt = nullptr;           // block counts / offsets of the collector's sweep
  int32_t* gc_off = nullptr;
  int NB = 1;                          // blocks per homologue (gnx_half.h), BW = W64 / NB words
  int NB_alloc = 1;                    // what the tables were sized for (NB <= NB_alloc)
  // words per block.  W64 / NB, or - GNX_BLOCK_LINES - whole 128-byte lines with a last block
  // that reaches past the homologue (NB * BW > W64: its tail is padding nobody reads)
  int BW = 0, BW_alloc = 0;
  hipStream_t stream3 = nullptr;       // the compaction of the id-ordered sort index
  hipEvent_t ev_compact = nullptr;
  bool alias_xo = true;          // blocks without a switch point are shared with the parent
  unsigned long long* xo_jobs_acc = nullptr;   // [2] gametes copied by the (wide, tail) launches
  bool genomes_assigned = false;

  // landscape
  float* rast = nullptr;       // [n_layers][H][W]

  // recombination paths
  int n_paths = 0;
  uint64_t* paths = nullptr;   // [n_paths][W64]
  int32_t* bp_off = nullptr;   // [n_paths+1]
  int32_t* bp_loci = nullptr;
  bool sparse_paths = false;

  // traits etc
  GnxTrait traits[GNX_MAX_TRAITS];
  // selected loci: all trait loci concatenated trait-major (n_tl of them), then the
  // deleterious loci; GnxSoA.tb holds every individual's alleles there and
  // path_sel[key] the homologue each cached recombination path is on there
  std::vector<int32_t> h_trait_loci[GNX_MAX_TRAITS];
  std::vector<int32_t> h_delet_loci;
  int n_tl = 0;
  int n_sel = 0, TW = 0;         // selected loci, u64 words per homologue of GnxSoA.tb
  int32_t* sel_loci = nullptr;   // device [n_sel]
  uint64_t* path_sel = nullptr;  // device [n_paths][TW]
  uint8_t* dom = nullptr;
  int n_delet = 0;
  int32_t* delet_loci = nullptr;
  double* delet_s = nullptr;

  // crossover jobs (csrc/gnx_xo.h).  Deferred mode (one GPU): the offspring of a step get
  // their genome rows and their crossover only after the step's death draws, survivors
  // only, on stream2 - under the next step's latency-bound kernels on `stream`.
  bool defer_xo = true;          // GNX_DEFER_XO=0: crossover of every birth at once
  bool xo_deferred = false;      // births of this step still wait for their crossover
  int64_t xo_first = 0, xo_B = 0;
  void* jobs[2]{};               // GnxXoJob [2 * cap] each, double-buffered
  int32_t* n_jobs_dev[2]{};
  void* jobs_bp[2]{};            // GnxJobBp beside every job: the switch points inside its block
  bool jobs_inline[2]{};         // ... written for this buffer's jobs (the fused builder does)
  int jobs_cur = 0;
  hipEvent_t ev_jobs = nullptr, ev_xo_done[2]{};
  hipEvent_t ev_counts = nullptr;   // the step's counts have reached pinned host memory
  bool xo_inflight[2]{};         // ev_xo_done[k] recorded and not yet joined
  hipEvent_t ev_xo_wide[2]{};    // end of the full-width share of a split launch
  bool xo_wide_inflight[2]{};
  int xo_split = 0;              // /1024 of the jobs that run at full width (0 = no split)
  int xo_last_split = 0;         // the split of the latest deferred launch
  bool xo_running = false;       // a crossover may still be running on stream2
  // How the deferred crossover shares the chip with the next step's small kernels
  // (measured: profiles/r02_xo_overlap_*.txt).  They are latency-bound chains (index
  // sampling, look-back sort and scans, spline gathers) and crawl 3-12x while a
  // bandwidth-bound kernel saturates HBM, and the crossover loses a third of its rate to
  // them.  Default: the crossover is launched as soon as its jobs are built, at full
  // width, and runs under the compaction, the next step's movement and sort keys; `stream`
  // then WAITS for it before the cell sort (xo_sort_waits).  GNX_XO_SORT_WAIT=0 lets the
  // whole next step run beside a narrow crossover (2 workgroups per CU): ~10 % more
  // individual-timesteps/s, but the crossover then stretches over the whole step.
  // xo_launch_policy 1 / 2 launch it after the next step's cell sort / pair sort instead.
  int xo_launch_policy = 0;
  int xo_launch_default = 0;     // ... as chosen at gnx_create (gnx_set_crossover_overlap(0) returns to it)
  bool xo_sort_waits = true;
  // where `stream` waits for the full-width crossover (GNX_XO_WAIT): 1 before the next cell
  // sort, 2 right after launching it (strictly serial), 3 after the compaction
  int xo_wait_at = 1;
  int xo_ready_buf = -1;         // jobs built, kernel not launched yet
  int64_t xo_ready_jobs = 0;
  int64_t last_xo_births = 0;    // births that went through the last crossover

  // hash grid for neighbour search
  double cs = 1, inv_cs = 1;
  int ncx = 1, ncy = 1, key_bits = 1;
  // cells that cover the mating radius: 1 (cell >= radius: uniform / inverse-distance index
  // sampling over the 3 x 3 block) or up to 8 (nearest-mate search: fine cells, ring by ring)
  int cell_ref = 1;
  uint32_t* key[2]{};
  int32_t* perm[2]{};
  int32_t* cell_start = nullptr;
  void* cand = nullptr;          // uint4 {x, y, tag, id_lo} per individual, sorted order
  uint32_t* tag = nullptr;       // per-individual mate-choice tag of this step (sorted order)
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  void* scan_tmp = nullptr;
  size_t scan_tmp_bytes = 0;
  void* sort64_tmp = nullptr;
  size_t sort64_tmp_bytes = 0;
  uint64_t* key64[2]{};          // 64-bit sort keys (focal ids of pairs; id -> slot lookups)
  int32_t* pairs2 = nullptr;     // pairs in ascending focal-id order
  int64_t* pair_goff = nullptr;  // tiled runs: global offspring offset of each local pair
  bool pair_goff_local = false;   // tile2, one tile: offspring ids from the local pair offsets
  // tile-major offspring ids (gnx_set_id_order, gnx_kernels_pop.hip): virtual tile and in-tile
  // rank of every pair, per-block and total counts per virtual tile, the tiles' base offsets
  int id_order = 0;
  bool pair_goff_ready = false;   // the tile-major pieces below hold this step's offsets (reset by the births)
  bool pair_goff_local_base = false;   // ... numbered from this device's own counts alone
  bool vt_fused = false;          // k_pair_compact classified this step's pairs (fixed births)
  int64_t vt_mul = 1;             // births per unit of vt_rank / vt_count: lambda, or 1 (Poisson)
  bool vt_weighted = false;       // this step's ranks came from k_pair_cls (Poisson births)
  uint8_t* vt_cls = nullptr;
  uint8_t* vt_scls = nullptr;
  unsigned long long* vt_blk_nz = nullptr;
  int32_t *vt_rank = nullptr, *vt_pblk = nullptr, *vt_blk_cnt = nullptr, *vt_blk_off = nullptr,
          *vt_count = nullptr;
  int64_t* vt_base = nullptr;
  int64_t n_births_pending = 0;  // births of the current pair list
  bool step_burn = false;        // gnx_step_begin: this step is a burn-in step
  bool births_ahead = false;     // k_offspring of the current pair list is already on the stream (gnx_l_offspring_ahead)
  // gamete requests (tiled runs)
  int64_t* req_pid = nullptr;
  int32_t* req_k = nullptr;
  int32_t* req_key = nullptr;
  uint8_t* req_start = nullptr;
  float* req_px = nullptr;
  float* req_py = nullptr;
  int32_t* req_count = nullptr;
  // tiling: uniform R x C grid of tiles over the landscape, this handle owns (r, c)
  bool tiled = false;
  int tile_R = 1, tile_C = 1, tile_r = 0, tile_c = 0;
  gnx_ind_rec* st_rec = nullptr;   // staged selection (migrants / halo)
  float* st_z = nullptr;
  uint64_t* st_geno = nullptr;
  int64_t* st_slots = nullptr;
  int64_t st_n = 0, st_cap = 0, st_geno_cap = 0;
  bool st_has_geno = false;
  int64_t birth_first_slot = 0;
  int64_t n_req = 0;
  // device-resident transport: the staged selection grouped by destination rank
  gnx_ind_rec* gp_rec = nullptr;
  float* gp_z = nullptr;
  int64_t* gp_slots = nullptr;
  int64_t gp_n = 0, gp_cap = 0;
  int32_t* tile_counts = nullptr;    // [GNX_MAX_TILES] per-destination counts
  void* rq_sorted = nullptr;         // gnx_gamete_req [n_req] grouped by owner rank
  int32_t* rq_k = nullptr;           // child index of each grouped request
  int64_t rq_cap = 0;
  uint64_t* gam_out = nullptr;       // gametes cut for other tiles
  int32_t* gam_slot = nullptr;
  int64_t gam_cap = 0;
  int64_t* chk = nullptr;            // [2] device-side record check: bad count, max id
  // tile2 protocol (gnx_tile.hip, "device-driven"): emigrants stay in their slots until the
  // cell sort, which gives them a key behind every cell (tile_evict individuals) and the
  // population shrinks by that many; gamete requests are counted with the pairs, so the
  // offspring kernel's own count is not read back (n_req_known >= 0)
  int64_t tile_evict = 0;
  float evict_box[4]{};              // own tile [x0, x1) x [y0, y1)
  int64_t n_req_known = -1;
  gnx_ind_rec* gh_rec = nullptr;     // ghosts routed to other tiles, grouped by rank
  int64_t gh_cap = 0;
  int32_t* route_cnt = nullptr;      // [4 * GNX_MAX_TILES + 8] counts / offsets of the routing
  int64_t route_n_mig = 0, route_n_gh = 0;
  int64_t prev_deaths = 0;           // deaths of the previous step (counter all-reduce)
  int32_t* h_route_pin = nullptr;    // pinned host words of the tile2 read-backs
  int32_t* h_route_pin_dev = nullptr;
  std::vector<int64_t> req_by_rank;  // gamete requests per owning rank (gnx_tile2_pairs)
  bool rq_is2 = false;               // rq_sorted holds 24-byte gnx_gamete_req2 records
  bool tile_req_on_device = false;   // gnx_tile2_pairs leaves the request counts in route_cnt (no wait)
  bool tile2_mode = false;           // the handle steps through the tile2 protocol (set by its entry points)
  bool req_cnt_zeroed = false;       // the request counters were cleared by k_tile2_zero
  bool tile_pairs_nowait = false;    // ... and does not wait for the pair count either (gnx_tile_step)
  bool tile_births_settled = false;  // the offspring with a remote gamete have re-read their rows

  // pairing / mating scratch (capacity cap_inds)
  int32_t* mate = nullptr;
  int32_t* flag = nullptr;
  int32_t* flag2 = nullptr;
  int32_t* scan = nullptr;
  int32_t* pairs = nullptr;      // [cap][2] slots
  int32_t* nbirths = nullptr;
  int32_t* boff = nullptr;
  int32_t* off_pair = nullptr;
  int32_t* off_parent = nullptr; // [cap][2] parent slots
  int32_t* off_keys = nullptr;   // [cap][2]
  uint8_t* off_start = nullptr;  // [cap][2]
  uint8_t* keep_in = nullptr;
  float* inj_a = nullptr;        // injected draws (theta)
  float* inj_b = nullptr;        // injected draws (dist)
  float* mid_x = nullptr;        // pair midpoints
  float* mid_y = nullptr;
  int64_t n_pairs = 0;
  // the two host read-backs of a step, each split into the half that enqueues and the half
  // that waits (gnx_step_begin / _mid / _end; several handles stepped side by side enqueue
  // all their first halves before any of them waits): what the first half left behind
  bool pairs_wait = false;           // gnx_l_find_pairs_enqueue ran, its count is not read yet
  int64_t pairs_seq = 0;
  bool pairs_with_top = false, pairs_with_density = false;
  bool mort_wait = false;            // gnx_l_mortality_enqueue ran, its counts are not read yet
  bool mort_xo = false, mort_fill = false, mort_ord_keep = false;
  int mort_has_rows = 0;
  int64_t mort_N = 0;

  // density
  GnxLattice lat;
  GnxSpline spl_N, spl_P;
  int32_t* bin_partials = nullptr;   // half-window bin counts of individuals [nby*nbx]
  int32_t* bins_P = nullptr;         // ... of pair midpoints
  bool bins_zeroed[2]{};             // [0] individuals, [1] pairs: already cleared by k_lattice
  bool nmax_zeroed = false;          // nmax_bits already cleared by k_lattice
  bool req_zeroed = false;           // req_count already cleared (tile2: k_tile2_zero)
  // The density path of one GPU without a counting pass (gnx_bins.h): the bins are counted
  // beside the step's serial chain - the adults on stream3 under the mate search, the pair
  // midpoints and their lattice on stream3 under k_offspring, the newborns by k_offspring
  // itself - into buffers of their own: fb[0], fb[1]
  // the individuals' (alternating from step to step: the lattice kernel of one step clears
  // the other step's buffer, it cannot clear the one its own workgroups still read), fb[2]
  // the pairs'.  The pairs' lattice runs on stream3 beside k_offspring.
  int32_t* fb[3]{};
  bool fb_zero[3]{};                 // buffer k holds zeros
  int fb_cur = 0;                    // the individuals' buffer of this step
  bool fb_adults = false;            // fb[fb_cur] holds the counts of ...
  int64_t fb_count = 0;              // ... this many individuals (adults, + newborns once born)
  unsigned long long* nmax2 = nullptr;           // [2] N.max() words, alternating like fb
  const unsigned long long* nmax_cur = nullptr;  // the word the death probabilities read
  bool nmax_ready = false;           // nmax_cur was filled by k_lattice_nmax
  bool last_N_fused = false;         // the last individuals' density came from fb (gnx_get_bins)
  hipEvent_t ev_pairs = nullptr, ev_latP = nullptr, ev_perm = nullptr, ev_binsN = nullptr;
  bool binsN_inflight = false;       // the adults are being counted on stream3
  bool latP_inflight = false;        // spl_P is being written on stream3
  // An index of the slots in ascending id order (ord[k] = slot of the k-th smallest id, k <
  // ord_n; slots appended since - this step's offspring - follow in slot order).  With it the
  // cell sort is a STABLE radix sort of that sequence by cell alone (2 passes instead of 4-5:
  // the id never enters the keys) while the slots themselves stay in cell order.  Kept up to
  // date by the permute kernel and by a compaction of its own behind the mortality
  // compaction; lost by anything else that moves slots (uploads, tile imports) and rebuilt
  // by one sort of (id, slot).
  bool ord_mode = true;          // GNX_ORD_SORT=0: always sort by (cell, id)
  bool ord_valid = false;
  int64_t ord_n = 0;
  int ord_cur = 0;
  int32_t* ord[2]{};
  int32_t* newslot = nullptr;    // [cap] where the last compaction put each slot (-1: dead)
  bool compact_fill = true;      // GNX_COMPACT_FILL=0 (read at gnx_create): always the stable copy
  bool jobs_self_scan = false;   // the job builder adds up the block counts itself (scan on stream3)
  bool ord_covers_xo = false;    // ev_ord was recorded behind a wait for the crossover in flight
  bool permute_split = true;        // GNX_PERMUTE_SPLIT=0 (read at gnx_create): one k_permute for every column
  bool perm_rest_inflight = false;  // k_permute_rest (stream3) has not been waited for
  bool perm_rest_pending = false;   // ... has not been launched yet
  bool perm_rest_late_ok = false;   // set by gnx_step around its cell sort: nothing reads the columns before the death probabilities
  bool perm_rest_late = false;      // this sort's columns follow beside the births (GNX_PERMUTE_REST_AT=2)
  GnxSoA perm_rest_a{}, perm_rest_b{};
  int64_t perm_rest_N = 0;
  hipEvent_t ev_perm_rest = nullptr;
  bool fb_pending = false;       // the adults' density bins are still to be counted (stream3)
  const float *fbp_x = nullptr, *fbp_y = nullptr;
  int64_t fbp_N = 0;
  hipEvent_t ev_alive = nullptr; // the death draws and their block counts are written
  int32_t* fill_cnt = nullptr;   // in-place compaction: the number of movers (device)
  hipEvent_t ev_fill = nullptr;  // its hole / mover lists are written (stream3)
  int64_t fill_guess = 0;        // slots the last mortality round emptied (sizes k_fill's grid)
  uint32_t* cell32 = nullptr;    // [cap] hash cell of each slot (k_move / k_keys)
  uint32_t* keyk[2]{};           // cells in id order / sorted
  int32_t* valk[2]{};            // id ranks in id order / sorted
  void* os_scratch = nullptr;    // gnx_os_sort32: histograms, look-back states, block counters
  uint32_t* os_ktmp = nullptr;   // ... and the pairs between two digit places
  int32_t* os_vtmp = nullptr;
  uint32_t* ord_state = nullptr; // k_ord_compact: [blk_stride] look-back words + [8] tickets, zero between launches
  int32_t* ord_cnt = nullptr;    // block counts / offsets of the index's own compaction
  int32_t* ord_off = nullptr;
  hipEvent_t ev_ord = nullptr;
  bool ord_inflight = false;     // the index's compaction runs on stream3
  bool keys_ordmode = false;     // k_move wrote cell32, not key64
  bool keys_fresh = false;
  // k_move also counted the digits of the cells it wrote into the head of os_scratch (the global
  // counts of the cell sort's passes, gnx_prim.hip): the sort needs no kernel in front of its
  // passes.  Whoever uses os_scratch otherwise, or drops the keys, clears the counts first
  // (gnx_os_hist_discard).
  bool hist_fresh = false;
  // gnx_walk: the next step's movement runs with this step's mortality (gnx_l_move_ahead)
  // gnx_walk, between two of its steps: the mortality leaves the dead where they are - no
  // compaction at all (k_fill_lists, k_fill: ~40 us alone, 130 beside the crossover, and the next
  // movement behind them).  The next step's movement skips the dead (their flags: h->flag), the
  // cell sort reads the living through the id-ordered index (which drops the dead) and k_permute
  // gathers them: the population is compact again behind the sort.  holes_N = the slots the
  // movement has to look at (gnx_l_mortality_enqueue: lazy).
  bool holes = false;
  int64_t holes_N = 0;
  // ... on tiles (gnx_tile_walk: a run of tile steps with nothing in between): the same, without
  // an index - the next step's routing skips the dead too, the imports are appended behind the
  // uncompacted stretch (holes_N grows with them; the first holes_flagged slots have flags), and
  // the cell sort gives the dead a key behind the emigrants': they leave with it.
  // the routing's counting pass inside the movement kernel (gnx_tile2_route_begin): a device copy
  // of the tile geometry, and whether the last movement counted
  RouteGeo* route_geo_dev = nullptr;
  uint64_t route_geo_epoch = 0;
  bool move_counts_routes = false;   // asked for by the caller of gnx_l_move
  bool move_counted_routes = false;  // ... and done
  bool tile_lazy_ok = false;     // set by gnx_tile_walk for every step but its last
  int64_t holes_flagged = 0;
  bool eager_move = false;       // set by gnx_walk for every step but the last
  bool moved_ahead = false;      // the coming step's age + movement are done, cell32 written
  hipEvent_t ev_move = nullptr;
  hipStream_t stream4 = nullptr; // ... on a stream of its own
  bool move_writes_keys = false;     // set around the movement of gnx_step           // k_move has written this step's sort keys
  int n_bin_blocks = 0;
  double* nodes = nullptr;           // [Jy][Jx] scratch node values
  double* K_over = nullptr;          // explicit K raster [H][W] (null: rast[K_layer] * K_factor)
  unsigned long long* nmax_bits = nullptr;
  double* p_death = nullptr;         // [cap]
  double* d_cell = nullptr;          // [cap]
  uint8_t* dead_in = nullptr;
  int32_t* counts_rast[2]{};         // per-cell individual counts (burn-in test)
  int counts_cur = 0;
  bool counts_init = false;
  double* red = nullptr;             // small reduction scratch [8]

  // block counts / offsets of the compactions (3 arrays of blk_stride entries each) and
  // their totals: cnt_dev[0] survivors, [1] rows freed, [2] surviving offspring that still
  // wait for their genome row and crossover
  int32_t* blk_cnt = nullptr;
  int32_t* blk_off = nullptr;
  int blk_stride = 0;
  int32_t* cnt_dev = nullptr;        // [4]
  // ticket counters of the kernels whose last workgroup scans the block counts
  // (gnx_compact.h: gnx_count_and_scan): [0] mortality, [1] pair list, [2] id-ordered index
  // (stream3), [3] block collector
  unsigned int* tickets = nullptr;   // [8], zero between launches

  // pinned host scratch for read-backs
  int64_t* h_pin = nullptr;          // [16]
  int64_t* h_pin_dev = nullptr;      // the same memory as the device sees it
  int64_t pin_seq = 0;               // sequence numbers of the polled read-backs
  void* h_stage = nullptr;           // pinned host staging buffer for per-step transfers
  size_t h_stage_bytes = 0;

  void* tile_comm = nullptr;           // gnx_comm.hip: the tile communicator (RCCL / local)
  // the device-driven step (gnx_dd.hip)
  GnxDD* dd = nullptr;                 // device
  GnxDDRec* dd_ring = nullptr;         // pinned host ring the steps publish into
  GnxDDRec* dd_ring_dev = nullptr;     // the same memory as the device sees it
  bool dd_active = false;              // between gnx_dd_enter and gnx_dd_leave
  int64_t dd_seq = 0;                  // steps enqueued since gnx_dd_enter
  int64_t dd_seen = 0;                 // ... whose record the host has taken into its totals
  int64_t dd_half_est = 0;             // free blocks the host can count on (lagged bound)
  int64_t dd_b_hi = 0;                 // largest births per step seen (the collector's bound)
  int64_t dd_use_hi = 0;               // most free blocks a step has taken
  int64_t dd_gc_seq = 0;               // steps enqueued before the last collection
  int64_t dd_top_last = 0, dd_top_seq = 0;   // the last record's stack height and step
  int64_t dd_est_seq = 0;              // the step whose record dd_half_est comes from
  int64_t dd_post_gc = 0;              // free blocks the last collection left (0: none yet)
  bool dd_gc_wait = false;             // the next step's record is waited for
  void* dd_graph[1]{};                 // DDExtra (gnx_dd.hip): captured graphs, events
  int32_t dd_err = 0;                  // sticky GNX_DD_ERR_* the steps have reported
  std::vector<int64_t> dd_hist;        // (N at start, births, deaths) of every step of the last gnx_walk
  bool dd_hist_closed = false;         // a step of this walk reported an error: no later step is recorded
  // bumped by whatever changes a by-value argument or a pointer of the step's kernels
  // (species parameters, traits, paths, K raster ...): captured graphs are dropped
  uint64_t cfg_epoch = 1;

  // profiling
  bool profiling = false;
  int profile_only = -1;
  GnxKernelTimer timers[GNX_K_COUNT];
  hipEvent_t ev_open = nullptr;
  std::vector<hipEvent_t> ev_free;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending[GNX_K_COUNT];
};

GnxTraitTab gnx_trait_tab(const gnx_state* h);

static inline GnxHalves gnx_halves(const gnx_state* h) {
  return GnxHalves{h->hmap, h->half_free, h->half_top, h->NB, h->BW > 0 ? h->BW : h->W64 / h->NB};
}
// Before a kernel that pops up to `blocks` physical blocks: the free stack holds that many
// (a mark-and-sweep collection runs first if the host cannot be sure, gnx_gc).
int gnx_half_reserve(gnx_state* h, int64_t blocks);
int gnx_gc(gnx_state* h);
// breakpoint offsets when blocks without a switch point may be shared with the parent
// (sparse paths only: the dense path table is not scanned for all-zero masks), else null
static inline const int32_t* gnx_alias_bp(const gnx_state* h) {
  return (h->alias_xo && h->sparse_paths) ? h->bp_off : nullptr;
}
static inline const int32_t* gnx_alias_loci(const gnx_state* h) {
  return (h->alias_xo && h->sparse_paths) ? h->bp_loci : nullptr;
}

// read-back of up to four device int32 counters: one tiny kernel writes them straight into
// pinned host memory h->h_pin[slot..] (an async D2H copy of 4 bytes is a blit kernel or an
// SDMA job of its own and queues behind whatever else runs, 50-900 us under load)
int gnx_publish(gnx_state* h, int slot, const int32_t* a, const int32_t* b = nullptr,
                const int32_t* c = nullptr, const int32_t* d = nullptr);

// RAII-less scoped timer helpers (events on h->stream)
// kernel: the family about to be timed, when the caller knows it - with only one family
// profiled (bench.py's timed region) the others record no event at all: an event record
// is a packet of its own on the stream, a few microseconds each
void gnx_time_begin(gnx_state* h, int kernel = -1);
void gnx_time_end(gnx_state* h, int kernel, double bytes);

// host <-> device copies through the pinned staging buffer (pageable hipMemcpy
// pins the user buffer on every call, which costs milliseconds per step)
int gnx_h2d(gnx_state* h, void* dst, const void* src, size_t bytes);
int gnx_d2h(gnx_state* h, void* dst, const void* src, size_t bytes);

// ---- launchers (gnx_kernels_*.hip) ---------------------------------------
int gnx_l_init_population(gnx_state* h, int64_t N);
int gnx_l_gather_e(gnx_state* h, int64_t first, int64_t n);
int gnx_l_age(gnx_state* h);
int gnx_l_move(gnx_state* h, bool inc_age, const float* inj_theta, const float* inj_dist,
               float* out_theta, float* out_dist, bool apply);
int gnx_l_move_ahead(gnx_state* h, int64_t N_all, const int32_t* d_alive, hipStream_t st);
int gnx_l_sort_by_cell(gnx_state* h, bool split_rest = false);
int gnx_wait_permute_rest(gnx_state* h, bool late_ok = false);
int gnx_permute_rest_launch(gnx_state* h);
// with_density: the n_pairs density (ops/demography.py:60-91) is launched before the host
// has read the pair count back, so the GPU works through the round trip
int gnx_l_find_pairs(gnx_state* h, const uint8_t* d_keep, int64_t* n_pairs_out,
                     bool with_density = false);
// ... in two halves: everything enqueued / the pair count read
int gnx_l_find_pairs_enqueue(gnx_state* h, const uint8_t* d_keep, bool with_density);
int gnx_l_find_pairs_finish(gnx_state* h, int64_t* n_pairs_out);
int gnx_l_births(gnx_state* h, int64_t* births_out);
int gnx_l_offspring_ahead(gnx_state* h, bool burn, bool inside_enqueue = false);
int gnx_l_pair_cls(gnx_state* h, int64_t P, bool local);
int gnx_vt_buffers(gnx_state* h);      // (allocated by gnx_set_id_order: never inside a stream capture)
int gnx_l_mate(gnx_state* h, bool burn, bool inject, int64_t B_inject, int64_t* births_out,
               int64_t id_base = -1, bool tiled = false);
int gnx_l_dispersal_inject(gnx_state* h, int64_t B, int A, const float* d_mx, const float* d_my,
                           const float* d_theta, const float* d_dist, float* d_ox, float* d_oy,
                           int32_t* d_used);
// crossover of every birth of the current step at once (rows for all of them)
int gnx_l_crossover_all(gnx_state* h, int64_t first_slot, int64_t B);
// tiled runs: row + local gamete, at once, of the n_req offspring whose mate is a ghost
int gnx_l_crossover_requests(gnx_state* h, int64_t first_slot, int64_t n_req);
// rows + crossover, now, of the offspring [first_slot, first_slot + B) that have no row yet
int gnx_l_crossover_pending(gnx_state* h, int64_t first_slot, int64_t B);
// crossover of the surviving offspring only (after k_alive + scan; gnx_l_mortality)
int gnx_l_crossover_survivors(gnx_state* h, int64_t first_slot, int64_t B, const int32_t* d_alive,
                              const int32_t* d_scan);
// every reader / writer of genome rows on `stream` goes through this first: a pending
// deferred crossover is carried out (for all pending offspring) and `stream` waits for
// the crossover in flight on stream2
int gnx_xo_join(gnx_state* h);
// only the first half: offspring still waiting for their crossover get it now (their slots
// are about to move); a crossover already in flight on stream2 is left alone
int gnx_xo_flush_deferred(gnx_state* h);
// launch the crossover whose jobs are ready (policy 1 / 2 hooks; no-op otherwise)
int gnx_xo_launch_pending(gnx_state* h);
// `stream` waits for the crossover in flight (not for one that is not launched yet)
int gnx_xo_wait_inflight(gnx_state* h);
int gnx_xo_prepare_jobs(gnx_state* h, int32_t** zero);
int gnx_xo_wait_wide(gnx_state* h);
double gnx_xo_bytes_per_birth(const gnx_state* h);
// selected-locus tables: rebuild sel_loci / path_sel / GnxSoA.tb after a change of the
// traits, the deleterious loci or the recombination paths
int gnx_l_rebuild_sel(gnx_state* h);
int gnx_l_path_sel(gnx_state* h);
// tb of slots [first, first+n) (or of first + list[q]) re-read from their genome rows
int gnx_l_tb_from_rows(gnx_state* h, int64_t first, int64_t n, const int32_t* d_list,
                       const int64_t* d_slots, bool join = true);
// tb of this step's offspring from their parents' tb and the paths' path_sel
int gnx_l_newborn_tb(gnx_state* h, int64_t first_slot, int64_t B);
// phenotypes of slots [first, first+n) from tb
int gnx_l_phenotype(gnx_state* h, int64_t first_slot, int64_t n, const int32_t* d_list = nullptr);
int gnx_l_assign_genomes(gnx_state* h, const int32_t* d_n_per_site);
int gnx_l_mutate(gnx_state* h, int n, const int64_t* d_slot, const int32_t* d_locus,
                 const uint8_t* d_hom);
// n_dev != null: n is an upper bound (grid size), the count itself is read on the device
int gnx_l_density(gnx_state* h, int64_t n, const float* d_x, const float* d_y, GnxSpline* spl,
                  const double* d_nodes_override, const int32_t* n_dev = nullptr);
// four words a counting kernel leaves behind its bins (tiles: the counters of the all-reduce)
struct GnxSetWords {
  int32_t* dst = nullptr;
  int32_t v[4] = {0, 0, 0, 0};
};
// two groups of device words for the host (pinned memory; a later wait covers them)
struct GnxPubWords {
  int n1 = 0;
  const int32_t* src1 = nullptr;
  int32_t* host1 = nullptr;
  int n2 = 0;
  const int32_t* src2 = nullptr;
  int32_t* host2 = nullptr;
};
int gnx_l_bins(gnx_state* h, int64_t n, const float* d_x, const float* d_y, const uint8_t* d_ghost,
               int32_t* d_bins, const int32_t* n_dev = nullptr, const GnxSetWords* sw = nullptr);
int gnx_l_spline(gnx_state* h, const int32_t* d_bins, GnxSpline* spl,
                 const double* d_nodes_override);
int gnx_l_spline_z(gnx_state* h, const int32_t* d_bins, GnxSpline* spl,
                   const double* d_nodes_override, bool housekeeping);
int gnx_l_raster(gnx_state* h, int which, double* d_out);
// the individuals' density of the step on one GPU: lattice + N.max() from the bins the
// step's kernels have counted, or (nobody counted: tiles, operator calls) bins + lattice
int gnx_l_density_N(gnx_state* h);
// the pairs' lattice from the bins k_pair_compact counted, on stream3
int gnx_l_lattice_P_async(gnx_state* h, int64_t n_max);
// the adults' bins (x, y: the sorted population), counted on stream3
int gnx_bins_adults_launch(gnx_state* h);
void gnx_bins_adults_drop(gnx_state* h);
int gnx_l_bins_adults_async(gnx_state* h, const float* d_x, const float* d_y, int64_t N);
int gnx_wait_latP(gnx_state* h);
bool gnx_fused_bins(const gnx_state* h);
int gnx_l_death_probs(gnx_state* h, bool with_selection);
int gnx_l_mortality(gnx_state* h, const uint8_t* d_dead_inject, int64_t* deaths_out);
// ... in two halves: death draws, compaction, crossover jobs and launch enqueued / the
// survivor counts read and the host's bookkeeping brought up to date
int gnx_l_mortality_enqueue(gnx_state* h, const uint8_t* d_dead_inject);
int gnx_l_mortality_finish(gnx_state* h, int64_t* deaths_out);
int gnx_l_spatial_diff(gnx_state* h, double* mean, double* sd, double* sums = nullptr);
int gnx_l_gather_genomes(gnx_state* h, int64_t n, const int64_t* d_slots, uint64_t* d_out);
// genomes d_in [n][2][W64] -> the rows of slots [first_slot, first_slot + n)
int gnx_l_scatter_genomes(gnx_state* h, int64_t n, const uint64_t* d_in, int64_t first_slot);

// ---- device-driven step (gnx_dd.hip): launchers with capacity-sized grids, counts in h->dd
// d_bins: density bins counted in the same launch (adults / pair midpoints), or null
int gnx_dd_l_sort(gnx_state* h, int32_t* d_bins, hipStream_t st);
int gnx_dd_l_pairs(gnx_state* h, int32_t* d_bins, hipStream_t st);
int gnx_dd_l_offspring(gnx_state* h, bool genomes, int32_t* d_bins, hipStream_t st);
int gnx_dd_l_bins_adults(gnx_state* h, int par, hipStream_t st);
int gnx_dd_l_density_pairs(gnx_state* h, hipStream_t st);
int gnx_dd_l_density_N(gnx_state* h, int par, hipStream_t st);
int gnx_dd_l_death_probs(gnx_state* h, bool with_selection, int par, hipStream_t st);
int gnx_dd_l_alive(gnx_state* h, bool xo, int buf, hipStream_t st);
int gnx_dd_l_jobs(gnx_state* h, int buf, hipStream_t st);
int gnx_dd_l_crossover(gnx_state* h, int buf, hipStream_t st);
int gnx_dd_l_fill_lists(gnx_state* h, int has_rows, hipStream_t st);
int gnx_dd_l_fill(gnx_state* h, int has_rows, bool xo, hipStream_t st);
int gnx_dd_l_ord_end(gnx_state* h, int has_rows, bool xo, hipStream_t st);
// the handle can take device-driven steps (else gnx_walk falls back to gnx_step)
bool gnx_dd_eligible(const gnx_state* h, bool burn);
int gnx_dd_leave(gnx_state* h);
void gnx_dd_destroy(gnx_state* h);

// look-back-free compaction (gnx_compact.h): block counts cnt[k * blk_stride + b], k < K
// <= 3, -> exclusive block offsets off[...], totals to out[0..2] on the device and to
// pinned host memory
int gnx_block_scan(gnx_state* h, int K, int64_t n_items, const int32_t* cnt, int32_t* off,
                   int32_t* out, int64_t* host, int64_t seq = 0, hipStream_t st = nullptr,
                   const int32_t* extra = nullptr);
// the digit counts k_move left in os_scratch are not going to be used: the scratch is zero again
int gnx_os_hist_discard(gnx_state* h, hipStream_t st = nullptr);
// the step's cell sort over the id-ordered index: Onesweep with one fill (gnx_prim.hip)
bool gnx_l_lattices_tiled(gnx_state* h, bool have_pairs, const GnxPubWords& pub);
void gnx_host_mark(int id);                       // GNX_HOST_TIMES=2 (gnx_api.hip)
extern double g_host_step_s, g_host_wait_s;      // GNX_HOST_TIMES=1 (gnx_api.hip)
bool gnx_host_times();
size_t gnx_os_scratch_bytes(size_t n, int end_bit);
size_t gnx_os_words_used64(size_t n, int end_bit);
int gnx_os_sort64_clean(void* scratch, void* tmp, const uint64_t* kin, uint64_t* kout,
                        const int32_t* vin, int32_t* vout, size_t n, int end_bit, hipStream_t s);
size_t gnx_os_words_used(size_t n, int end_bit, int geometry = 0);
int gnx_os_keys_hist(void* scratch, unsigned int* ticket, int64_t N, int64_t ord_n,
                     const int32_t* ord, const uint32_t* cell32, uint32_t* key, int32_t* val,
                     int end_bit, hipStream_t s, const GnxDD* dd = nullptr, int geometry = 0);
void gnx_os_digits(int end_bit, int* places, int* rb);
int gnx_os_sort32_gather(void* scratch, uint32_t* ktmp, int32_t* vtmp, uint32_t* kout, int32_t* vout,
                         size_t n, int end_bit, const int32_t* ord, int64_t ord_n,
                         const uint32_t* cell32, hipStream_t s);
int gnx_os_sort32_ranked(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin,
                         uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, int geometry = 0);
int gnx_os_sort32(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin,
                  uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                  hipStream_t s, int variant);
int gnx_prim_sort32_bits(void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout,
                         const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, bool alone);
// the host spins on pinned word h_pin[slot + 3] until the scan kernel given `seq` has
// published its totals there (falls back to a stream sync after a while): no wait for the
// kernels queued behind the scan, no driver wake-up latency
int gnx_wait_published(gnx_state* h, int slot, int64_t seq);

// rocPRIM wrappers (gnx_prim.hip)
int gnx_prim_sort_bytes(size_t n, int bits, size_t* bytes);
int gnx_prim_sort(void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout, const int32_t* vin,
                  int32_t* vout, size_t n, int bits, hipStream_t s);
int gnx_prim_sort64_bytes(size_t n, size_t* bytes);
int gnx_prim_sort64(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                    const int32_t* vin, int32_t* vout, size_t n, hipStream_t s);
int gnx_prim_sort64_bits(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                         const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, bool alone = true);
int gnx_prim_scan_bytes(size_t n, size_t* bytes);
int gnx_prim_scan(void* tmp, size_t bytes, const int32_t* in, int32_t* out, size_t n,
                  hipStream_t s);

// the slots in use: the population, or - uncompacted (holes) - the stretch it is spread over
static inline int64_t gnx_extent(const gnx_state* h) { return h->holes ? h->holes_N : h->N; }

static inline int gnx_grid(int64_t n, int block, int max_blocks = 1 << 20) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}
