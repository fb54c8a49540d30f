// Spatial tiling of one species over several GPUs.
//
// The landscape is cut into a uniform R x C grid of tiles; one process/GPU owns
// one tile and every individual whose (x, y) lies in it.  Every rank keeps the
// whole (small) rasters and works in global coordinates.  Per time step the
// host layer (geonomics_amd/parallel.py, torch.distributed over RCCL) moves:
//   1. migrants   - individuals that left their tile, with their genomes;
//   2. halo       - light copies (x, y, age, sex, id) of the individuals in the hash
//                   cells within 2 cells of a neighbour tile ("ghosts"): candidates
//                   for the mate search AND focal individuals whose own choice
//                   is recomputed locally, so that the reciprocal-pair rule
//                   gives the same answer on both sides of a border;
//   3. pair lists - focal ids + birth counts, all-gathered, so that offspring
//                   ids are the same as on one GPU (pairs ordered by focal id);
//   4. gametes    - the gamete of a ghost mate is cut from its genome on the
//                   tile that owns it and shipped back (L/8 bytes);
//   5. density    - the integer half-window bin counts, all-reduced (sum).
// All random draws are keyed by individual id, all choices are order-
// independent, so a tiled run reproduces the single-GPU run bit for bit.
#include <algorithm>
#include <atomic>
#include "gnx_internal.h"
#include "gnx_rng.h"

#include "gnx_xo.h"

struct TileBox {
  float x0, y0, x1, y1;   // own tile [x0,x1) x [y0,y1)
  int r, c, R, C;
};

static TileBox tile_box(const gnx_state* h) {
  TileBox t;
  float tw = (float)h->cfg.W / h->tile_C, th = (float)h->cfg.H / h->tile_R;
  t.x0 = h->tile_c * tw;
  t.x1 = (h->tile_c + 1) * tw;
  t.y0 = h->tile_r * th;
  t.y1 = (h->tile_r + 1) * th;
  t.r = h->tile_r;
  t.c = h->tile_c;
  t.R = h->tile_R;
  t.C = h->tile_C;
  return t;
}

extern "C" int gnx_tile_set(gnx_state* h, int32_t R, int32_t C, int32_t r, int32_t c) {
  if (R < 1 || C < 1 || r < 0 || r >= R || c < 0 || c >= C || h->cfg.W % C || h->cfg.H % R) {
    gnx_set_error("gnx_tile_set: bad tile grid %dx%d (%d,%d) for a %dx%d landscape", R, C, r, c,
                  h->cfg.W, h->cfg.H);
    return 1;
  }
  h->route_geo_epoch = 0;       // (the device copy of the routing geometry is stale)
  h->tile_R = R;
  h->tile_C = C;
  h->tile_r = r;
  h->tile_c = c;
  h->tiled = R * C > 1;
  return 0;
}

// ---------------------------------------------------------------- pack / unpack
__global__ void k_mark_out(int64_t N, const float* x, const float* y, TileBox t, int32_t* flag,
                           uint8_t* dead) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  bool out = x[i] < t.x0 || x[i] >= t.x1 || y[i] < t.y0 || y[i] >= t.y1;
  flag[i] = out ? 1 : 0;
  dead[i] = out ? 1 : 0;
}

// Halo: whole hash cells.  A neighbour tile needs every individual whose cell lies
// within 2 cells of the tile's own cell range: ring 1 completes the 3x3 candidate
// blocks of the neighbour's own focal individuals, ring 2 those of the ghosts they can
// choose (a ghost's own choice decides the reciprocal-pair rule), so that the index
// sampling of the mate search sees the same candidate lists on every tile.
// neighbour bit k = (dy+1)*3 + (dx+1), dy,dx in {-1,0,1} (bit 4 unused)
struct HaloSpans {
  int cx0[3], cx1[3], cy0[3], cy1[3];   // cell spans (+/- 2 rings) of tile columns c-1..c+1, rows r-1..r+1
  int okx[3], oky[3];                   // that column / row exists
};

__global__ void k_mark_halo(int64_t N, const float* x, const float* y, const uint8_t* ghost,
                            HaloSpans sp, double inv_cs, int ncx, int ncy, int32_t* flag,
                            int32_t* mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int m = 0;
  if (!ghost[i]) {
    const int cx = min(ncx - 1, (int)((double)x[i] * inv_cs));
    const int cy = min(ncy - 1, (int)((double)y[i] * inv_cs));
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx) {
        if (dx == 1 && dy == 1) continue;
        if (sp.okx[dx] && sp.oky[dy] && cx >= sp.cx0[dx] && cx <= sp.cx1[dx] &&
            cy >= sp.cy0[dy] && cy <= sp.cy1[dy])
          m |= 1 << (dy * 3 + dx);
      }
  }
  mask[i] = m;
  flag[i] = m ? 1 : 0;
}

// Panmixia (mating_radius None, structs/species.py:2178-2194: both members of a pair drawn
// uniformly from the WHOLE population): every other tile needs every individual - the halo is
// everybody, to every neighbour bit.
__global__ void k_mark_everybody(int64_t N, const uint8_t* ghost, int32_t* flag, int32_t* mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int m = ghost[i] ? 0 : 0x1EF;
  mask[i] = m;
  flag[i] = m ? 1 : 0;
}

static HaloSpans halo_spans(const gnx_state* h) {
  HaloSpans sp;
  const int tw = h->cfg.W / h->tile_C, th = h->cfg.H / h->tile_R;
  for (int d = 0; d < 3; ++d) {
    const int cc = h->tile_c + d - 1, rr = h->tile_r + d - 1;
    sp.okx[d] = cc >= 0 && cc < h->tile_C;
    sp.oky[d] = rr >= 0 && rr < h->tile_R;
    const float x0 = (float)(cc * tw), x1 = nextafterf((float)((cc + 1) * tw), 0.f);
    const float y0 = (float)(rr * th), y1 = nextafterf((float)((rr + 1) * th), 0.f);
    const int ring = 2 * h->cell_ref;       // two mating radii, in cells
    sp.cx0[d] = std::min(h->ncx - 1, (int)((double)x0 * h->inv_cs)) - ring;
    sp.cx1[d] = std::min(h->ncx - 1, (int)((double)x1 * h->inv_cs)) + ring;
    sp.cy0[d] = std::min(h->ncy - 1, (int)((double)y0 * h->inv_cs)) - ring;
    sp.cy1[d] = std::min(h->ncy - 1, (int)((double)y1 * h->inv_cs)) + ring;
  }
  return sp;
}

__global__ void k_pack(int64_t N, int64_t cap, const int32_t* flag, const int32_t* scan,
                       const int32_t* mask, GnxSoA s, int n_traits, gnx_ind_rec* rec, float* zrec,
                       int64_t* slots) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || !flag[i]) return;
  int64_t q = scan[i];
  gnx_ind_rec r;
  r.x = s.x[i];
  r.y = s.y[i];
  r.age = s.age[i];
  r.sex = s.sex[i];
  r.id = s.id[i];
  r.fit = s.fit[i];
  r.nbr_mask = mask ? mask[i] : 0;
  rec[q] = r;
  if (zrec)
    for (int t = 0; t < n_traits; ++t) zrec[q * n_traits + t] = s.z[(int64_t)t * cap + i];
  if (slots) slots[q] = i;
}

__global__ void k_unpack(int64_t N, int64_t n, int64_t cap, GnxSoA s, const gnx_ind_rec* rec,
                         const float* zrec, int n_traits, int n_layers, const float* rast, int W,
                         int H, const int32_t* free_rows, int64_t n_free, int has_rows, int ghost,
                         GnxHalves Hv) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int64_t slot = N + k;
  gnx_ind_rec r = rec[k];
  s.x[slot] = r.x;
  s.y[slot] = r.y;
  s.age[slot] = r.age;
  s.sex[slot] = (uint8_t)r.sex;
  s.id[slot] = r.id;
  s.fit[slot] = r.fit;
  s.ghost[slot] = (uint8_t)ghost;
  s.grow[slot] = (has_rows && !ghost) ? free_rows[n_free - 1 - k] : -1;
  if (has_rows && !ghost)        // (uniform over the launch) blocks for the genome
    for (int q = 0; q < 2 * Hv.NB; ++q)
      gnx_half_new(Hv, (int64_t)s.grow[slot] * 2 * Hv.NB + q, true);
  for (int t = 0; t < n_traits; ++t)
    s.z[(int64_t)t * cap + slot] = zrec ? zrec[k * n_traits + t] : 0.f;
  int cx = (int)r.x, cy = (int)r.y;
  for (int l = 0; l < n_layers; ++l)
    s.e[(int64_t)l * cap + slot] = rast[((int64_t)l * H + cy) * W + cx];
}

template <typename T>
static int dalloc_t(T** p, size_t n) {
  *p = nullptr;
  HIPCHK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
  return 0;
}

static bool has_rows(const gnx_state* h) { return h->genomes_assigned && h->cfg.L > 0; }

static int stage_selection(gnx_state* h, const int32_t* d_mask, bool with_z, bool with_geno,
                           int64_t* n_out) {
  // flag[] is set; pack the flagged individuals into the staging buffers
  int64_t N = h->N;
  GnxSoA s = h->soa[h->cur];
  HIPCHK(hipMemsetAsync(h->flag + N, 0, sizeof(int32_t), h->stream));
  GNXCHK(gnx_prim_scan(h->scan_tmp, h->scan_tmp_bytes, h->flag, h->scan, (size_t)N + 1,
                       h->stream));
  GNXCHK(gnx_publish(h, 0, h->scan + N));
  HIPCHK(hipStreamSynchronize(h->stream));
  int64_t n = h->h_pin[0];
  *n_out = n;
  h->st_n = n;
  h->st_has_geno = false;
  if (n == 0) return 0;
  // grow-only staging buffers (no hipMalloc/hipFree on the per-step path)
  if (n > h->st_cap) {
    (void)hipFree(h->st_rec);
    (void)hipFree(h->st_z);
    (void)hipFree(h->st_slots);
    h->st_cap = n + n / 4 + 1024;
    GNXCHK(dalloc_t(&h->st_rec, (size_t)h->st_cap));
    GNXCHK(dalloc_t(&h->st_z, (size_t)h->st_cap * std::max(h->cfg.n_traits, 1)));
    GNXCHK(dalloc_t(&h->st_slots, (size_t)h->st_cap));
  }
  hipLaunchKernelGGL(k_pack, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, h->cfg.cap_inds,
                     h->flag, h->scan, d_mask, s, h->cfg.n_traits, h->st_rec,
                     (with_z && h->cfg.n_traits) ? h->st_z : nullptr, h->st_slots);
  if (with_geno && has_rows(h)) {
    if (n > h->st_geno_cap) {
      (void)hipFree(h->st_geno);
      h->st_geno_cap = n + n / 4 + 64;
      GNXCHK(dalloc_t(&h->st_geno, (size_t)h->st_geno_cap * 2 * h->W64));
    }
    GNXCHK(gnx_l_gather_genomes(h, n, h->st_slots, h->st_geno));
    h->st_has_geno = true;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// individuals that left the tile: pack them (records, phenotypes, genomes) and
// remove them from this tile (their genome rows are freed)
extern "C" int gnx_tile_export_migrants(gnx_state* h, int64_t* n_out) {
  *n_out = 0;
  if (h->n_ghost) {
    gnx_set_error("gnx_tile_export_migrants: ghosts are resident");
    return 1;
  }
  int64_t N = h->N;
  if (N == 0) return 0;
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL(k_mark_out, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s.x, s.y,
                     tile_box(h), h->flag, h->dead_in);
  GNXCHK(stage_selection(h, nullptr, true, true, n_out));
  if (*n_out > 0) {
    int64_t D = 0;
    GNXCHK(gnx_l_mortality(h, h->dead_in, &D));
    HIPCHK(hipStreamSynchronize(h->stream));     // tile calls return with the stream idle
  }
  return 0;
}

extern "C" int gnx_tile_export_halo(gnx_state* h, int64_t* n_out) {
  *n_out = 0;
  int64_t N = h->N;
  if (N == 0) return 0;
  GnxSoA s = h->soa[h->cur];
  if (h->have_sp && h->sp.mating_radius < 0)
    hipLaunchKernelGGL(k_mark_everybody, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s.ghost,
                       h->flag, h->mate);
  else
    hipLaunchKernelGGL(k_mark_halo, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s.x, s.y,
                       s.ghost, halo_spans(h), h->inv_cs, h->ncx, h->ncy, h->flag, h->mate);
  return stage_selection(h, h->mate, false, false, n_out);
}

// read the staged selection (after export_migrants / export_halo)
extern "C" int gnx_tile_get_staged(gnx_state* h, gnx_ind_rec* rec, float* z, uint64_t* geno) {
  int64_t n = h->st_n;
  if (n == 0) return 0;
  if (rec) GNXCHK(gnx_d2h(h, rec, h->st_rec, n * sizeof(gnx_ind_rec)));
  if (z && h->st_z && h->cfg.n_traits)
    GNXCHK(gnx_d2h(h, z, h->st_z, n * h->cfg.n_traits * sizeof(float)));
  if (geno && h->st_has_geno)
    GNXCHK(gnx_d2h(h, geno, h->st_geno, (size_t)n * 2 * h->W64 * 8));
  return 0;
}

static int import_common(gnx_state* h, int64_t n, const gnx_ind_rec* rec, const float* z,
                         const uint64_t* geno, int ghost) {
  if (n == 0) return 0;
  const gnx_config& c = h->cfg;
  const bool rows = has_rows(h) && !ghost;
  if (h->N + n > c.cap_inds || (rows && n > h->n_free)) {
    gnx_set_error("capacity exceeded importing %lld individuals (N=%lld cap=%lld free rows %lld)",
                  (long long)n, (long long)h->N, (long long)c.cap_inds, (long long)h->n_free);
    return 2;
  }
  for (int64_t k = 0; k < n; ++k)
    if (!(rec[k].x >= 0 && rec[k].x < c.W && rec[k].y >= 0 && rec[k].y < c.H)) {
      gnx_set_error("import: record %lld is off the landscape", (long long)k);
      return 1;
    }
  gnx_ind_rec* d_rec = nullptr;
  float* d_z = nullptr;
  u64* d_g = nullptr;
  GNXCHK(dalloc_t(&d_rec, (size_t)n));
  GNXCHK(gnx_h2d(h, d_rec, rec, n * sizeof(gnx_ind_rec)));
  if (z && c.n_traits) {
    GNXCHK(dalloc_t(&d_z, (size_t)n * c.n_traits));
    GNXCHK(gnx_h2d(h, d_z, z, n * c.n_traits * sizeof(float)));
  }
  GnxSoA s = h->soa[h->cur];
  if (rows && !ghost) GNXCHK(gnx_half_reserve(h, 2 * (int64_t)h->NB * n));
  hipLaunchKernelGGL(k_unpack, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, h->N, n,
                     c.cap_inds, s, d_rec, d_z, c.n_traits, c.n_layers, h->rast, c.W, c.H,
                     h->free_rows, h->n_free, rows ? 1 : 0, ghost,
                     gnx_halves(h));
  if (rows) {
    if (!geno) {
      gnx_set_error("import: genomes are assigned on this tile but none were sent");
      return 1;
    }
    GNXCHK(dalloc_t(&d_g, (size_t)n * 2 * h->W64));
    GNXCHK(gnx_h2d(h, d_g, geno, (size_t)n * 2 * h->W64 * 8));
    GNXCHK(gnx_l_scatter_genomes(h, n, (const uint64_t*)d_g, h->N));
    h->n_free -= n;
    GNXCHK(gnx_l_tb_from_rows(h, h->N, n, nullptr, nullptr));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  (void)hipFree(d_rec);
  (void)hipFree(d_z);
  (void)hipFree(d_g);
  h->N += n;
  h->ord_valid = false;          // arrivals carry any id: no id-ordered index
  if (ghost) h->n_ghost += n;
  for (int64_t k = 0; k < n; ++k) h->max_id = std::max<int64_t>(h->max_id, rec[k].id);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int gnx_tile_import(gnx_state* h, int64_t n, const gnx_ind_rec* rec, const float* z,
                               const uint64_t* geno) {
  return import_common(h, n, rec, z, geno, 0);
}

extern "C" int gnx_tile_import_ghosts(gnx_state* h, int64_t n, const gnx_ind_rec* rec) {
  return import_common(h, n, rec, nullptr, nullptr, 1);
}

// ---------------------------------------------------------------- step phases
extern "C" int gnx_tile_pairs(gnx_state* h, int32_t burn, int64_t* n_pairs, int64_t* n_births) {
  if (!h->have_sp) {
    gnx_set_error("species parameters not set");
    return 1;
  }
  // (panmixia: every tile holds everybody - its own individuals and all the others as ghosts,
  // gnx_tile_export_halo - in the canonical (hash cell, id) order, so the trials' draws name the
  // same individuals on every tile as on one device; a pair belongs to the tile that owns its
  // focal individual (k_pair_flags).  The Python-driven protocol only: gnx_tile2_pairs refuses.)
  int64_t P = 0, B = 0;
  GNXCHK(gnx_l_sort_by_cell(h));
  GNXCHK(gnx_l_find_pairs(h, nullptr, &P));
  GNXCHK(gnx_l_bins(h, P, h->mid_x, h->mid_y, nullptr, h->bins_P));
  GNXCHK(gnx_l_births(h, &B));
  *n_pairs = P;
  *n_births = B;
  return 0;
}

// order keys (ascending: hash cell << 40 | id of the focal individual) and birth counts of
// the local pair list
extern "C" int gnx_tile_pair_info(gnx_state* h, int64_t* focal_ids, int32_t* n_births) {
  int64_t P = h->n_pairs;
  if (P == 0) return 0;
  // gnx_l_find_pairs left the pairs' order keys in key64[0]
  GNXCHK(gnx_d2h(h, focal_ids, h->key64[0], P * sizeof(int64_t)));
  if (h->sp.n_births_fixed)
    for (int64_t p = 0; p < P; ++p) n_births[p] = (int32_t)h->sp.n_births_lambda;
  else
    GNXCHK(gnx_d2h(h, n_births, h->nbirths, P * sizeof(int32_t)));
  return 0;
}

extern "C" int gnx_get_bins(gnx_state* h, int32_t which, int32_t* out) {
  size_t nb = (size_t)h->lat.nbx * h->lat.nby;
  HIPCHK(hipStreamSynchronize(h->stream));
  // the individuals' counts of the last density: where the step's own kernels counted them
  // (one GPU, gnx_bins.h; intact until the next step's lattice), or the counting pass's bins
  const int32_t* src = which ? h->bins_P : (h->last_N_fused ? h->fb[h->fb_cur ^ 1] : h->bin_partials);
  GNXCHK(gnx_d2h(h, out, src, nb * sizeof(int32_t)));
  return 0;
}

extern "C" int gnx_set_bins(gnx_state* h, int32_t which, const int32_t* in) {
  size_t nb = (size_t)h->lat.nbx * h->lat.nby;
  GNXCHK(gnx_h2d(h, which ? h->bins_P : h->bin_partials, in, nb * sizeof(int32_t)));
  h->bins_zeroed[which ? 1 : 0] = false;
  if (!which) h->last_N_fused = false;
  return 0;
}

extern "C" int gnx_density_bin_count(gnx_state* h) { return h->lat.nbx * h->lat.nby; }

// While the crossover launched by gnx_tile_offspring(_dev) is in flight on the handle's
// stream, the entry points of the gamete service run on a second stream (they touch
// parent rows and the ghost-mate halves of child rows only, which the crossover leaves
// alone); gnx_tile_finish_births joins the two.
struct SideStream {
  gnx_state* h;
  hipStream_t saved;
  explicit SideStream(gnx_state* hh) : h(hh), saved(hh->stream) {
    if (h->xo_pending && h->stream2) h->stream = h->stream2;
  }
  ~SideStream() { h->stream = saved; }
};

extern "C" int gnx_tile_offspring(gnx_state* h, int32_t burn, int64_t id_base,
                                  const int64_t* pair_goff, int64_t* n_requests) {
  *n_requests = 0;
  int64_t P = h->n_pairs, B = 0;
  h->birth_first_slot = h->N;
  h->n_req = 0;
  h->tile_births_settled = false;
  // (no offsets: tile-major ids, the pieces were made by gnx_tile2_vt_counts / _vt_bases)
  if (P > 0 && pair_goff)
    GNXCHK(gnx_h2d(h, h->pair_goff, pair_goff, P * sizeof(int64_t)));
  GNXCHK(gnx_l_mate(h, burn != 0, false, 0, &B, id_base, true));
  h->last_births = B;
  if (B == 0) h->n_req = 0;
  *n_requests = h->n_req;
  h->xo_pending = B > 0 && !burn && has_rows(h);
  return 0;
}

// The gamete service reads parents' genome blocks.  Last step's deferred crossover, which
// writes the blocks of last step's surviving newborns (parents already), may still be running
// on stream2 (gnx_set_crossover_overlap(1), GNX_XO_SORT_WAIT=0, a split launch): the serving
// stream waits for it unless it IS stream2 (SideStream: in order behind the crossover anyway).
static int serve_wait_crossover(gnx_state* h, hipStream_t main_stream) {
  if (h->stream == h->stream2 && h->stream2 != nullptr) {
    // a crossover not launched yet would be launched on stream2 behind `main_stream`'s work
    hipStream_t cur = h->stream;
    h->stream = main_stream;
    int rc = gnx_xo_launch_pending(h);
    h->stream = cur;
    return rc;
  }
  GNXCHK(gnx_xo_launch_pending(h));
  return gnx_xo_wait_inflight(h);
}

extern "C" int gnx_tile_get_requests(gnx_state* h, int64_t* pid, int32_t* child_k, int32_t* key,
                                     uint8_t* start, float* px, float* py) {
  SideStream side(h);
  int64_t n = h->n_req;
  if (n == 0) return 0;
  GNXCHK(gnx_d2h(h, pid, h->req_pid, n * 8));
  GNXCHK(gnx_d2h(h, child_k, h->req_k, n * 4));
  GNXCHK(gnx_d2h(h, key, h->req_key, n * 4));
  GNXCHK(gnx_d2h(h, start, h->req_start, n));
  GNXCHK(gnx_d2h(h, px, h->req_px, n * 4));
  GNXCHK(gnx_d2h(h, py, h->req_py, n * 4));
  return 0;
}

// ---------------------------------------------------------------- gamete service
__global__ void k_id_keys(int64_t N, const int64_t* id, uint64_t* key, int32_t* slot) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  key[i] = (uint64_t)id[i];
  slot[i] = (int32_t)i;
}

__global__ void k_lookup(int64_t n, const int64_t* want, int64_t N, const uint64_t* sorted_ids,
                         const int32_t* sorted_slots, int32_t* out_slot, int32_t* n_missing) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  uint64_t w = (uint64_t)want[q];
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (sorted_ids[mid] < w) lo = mid + 1; else hi = mid;
  }
  if (lo < N && sorted_ids[lo] == w) {
    out_slot[q] = sorted_slots[lo];
  } else {
    out_slot[q] = -1;
    atomicAdd(n_missing, 1);
  }
}

// gamete of parent slot[q] along path key[q] from start homologue start[q]
// (ops/mating.py:165-168), one wave per gamete, written to out[q][W16]
__global__ void __launch_bounds__(256)
k_make_gametes(int64_t n, int W16, const u64x2* __restrict__ G, const int32_t* __restrict__ grow,
               GnxHalves H, const int32_t* __restrict__ slot, const int32_t* __restrict__ keys,
               const uint8_t* __restrict__ starts, const u64x2* __restrict__ paths,
               u64x2* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= n) return;
  const int prow = grow[slot[q]];
  const u64 s = starts[q] ? ~0ull : 0ull;
  const int64_t lh0 = (int64_t)prow * 2;
  const u64x2* pm = paths + (int64_t)keys[q] * W16;
  for (int c = lane; c < W16; c += 64) {
    u64x2 m = pm[c];
    m.a ^= s;
    m.b ^= s;
    const u64x2 a = G[gnx_chunk_at(H, lh0, c)], b = G[gnx_chunk_at(H, lh0 + 1, c)];
    u64x2 o;
    o.a = (a.a & ~m.a) | (b.a & m.a);
    o.b = (a.b & ~m.b) | (b.b & m.b);
    out[q * W16 + c] = o;
  }
}

extern "C" int gnx_tile_serve_gametes(gnx_state* h, int64_t n, const int64_t* parent_ids,
                                      const int32_t* keys, const uint8_t* starts,
                                      uint64_t* out) {
  SideStream side(h);
  if (n == 0) return 0;
  if (!has_rows(h) || h->n_paths == 0) {
    gnx_set_error("gnx_tile_serve_gametes: genomes / paths not set");
    return 1;
  }
  for (int64_t q = 0; q < n; ++q)
    if (keys[q] < 0 || keys[q] >= h->n_paths || starts[q] > 1) {
      gnx_set_error("gnx_tile_serve_gametes: request %lld out of range", (long long)q);
      return 1;
    }
  int64_t N = h->N;
  GnxSoA s = h->soa[h->cur];
  GNXCHK(serve_wait_crossover(h, side.saved));
  hipLaunchKernelGGL(k_id_keys, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s.id,
                     h->key64[0], h->perm[0]);
  GNXCHK(gnx_prim_sort64(h->sort64_tmp, h->sort64_tmp_bytes, h->key64[0], h->key64[1], h->perm[0],
                         h->perm[1], (size_t)N, h->stream));
  int64_t* d_ids = nullptr;
  int32_t *d_keys = nullptr, *d_slot = nullptr, *d_miss = nullptr;
  uint8_t* d_st = nullptr;
  u64* d_out = nullptr;
  GNXCHK(dalloc_t(&d_ids, (size_t)n));
  GNXCHK(dalloc_t(&d_keys, (size_t)n));
  GNXCHK(dalloc_t(&d_slot, (size_t)n));
  GNXCHK(dalloc_t(&d_miss, 1));
  GNXCHK(dalloc_t(&d_st, (size_t)n));
  GNXCHK(dalloc_t(&d_out, (size_t)n * h->W64));
  GNXCHK(gnx_h2d(h, d_ids, parent_ids, n * 8));
  GNXCHK(gnx_h2d(h, d_keys, keys, n * 4));
  GNXCHK(gnx_h2d(h, d_st, starts, n));
  HIPCHK(hipMemsetAsync(d_miss, 0, 4, h->stream));
  hipLaunchKernelGGL(k_lookup, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, d_ids, N,
                     h->key64[1], h->perm[1], d_slot, d_miss);
  int miss = 0;
  GNXCHK(gnx_d2h(h, &miss, d_miss, 4));
  int rc = 0;
  if (miss) {
    gnx_set_error("gnx_tile_serve_gametes: %d requested parents do not live on this tile", miss);
    rc = 1;
  } else {
    const int W16 = h->W64 / 2;
    hipLaunchKernelGGL(k_make_gametes, dim3(gnx_grid(n * 64, 256)), dim3(256), 0, h->stream, n, W16,
                       (const u64x2*)h->G, s.grow, gnx_halves(h), d_slot, d_keys, d_st, (const u64x2*)h->paths,
                       (u64x2*)d_out);
    rc = gnx_d2h(h, out, d_out, (size_t)n * h->W64 * 8);
  }
  for (void* p : {(void*)d_ids, (void*)d_keys, (void*)d_slot, (void*)d_miss, (void*)d_st,
                  (void*)d_out})
    (void)hipFree(p);
  return rc;
}

__global__ void k_put_gametes(int64_t n, int W16, const u64x2* in, u64x2* G, const int32_t* grow,
                              GnxHalves H, int64_t first_slot, const int32_t* child_k) {
  const int64_t total = n * (int64_t)W16;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    int64_t q = g / W16;
    int64_t c = g - q * W16;
    // the mate is pair[1] -> the child's homologue 1 (ops/mating.py:169)
    G[gnx_chunk_at(H, (int64_t)grow[first_slot + child_k[q]] * 2 + 1, (int)c)] = in[g];
  }
}

extern "C" int gnx_tile_put_gametes(gnx_state* h, int64_t n, const int32_t* child_k,
                                    const uint64_t* data) {
  SideStream side(h);
  if (n == 0) return 0;
  for (int64_t q = 0; q < n; ++q)
    if (child_k[q] < 0 || child_k[q] >= h->last_births) {
      gnx_set_error("gnx_tile_put_gametes: child index out of range");
      return 1;
    }
  int32_t* d_k = nullptr;
  u64* d_in = nullptr;
  GNXCHK(dalloc_t(&d_k, (size_t)n));
  GNXCHK(dalloc_t(&d_in, (size_t)n * h->W64));
  GNXCHK(gnx_h2d(h, d_k, child_k, n * 4));
  GNXCHK(gnx_h2d(h, d_in, data, (size_t)n * h->W64 * 8));
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_put_gametes, dim3(gnx_grid(n * W16, 256, 256 * 32)), dim3(256), 0, h->stream,
                     n, W16, (const u64x2*)d_in, (u64x2*)h->G, h->soa[h->cur].grow,
                     gnx_halves(h), h->birth_first_slot, d_k);
  HIPCHK(hipStreamSynchronize(h->stream));
  (void)hipFree(d_k);
  (void)hipFree(d_in);
  HIPCHK(hipGetLastError());
  return 0;
}

// phenotypes of this step's offspring (all gametes are in place) and the bins
// of the tile's own individuals (ghosts skipped) for the N density
static int tile_finish_births(gnx_state* h, int32_t burn, const GnxSetWords* sw);
extern "C" int gnx_tile_finish_births(gnx_state* h, int32_t burn) {
  return tile_finish_births(h, burn, nullptr);
}

// the offspring that took a remote gamete re-read their alleles at the selected loci (and their
// phenotype) from their finished rows
static int tile_settle_births(gnx_state* h, int32_t burn) {
  // the service entry points return with stream2 idle; what follows is ordered behind
  // the crossover on the main stream
  h->xo_pending = false;
  if (h->tile_births_settled) return 0;
  h->tile_births_settled = true;
  int64_t B = h->last_births;
  if (B > 0 && !burn && has_rows(h)) {
    // local gametes took their alleles at the selected loci from the parents' compact
    // tables (k_newborn_tb); offspring that received a remote gamete re-read theirs from
    // the row the gamete was put in
    // (their rows are complete: the local gametes were cut on this stream, the puts were
    // waited for; no join, the other offspring's crossover stays deferred)
    GNXCHK(gnx_l_tb_from_rows(h, h->birth_first_slot, h->n_req, h->req_k, nullptr, false));
    // (everybody else's phenotype came with k_offspring)
    static const bool tile_fuse = !(getenv("GNX_TILE_FUSE_TB") && atoi(getenv("GNX_TILE_FUSE_TB")) == 0);
    if (tile_fuse)
      GNXCHK(gnx_l_phenotype(h, h->birth_first_slot, h->n_req, h->req_k));
    else
      GNXCHK(gnx_l_phenotype(h, h->birth_first_slot, B));
  }
  return 0;
}

// gnx_tile_step_begin ends with this: the step's offspring are complete before the host looks at
// them (mutations, pedigree records); gnx_tile2_finish_births finds it done
extern "C" int gnx_tile2_settle_births(gnx_state* h, int32_t burn) { return tile_settle_births(h, burn); }

static int tile_finish_births(gnx_state* h, int32_t burn, const GnxSetWords* sw) {
  GNXCHK(tile_settle_births(h, burn));
  h->tile_births_settled = false;
  GnxSoA s = h->soa[h->cur];
  h->last_N_fused = false;
  GNXCHK(gnx_l_bins(h, h->N, s.x, s.y, s.ghost, h->bin_partials, nullptr, sw));
  return 0;
}

// N and n_pairs splines from the (all-reduced) bins, death probabilities,
// mortality; ghosts are dropped
extern "C" int gnx_tile_die(gnx_state* h, int32_t burn, int32_t with_selection,
                            int32_t have_pairs) {
  if (have_pairs)
    GNXCHK(gnx_l_spline(h, h->bins_P, &h->spl_P, nullptr));
  else
    h->spl_P.valid = false;
  GNXCHK(gnx_l_spline(h, h->bin_partials, &h->spl_N, nullptr));
  GNXCHK(gnx_l_death_probs(h, with_selection != 0 && !burn));
  int64_t D = 0;
  GNXCHK(gnx_l_mortality(h, nullptr, &D));
  HIPCHK(hipStreamSynchronize(h->stream));       // tile calls return with the stream idle
  h->last_deaths = D;
  return 0;
}

extern "C" int gnx_set_max_id(gnx_state* h, int64_t max_id) {
  h->max_id = max_id;
  return 0;
}

// =====================================================================================
// Device-resident transport.  RCCL moves GPU memory, so the payloads of a step
// never visit the host: the staged selection is grouped by destination rank on
// the device, the host layer wraps the device addresses returned here in
// tensors and hands them to isend/irecv, and the receiving side imports
// straight from the buffers RCCL filled.  Only the per-destination counts
// (R*C integers) cross PCIe.
// =====================================================================================
__device__ __forceinline__ int tile_index(float v, int tw, int n) {
  return gnx_tile_index(v, tw, n);          // (gnx_internal.h: exact on the integer boundaries)
}

__global__ void k_dest_owner(int64_t n, const gnx_ind_rec* rec, int tw, int th, int R, int C,
                             uint32_t* key, int32_t* idx) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  key[q] = (uint32_t)(tile_index(rec[q].y, th, R) * C + tile_index(rec[q].x, tw, C));
  idx[q] = (int32_t)q;
}

__global__ void k_req_owner(int64_t n, const float* px, const float* py, int tw, int th, int R,
                            int C, uint32_t* key, int32_t* idx) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  key[q] = (uint32_t)(tile_index(py[q], th, R) * C + tile_index(px[q], tw, C));
  idx[q] = (int32_t)q;
}

__global__ void k_count_keys(int64_t m, const uint32_t* key, int32_t* counts) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  // sorted keys: one atomic per run boundary would do, but m is a few 10^4
  atomicAdd(&counts[key[q]], 1);
}

__global__ void k_gather_staged(int64_t m, const int32_t* idx, const gnx_ind_rec* rec,
                                const float* z, const int64_t* slots, int n_traits,
                                gnx_ind_rec* rec_o, float* z_o, int64_t* slots_o) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  int i = idx[q];
  rec_o[q] = rec[i];
  slots_o[q] = slots[i];
  if (z)
    for (int t = 0; t < n_traits; ++t) z_o[q * n_traits + t] = z[(int64_t)i * n_traits + t];
}

static int n_tiles(const gnx_state* h) { return h->tile_R * h->tile_C; }

// sort (key, idx)[m] by key (stable), count per destination -> host counts[R*C]
static int group_by_key(gnx_state* h, int64_t m, int64_t* counts) {
  const int nt = n_tiles(h);
  if (!h->tile_counts) GNXCHK(dalloc_t(&h->tile_counts, (size_t)GNX_MAX_TILES * 2));
  HIPCHK(hipMemsetAsync(h->tile_counts, 0, nt * sizeof(int32_t), h->stream));
  if (m > 0) {
    int bits = 1;
    while ((1 << bits) < nt) ++bits;
    GNXCHK(gnx_prim_sort(h->sort_tmp, h->sort_tmp_bytes, h->key[0], h->key[1], h->perm[0],
                         h->perm[1], (size_t)m, bits, h->stream));
    hipLaunchKernelGGL(k_count_keys, dim3(gnx_grid(m, 256)), dim3(256), 0, h->stream, m,
                       h->key[1], h->tile_counts);
  }
  std::vector<int32_t> c(nt);
  GNXCHK(gnx_d2h(h, c.data(), h->tile_counts, nt * sizeof(int32_t)));
  for (int p = 0; p < nt; ++p) counts[p] = c[p];
  return 0;
}

static int grouped_gather(gnx_state* h, int64_t m, bool with_z) {
  h->gp_n = m;
  if (m == 0) return 0;
  if (m > h->gp_cap) {
    (void)hipFree(h->gp_rec);
    (void)hipFree(h->gp_z);
    (void)hipFree(h->gp_slots);
    h->gp_cap = m + m / 4 + 1024;
    GNXCHK(dalloc_t(&h->gp_rec, (size_t)h->gp_cap));
    GNXCHK(dalloc_t(&h->gp_z, (size_t)h->gp_cap * std::max(h->cfg.n_traits, 1)));
    GNXCHK(dalloc_t(&h->gp_slots, (size_t)h->gp_cap));
  }
  hipLaunchKernelGGL(k_gather_staged, dim3(gnx_grid(m, 256)), dim3(256), 0, h->stream, m,
                     h->perm[1], h->st_rec, (with_z && h->cfg.n_traits) ? h->st_z : nullptr,
                     h->st_slots, h->cfg.n_traits, h->gp_rec, h->gp_z, h->gp_slots);
  HIPCHK(hipGetLastError());
  return 0;
}

static int check_tiles(gnx_state* h, const char* who) {
  if (n_tiles(h) > GNX_MAX_TILES) {
    gnx_set_error("%s: more than %d tiles", who, GNX_MAX_TILES);
    return 1;
  }
  return 0;
}

// migrants grouped by the rank that owns their new position; they leave this tile
extern "C" int gnx_tile_export_migrants_dev(gnx_state* h, int64_t* counts /*[R*C]*/) {
  GNXCHK(check_tiles(h, "gnx_tile_export_migrants_dev"));
  for (int p = 0; p < n_tiles(h); ++p) counts[p] = 0;
  h->gp_n = 0;
  h->st_has_geno = false;
  if (h->n_ghost) {
    gnx_set_error("gnx_tile_export_migrants_dev: ghosts are resident");
    return 1;
  }
  int64_t N = h->N, n = 0;
  if (N == 0) return 0;
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL(k_mark_out, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s.x, s.y,
                     tile_box(h), h->flag, h->dead_in);
  GNXCHK(stage_selection(h, nullptr, true, false, &n));
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_dest_owner, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, h->st_rec,
                     h->cfg.W / h->tile_C, h->cfg.H / h->tile_R, h->tile_R, h->tile_C, h->key[0],
                     h->perm[0]);
  GNXCHK(group_by_key(h, n, counts));
  GNXCHK(grouped_gather(h, n, true));
  if (has_rows(h)) {
    if (n > h->st_geno_cap) {
      (void)hipFree(h->st_geno);
      h->st_geno_cap = n + n / 4 + 64;
      GNXCHK(dalloc_t(&h->st_geno, (size_t)h->st_geno_cap * 2 * h->W64));
    }
    GNXCHK(gnx_l_gather_genomes(h, n, h->gp_slots, h->st_geno));
    h->st_has_geno = true;
  }
  int64_t D = 0;
  GNXCHK(gnx_l_mortality(h, h->dead_in, &D));
  HIPCHK(hipStreamSynchronize(h->stream));       // tile calls return with the stream idle
  return 0;
}

// halo records, one copy per neighbour tile that needs them, grouped by rank.  Two
// passes over the individuals with wave-aggregated atomics (count per destination, then
// append behind the destination's offset): the order inside a group is whatever the
// atomics give, which nobody depends on - the receiver sorts by (cell, id).
__device__ __forceinline__ int halo_mask_of(float x, float y, const HaloSpans& sp, double inv_cs,
                                            int ncx, int ncy) {
  const int cx = min(ncx - 1, (int)((double)x * inv_cs));
  const int cy = min(ncy - 1, (int)((double)y * inv_cs));
  int m = 0;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      if (dx == 1 && dy == 1) continue;
      if (sp.okx[dx] && sp.oky[dy] && cx >= sp.cx0[dx] && cx <= sp.cx1[dx] &&
          cy >= sp.cy0[dy] && cy <= sp.cy1[dy])
        m |= 1 << (dy * 3 + dx);
    }
  return m;
}

template <bool WRITE>
__global__ void __launch_bounds__(256)
k_halo_direct(int64_t N, GnxSoA s, HaloSpans sp, double inv_cs, int ncx, int ncy, int tr, int tc,
              int C, int32_t* __restrict__ counts, const int32_t* __restrict__ offs,
              gnx_ind_rec* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int m = 0;
  if (i < N && !s.ghost[i]) m = halo_mask_of(s.x[i], s.y[i], sp, inv_cs, ncx, ncy);
  if (__ballot(m != 0) == 0ull) return;          // wave-uniform: nobody near a border
  for (int k = 0; k < 9; ++k) {
    if (k == 4) continue;
    const unsigned long long b = __ballot((m >> k) & 1);
    if (b == 0ull) continue;
    const int rank = (tr + k / 3 - 1) * C + (tc + k % 3 - 1);
    int base = 0;
    if (lane == __ffsll((long long)b) - 1) base = atomicAdd(&counts[rank], __popcll(b));
    if (WRITE) {
      base = __shfl(base, __ffsll((long long)b) - 1);
      if ((m >> k) & 1) {
        gnx_ind_rec r;
        r.x = s.x[i];
        r.y = s.y[i];
        r.age = s.age[i];
        r.sex = s.sex[i];
        r.id = s.id[i];
        r.fit = s.fit[i];
        r.nbr_mask = m;
        out[offs[rank] + base + __popcll(b & ((1ull << lane) - 1ull))] = r;
      }
    }
  }
}

extern "C" int gnx_tile_export_halo_dev(gnx_state* h, int64_t* counts) {
  GNXCHK(check_tiles(h, "gnx_tile_export_halo_dev"));
  const int nt = n_tiles(h);
  for (int p = 0; p < nt; ++p) counts[p] = 0;
  h->gp_n = 0;
  h->st_has_geno = false;
  const int64_t N = h->N;
  if (N == 0) return 0;
  if (!h->tile_counts) GNXCHK(dalloc_t(&h->tile_counts, (size_t)GNX_MAX_TILES * 2));
  GnxSoA s = h->soa[h->cur];
  const HaloSpans sp = halo_spans(h);
  int32_t* cnt = h->tile_counts;
  int32_t* off = h->tile_counts + GNX_MAX_TILES;
  HIPCHK(hipMemsetAsync(cnt, 0, nt * sizeof(int32_t), h->stream));
  hipLaunchKernelGGL(k_halo_direct<false>, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s,
                     sp, h->inv_cs, h->ncx, h->ncy, h->tile_r, h->tile_c, h->tile_C, cnt, nullptr,
                     nullptr);
  std::vector<int32_t> c(nt), o(nt);
  GNXCHK(gnx_d2h(h, c.data(), cnt, nt * sizeof(int32_t)));
  int64_t m = 0;
  for (int p = 0; p < nt; ++p) {
    o[p] = (int32_t)m;
    counts[p] = c[p];
    m += c[p];
  }
  if (m == 0) return 0;
  if (m > h->gp_cap) {
    (void)hipFree(h->gp_rec);
    (void)hipFree(h->gp_z);
    (void)hipFree(h->gp_slots);
    h->gp_cap = m + m / 4 + 1024;
    GNXCHK(dalloc_t(&h->gp_rec, (size_t)h->gp_cap));
    GNXCHK(dalloc_t(&h->gp_z, (size_t)h->gp_cap * std::max(h->cfg.n_traits, 1)));
    GNXCHK(dalloc_t(&h->gp_slots, (size_t)h->gp_cap));
  }
  GNXCHK(gnx_h2d(h, off, o.data(), nt * sizeof(int32_t)));
  HIPCHK(hipMemsetAsync(cnt, 0, nt * sizeof(int32_t), h->stream));
  hipLaunchKernelGGL(k_halo_direct<true>, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s,
                     sp, h->inv_cs, h->ncx, h->ncy, h->tile_r, h->tile_c, h->tile_C, cnt, off,
                     h->gp_rec);
  HIPCHK(hipGetLastError());
  h->gp_n = m;
  return 0;
}

// device addresses of the grouped selection: rec [n], z [n][n_traits] (migrants),
// geno [n][2][W64] (migrants of a species with genomes); null when absent
extern "C" int gnx_tile_staged_ptrs(gnx_state* h, void** rec, void** z, void** geno) {
  HIPCHK(hipStreamSynchronize(h->stream));
  *rec = h->gp_n ? (void*)h->gp_rec : nullptr;
  *z = (h->gp_n && h->cfg.n_traits) ? (void*)h->gp_z : nullptr;
  *geno = (h->gp_n && h->st_has_geno) ? (void*)h->st_geno : nullptr;
  return 0;
}

__global__ void k_check_rec(int64_t n, const gnx_ind_rec* rec, int W, int H, int64_t* chk) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  gnx_ind_rec r = rec[k];
  if (!(r.x >= 0 && r.x < W && r.y >= 0 && r.y < H)) atomicAdd((unsigned long long*)&chk[0], 1ull);
  atomicMax((long long*)&chk[1], (long long)r.id);
}

static int import_device(gnx_state* h, int64_t n, const gnx_ind_rec* d_rec, const float* d_z,
                         const uint64_t* d_g, int ghost) {
  if (n == 0) return 0;
  const gnx_config& c = h->cfg;
  const bool rows = has_rows(h) && !ghost;
  if (h->N + n > c.cap_inds || (rows && n > h->n_free)) {
    gnx_set_error("capacity exceeded importing %lld individuals (N=%lld cap=%lld free rows %lld)",
                  (long long)n, (long long)h->N, (long long)c.cap_inds, (long long)h->n_free);
    return 2;
  }
  if (rows && !d_g) {
    gnx_set_error("import: genomes are assigned on this tile but none were sent");
    return 1;
  }
  if (!h->chk) GNXCHK(dalloc_t(&h->chk, 2));
  int64_t init[2] = {0, h->max_id};
  GNXCHK(gnx_h2d(h, h->chk, init, sizeof(init)));
  hipLaunchKernelGGL(k_check_rec, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, d_rec, c.W,
                     c.H, h->chk);
  int64_t res[2];
  GNXCHK(gnx_d2h(h, res, h->chk, sizeof(res)));
  if (res[0]) {
    gnx_set_error("import: %lld records are off the landscape", (long long)res[0]);
    return 1;
  }
  GnxSoA s = h->soa[h->cur];
  if (rows && !ghost) GNXCHK(gnx_half_reserve(h, 2 * (int64_t)h->NB * n));
  hipLaunchKernelGGL(k_unpack, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, h->N, n,
                     c.cap_inds, s, d_rec, (d_z && c.n_traits) ? d_z : nullptr, c.n_traits,
                     c.n_layers, h->rast, c.W, c.H, h->free_rows, h->n_free, rows ? 1 : 0, ghost,
                     gnx_halves(h));
  if (rows) {
    GNXCHK(gnx_l_scatter_genomes(h, n, (const uint64_t*)d_g, h->N));
    h->n_free -= n;
    GNXCHK(gnx_l_tb_from_rows(h, h->N, n, nullptr, nullptr));
  }
  // the source buffers belong to the caller: finish reading them before returning
  HIPCHK(hipStreamSynchronize(h->stream));
  h->N += n;
  h->ord_valid = false;          // arrivals carry any id: no id-ordered index
  if (ghost) h->n_ghost += n;
  h->max_id = std::max(h->max_id, res[1]);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int gnx_tile_import_dev(gnx_state* h, int64_t n, const void* rec, const void* z,
                                   const void* geno) {
  return import_device(h, n, (const gnx_ind_rec*)rec, (const float*)z, (const uint64_t*)geno, 0);
}

extern "C" int gnx_tile_import_ghosts_dev(gnx_state* h, int64_t n, const void* rec) {
  return import_device(h, n, (const gnx_ind_rec*)rec, nullptr, nullptr, 1);
}

// order keys (ascending, int64 [P]) and birth counts (int32 [P]; null when every
// pair has the fixed n_births) of the local pair list, on the device
extern "C" int gnx_tile_pair_ptrs(gnx_state* h, int64_t* n_pairs, void** focal_ids,
                                  void** n_births) {
  const int64_t P = h->n_pairs;
  *n_pairs = P;
  *focal_ids = nullptr;
  *n_births = nullptr;
  if (P == 0) return 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  *focal_ids = h->key64[0];
  if (!h->sp.n_births_fixed) *n_births = h->nbirths;
  return 0;
}

extern "C" int gnx_tile_offspring_dev(gnx_state* h, int32_t burn, int64_t id_base,
                                      const void* pair_goff_dev, int64_t* n_requests) {
  *n_requests = 0;
  int64_t P = h->n_pairs, B = 0;
  h->birth_first_slot = h->N;
  h->n_req = 0;
  h->tile_births_settled = false;
  if (P > 0 && pair_goff_dev) {
    HIPCHK(hipMemcpyAsync(h->pair_goff, pair_goff_dev, P * sizeof(int64_t),
                          hipMemcpyDeviceToDevice, h->stream));
    // the source belongs to the caller (a tensor it may release on return)
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  GNXCHK(gnx_l_mate(h, burn != 0, false, 0, &B, id_base, true));
  h->last_births = B;
  if (B == 0) h->n_req = 0;
  *n_requests = h->n_req;
  h->xo_pending = B > 0 && !burn && has_rows(h);     // the crossover is still running
  return 0;
}

// ---- gametes of ghost mates, device to device
__global__ void k_pack_requests(int64_t n, const int32_t* idx, const int64_t* pid,
                                const int32_t* key, const uint8_t* start, const int32_t* child_k,
                                gnx_gamete_req* out, int32_t* k_out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int i = idx[q];
  gnx_gamete_req r;
  r.parent_id = pid[i];
  r.key = key[i];
  r.start = start[i];
  out[q] = r;
  k_out[q] = child_k[i];
}

// requests of this step grouped by the rank that owns the ghost mate
extern "C" int gnx_tile_group_requests(gnx_state* h, int64_t* counts, void** req_dev) {
  SideStream side(h);
  GNXCHK(check_tiles(h, "gnx_tile_group_requests"));
  for (int p = 0; p < n_tiles(h); ++p) counts[p] = 0;
  *req_dev = nullptr;
  const int64_t n = h->n_req;
  if (n == 0) return 0;
  if (n > h->rq_cap) {
    (void)hipFree(h->rq_sorted);
    (void)hipFree(h->rq_k);
    h->rq_cap = n + n / 4 + 1024;
    gnx_gamete_req* p = nullptr;
    GNXCHK(dalloc_t(&p, (size_t)h->rq_cap));
    h->rq_sorted = p;
    GNXCHK(dalloc_t(&h->rq_k, (size_t)h->rq_cap));
  }
  hipLaunchKernelGGL(k_req_owner, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, h->req_px,
                     h->req_py, h->cfg.W / h->tile_C, h->cfg.H / h->tile_R, h->tile_R, h->tile_C,
                     h->key[0], h->perm[0]);
  GNXCHK(group_by_key(h, n, counts));
  hipLaunchKernelGGL(k_pack_requests, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n,
                     h->perm[1], h->req_pid, h->req_key, h->req_start, h->req_k,
                     (gnx_gamete_req*)h->rq_sorted, h->rq_k);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  *req_dev = h->rq_sorted;
  return 0;
}

__global__ void k_lookup_req(int64_t n, const gnx_gamete_req* req, int64_t N, int n_paths,
                             const uint64_t* sorted_ids, const int32_t* sorted_slots,
                             int32_t* out_slot, int32_t* n_bad) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const gnx_gamete_req r = req[q];
  uint64_t w = (uint64_t)r.parent_id;
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (sorted_ids[mid] < w) lo = mid + 1; else hi = mid;
  }
  const bool ok = lo < N && sorted_ids[lo] == w && r.key >= 0 && r.key < n_paths &&
                  (r.start == 0 || r.start == 1);
  out_slot[q] = ok ? sorted_slots[lo] : -1;
  if (!ok) atomicAdd(n_bad, 1);
}

__global__ void __launch_bounds__(256)
k_make_gametes_req(int64_t n, int W16, const u64x2* __restrict__ G,
                   const int32_t* __restrict__ grow, GnxHalves H,
                   const int32_t* __restrict__ slot, const gnx_gamete_req* __restrict__ req, const u64x2* __restrict__ paths,
                   u64x2* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= n) return;
  const int prow = grow[slot[q]];
  const gnx_gamete_req r = req[q];
  const u64 s = r.start ? ~0ull : 0ull;
  const int64_t lh0 = (int64_t)prow * 2;
  const u64x2* pm = paths + (int64_t)r.key * W16;
  for (int c = lane; c < W16; c += 64) {
    u64x2 m = pm[c];
    m.a ^= s;
    m.b ^= s;
    const u64x2 a = G[gnx_chunk_at(H, lh0, c)], b = G[gnx_chunk_at(H, lh0 + 1, c)];
    u64x2 o;
    o.a = (a.a & ~m.a) | (b.a & m.a);
    o.b = (a.b & ~m.b) | (b.b & m.b);
    out[q * W16 + c] = o;
  }
}

// cut the requested gametes (requests in device memory, as received) into a
// device buffer owned by the handle; *out_dev stays valid until the next call
extern "C" int gnx_tile_serve_gametes_dev(gnx_state* h, int64_t n, const void* req_dev,
                                          void** out_dev) {
  SideStream side(h);
  *out_dev = nullptr;
  if (n == 0) return 0;
  if (!has_rows(h) || h->n_paths == 0) {
    gnx_set_error("gnx_tile_serve_gametes_dev: genomes / paths not set");
    return 1;
  }
  if (n > h->gam_cap) {
    (void)hipFree(h->gam_out);
    (void)hipFree(h->gam_slot);
    h->gam_cap = n + n / 4 + 256;
    GNXCHK(dalloc_t(&h->gam_out, (size_t)h->gam_cap * h->W64));
    GNXCHK(dalloc_t(&h->gam_slot, (size_t)h->gam_cap + 1));
  }
  const int64_t N = h->N;
  GnxSoA s = h->soa[h->cur];
  GNXCHK(serve_wait_crossover(h, side.saved));
  hipLaunchKernelGGL(k_id_keys, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, s.id,
                     h->key64[0], h->perm[0]);
  GNXCHK(gnx_prim_sort64(h->sort64_tmp, h->sort64_tmp_bytes, h->key64[0], h->key64[1], h->perm[0],
                         h->perm[1], (size_t)N, h->stream));
  int32_t* d_bad = h->gam_slot + h->gam_cap;
  HIPCHK(hipMemsetAsync(d_bad, 0, 4, h->stream));
  hipLaunchKernelGGL(k_lookup_req, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n,
                     (const gnx_gamete_req*)req_dev, N, h->n_paths, h->key64[1], h->perm[1],
                     h->gam_slot, d_bad);
  int bad = 0;
  GNXCHK(gnx_d2h(h, &bad, d_bad, 4));
  if (bad) {
    gnx_set_error("gnx_tile_serve_gametes_dev: %d requests name a parent that does not live on "
                  "this tile, or a path / start out of range", bad);
    return 1;
  }
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_make_gametes_req, dim3(gnx_grid(n * 64, 256)), dim3(256), 0, h->stream, n,
                     W16, (const u64x2*)h->G, s.grow, gnx_halves(h), h->gam_slot,
                     (const gnx_gamete_req*)req_dev,
                     (const u64x2*)h->paths, (u64x2*)h->gam_out);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  *out_dev = h->gam_out;
  return 0;
}

// the gametes answering this tile's grouped requests, in the grouped order
extern "C" int gnx_tile_put_gametes_dev(gnx_state* h, int64_t n, const void* data_dev) {
  SideStream side(h);
  if (n == 0) return 0;
  if (n != h->n_req) {
    gnx_set_error("gnx_tile_put_gametes_dev: %lld gametes for %lld requests", (long long)n,
                  (long long)h->n_req);
    return 1;
  }
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_put_gametes, dim3(gnx_grid(n * W16, 256, 256 * 32)), dim3(256), 0, h->stream,
                     n, W16, (const u64x2*)data_dev, (u64x2*)h->G, h->soa[h->cur].grow,
                     gnx_halves(h), h->birth_first_slot, h->rq_k);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  return 0;
}

// both density bin fields, contiguous int32 [2][bin_count]: individuals, pair midpoints
extern "C" int gnx_tile_bins_ptr(gnx_state* h, void** bins, int64_t* n_total) {
  HIPCHK(hipStreamSynchronize(h->stream));
  *bins = h->bin_partials;
  *n_total = 2 * (int64_t)h->lat.nbx * h->lat.nby;
  return 0;
}

// =====================================================================================
// tile2: the device-driven protocol (include/gnx_hip.h).  Per step the host waits for the
// device three times (routing counts, pair / request counts, survivor count); payloads are
// grouped, exchanged and imported in device memory, and nothing else blocks the host.
// =====================================================================================
static int route_geo(const gnx_state* h, RouteGeo* g) {
  if (h->tile_R > GNX_TILE_DIM || h->tile_C > GNX_TILE_DIM) {
    gnx_set_error("tile2: at most %d tiles per axis", GNX_TILE_DIM);
    return 1;
  }
  g->R = h->tile_R;
  g->C = h->tile_C;
  g->tw = h->cfg.W / h->tile_C;
  g->th = h->cfg.H / h->tile_R;
  g->me = h->tile_r * h->tile_C + h->tile_c;
  g->ncx = h->ncx;
  g->ncy = h->ncy;
  g->inv_cs = h->inv_cs;
  const int ring = 2 * h->cell_ref;       // two mating radii, in cells (halo_spans)
  for (int c = 0; c < g->C; ++c) {
    const float x0 = (float)(c * g->tw), x1 = nextafterf((float)((c + 1) * g->tw), 0.f);
    g->cx0[c] = std::min(h->ncx - 1, (int)((double)x0 * h->inv_cs)) - ring;
    g->cx1[c] = std::min(h->ncx - 1, (int)((double)x1 * h->inv_cs)) + ring;
  }
  for (int r = 0; r < g->R; ++r) {
    const float y0 = (float)(r * g->th), y1 = nextafterf((float)((r + 1) * g->th), 0.f);
    g->cy0[r] = std::min(h->ncy - 1, (int)((double)y0 * h->inv_cs)) - ring;
    g->cy1[r] = std::min(h->ncy - 1, (int)((double)y1 * h->inv_cs)) + ring;
  }
  return 0;
}

// Where every individual of the tile goes after the movement: to the tile that owns its
// position if that is another one (direction 4 below), and as a ghost to every tile
// around its OWNER whose widened cell span holds its cell.  Pass 1 counts per destination
// rank (cnt[0 .. T) migrants, cnt[T .. 2T) ghosts), pass 2 writes behind the groups'
// offsets (order inside a group: whatever the atomics give; receivers sort by cell and id).
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_route(int64_t N, int64_t cap, GnxSoA s, RouteGeo g, int n_traits, int32_t* __restrict__ cnt,
        const int32_t* __restrict__ offs, gnx_ind_rec* __restrict__ mig_rec,
        float* __restrict__ mig_z, int64_t* __restrict__ mig_slot,
        gnx_ind_rec* __restrict__ gh_rec, int64_t mig_cap, int64_t gh_cap,
        const int32_t* __restrict__ alive) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (alive: the last mortality left its dead in place - gnx_tile_walk)
  const bool act = i < N && !s.ghost[i] && (!alive || (alive[i] & 1) != 0);
  const int T = g.R * g.C;
  float x = 0.f, y = 0.f;
  int oc = 0, orow = 0, cx = 0, cy = 0;
  if (act) {
    x = s.x[i];
    y = s.y[i];
    oc = tile_index(x, g.tw, g.C);
    orow = tile_index(y, g.th, g.R);
    cx = min(g.ncx - 1, (int)((double)x * g.inv_cs));
    cy = min(g.ncy - 1, (int)((double)y * g.inv_cs));
  }
  const int owner = orow * g.C + oc;
  gnx_ind_rec r;
  if (WRITE && act) {
    r.x = x;
    r.y = y;
    r.age = s.age[i];
    r.sex = s.sex[i];
    r.id = s.id[i];
    r.fit = s.fit[i];
    r.nbr_mask = 0;
  }
  for (int k = 0; k < 9; ++k) {
    int dest = -1;
    if (act) {
      if (k == 4) {
        if (owner != g.me) dest = owner;                    // a migrant
      } else {
        const int rr = orow + k / 3 - 1, cc = oc + k % 3 - 1;
        if (rr >= 0 && rr < g.R && cc >= 0 && cc < g.C && cx >= g.cx0[cc] && cx <= g.cx1[cc] &&
            cy >= g.cy0[rr] && cy <= g.cy1[rr])
          dest = rr * g.C + cc;                             // a ghost there
      }
    }
    if (__ballot(dest >= 0) == 0ull) continue;
    const int grp = (k == 4) ? 0 : T;
    const int idx = gnx_route_append(cnt + grp, dest);
    if (WRITE && dest >= 0) {
      const int64_t q = (int64_t)offs[grp + dest] + idx;
      if (k == 4) {
        if (q < mig_cap) {
          mig_rec[q] = r;
          mig_slot[q] = i;
          if (mig_z)
            for (int t = 0; t < n_traits; ++t) mig_z[q * n_traits + t] = s.z[(int64_t)t * cap + i];
        }
      } else if (q < gh_cap) {
        gh_rec[q] = r;
      }
    }
  }
}

// exclusive offsets of the 2 x T groups, totals, and all of it to pinned host memory
__global__ void k_route_offsets(int T, int32_t* cnt /*[2T counts | 2T offsets | 2 totals]*/,
                                int32_t* host, int n_extra, int seq, int zero_counts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int grp = 0; grp < 2; ++grp) {
    int run = 0;
    for (int p = 0; p < T; ++p) {
      const int v = cnt[grp * T + p];
      cnt[2 * T + grp * T + p] = run;
      run += v;
      if (zero_counts) cnt[grp * T + p] = 0;      // (the pass that fills counts again from zero)
    }
    cnt[4 * T + grp] = run;
  }
  if (!host) return;
  for (int p = 0; p < 2 * T + n_extra; ++p)
    __hip_atomic_store(&host[p], cnt[p < 2 * T ? p : 4 * T + (p - 2 * T)], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(&host[2 * GNX_MAX_TILES + 7], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// plain counters -> pinned host words behind whatever the stream holds so far
__global__ void k_publish_words(int n, const int32_t* src, int32_t* host, int seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int p = 0; p < n; ++p)
    __hip_atomic_store(&host[p], src[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(&host[2 * GNX_MAX_TILES + 7], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the step's small accumulators in one launch instead of a fill kernel each: the deferred
// checks, both density fields' bins with the four counter words behind them, the N.max()
// word, the gamete-request counter
__global__ void k_tile2_zero(int32_t* p0, int n0, int32_t* p1, int n1, int32_t* p2, int n2,
                             int32_t* p3, int n3, int32_t* p4, int n4) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) p1[i] = 0;
  // (the routing's counters and the gamete-request counts: a fill kernel each otherwise - and a
  // hipMemsetAsync of an odd size is up to THREE of them, 5 us apiece on the step's chain)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p4[i] = 0;
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < n0) p0[threadIdx.x] = 0;
    if ((int)threadIdx.x < n2) p2[threadIdx.x] = 0;
    if ((int)threadIdx.x < n3) p3[threadIdx.x] = 0;
  }
}

// two groups of words for the host that needs no sequence number (a later wait covers them)
__global__ void k_publish_words2(int n1, const int32_t* src1, int32_t* host1, int n2,
                                 const int32_t* src2, int32_t* host2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int p = 0; p < n1; ++p)
    __hip_atomic_store(&host1[p], src1[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int p = 0; p < n2; ++p)
    __hip_atomic_store(&host2[p], src2[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define GNX_REQ_CNT(h) ((h)->route_cnt + 4 * n_tiles(h) + 8)
static int tile2_buffers(gnx_state* h) {
  h->tile2_mode = true;         // (the step's density comes from the tile protocol's own counting pass)
  if (!h->route_cnt) {
    // [2T counts | 2T offsets | 2 totals ...] of the routing, and behind those 4T + 8 words the
    // gamete-request counts per owning rank (GNX_REQ_CNT)
    GNXCHK(dalloc_t(&h->route_cnt, (size_t)5 * GNX_MAX_TILES + 8));
    HIPCHK(hipMemset(h->route_cnt, 0, ((size_t)5 * GNX_MAX_TILES + 8) * sizeof(int32_t)));
    HIPCHK(hipHostMalloc((void**)&h->h_route_pin, (2 * GNX_MAX_TILES + 8) * sizeof(int32_t),
                         hipHostMallocCoherent | hipHostMallocMapped));
    memset(h->h_route_pin, 0, (2 * GNX_MAX_TILES + 8) * sizeof(int32_t));
    HIPCHK(hipHostGetDevicePointer((void**)&h->h_route_pin_dev, h->h_route_pin, 0));
  }
  if (!h->chk) GNXCHK(dalloc_t(&h->chk, 2));
  return 0;
}

// the host spins on the sequence word the publishing kernel writes last
static int tile2_wait(gnx_state* h, int seq) {
  volatile int32_t* word = h->h_route_pin + 2 * GNX_MAX_TILES + 7;
  for (long spin = 0;; ++spin) {
    if (*word == seq) break;
    if (spin > 2000000) {                 // (~20 ms of spinning: fall back to the driver)
      HIPCHK(hipStreamSynchronize(h->stream));
      if (*word != seq) {
        gnx_set_error("tile2: the device counters never arrived");
        return 1;
      }
      break;
    }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

extern "C" int gnx_stream_ptr(gnx_state* h, void** stream) {
  *stream = (void*)h->stream;
  return 0;
}

// The routing in two halves (gnx_tile_step merges the wait for this tile's counts with the
// exchange of everybody's): _begin enqueues age + movement and the counting pass and leaves
// the 2T counts in device memory (*counts_dev, int32: migrants per rank, ghosts per rank);
// _finish takes them from the host (counts[2T]) and enqueues the pass that fills the staging
// buffers.  gnx_tile2_move_route = _begin, wait, _finish.
extern "C" int gnx_tile2_route_begin(gnx_state* h, int32_t move, void** counts_dev) {
  GNXCHK(check_tiles(h, "gnx_tile2_move_route"));
  const int T = n_tiles(h);
  *counts_dev = nullptr;
  h->route_n_mig = h->route_n_gh = 0;
  h->st_has_geno = false;
  if (h->n_ghost) {
    gnx_set_error("gnx_tile2_move_route: ghosts are resident");
    return 1;
  }
  GNXCHK(tile2_buffers(h));
  {
    // (everything that read last step's values ran before this on the same stream)
    const int nb = h->lat.nbx * h->lat.nby;
    const bool bins = h->have_sp && h->bin_partials != nullptr;
    const int T0 = n_tiles(h);
    hipLaunchKernelGGL(k_tile2_zero, dim3(8), dim3(256), 0, h->stream, (int32_t*)h->chk, 4,
                       bins ? h->bin_partials : nullptr, bins ? 2 * nb + 4 : 0,
                       (int32_t*)h->nmax_bits, h->nmax_bits ? 2 : 0, h->req_count,
                       h->req_count ? 1 : 0, h->route_cnt, 5 * T0 + 8);
    if (bins) h->bins_zeroed[0] = h->bins_zeroed[1] = true;
    h->req_cnt_zeroed = true;
    if (h->nmax_bits) h->nmax_zeroed = true;
    h->req_zeroed = h->req_count != nullptr;
  }
  h->move_counted_routes = false;
  if (move && h->sp.move) {
    // (one tile: nobody arrives between the movement and the cell sort - the movement writes
    // the sort's keys as in gnx_step)
    h->move_writes_keys = T == 1 && h->sp.mating_radius >= 0;
    // several tiles: the routing's counting pass rides in the movement kernel (it has the new
    // positions in registers; k_route<false> read them back: 34 us of a tile-step) - GNX_ROUTE_IN_MOVE=0: off
    static const bool in_move = !(getenv("GNX_ROUTE_IN_MOVE") && atoi(getenv("GNX_ROUTE_IN_MOVE")) == 0);
    if (in_move && T > 1 && h->N > 0) {
      if (!h->route_geo_dev) HIPCHK(hipMalloc((void**)&h->route_geo_dev, sizeof(RouteGeo)));
      if (h->route_geo_epoch != h->cfg_epoch) {
        RouteGeo g;
        GNXCHK(route_geo(h, &g));
        HIPCHK(hipMemcpyAsync(h->route_geo_dev, &g, sizeof(g), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));          // (`g` is on this frame; once per geometry)
        h->route_geo_epoch = h->cfg_epoch;
      }
      h->move_counts_routes = true;
    }
    const int rc_move = gnx_l_move(h, true, nullptr, nullptr, nullptr, nullptr, true);
    h->move_writes_keys = false;
    h->move_counts_routes = false;
    GNXCHK(rc_move);
  } else {
    GNXCHK(gnx_l_age(h));
  }
  const int64_t N = gnx_extent(h);       // (every slot of an uncompacted population; the dead are skipped)
  *counts_dev = h->route_cnt;             // (zeroed by k_tile2_zero above)
  if (h->N == 0 || T == 1) return 0;       // one tile: nobody leaves, nobody borders
  RouteGeo g;
  GNXCHK(route_geo(h, &g));
  GnxSoA s = h->soa[h->cur];
  const int nt = h->cfg.n_traits;
  if (h->move_counted_routes) return 0;    // (the movement kernel has counted)
  hipLaunchKernelGGL(k_route<false>, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N,
                     h->cfg.cap_inds, s, g, nt, h->route_cnt, (const int32_t*)nullptr,
                     (gnx_ind_rec*)nullptr, (float*)nullptr, (int64_t*)nullptr,
                     (gnx_ind_rec*)nullptr, (int64_t)0, (int64_t)0,
                     h->holes ? (const int32_t*)h->flag : (const int32_t*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int gnx_tile2_route_finish(gnx_state* h, const int64_t* counts) {
  const int T = n_tiles(h);
  const int64_t N = gnx_extent(h);
  if (h->N == 0 || T == 1) return 0;
  RouteGeo g;
  GNXCHK(route_geo(h, &g));
  GnxSoA s = h->soa[h->cur];
  const int nt = h->cfg.n_traits;
  int64_t n_mig = 0, n_gh = 0;
  for (int p = 0; p < T; ++p) {
    n_mig += counts[p];
    n_gh += counts[T + p];
  }
  if (counts[g.me] != 0) {
    gnx_set_error("tile2: %lld migrants routed to their own tile", (long long)counts[g.me]);
    return 1;
  }
  h->route_n_mig = n_mig;
  h->route_n_gh = n_gh;
  if (n_mig == 0 && n_gh == 0) return 0;
  // grow-only staging (the host knows the totals before anything is written)
  if (n_mig > h->gp_cap) {
    (void)hipFree(h->gp_rec);
    (void)hipFree(h->gp_z);
    (void)hipFree(h->gp_slots);
    h->gp_cap = n_mig + n_mig / 4 + 1024;
    GNXCHK(dalloc_t(&h->gp_rec, (size_t)h->gp_cap));
    GNXCHK(dalloc_t(&h->gp_z, (size_t)h->gp_cap * std::max(nt, 1)));
    GNXCHK(dalloc_t(&h->gp_slots, (size_t)h->gp_cap));
  }
  if (n_gh > h->gh_cap) {
    (void)hipFree(h->gh_rec);
    h->gh_cap = n_gh + n_gh / 4 + 4096;
    GNXCHK(dalloc_t(&h->gh_rec, (size_t)h->gh_cap));
  }
  // the per-destination offsets of both groups (the counting pass left the counts), then the
  // counters again for the pass that fills
  hipLaunchKernelGGL(k_route_offsets, dim3(1), dim3(64), 0, h->stream, T, h->route_cnt,
                     (int32_t*)nullptr, 0, 0, 1);
  hipLaunchKernelGGL(k_route<true>, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N,
                     h->cfg.cap_inds, s, g, nt, h->route_cnt,
                     (const int32_t*)(h->route_cnt + 2 * T), h->gp_rec, nt ? h->gp_z : nullptr,
                     h->gp_slots, h->gh_rec, h->gp_cap, h->gh_cap,
                     h->holes ? (const int32_t*)h->flag : (const int32_t*)nullptr);
  if (n_mig > 0 && has_rows(h)) {
    if (n_mig > h->st_geno_cap) {
      (void)hipFree(h->st_geno);
      h->st_geno_cap = n_mig + n_mig / 4 + 64;
      GNXCHK(dalloc_t(&h->st_geno, (size_t)h->st_geno_cap * 2 * h->W64));
    }
    GNXCHK(gnx_l_gather_genomes(h, n_mig, h->gp_slots, h->st_geno));
    h->st_has_geno = true;
  }
  HIPCHK(hipGetLastError());
  // the emigrants leave with the cell sort (gnx_l_sort_by_cell)
  h->tile_evict = n_mig;
  const TileBox tb = tile_box(h);
  h->evict_box[0] = tb.x0;
  h->evict_box[1] = tb.x1;
  h->evict_box[2] = tb.y0;
  h->evict_box[3] = tb.y1;
  h->fb_adults = false;
  return 0;
}

extern "C" int gnx_tile2_move_route(gnx_state* h, int32_t move, int64_t* counts) {
  const int T = n_tiles(h);
  for (int p = 0; p < 2 * T; ++p) counts[p] = 0;
  void* d_counts = nullptr;
  GNXCHK(gnx_tile2_route_begin(h, move, &d_counts));
  if (h->N == 0 || T == 1) return 0;
  const int seq = (int)(++h->pin_seq & 0x3fffffff);
  hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, h->stream, 2 * T,
                     (const int32_t*)h->route_cnt, h->h_route_pin_dev, seq);
  HIPCHK(hipGetLastError());
  GNXCHK(tile2_wait(h, seq));                                   // wait 1 of the step
  for (int p = 0; p < 2 * T; ++p) counts[p] = h->h_route_pin[p];
  return gnx_tile2_route_finish(h, counts);
}

extern "C" int gnx_tile2_route_ptrs(gnx_state* h, void** mig_rec, void** mig_z, void** mig_geno,
                                    void** ghost_rec) {
  *mig_rec = h->route_n_mig ? (void*)h->gp_rec : nullptr;
  *mig_z = (h->route_n_mig && h->cfg.n_traits) ? (void*)h->gp_z : nullptr;
  *mig_geno = (h->route_n_mig && h->st_has_geno) ? (void*)h->st_geno : nullptr;
  *ghost_rec = h->route_n_gh ? (void*)h->gh_rec : nullptr;
  return 0;
}

// records off the landscape are counted on the device (chk[0]) and reported by gnx_tile2_die
__global__ void k_check_rec2(int64_t n, const gnx_ind_rec* rec, int W, int H, int64_t* chk) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const gnx_ind_rec r = rec[k];
  if (!(r.x >= 0 && r.x < W && r.y >= 0 && r.y < H)) atomicAdd((unsigned long long*)&chk[0], 1ull);
}

static int import2(gnx_state* h, int64_t n, const gnx_ind_rec* d_rec, const float* d_z,
                   const uint64_t* d_g, int ghost) {
  if (n == 0) return 0;
  const gnx_config& c = h->cfg;
  const bool rows = has_rows(h) && !ghost;
  // (an uncompacted population - gnx_tile_walk: the arrivals go behind the stretch it is spread over)
  const int64_t first = gnx_extent(h);
  if (first + n > c.cap_inds || (rows && n > h->n_free)) {
    gnx_set_error("capacity exceeded importing %lld individuals (N=%lld cap=%lld free rows %lld)",
                  (long long)n, (long long)first, (long long)c.cap_inds, (long long)h->n_free);
    return 2;
  }
  if (rows && !d_g) {
    gnx_set_error("import: genomes are assigned on this tile but none were sent");
    return 1;
  }
  hipLaunchKernelGGL(k_check_rec2, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, d_rec, c.W,
                     c.H, h->chk);
  GnxSoA s = h->soa[h->cur];
  if (rows) GNXCHK(gnx_half_reserve(h, 2 * (int64_t)h->NB * n));
  hipLaunchKernelGGL(k_unpack, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, first, n,
                     c.cap_inds, s, d_rec, (d_z && c.n_traits) ? d_z : nullptr, c.n_traits,
                     c.n_layers, h->rast, c.W, c.H, h->free_rows, h->n_free, rows ? 1 : 0, ghost,
                     gnx_halves(h));
  if (rows) {
    GNXCHK(gnx_l_scatter_genomes(h, n, (const uint64_t*)d_g, first));
    h->n_free -= n;
    GNXCHK(gnx_l_tb_from_rows(h, first, n, nullptr, nullptr));
  }
  h->N += n;
  if (h->holes) h->holes_N += n;
  h->ord_valid = false;
  if (ghost) h->n_ghost += n;
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int gnx_tile2_import(gnx_state* h, int64_t n_mig, const void* rec, const void* z,
                                const void* geno, int64_t n_ghost, const void* ghost_rec) {
  GNXCHK(tile2_buffers(h));
  GNXCHK(import2(h, n_mig, (const gnx_ind_rec*)rec, (const float*)z, (const uint64_t*)geno, 0));
  GNXCHK(import2(h, n_ghost, (const gnx_ind_rec*)ghost_rec, nullptr, nullptr, 1));
  return 0;
}

// gamete requests this tile is going to make, per owning rank: births of the pairs whose
// mate is a ghost (the mate's position names its owner)
__global__ void k_req_count(int64_t P, const int32_t* __restrict__ pairs,
                            const uint8_t* __restrict__ ghost, const float* __restrict__ x,
                            const float* __restrict__ y, int tw, int th, int R, int C,
                            int fixed_nb, const int32_t* __restrict__ nbirths,
                            int32_t* __restrict__ cnt, const int32_t* __restrict__ P_dev) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (P_dev) P = *P_dev;
  if (p >= P) return;
  const int m = pairs[2 * p + 1];
  if (!ghost[m]) return;
  const int owner = tile_index(y[m], th, R) * C + tile_index(x[m], tw, C);
  atomicAdd(&cnt[owner], fixed_nb > 0 ? fixed_nb : nbirths[p]);
}

extern "C" int gnx_tile2_pairs(gnx_state* h, int32_t burn, int64_t* counts) {
  if (!h->have_sp) {
    gnx_set_error("species parameters not set");
    return 1;
  }
  if (h->sp.mating_radius < 0) {
    gnx_set_error("panmixia (mating_radius None) on tiles runs through the Python-driven protocol "
                  "(gnx_tile_pairs: every tile holds everybody's record), not through gnx_tile_step");
    return 1;
  }
  GNXCHK(tile2_buffers(h));
  const int T = n_tiles(h);
  for (int p = 0; p < 2 + T; ++p) counts[p] = 0;
  h->req_by_rank.assign(T, 0);
  int64_t P = 0, B = 0;
  // (emigrants leave here; where the sort runs over the id-ordered index - one tile - the columns
  // the mate search does not read follow on the side stream as in gnx_step)
  GNXCHK(gnx_l_sort_by_cell(h, true));
  const bool genomes = !burn && h->cfg.L > 0 && h->genomes_assigned;
  if (h->tile_pairs_nowait && h->sp.n_births_fixed && h->N == 0) {
    // an empty tile: the pair count that travels is a clean zero
    HIPCHK(hipMemsetAsync(h->cnt_dev, 0, sizeof(int32_t), h->stream));
    h->n_pairs = 0;
    h->pairs_wait = false;
    h->n_req_known = 0;
    counts[0] = counts[1] = -1;
    return 0;
  }
  if (h->tile_pairs_nowait && h->sp.n_births_fixed) {
    // gnx_tile_step on several tiles, a fixed number of births per pair: nothing here needs the
    // pair count on the host - the pairs' density bins and the request counts read it on the
    // device (grids sized by the population), and it reaches the host with the count exchange
    // that follows (gnx_tile2_pairs_settle)
    const int rc_enq = gnx_l_find_pairs_enqueue(h, nullptr, false);
    GNXCHK(gnx_wait_permute_rest(h));
    GNXCHK(rc_enq);
    GNXCHK(gnx_l_bins(h, h->N, h->mid_x, h->mid_y, nullptr, h->bins_P, h->cnt_dev));
    h->n_req_known = 0;
    if (T > 1 && genomes) {
      GnxSoA s = h->soa[h->cur];
      if (!h->req_cnt_zeroed)
        HIPCHK(hipMemsetAsync(GNX_REQ_CNT(h), 0, (size_t)T * sizeof(int32_t), h->stream));
      h->req_cnt_zeroed = false;
      hipLaunchKernelGGL(k_req_count, dim3(gnx_grid(h->N, 256)), dim3(256), 0, h->stream, h->N,
                         h->pairs, s.ghost, s.x, s.y, h->cfg.W / h->tile_C, h->cfg.H / h->tile_R,
                         h->tile_R, h->tile_C, (int)h->sp.n_births_lambda, h->nbirths,
                         GNX_REQ_CNT(h), (const int32_t*)h->cnt_dev);
      HIPCHK(hipGetLastError());
      h->n_req_known = -2;
    }
    counts[0] = counts[1] = -1;
    return 0;
  }
  const int rc_fp = gnx_l_find_pairs(h, nullptr, &P);      // wait 2 of the step: the pair count
  GNXCHK(gnx_wait_permute_rest(h));
  GNXCHK(rc_fp);
  GNXCHK(gnx_l_bins(h, P, h->mid_x, h->mid_y, nullptr, h->bins_P));
  GNXCHK(gnx_l_births(h, &B));
  counts[0] = P;
  counts[1] = B;
  h->n_req_known = 0;
  if (P > 0 && T > 1 && genomes) {
    GnxSoA s = h->soa[h->cur];
    // (the request counters were zeroed with the step's other accumulators, k_tile2_zero - or,
    // a caller that skipped gnx_tile2_route_begin, here)
    if (!h->req_cnt_zeroed)
      HIPCHK(hipMemsetAsync(GNX_REQ_CNT(h), 0, (size_t)T * sizeof(int32_t), h->stream));
    h->req_cnt_zeroed = false;
    hipLaunchKernelGGL(k_req_count, dim3(gnx_grid(P, 256)), dim3(256), 0, h->stream, P, h->pairs,
                       s.ghost, s.x, s.y, h->cfg.W / h->tile_C, h->cfg.H / h->tile_R, h->tile_R,
                       h->tile_C, h->sp.n_births_fixed ? (int)h->sp.n_births_lambda : 0,
                       h->nbirths, GNX_REQ_CNT(h), (const int32_t*)nullptr);
    HIPCHK(hipGetLastError());
    if (h->tile_req_on_device) {
      // (gnx_tile_step: the counts travel with its count exchange - gnx_tile2_set_requests)
      h->n_req_known = -2;
      return 0;
    }
    const int seq = (int)(++h->pin_seq & 0x3fffffff);
    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, h->stream, T,
                       (const int32_t*)GNX_REQ_CNT(h), h->h_route_pin_dev, seq);
    HIPCHK(hipGetLastError());
    GNXCHK(tile2_wait(h, seq));                  // (a few microseconds behind wait 2)
    for (int p = 0; p < T; ++p) {
      counts[2 + p] = h->h_route_pin[p];
      h->req_by_rank[p] = h->h_route_pin[p];
      h->n_req_known += h->h_route_pin[p];
    }
  }
  return 0;
}

// gnx_tile_step, a fixed number of births per pair: gnx_tile2_pairs does not wait for the pair
// count (mode 1); it arrives with the count exchange and _settle does the host's bookkeeping
extern "C" int gnx_tile2_pairs_mode(gnx_state* h, int32_t nowait) {
  h->tile_pairs_nowait = nowait != 0;
  return 0;
}

extern "C" int gnx_tile2_pairs_settle(gnx_state* h, int32_t burn, int64_t P, int64_t* births) {
  (void)burn;
  int64_t P_pub = 0, B = 0;
  // (the exchange that carried P waited for the stream: the published count is there)
  GNXCHK(gnx_l_find_pairs_finish(h, &P_pub));
  if (P_pub != P) {
    gnx_set_error("tile2: the pair count that travelled (%lld) is not the one published (%lld)",
                  (long long)P, (long long)P_pub);
    return 1;
  }
  GNXCHK(gnx_l_births(h, &B));
  *births = B;
  return 0;
}

// Tile-major offspring ids through the Python-driven protocol (TiledStepper._step_v2): this
// tile's births per virtual tile (one wait), then the global bases before the births
extern "C" int gnx_tile2_vt_counts(gnx_state* h, int64_t* counts) {
  if (h->id_order != 1) {
    gnx_set_error("gnx_tile2_vt_counts: offspring ids are not tile-major (gnx_set_id_order)");
    return 1;
  }
  GNXCHK(gnx_l_pair_cls(h, h->n_pairs, n_tiles(h) == 1));
  int32_t tmp[64];
  GNXCHK(gnx_d2h(h, tmp, h->vt_count, sizeof(tmp)));
  for (int q = 0; q < 64; ++q) counts[q] = (int64_t)tmp[q] * h->vt_mul;
  return 0;
}

extern "C" int gnx_tile2_vt_bases(gnx_state* h, const int64_t* bases) {
  return gnx_h2d(h, h->vt_base, bases, 64 * sizeof(int64_t));
}

// gnx_tile_step: the request counts gnx_tile2_pairs left on the device (*counts_dev, int32 [T])
// have reached the host with the count exchange
extern "C" int gnx_tile2_requests_dev(gnx_state* h, int32_t on, void** counts_dev) {
  h->tile_req_on_device = on != 0;
  if (counts_dev) {
    GNXCHK(tile2_buffers(h));
    *counts_dev = GNX_REQ_CNT(h);
  }
  return 0;
}

extern "C" int gnx_tile2_set_requests(gnx_state* h, const int64_t* req /*[T]*/) {
  const int T = n_tiles(h);
  h->req_by_rank.assign(T, 0);
  h->n_req_known = 0;
  for (int p = 0; p < T; ++p) {
    h->req_by_rank[p] = req[p];
    h->n_req_known += req[p];
  }
  return 0;
}

__global__ void k_pack_requests2(int64_t n, const int32_t* idx, const int64_t* pid,
                                 const int32_t* key, const uint8_t* start, const float* px,
                                 const float* py, const int32_t* child_k, gnx_gamete_req2* out,
                                 int32_t* k_out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int i = idx[q];
  gnx_gamete_req2 r;
  r.parent_id = pid[i];
  r.key = key[i];
  r.start = start[i];
  r.px = px[i];
  r.py = py[i];
  out[q] = r;
  k_out[q] = child_k[i];
}

extern "C" int gnx_tile2_offspring(gnx_state* h, int32_t burn, int64_t id_base,
                                   const void* pair_goff_dev, void** req_dev) {
  *req_dev = nullptr;
  int64_t P = h->n_pairs, B = 0;
  h->birth_first_slot = h->N;
  h->n_req = 0;
  h->tile_births_settled = false;
  if (P > 0 && pair_goff_dev && pair_goff_dev != (const void*)h->pair_goff)
    HIPCHK(hipMemcpyAsync(h->pair_goff, pair_goff_dev, P * sizeof(int64_t),
                          hipMemcpyDeviceToDevice, h->stream));
  // (no offsets handed in - one tile: the pairs' global offsets are their local ones)
  h->pair_goff_local = pair_goff_dev == nullptr;
  int rc = gnx_l_mate(h, burn != 0, false, 0, &B, id_base, true);
  h->pair_goff_local = false;
  h->n_req_known = -1;
  GNXCHK(rc);
  h->last_births = B;
  if (B == 0) h->n_req = 0;
  h->xo_pending = false;                    // (tile2 serves on the main stream)
  const int64_t n = h->n_req;
  if (n == 0) return 0;
  if (n > h->rq_cap) {
    (void)hipFree(h->rq_sorted);
    (void)hipFree(h->rq_k);
    h->rq_cap = n + n / 4 + 1024;
    gnx_gamete_req2* p = nullptr;           // (24-byte records fit the 16-byte ones' buffer too)
    GNXCHK(dalloc_t(&p, (size_t)h->rq_cap));
    h->rq_sorted = p;
    GNXCHK(dalloc_t(&h->rq_k, (size_t)h->rq_cap));
    h->rq_is2 = true;
  } else if (!h->rq_is2) {                  // sized for 16-byte records by the other protocol
    (void)hipFree(h->rq_sorted);
    gnx_gamete_req2* p = nullptr;
    GNXCHK(dalloc_t(&p, (size_t)h->rq_cap));
    h->rq_sorted = p;
    h->rq_is2 = true;
  }
  // grouped by the rank that owns the ghost mate (counts: gnx_tile2_pairs, on the host)
  hipLaunchKernelGGL(k_req_owner, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, h->req_px,
                     h->req_py, h->cfg.W / h->tile_C, h->cfg.H / h->tile_R, h->tile_R, h->tile_C,
                     h->key[0], h->perm[0]);
  int bits = 1;
  while ((1 << bits) < n_tiles(h)) ++bits;
  GNXCHK(gnx_prim_sort(h->sort_tmp, h->sort_tmp_bytes, h->key[0], h->key[1], h->perm[0],
                       h->perm[1], (size_t)n, bits, h->stream));
  hipLaunchKernelGGL(k_pack_requests2, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n,
                     h->perm[1], h->req_pid, h->req_key, h->req_start, h->req_px, h->req_py,
                     h->req_k, (gnx_gamete_req2*)h->rq_sorted, h->rq_k);
  HIPCHK(hipGetLastError());
  *req_dev = h->rq_sorted;
  return 0;
}

// the requested parent's slot: the population is sorted by (hash cell, id), so the
// parent's position names its cell and a binary search inside the cell finds the id
// (round 2: a 64-bit sort of every id of the tile per step)
__global__ void k_lookup_req2(int64_t n, const gnx_gamete_req2* __restrict__ req, int n_paths,
                              double inv_cs, int ncx, int ncy,
                              const int32_t* __restrict__ cell_start,
                              const int64_t* __restrict__ id, const uint8_t* __restrict__ ghost,
                              int32_t* __restrict__ out_slot, int64_t* __restrict__ chk) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const gnx_gamete_req2 r = req[q];
  const int cx = min(ncx - 1, max(0, (int)((double)r.px * inv_cs)));
  const int cy = min(ncy - 1, max(0, (int)((double)r.py * inv_cs)));
  const int c = cy * ncx + cx;
  int lo = cell_start[c], hi = cell_start[c + 1];
  const int end = hi;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (id[mid] < r.parent_id) lo = mid + 1; else hi = mid;
  }
  const bool ok = lo < end && id[lo] == r.parent_id && !ghost[lo] && r.key >= 0 &&
                  r.key < n_paths && (r.start == 0 || r.start == 1);
  out_slot[q] = ok ? lo : 0;
  if (!ok) atomicAdd((unsigned long long*)&chk[1], 1ull);
}

__global__ void __launch_bounds__(256)
k_make_gametes_req2(int64_t n, int W16, const u64x2* __restrict__ G,
                    const int32_t* __restrict__ grow, GnxHalves H,
                    const int32_t* __restrict__ slot, const gnx_gamete_req2* __restrict__ req,
                    const u64x2* __restrict__ paths, u64x2* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= n) return;
  const int prow = grow[slot[q]];
  const gnx_gamete_req2 r = req[q];
  const u64 s = r.start ? ~0ull : 0ull;
  const int64_t lh0 = (int64_t)max(prow, 0) * 2;
  const int key = min(max(r.key, 0), 0x7fffffff);
  const u64x2* pm = paths + (int64_t)key * W16;
  for (int c = lane; c < W16; c += 64) {
    u64x2 m = pm[c];
    m.a ^= s;
    m.b ^= s;
    const u64x2 a = G[gnx_chunk_at(H, lh0, c)], b = G[gnx_chunk_at(H, lh0 + 1, c)];
    u64x2 o;
    o.a = (a.a & ~m.a) | (b.a & m.a);
    o.b = (a.b & ~m.b) | (b.b & m.b);
    out[q * W16 + c] = o;
  }
}

extern "C" int gnx_tile2_serve(gnx_state* h, int64_t n, const void* req_dev, void** out_dev) {
  *out_dev = nullptr;
  if (n == 0) return 0;
  if (!has_rows(h) || h->n_paths == 0) {
    gnx_set_error("gnx_tile2_serve: genomes / paths not set");
    return 1;
  }
  GNXCHK(tile2_buffers(h));
  if (n > h->gam_cap) {
    (void)hipFree(h->gam_out);
    (void)hipFree(h->gam_slot);
    h->gam_cap = n + n / 4 + 256;
    GNXCHK(dalloc_t(&h->gam_out, (size_t)h->gam_cap * h->W64));
    GNXCHK(dalloc_t(&h->gam_slot, (size_t)h->gam_cap + 1));
  }
  // last step's deferred crossover wrote blocks of individuals that are parents now
  GNXCHK(gnx_xo_launch_pending(h));
  GNXCHK(gnx_xo_wait_inflight(h));
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL(k_lookup_req2, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n,
                     (const gnx_gamete_req2*)req_dev, h->n_paths, h->inv_cs, h->ncx, h->ncy,
                     h->cell_start, s.id, s.ghost, h->gam_slot, h->chk);
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_make_gametes_req2, dim3(gnx_grid(n * 64, 256)), dim3(256), 0, h->stream, n,
                     W16, (const u64x2*)h->G, s.grow, gnx_halves(h), h->gam_slot,
                     (const gnx_gamete_req2*)req_dev, (const u64x2*)h->paths, (u64x2*)h->gam_out);
  HIPCHK(hipGetLastError());
  *out_dev = h->gam_out;
  return 0;
}

extern "C" int gnx_tile2_put(gnx_state* h, int64_t n, const void* data_dev) {
  if (n == 0) return 0;
  if (n != h->n_req) {
    gnx_set_error("gnx_tile2_put: %lld gametes for %lld requests", (long long)n,
                  (long long)h->n_req);
    return 1;
  }
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_put_gametes, dim3(gnx_grid(n * W16, 256, 256 * 32)), dim3(256), 0, h->stream,
                     n, W16, (const u64x2*)data_dev, (u64x2*)h->G, h->soa[h->cur].grow,
                     gnx_halves(h), h->birth_first_slot, h->rq_k);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int gnx_tile2_finish_births(gnx_state* h, int32_t burn, void** reduce_dev,
                                       int64_t* n_words) {
  const int64_t nb = (int64_t)h->lat.nbx * h->lat.nby;
  // this tile's own individuals (ghosts are resident), births, deaths of the previous step:
  // the kernel that counts the individuals leaves them behind the bins
  GnxSetWords sw;
  sw.dst = h->bin_partials + 2 * nb;
  sw.v[0] = (int32_t)(h->N - h->n_ghost);
  sw.v[1] = (int32_t)h->last_births;
  sw.v[2] = (int32_t)h->prev_deaths;
  sw.v[3] = 0;
  GNXCHK(tile_finish_births(h, burn, &sw));
  *reduce_dev = h->bin_partials;
  *n_words = 2 * nb + 4;
  return 0;
}

extern "C" int gnx_tile2_die(gnx_state* h, int32_t burn, int32_t with_selection,
                             int32_t have_pairs, int64_t* totals) {
  GNXCHK(tile2_buffers(h));
  const int64_t nb = (int64_t)h->lat.nbx * h->lat.nby;
  // the reduced counter words and the deferred checks ride to the host behind the
  // mortality's own wait (wait 3 of the step)
  GnxPubWords pub;
  pub.n1 = 4;
  pub.src1 = h->bin_partials + 2 * nb;
  pub.host1 = h->h_route_pin_dev;
  pub.n2 = 4;
  pub.src2 = (const int32_t*)h->chk;
  pub.host2 = h->h_route_pin_dev + 8;
  if (!gnx_l_lattices_tiled(h, have_pairs != 0, pub)) {
    hipLaunchKernelGGL(k_publish_words2, dim3(1), dim3(64), 0, h->stream, 4,
                       (const int32_t*)(h->bin_partials + 2 * nb), h->h_route_pin_dev, 4,
                       (const int32_t*)h->chk, h->h_route_pin_dev + 8);
    if (have_pairs)
      GNXCHK(gnx_l_spline(h, h->bins_P, &h->spl_P, nullptr));
    else
      h->spl_P.valid = false;
    GNXCHK(gnx_l_spline(h, h->bin_partials, &h->spl_N, nullptr));
  }
  GNXCHK(gnx_l_death_probs(h, with_selection != 0 && !burn));
  const int64_t N_before = h->N;
  int64_t D = 0;
  GNXCHK(gnx_l_mortality(h, nullptr, &D));
  h->last_deaths = D;
  h->prev_deaths = D;
  // (the words were published before the kernels the mortality waited for - unless this tile
  // is empty: gnx_l_death_probs and gnx_l_mortality return at once for N == 0, a legitimate
  // state of one tile of many, and nothing has waited for the stream yet)
  if (N_before == 0) HIPCHK(hipStreamSynchronize(h->stream));
  std::atomic_thread_fence(std::memory_order_acquire);
  for (int k = 0; k < 3; ++k) totals[k] = h->h_route_pin[k];
  const int64_t bad_rec = (int64_t)(uint32_t)h->h_route_pin[8] | ((int64_t)h->h_route_pin[9] << 32);
  const int64_t bad_req = (int64_t)(uint32_t)h->h_route_pin[10] | ((int64_t)h->h_route_pin[11] << 32);
  if (bad_rec || bad_req) {
    gnx_set_error("tile2: %lld imported records off the landscape, %lld gamete requests for a "
                  "parent that does not live on this tile (or a path / start out of range)",
                  (long long)bad_rec, (long long)bad_req);
    return 1;
  }
  return 0;
}

extern "C" int gnx_tile_pair_ptrs_nosync(gnx_state* h, int64_t* n_pairs, void** focal_ids,
                                         void** n_births) {
  const int64_t P = h->n_pairs;
  *n_pairs = P;
  *focal_ids = P ? (void*)h->key64[0] : nullptr;
  *n_births = (P && !h->sp.n_births_fixed) ? (void*)h->nbirths : nullptr;
  return 0;
}
