// The device-driven step: gnx_walk (include/gnx_hip.h).
//
// gnx_step (gnx_api.hip) keeps the population's counts on the host: two read-backs and ~45
// runtime calls per step.  For the metric workload (10^6 individuals) the host is not what the
// step waits for; for BASELINE configs[1] and [2] (10^5 individuals) it is - the kernels of a
// step take ~150 us on the GPU and the host needs 185 us to enqueue them.  Here the counts live
// in a device block (GnxDD, gnx_internal.h), every kernel reads what it needs from it, grids
// are sized by the handle's capacity, and nothing is read back: one step = one HIP graph,
// captured once per (buffer parity, burn, selection) and replayed - one runtime call per step.
// The same kernels, the same draws (keyed by id and step), the same canonical orders: a
// population walked this way equals the one gnx_step produces, id by id
// (tests/test_gpu_deferred.py::test_small_path_equals_default_path).
//
// Reference: the loop this replaces is Model.walk -> _do_timestep over the function queue
// (sim/model.py:603-667, 699-744, 966-1161), T times.
#include <atomic>
#include <chrono>
#include <map>
#include "gnx_internal.h"

bool gnx_fused_bins(const gnx_state* h);

namespace {
using GraphMap = std::map<uint32_t, hipGraphExec_t>;

struct DDExtra {             // host-side state of the device-driven mode
  GraphMap graphs;
  hipEvent_t ev[8]{};
  bool have_ev = false;
  // Streams the step is CAPTURED on (the graph is launched on the handle's own stream).  They
  // are non-blocking: while a blocking stream captures, any hipMemcpy on the legacy stream -
  // another handle's, on another host thread - fails ("would make the legacy stream depend on
  // a capturing blocking stream").
  hipStream_t cap[3]{};
  bool capturing = false;
  uint64_t epoch = 0;        // h's configuration epoch the graphs were captured under
};

DDExtra* extra(gnx_state* h) {
  if (!h->dd_graph[0]) h->dd_graph[0] = new DDExtra();
  return (DDExtra*)h->dd_graph[0];
}

bool env_on(const char* name, bool dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) != 0 : dflt;
}
}  // namespace

unsigned gnx_order_event_flags();

bool gnx_dd_eligible(const gnx_state* h, bool burn) {
  // GNX_DD=0: never; 2: whatever the size.  By default populations whose step the host cannot
  // enqueue as fast as the GPU runs it (capacity up to GNX_DD_MAX_CAP slots, 600 000): at 10^6
  // individuals the host-driven step is the faster one - its crossover runs under the next
  // step's movement and its grids fit the population (0.60 against 0.70 ms per step)
  const int mode = getenv("GNX_DD") ? atoi(getenv("GNX_DD")) : 1;
  static const int64_t max_cap = getenv("GNX_DD_MAX_CAP") ? atoll(getenv("GNX_DD_MAX_CAP")) : 600000;
  if (mode == 0 || !h->have_sp || h->tiled || h->profiling) return false;
  if (mode != 2 && h->cfg.cap_inds > max_cap) return false;
  const gnx_species_params& sp = h->sp;
  if (sp.mating_radius < 0 || !sp.n_births_fixed || !sp.move) return false;
  if (!h->ord_mode || h->key_bits > 24 || !h->compact_fill || !h->defer_xo) return false;
  if (!h->stream2 || !h->stream3 || !gnx_fused_bins(h)) return false;
  // (tile-major offspring ids - gnx_set_id_order(1), the Model API's default - ride along: the
  // classification inside k_pair_compact, one more node for the blocks' offsets)
  if (h->NB > GNX_MAX_NB || h->n_ghost != 0) return false;
  if (h->cfg.cap_inds >= (1ll << 30)) return false;
  // (a launch policy other than the handle's own default was asked for: the host-driven step honours it)
  if ((h->xo_launch_policy != 0 && h->xo_launch_policy != h->xo_launch_default) || h->xo_split != 0)
    return false;
  (void)burn;
  return true;
}

static int dd_events(gnx_state* h) {
  DDExtra* x = extra(h);
  if (x->have_ev) return 0;
  for (int k = 0; k < 8; ++k) HIPCHK(hipEventCreateWithFlags(&x->ev[k], gnx_order_event_flags()));
  for (int k = 0; k < 3; ++k) HIPCHK(hipStreamCreateWithFlags(&x->cap[k], hipStreamNonBlocking));
  x->have_ev = true;
  return 0;
}

// one step, enqueued on the handle's three streams (inside a stream capture: the body of the
// step's graph).  Host state only flips its buffer parities.
static int dd_enqueue_step(gnx_state* h, bool burn, bool sel) {
  DDExtra* x = extra(h);
  const gnx_config& c = h->cfg;
  // The whole step on ONE stream: a linear graph replays at ~2 us per node, one with side
  // branches at ~30 (ROCm 7.2, tools/launch_micro.hip and DESIGN 4.4: 0.21 against 1.0 ms per
  // step at 10^5 individuals) - and an event hand-over costs the host 12 us where a launch
  // costs 3.  GNX_DD_STREAMS=3: the densities and the lists on the side streams as in gnx_step.
  const bool one = !(getenv("GNX_DD_STREAMS") && atoi(getenv("GNX_DD_STREAMS")) == 3);
  hipStream_t s1 = x->capturing ? x->cap[0] : h->stream;
  hipStream_t s2 = one ? s1 : (x->capturing ? x->cap[1] : h->stream2);
  hipStream_t s3 = one ? s1 : (x->capturing ? x->cap[2] : h->stream3);
  hipStream_t own = h->stream;
  h->stream = s1;                 // (the launchers that take no stream argument)
  struct Restore {
    gnx_state* h;
    hipStream_t s;
    ~Restore() { h->stream = s; }
  } restore{h, own};
  const bool genomes = !burn && c.L > 0 && h->genomes_assigned;
  const bool xo = genomes;
  const int has_rows = (h->genomes_assigned && c.L > 0) ? 1 : 0;
  const int par = h->fb_cur;
  const int buf = h->jobs_cur;
  // age, movement, environment, hash cells (Species._set_age_stage, _do_movement)
  h->move_writes_keys = true;
  int rc = gnx_l_move(h, true, nullptr, nullptr, nullptr, nullptr, true);
  h->move_writes_keys = false;
  GNXCHK(rc);
  // mating pairs over the cell-sorted population; the adults' density bins beside the search
  // (one stream: the two fields' bins are counted by k_permute and k_pair_compact themselves -
  // two launches of ~5 us less on a chain of ~20; GNX_DD_FUSE_BINS=0: launches of their own)
  const bool fuse_bins = one && !(getenv("GNX_DD_FUSE_BINS") && atoi(getenv("GNX_DD_FUSE_BINS")) == 0);
  GNXCHK(gnx_dd_l_sort(h, fuse_bins ? h->fb[par] : nullptr, s1));
  HIPCHK(hipEventRecord(x->ev[0], s1));
  HIPCHK(hipStreamWaitEvent(s3, x->ev[0], 0));
  if (!fuse_bins) GNXCHK(gnx_dd_l_bins_adults(h, par, s3));
  GNXCHK(gnx_dd_l_pairs(h, fuse_bins ? h->fb[2] : nullptr, s1));
  // the pairs' density beside the births
  HIPCHK(hipEventRecord(x->ev[1], s1));
  HIPCHK(hipStreamWaitEvent(s3, x->ev[1], 0));
  if (!fuse_bins) GNXCHK(gnx_dd_l_density_pairs(h, s3));
  HIPCHK(hipEventRecord(x->ev[2], s3));
  GNXCHK(gnx_dd_l_offspring(h, genomes, h->fb[par], s1));
  // densities, death probabilities, death draws
  HIPCHK(hipStreamWaitEvent(s1, x->ev[2], 0));
  GNXCHK(gnx_dd_l_density_N(h, par, s1));
  GNXCHK(gnx_dd_l_death_probs(h, sel && !burn, par, s1));
  GNXCHK(gnx_dd_l_alive(h, xo, buf, s1));
  // the compaction's lists beside the crossover's job builder; the crossover beside the
  // compaction and the index's own compaction
  HIPCHK(hipEventRecord(x->ev[3], s1));
  HIPCHK(hipStreamWaitEvent(s3, x->ev[3], 0));
  GNXCHK(gnx_dd_l_fill_lists(h, has_rows, s3));
  HIPCHK(hipEventRecord(x->ev[4], s3));
  if (xo) {
    GNXCHK(gnx_dd_l_jobs(h, buf, s1));
    HIPCHK(hipEventRecord(x->ev[5], s1));
    HIPCHK(hipStreamWaitEvent(s2, x->ev[5], 0));
    GNXCHK(gnx_dd_l_crossover(h, buf, s2));
    HIPCHK(hipEventRecord(x->ev[6], s2));
  }
  HIPCHK(hipStreamWaitEvent(s1, x->ev[4], 0));
  GNXCHK(gnx_dd_l_fill(h, has_rows, xo, s1));
  if (xo) HIPCHK(hipStreamWaitEvent(s1, x->ev[6], 0));
  GNXCHK(gnx_dd_l_ord_end(h, has_rows, xo, s1));
  if (xo) h->jobs_cur ^= 1;
  h->fb_cur ^= 1;
  return 0;
}

// what a replayed graph would have flipped on the host
static void dd_flip(gnx_state* h, bool burn) {
  const bool xo = !burn && h->cfg.L > 0 && h->genomes_assigned;
  h->cur ^= 1;                  // the cell sort (the compaction is in place)
  if (xo) h->jobs_cur ^= 1;
  h->fb_cur ^= 1;               // (ord_cur flips twice per step)
}

static uint32_t dd_key(const gnx_state* h, bool burn, bool sel, int k) {
  return (uint32_t)h->cur | (uint32_t)h->ord_cur << 1 | (uint32_t)h->jobs_cur << 2 |
         (uint32_t)h->fb_cur << 3 | (burn ? 16u : 0u) | (sel ? 32u : 0u) |
         (h->genomes_assigned ? 64u : 0u) | (uint32_t)k << 8;
}

static void dd_drop_graphs(gnx_state* h) {
  DDExtra* x = extra(h);
  for (auto& kv : x->graphs) (void)hipGraphExecDestroy(kv.second);
  x->graphs.clear();
}

// k consecutive steps in ONE graph (between two replays the GPU idles ~9 us - the start of a
// graph - whatever it holds: four steps per graph share that)
static int dd_launch_step(gnx_state* h, bool burn, bool sel, int k) {
  const bool use_graph = env_on("GNX_DD_GRAPH", true);
  if (!use_graph) {
    for (int q = 0; q < k; ++q) GNXCHK(dd_enqueue_step(h, burn, sel));
    return 0;
  }
  DDExtra* x = extra(h);
  const uint32_t key = dd_key(h, burn, sel, k);
  auto it = x->graphs.find(key);
  if (it == x->graphs.end()) {
    const auto tc0 = std::chrono::steady_clock::now();
    hipGraph_t g = nullptr;
    const int cur0 = h->cur, ord0 = h->ord_cur, jobs0 = h->jobs_cur, fb0 = h->fb_cur;
    HIPCHK(hipStreamBeginCapture(x->cap[0], hipStreamCaptureModeThreadLocal));
    x->capturing = true;
    int rc = 0;
    for (int q = 0; q < k && !rc; ++q) rc = dd_enqueue_step(h, burn, sel);
    x->capturing = false;
    hipError_t e = hipStreamEndCapture(x->cap[0], &g);
    // (the capture enqueued nothing: the host's parities are those before it)
    h->cur = cur0;
    h->ord_cur = ord0;
    h->jobs_cur = jobs0;
    h->fb_cur = fb0;
    if (rc) {
      if (g) (void)hipGraphDestroy(g);
      return rc;
    }
    if (e != hipSuccess) {
      gnx_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
      return 1;
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
      gnx_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
      return 1;
    }
    it = x->graphs.emplace(key, ex).first;
    if (env_on("GNX_DD_DEBUG", false))
      fprintf(stderr, "[gnx dd] graph %u captured and instantiated in %.2f ms\n", key,
              1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count());
  }
  HIPCHK(hipGraphLaunch(it->second, h->stream));
  for (int q = 0; q < k; ++q) dd_flip(h, burn);
  return 0;
}

// records the steps have left in pinned memory, in order, as far as they have arrived
static void dd_consume(gnx_state* h) {
  while (h->dd_seen < h->dd_seq) {
    const GnxDDRec* r = h->dd_ring + (h->dd_seen % GNX_DD_RING);
    const int64_t want = h->dd_seen + 1;
    if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) != want) break;
    const int64_t deaths = (int64_t)r->N0 + r->B - r->S;
    h->tot[0] += 1;
    h->tot[1] += r->N0;
    h->tot[2] += r->B;
    h->tot[3] += deaths;
    h->tot[4] += r->xo;
    h->tot[5] += 1;                             // steps taken the device-driven way
    h->last_births = r->B;
    h->last_deaths = deaths;
    h->last_xo_births = r->xo;
    h->n_pairs = r->P;
    h->fill_guess = deaths;
    h->dd_b_hi = std::max<int64_t>(h->dd_b_hi, r->B);
    // (blocks only come back through the collector: between two consecutive records with no
    // collection in between the stack's height falls by what the step took)
    if (h->dd_top_seq == want - 1 && h->dd_gc_seq != want - 1 && h->dd_top_seq > 0)
      h->dd_use_hi = std::max<int64_t>(h->dd_use_hi, h->dd_top_last - r->half_top);
    if (h->dd_gc_seq == want - 1 && h->dd_gc_seq > 0)
      h->dd_post_gc = r->half_top + h->dd_use_hi;         // the height that collection left
    h->dd_top_last = r->half_top;
    h->dd_top_seq = want;
    // (a step that ran before the last collection tells nothing about the stack behind it)
    if (want > h->dd_gc_seq) {
      h->dd_half_est = r->half_top;
      h->dd_est_seq = want;
    }
    if (r->err) h->dd_err |= r->err;            // (sticky; looked at by dd_check)
    // a step that reported an error dropped births or shared genome rows, and every step
    // enqueued behind it started from that state: none of them enters the walk's history
    // (gnx_walk_history then ends with the last step that completed cleanly)
    if (!h->dd_err && !h->dd_hist_closed) {
      h->dd_hist.push_back(r->N0);
      h->dd_hist.push_back(r->B);
      h->dd_hist.push_back(deaths);
    } else {
      h->dd_hist_closed = true;
    }
    h->dd_seen = want;
  }
}

static int dd_check(gnx_state* h) {
  const int64_t err = h->dd_err;
  if (!err) return 0;
  h->dd_err = 0;
  if (err & GNX_DD_ERR_SLOTS)
    gnx_set_error("capacity exceeded: the population and its births outgrew cap_inds=%lld",
                  (long long)h->cfg.cap_inds);
  else if (err & GNX_DD_ERR_ROWS)
    gnx_set_error("capacity exceeded: surviving offspring outnumbered the free genome rows");
  else
    gnx_set_error("the stack of free genome blocks ran dry (device-driven step)");
  return 2;
}

static int dd_wait_all_seen(gnx_state* h) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int spin = 0; h->dd_seen < h->dd_seq; ++spin) {
    dd_consume(h);
    if ((spin & 255) == 255 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
      HIPCHK(hipStreamSynchronize(h->stream));
      dd_consume(h);
      if (h->dd_seen < h->dd_seq) {
        gnx_set_error("device-driven step: a step's record never arrived");
        return 1;
      }
    }
    __builtin_ia32_pause();
  }
  return 0;
}

static int dd_enter(gnx_state* h) {
  const auto te0 = std::chrono::steady_clock::now();
  struct Report {
    std::chrono::steady_clock::time_point t0;
    ~Report() {
      if (env_on("GNX_DD_DEBUG", false))
        fprintf(stderr, "[gnx dd] enter took %.2f ms\n",
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
  } report{te0};
  GNXCHK(gnx_xo_join(h));
  GNXCHK(gnx_wait_permute_rest(h));
  gnx_bins_adults_drop(h);
  HIPCHK(hipStreamSynchronize(h->stream3));
  HIPCHK(hipStreamSynchronize(h->stream2));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->latP_inflight = h->binsN_inflight = h->ord_inflight = false;
  GNXCHK(dd_events(h));
  if (!h->dd) {
    HIPCHK(hipMalloc((void**)&h->dd, sizeof(GnxDD)));
    HIPCHK(hipHostMalloc((void**)&h->dd_ring, GNX_DD_RING * sizeof(GnxDDRec),
                         hipHostMallocCoherent | hipHostMallocMapped));
    memset(h->dd_ring, 0, GNX_DD_RING * sizeof(GnxDDRec));
    HIPCHK(hipHostGetDevicePointer((void**)&h->dd_ring_dev, h->dd_ring, 0));
  }
  memset(h->dd_ring, 0, GNX_DD_RING * sizeof(GnxDDRec));
  GnxDD d{};
  d.N = (int32_t)h->N;
  d.n_free = (int32_t)h->n_free;
  d.ord_n = (int32_t)h->N;
  d.max_id = h->max_id;
  d.step = h->step;
  HIPCHK(hipMemcpy(h->dd, &d, sizeof(d), hipMemcpyHostToDevice));
  const size_t nbins = (size_t)h->lat.nbx * h->lat.nby * sizeof(int32_t);
  for (int k = 0; k < 3; ++k) HIPCHK(hipMemset(h->fb[k], 0, nbins));
  HIPCHK(hipMemset(h->nmax2, 0, 2 * sizeof(unsigned long long)));
  h->dd_half_est = 0;
  if (h->half_top) {
    int32_t top = 0;
    HIPCHK(hipMemcpy(&top, h->half_top, sizeof(top), hipMemcpyDeviceToHost));
    h->dd_half_est = top;
  }
  h->dd_seq = h->dd_seen = 0;
  h->dd_b_hi = std::max<int64_t>(h->last_births, h->N / 4);
  h->dd_use_hi = 0;
  h->dd_gc_seq = 0;
  h->dd_est_seq = 0;
  h->dd_top_seq = 0;
  h->dd_top_last = 0;
  h->dd_post_gc = 0;
  h->dd_gc_wait = false;
  h->dd_err = 0;
  h->dd_active = true;
  return 0;
}

int gnx_dd_leave(gnx_state* h) {
  if (!h->dd_active) return 0;
  if (env_on("GNX_DD_DEBUG", false))
    fprintf(stderr, "[gnx dd] leave: %lld steps enqueued, %lld seen, collections so far %lld\n",
            (long long)h->dd_seq, (long long)h->dd_seen, (long long)h->gc_runs);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream2));
  HIPCHK(hipStreamSynchronize(h->stream3));
  dd_consume(h);
  GnxDD d{};
  HIPCHK(hipMemcpy(&d, h->dd, sizeof(d), hipMemcpyDeviceToHost));
  h->N = d.N;
  h->n_free = d.n_free;
  h->max_id = d.max_id;
  h->step = d.step;
  h->ord_n = d.ord_n;
  h->ord_valid = true;
  h->n_ghost = 0;
  if (h->half_top) {
    int32_t top = 0;
    HIPCHK(hipMemcpy(&top, h->half_top, sizeof(top), hipMemcpyDeviceToHost));
    h->half_free_est = top;
  }
  h->xo_deferred = false;
  h->xo_ready_buf = -1;
  h->xo_running = false;
  for (int k = 0; k < 2; ++k) h->xo_inflight[k] = h->xo_wide_inflight[k] = false;
  h->keys_fresh = false;
  GNXCHK(gnx_os_hist_discard(h));       // (the device-driven sort counts its own digits: k_keys_hist)
  h->fb_adults = h->fb_pending = false;
  // the last step read fb[fb_cur ^ 1] and cleared fb[fb_cur] and the pairs' bins
  h->fb_zero[h->fb_cur] = true;
  h->fb_zero[h->fb_cur ^ 1] = false;
  h->fb_zero[2] = true;
  h->nmax_cur = h->nmax2 + (h->fb_cur ^ 1);
  h->nmax_ready = false;
  h->last_N_fused = true;
  h->spl_N.valid = h->spl_P.valid = d.seq > 0 ? true : h->spl_N.valid;
  h->pairs_wait = h->mort_wait = false;
  h->n_births_pending = 0;
  h->perm_rest_pending = h->perm_rest_inflight = false;
  h->dd_active = false;
  return dd_check(h);
}

void gnx_dd_destroy(gnx_state* h) {
  if (h->dd_graph[0]) {
    dd_drop_graphs(h);
    DDExtra* x = extra(h);
    if (x->have_ev) {
      for (int k = 0; k < 8; ++k) (void)hipEventDestroy(x->ev[k]);
      for (int k = 0; k < 3; ++k) (void)hipStreamDestroy(x->cap[k]);
    }
    delete x;
    h->dd_graph[0] = nullptr;
  }
  if (h->dd) (void)hipFree(h->dd);
  if (h->dd_ring) (void)hipHostFree(h->dd_ring);
  h->dd = nullptr;
  h->dd_ring = nullptr;
}

// The free-block stack holds what the next step can take at most; else the collector is
// enqueued in front of it (gnx_gc: behind the steps enqueued so far, nobody waits for it).
// The host reasons about the stack's height at the TAIL of its queue: the height the last
// record reported, less what the steps enqueued since can have taken.
static int64_t dd_per_step(const gnx_state* h) {
  // what a step takes: every block of every birth's two gametes at most; once steps have been
  // observed, twice the most any of them took (a step takes one block per switch point of its
  // surviving offspring's gametes - a sum of ~10^4 .. 10^5 draws; should a step ever want more
  // than the stack holds, its kernels stay inside the stack and the walk ends with an error)
  const int64_t worst = 2ll * h->NB * (h->dd_b_hi + h->dd_b_hi / 4 + 1024);
  return h->dd_use_hi > 0 ? std::min(worst, 2 * h->dd_use_hi + 4096) : worst;
}

// how many of the next `want` steps the stack of free blocks is certain to carry without a
// collection (0: the collector has to run first)
static int64_t dd_steps_in_stock(const gnx_state* h, bool burn, int64_t want) {
  if (burn || !h->half_top || !h->genomes_assigned || h->cfg.L <= 0) return want;
  const int64_t per_step = dd_per_step(h);
  const int64_t since = h->dd_seq - std::max(h->dd_est_seq, h->dd_gc_seq);
  const int64_t room = (h->dd_half_est - since * per_step) / per_step - 1;
  return std::max<int64_t>(0, std::min(want, room));
}

static int dd_blocks(gnx_state* h, bool burn) {
  if (burn || !h->half_top || !h->genomes_assigned || h->cfg.L <= 0) return 0;
  const int64_t per_step = dd_per_step(h);
  // steps enqueued behind the state dd_half_est describes (a record's, or a collection's)
  const int64_t since = h->dd_seq - std::max(h->dd_est_seq, h->dd_gc_seq);
  if (h->dd_half_est - since * per_step >= 2 * per_step) return 0;
  GNXCHK(gnx_gc(h));
  h->dd_gc_seq = h->dd_seq;
  if (h->dd_post_gc > 0) {
    // (the living change slowly: about what the last collection left, less a tenth)
    h->dd_half_est = h->dd_post_gc - h->dd_post_gc / 10;
  } else {
    // the first collection of this walk: the step behind it reports the height before the host
    // enqueues any further (one wait, once)
    h->dd_half_est = 1ll << 50;
    h->dd_gc_wait = true;
  }
  return 0;
}

// up to `most` steps; *taken = how many were enqueued (GNX_DD_STEPS_PER_GRAPH at a time, 4 by
// default, while the free blocks are certain to last; one otherwise)
static int dd_some(gnx_state* h, bool burn, bool sel, int64_t most, int64_t* taken) {
  static const int per_graph =
      std::max(1, std::min(16, getenv("GNX_DD_STEPS_PER_GRAPH") ? atoi(getenv("GNX_DD_STEPS_PER_GRAPH")) : 4));
  dd_consume(h);
  GNXCHK(dd_check(h));
  GNXCHK(dd_blocks(h, burn));
  // (the ring holds GNX_DD_RING records: never run further ahead of the device than that)
  if (h->dd_seq - h->dd_seen > GNX_DD_RING / 4) GNXCHK(dd_wait_all_seen(h));
  int k = 1;
  if (!h->dd_gc_wait && most >= per_graph && dd_steps_in_stock(h, burn, per_graph) >= per_graph)
    k = per_graph;
  GNXCHK(dd_launch_step(h, burn, sel, k));
  h->dd_seq += k;
  *taken = k;
  if (h->dd_gc_wait) {
    h->dd_gc_wait = false;
    GNXCHK(dd_wait_all_seen(h));
  }
  return 0;
}

extern double g_host_step_s;
extern long long g_host_steps;

// a handle that cannot take device-driven steps (or is not in a steady state yet) walks
// through gnx_step
static int walk_prepare(gnx_state* h, int64_t* T, bool burn, bool sel, bool* dd_ok) {
  *dd_ok = gnx_dd_eligible(h, burn) && *T > 0;
  if (!*dd_ok) return 0;
  DDExtra* x = extra(h);
  if (x->epoch != h->cfg_epoch) {          // parameters, traits or paths were set since
    dd_drop_graphs(h);
    x->epoch = h->cfg_epoch;
  }
  // the id-ordered index and the densities' buffers are in their steady state after one step
  // of the host-driven kind
  if (!h->ord_valid || h->ord_n != h->N || h->N == 0) {
    GNXCHK(gnx_step(h, burn ? 1 : 0, sel ? 1 : 0));
    *T -= 1;
    if (h->N == 0 || !h->ord_valid || h->ord_n != h->N) *dd_ok = false;
  }
  return 0;
}

extern "C" int gnx_walk_many(gnx_state** hs, int32_t n, int64_t T, int32_t burn,
                             int32_t with_selection);

extern "C" int gnx_walk(gnx_state* h, int64_t T, int32_t burn, int32_t with_selection) {
  gnx_state* hs[1] = {h};
  return gnx_walk_many(hs, 1, T, burn, with_selection);
}

// T steps of n independent handles (the iterations of one model, sim/model.py:866-953): step
// t of every handle is enqueued before step t + 1 of any - one graph launch each, on the
// handles' own streams - so their kernels share the chip.
extern "C" int gnx_walk_many(gnx_state** hs, int32_t n, int64_t T, int32_t burn,
                             int32_t with_selection) {
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<int64_t> left(n, T);
  std::vector<char> dd(n, 0);
  for (int k = 0; k < n; ++k) {
    bool ok = false;
    hs[k]->dd_hist.clear();
    hs[k]->dd_hist_closed = false;
    const int64_t n0 = hs[k]->N - hs[k]->n_ghost;
    GNXCHK(walk_prepare(hs[k], &left[k], burn != 0, with_selection != 0, &ok));
    if (left[k] < T) {
      hs[k]->dd_hist.push_back(n0);
      hs[k]->dd_hist.push_back(hs[k]->last_births);
      hs[k]->dd_hist.push_back(hs[k]->last_deaths);
    }
    dd[k] = ok && left[k] > 0;
    if (dd[k]) {
      const int rc_enter = dd_enter(hs[k]);
      if (rc_enter) {
        // the handles entered so far go back to their host-driven state before the error leaves
        dd[k] = 0;
        for (int q = 0; q < k; ++q)
          if (dd[q]) (void)gnx_dd_leave(hs[q]);
        return rc_enter;
      }
    }
  }
  int rc = 0;
  for (bool any = true; any && !rc;) {
    any = false;
    for (int k = 0; k < n && !rc; ++k) {
      if (left[k] <= 0) continue;
      int64_t taken = 1;
      if (dd[k]) {
        rc = dd_some(hs[k], burn != 0, with_selection != 0, left[k], &taken);
      } else {
        const int64_t n0 = hs[k]->N - hs[k]->n_ghost;
        // (every step but the walk's last moves the population for the next one with its own
        // mortality: nobody looks at it in between - gnx_l_move_ahead)
        hs[k]->eager_move = left[k] > 1;
        rc = gnx_step(hs[k], burn, with_selection);
        hs[k]->eager_move = false;
        if (!rc) {                  // (a step that failed half-way is no step of the history)
          hs[k]->dd_hist.push_back(n0);
          hs[k]->dd_hist.push_back(hs[k]->last_births);
          hs[k]->dd_hist.push_back(hs[k]->last_deaths);
        }
      }
      left[k] -= taken;
      any = any || left[k] > 0;
    }
  }
  for (int k = 0; k < n; ++k)
    if (dd[k]) {
      const int r2 = gnx_dd_leave(hs[k]);
      if (!rc) rc = r2;
    }
  if (gnx_host_times()) {
    g_host_step_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_host_steps += T * n;
  }
  return rc;
}

// per-step records of the last gnx_walk: population at the start of the step, births, deaths -
// Species.Nt / n_births / n_deaths of the reference (structs/species.py:374-380) without a
// read-back per step.  Returns how many steps were written (the last max_steps of the walk).
extern "C" int64_t gnx_walk_history(gnx_state* h, int64_t max_steps, int64_t* n_start,
                                    int64_t* births, int64_t* deaths) {
  const int64_t total = (int64_t)h->dd_hist.size() / 3;
  const int64_t have = std::min<int64_t>(total, max_steps);
  for (int64_t q = 0; q < have; ++q) {
    const int64_t k = total - have + q;
    n_start[q] = h->dd_hist[3 * k];
    births[q] = h->dd_hist[3 * k + 1];
    deaths[q] = h->dd_hist[3 * k + 2];
  }
  return have;
}
