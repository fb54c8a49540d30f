// One tiled time step as ONE call into the library, the exchanges issued by the library itself
// on the handle's own stream: gnx_tile_step (include/gnx_hip.h).
//
// Round 3's tiled step was driven from Python (geonomics_amd/parallel.py: TiledStepper._step_v2):
// the tile2 entry points of gnx_tile.hip composed with torch.distributed collectives - two
// gloo all-gathers of counts, batches of P2POp sends, an all-gather of pair keys with
// torch.searchsorted behind it, ~10 hand-overs between torch's stream and the library's.
// Here the same protocol runs in C: neighbour exchanges are grouped
// ncclSend / ncclRecv on h->stream (RCCL over xGMI: point-to-point links, so the byte movers
// are p2p messages between neighbour tiles and the only collectives are KB-sized), the count
// exchanges are a KB-sized ncclAllGather whose result reaches the host through pinned memory,
// the pairs' global offspring offsets come from a kernel of binary searches.  No torch, no
// Python between the phases of a step.
//
// The reference has no counterpart: it is a single process (sim/model.py:924-925 is a TODO
// about farming iterations out); the partitioning is SURVEY 8(e)'s.
//
// Transports: RCCL (librccl, loaded when a communicator is made), and - for the tests on a
// one-GPU box, where RCCL refuses two ranks on one device - "local": the tiles are handles
// of one process driven by one host thread each, and the exchanges are device-to-device
// copies between them behind a barrier of the threads.  Everything but the ncclSend /
// ncclRecv / ncclAllGather / ncclAllReduce calls themselves is the same code.
#include <dlfcn.h>
#include <pthread.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <rccl/rccl.h>
#include "gnx_internal.h"

namespace {

// ---- librccl, loaded on demand (the library itself does not link it) ----------------------
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib) return 0;
  void* lib = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) {
    gnx_set_error("librccl.so not found (%s)", dlerror());
    return 1;
  }
#define GNX_SYM(f)                                                              \
  g_rccl.f = (decltype(g_rccl.f))dlsym(lib, "nccl" #f);                         \
  if (!g_rccl.f) {                                                              \
    gnx_set_error("librccl.so has no nccl" #f);                                 \
    return 1;                                                                   \
  }
  GNX_SYM(GetUniqueId) GNX_SYM(CommInitRank) GNX_SYM(CommDestroy) GNX_SYM(GetErrorString)
  GNX_SYM(GroupStart) GNX_SYM(GroupEnd) GNX_SYM(Send) GNX_SYM(Recv) GNX_SYM(AllGather)
  GNX_SYM(AllReduce) GNX_SYM(CommCount) GNX_SYM(CommUserRank) GNX_SYM(CommCuDevice)
#undef GNX_SYM
  g_rccl.lib = lib;
  return 0;
}

#define NCCLCHK(expr)                                                                   \
  do {                                                                                  \
    ncclResult_t _r = (expr);                                                           \
    if (_r != ncclSuccess) {                                                            \
      gnx_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, \
                    __LINE__);                                                          \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

// ---- the local transport's meeting point ----------------------------------------------------
struct LocalGroup {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0;
  long generation = 0;
  bool aborted = false;
  // what the ranks post for each other between two barriers
  std::vector<std::vector<int64_t>> vec;           // small host vectors
  std::vector<std::vector<const void*>> ptr;       // device pointers of the posted parts
  // Like RCCL the local transport is ordered on streams, not by draining them: a rank records
  // `posted` behind the kernels that filled what it posts, the readers' streams wait for it,
  // copy, and record `done`, which the poster's stream waits for before it touches the buffers
  // again.  The threads still meet twice per exchange (pointers out, events recorded) - on the
  // host only.  GNX_LOCAL_SYNC=0 selects it; the default still drains the stream on both sides of
  // every exchange: two tiles sharing ONE GPU step no faster either way (1.84-2.14 against 1.78-1.81
  // ms, profiles/r05_ab_runs.txt - what the threads wait for on a shared GPU is each other's kernels).
  std::vector<hipEvent_t> posted, done;
  explicit LocalGroup(int w) : world(w), vec(w), ptr(w), posted(w, nullptr), done(w, nullptr) {}
  int barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return 1;
    const long gen = generation;
    if (++waiting == world) {
      waiting = 0;
      ++generation;
      cv.notify_all();
      return 0;
    }
    const bool ok = cv.wait_for(lk, std::chrono::seconds(60),
                                [&] { return generation != gen || aborted; });
    if (!ok || aborted) {
      aborted = true;
      cv.notify_all();
      return 1;
    }
    return 0;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
};

enum { COMM_SINGLE = 0, COMM_RCCL = 1, COMM_LOCAL = 2 };
enum { RB_MIG_REC, RB_MIG_Z, RB_MIG_GENO, RB_GHOST, RB_REQ, RB_GAMETES, RB_KEYS, RB_GOFF, RB_PAD,
       RB_VEC_SEND, RB_VEC_RECV, RB_COUNT };

struct Comm {
  int kind = COMM_SINGLE, rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  LocalGroup* grp = nullptr;
  void* rbuf[RB_COUNT]{};
  size_t rcap[RB_COUNT]{};
  int64_t* pin = nullptr;          // pinned host words [2][pin_words]: vectors out / in
  int64_t* pin_dev = nullptr;
  int64_t pin_words = 0;
  // between gnx_tile_step_begin and _end (the births' host hooks run there)
  bool mid = false;
  int64_t mid_pairs = 0, mid_births = 0;
  int64_t pre = -1;                // global population before the last step's deaths
  int64_t bytes_sent = 0;
  int64_t steps = 0;
  // host wall time of a tiled step's phases, summed over the steps (gnx_comm_info): [0] age +
  // movement + routing counts + count exchange (host wait 1), [1] migrant / ghost exchange +
  // import, [2] cell sort + pairs + second count exchange (host wait 2), [3] births + gamete
  // service, [4] density all-reduce + death probabilities + mortality (host wait 3)
  double phase_s[5]{};
  // GNX_COMM_FORCE_RCCL=1 at gnx_comm_init_rccl: a ONE-rank communicator goes through the RCCL
  // calls too (all-gather, all-reduce, sends and receives to itself) instead of the shortcuts
  // - what a one-GPU box can run of them
  bool forced = false;
};

Comm* comm_of(gnx_state* h) { return (Comm*)h->tile_comm; }

// host wall clock of a tiled step's phases (Comm::phase_s)
struct PhaseClock {
  Comm* c;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  explicit PhaseClock(Comm* c_) : c(c_) {}
  void mark(int k) {
    const auto n = std::chrono::steady_clock::now();
    c->phase_s[k] += std::chrono::duration<double>(n - t).count();
    t = n;
  }
};

int rb_need(Comm* c, int k, size_t bytes) {
  if (bytes <= c->rcap[k]) return 0;
  if (c->rbuf[k]) HIPCHK(hipFree(c->rbuf[k]));       // (waits for the device: nobody reads it)
  c->rcap[k] = bytes + bytes / 4 + 4096;
  HIPCHK(hipMalloc(&c->rbuf[k], c->rcap[k]));
  return 0;
}

__global__ void k_copy_i64(int n, const int64_t* __restrict__ src, int64_t* __restrict__ dst) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) dst[k] = src[k];
}

__global__ void k_sum_i32(int n, int world, const int32_t* __restrict__ all, int32_t* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int32_t s = 0;
  for (int r = 0; r < world; ++r) s += all[(int64_t)r * n + k];
  out[k] = s;
}

__global__ void k_copy_i32(int n, const int32_t* __restrict__ src, int32_t* __restrict__ dst, int zero) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) dst[k] = zero ? 0 : src[k];
}

__global__ void k_widen_i32(int n, const int32_t* __restrict__ src, int64_t* __restrict__ dst) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) dst[k] = src[k];
}

// every rank's vector of n int64 words -> out[world][n] on the host; the gathered words stay in
// device memory too (rbuf[RB_VEC_RECV]).  The last n_tail words of the vector come from DEVICE
// memory (int32 `tail`: counts a kernel has just left there), the others from the host.
// RCCL: one KB-sized all-gather on the handle's stream and one wait for it; the words travel
// to and from the host through pinned memory.
int host_allgather(gnx_state* h, const int64_t* vec, int n, int64_t* out,
                   const int32_t* tail = nullptr, int n_tail = 0) {
  Comm* c = comm_of(h);
  const int w = c->world;
  const int n_host = n - n_tail;
  if ((w == 1 && !c->forced) || c->kind == COMM_LOCAL) {
    std::vector<int64_t> mine(vec, vec + n_host);
    mine.resize(n, 0);
    if (n_tail > 0) {
      std::vector<int32_t> t32(n_tail);
      HIPCHK(hipMemcpyAsync(t32.data(), tail, (size_t)n_tail * 4, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      for (int k = 0; k < n_tail; ++k) mine[n_host + k] = t32[k];
    }
    if (w == 1) {
      for (int k = 0; k < n; ++k) out[k] = mine[k];
      return 0;
    }
    GNXCHK(rb_need(c, RB_VEC_RECV, (size_t)w * n * 8));
    LocalGroup* g = c->grp;
    g->vec[c->rank] = mine;
    if (g->barrier()) {
      gnx_set_error("local tile group: a rank failed or never arrived");
      return 1;
    }
    for (int r = 0; r < w; ++r)
      for (int k = 0; k < n; ++k) out[(int64_t)r * n + k] = g->vec[r][k];
    if (g->barrier()) return 1;
    HIPCHK(hipMemcpyAsync(c->rbuf[RB_VEC_RECV], out, (size_t)w * n * 8, hipMemcpyHostToDevice,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));          // (`out` is the caller's)
    return 0;
  }
  if (n_host > c->pin_words || (int64_t)w * n > c->pin_words) {
    gnx_set_error("tile communicator: a gather of %d x %d words does not fit its %lld pinned words",
                  w, n, (long long)c->pin_words);
    return 1;
  }
  GNXCHK(rb_need(c, RB_VEC_RECV, (size_t)w * n * 8));
  GNXCHK(rb_need(c, RB_VEC_SEND, (size_t)n * 8));
  for (int k = 0; k < n_host; ++k) c->pin[k] = vec[k];
  if (n_host > 0)
    hipLaunchKernelGGL(k_copy_i64, dim3((n_host + 63) / 64), dim3(64), 0, h->stream, n_host,
                       (const int64_t*)c->pin_dev, (int64_t*)c->rbuf[RB_VEC_SEND]);
  if (n_tail > 0)
    hipLaunchKernelGGL(k_widen_i32, dim3((n_tail + 63) / 64), dim3(64), 0, h->stream, n_tail, tail,
                       (int64_t*)c->rbuf[RB_VEC_SEND] + n_host);
  NCCLCHK(g_rccl.AllGather(c->rbuf[RB_VEC_SEND], c->rbuf[RB_VEC_RECV], (size_t)n, ncclInt64,
                           c->nccl, h->stream));
  hipLaunchKernelGGL(k_copy_i64, dim3((w * n + 63) / 64), dim3(64), 0, h->stream, w * n,
                     (const int64_t*)c->rbuf[RB_VEC_RECV], c->pin_dev + c->pin_words);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < w * n; ++k) out[k] = c->pin[c->pin_words + k];
  return 0;
}

struct Part {
  const void* send;     // grouped by destination rank
  size_t unit;          // bytes per element
  int rb;               // receive buffer
};

// mat[src * w + dst] = elements src sends to dst; every part moves with the same counts.  One
// group of sends and receives for all parts (RCCL), nothing waits for the device.
int exchange(gnx_state* h, const Part* parts, int n_parts, const int64_t* mat) {
  Comm* c = comm_of(h);
  const int w = c->world, me = c->rank;
  int64_t n_in = 0;
  for (int r = 0; r < w; ++r) n_in += mat[(int64_t)r * w + me];
  for (int k = 0; k < n_parts; ++k) GNXCHK(rb_need(c, parts[k].rb, (size_t)n_in * parts[k].unit));
  if (c->kind == COMM_LOCAL) {
    LocalGroup* g = c->grp;
    static const bool drain = !(getenv("GNX_LOCAL_SYNC") && atoi(getenv("GNX_LOCAL_SYNC")) == 0);
    if (drain) HIPCHK(hipStreamSynchronize(h->stream));          // what this rank posts is written
    else HIPCHK(hipEventRecord(g->posted[me], h->stream));
    g->ptr[me].assign(n_parts, nullptr);
    for (int k = 0; k < n_parts; ++k) g->ptr[me][k] = parts[k].send;
    if (g->barrier()) {
      gnx_set_error("local tile group: a rank failed or never arrived");
      return 1;
    }
    int64_t roff = 0;
    for (int peer = 0; peer < w; ++peer) {
      const int64_t n = mat[(int64_t)peer * w + me];
      int64_t soff = 0;
      for (int q = 0; q < me; ++q) soff += mat[(int64_t)peer * w + q];
      if (n > 0) {
        if (!drain && peer != me) HIPCHK(hipStreamWaitEvent(h->stream, g->posted[peer], 0));
        for (int k = 0; k < n_parts; ++k) {
          c->bytes_sent += n * (int64_t)parts[k].unit;          // (counted by the receiver here)
          HIPCHK(hipMemcpyAsync((char*)c->rbuf[parts[k].rb] + roff * parts[k].unit,
                                (const char*)g->ptr[peer][k] + soff * parts[k].unit,
                                (size_t)n * parts[k].unit, hipMemcpyDeviceToDevice, h->stream));
        }
      }
      roff += n;
    }
    if (drain) HIPCHK(hipStreamSynchronize(h->stream));          // nobody reuses a posted buffer before this
    else HIPCHK(hipEventRecord(g->done[me], h->stream));
    if (g->barrier()) return 1;
    if (!drain)
      for (int peer = 0; peer < w; ++peer)         // whoever read from this rank has finished
        if (peer != me && mat[(int64_t)me * w + peer] > 0)
          HIPCHK(hipStreamWaitEvent(h->stream, g->done[peer], 0));
    return 0;
  }
  if (c->kind == COMM_SINGLE) {                       // one rank, no RCCL: what it sends itself
    for (int k = 0; k < n_parts; ++k)
      if (mat[0] > 0)
        HIPCHK(hipMemcpyAsync(c->rbuf[parts[k].rb], parts[k].send, (size_t)mat[0] * parts[k].unit,
                              hipMemcpyDeviceToDevice, h->stream));
    return 0;
  }
  // (what this rank sends itself is a plain copy, outside the group)
  {
    int64_t roff = 0, soff = 0;
    for (int peer = 0; peer < w; ++peer) {
      const int64_t n_out = mat[(int64_t)me * w + peer], n_from = mat[(int64_t)peer * w + me];
      if (peer == me && !c->forced && n_out > 0)
        for (int k = 0; k < n_parts; ++k)
          HIPCHK(hipMemcpyAsync((char*)c->rbuf[parts[k].rb] + roff * parts[k].unit,
                                (const char*)parts[k].send + soff * parts[k].unit,
                                (size_t)n_out * parts[k].unit, hipMemcpyDeviceToDevice, h->stream));
      soff += n_out;
      roff += n_from;
    }
  }
  NCCLCHK(g_rccl.GroupStart());
  // an error between GroupStart and GroupEnd must not leave the group open: the first failure
  // is remembered, the group is closed, then it is reported
  ncclResult_t bad = ncclSuccess;
  const char* what = "";
  int64_t roff = 0, soff = 0;
  for (int peer = 0; peer < w && bad == ncclSuccess; ++peer) {
    const int64_t n_out = mat[(int64_t)me * w + peer], n_from = mat[(int64_t)peer * w + me];
    for (int k = 0; k < n_parts && bad == ncclSuccess; ++k) {
      if (peer == me && !c->forced) continue;
      if (n_out > 0) {
        bad = g_rccl.Send((const char*)parts[k].send + soff * parts[k].unit,
                          (size_t)n_out * parts[k].unit, ncclChar, peer, c->nccl, h->stream);
        what = "ncclSend";
        c->bytes_sent += n_out * (int64_t)parts[k].unit;
      }
      if (n_from > 0 && bad == ncclSuccess) {
        bad = g_rccl.Recv((char*)c->rbuf[parts[k].rb] + roff * parts[k].unit,
                          (size_t)n_from * parts[k].unit, ncclChar, peer, c->nccl, h->stream);
        what = "ncclRecv";
      }
    }
    soff += n_out;
    roff += n_from;
  }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (bad != ncclSuccess) {
    gnx_set_error("%s failed: %s (tile exchange)", what, g_rccl.GetErrorString(bad));
    return 1;
  }
  NCCLCHK(end);
  return 0;
}

// int32 words summed over the ranks, in place (both density fields and the counters)
int allreduce_i32(gnx_state* h, int32_t* buf, int64_t n) {
  Comm* c = comm_of(h);
  const int w = c->world, me = c->rank;
  if (w == 1 && !c->forced) return 0;
  if (c->kind == COMM_LOCAL) {
    LocalGroup* g = c->grp;
    static const bool drain = !(getenv("GNX_LOCAL_SYNC") && atoi(getenv("GNX_LOCAL_SYNC")) == 0);
    GNXCHK(rb_need(c, RB_PAD, (size_t)w * n * 4));
    if (drain) HIPCHK(hipStreamSynchronize(h->stream));
    else HIPCHK(hipEventRecord(g->posted[me], h->stream));
    g->ptr[me].assign(1, buf);
    if (g->barrier()) {
      gnx_set_error("local tile group: a rank failed or never arrived");
      return 1;
    }
    for (int r = 0; r < w; ++r) {
      if (!drain && r != me) HIPCHK(hipStreamWaitEvent(h->stream, g->posted[r], 0));
      HIPCHK(hipMemcpyAsync((int32_t*)c->rbuf[RB_PAD] + (int64_t)r * n, g->ptr[r][0], (size_t)n * 4,
                            hipMemcpyDeviceToDevice, h->stream));
    }
    if (drain) HIPCHK(hipStreamSynchronize(h->stream));
    else HIPCHK(hipEventRecord(g->done[me], h->stream));
    if (g->barrier()) return 1;                       // everybody has read everybody's words
    if (!drain)
      for (int r = 0; r < w; ++r)                     // ... before this rank's sum overwrites its own
        if (r != me) HIPCHK(hipStreamWaitEvent(h->stream, g->done[r], 0));
    hipLaunchKernelGGL(k_sum_i32, dim3((int)((n + 255) / 256)), dim3(256), 0, h->stream, (int)n, w,
                       (const int32_t*)c->rbuf[RB_PAD], buf);
    HIPCHK(hipGetLastError());
    return 0;
  }
  NCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclInt32, ncclSum, c->nccl, h->stream));
  return 0;
}

int comm_new(gnx_state* h, Comm** out, int world) {
  if (h->tile_comm) {
    gnx_set_error("this handle already has a tile communicator");
    return 1;
  }
  Comm* c = new Comm();
  // the largest gather of a step: every rank's [P, B | 64 virtual-tile counts | world request
  // counts] (gnx_tile_step), read back as world x that many words
  c->pin_words = std::max<int64_t>(4096, (int64_t)world * (2 + 64 + world) + 64);
  if (hipHostMalloc((void**)&c->pin, 2 * c->pin_words * sizeof(int64_t),
                    hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->pin_dev, c->pin, 0) != hipSuccess) {
    delete c;
    gnx_set_error("pinned memory for the tile communicator");
    return 1;
  }
  h->tile_comm = c;
  *out = c;
  return 0;
}

}  // namespace

// ---------------------------------------------------------------- communicators
extern "C" int gnx_comm_unique_id(uint8_t* out128) {
  GNXCHK(rccl_load());
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, sizeof(id));
  return 0;
}

// librccl can be loaded and has every entry point the transport uses: what the ranks tell each
// other BEFORE any of them enters ncclCommInitRank (a collective nobody can be called back
// from: a rank that cannot follow would leave the others inside it)
extern "C" int gnx_comm_probe(void) { return rccl_load(); }

namespace {
// ncclCommInitRank with a deadline.  The call blocks until every rank has arrived; a rank that
// died on its way leaves the others inside it for ever.  Past the deadline (GNX_COMM_INIT_TIMEOUT_S,
// default 180 s) this rank says why and ends the PROCESS with a non-zero code - the call cannot be
// cancelled, and a process that has touched the GPU must not be replaced by another program.
struct InitJob {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  ncclResult_t res = ncclSuccess;
  ncclComm_t comm = nullptr;
};
void* init_thread(void* arg);
struct InitArgs {
  InitJob* job;
  int device, world, rank;
  ncclUniqueId id;
};
void* init_thread(void* arg) {
  InitArgs* a = (InitArgs*)arg;
  (void)hipSetDevice(a->device);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&comm, a->world, a->id, a->rank);
  {
    std::lock_guard<std::mutex> lk(a->job->mu);
    a->job->res = r;
    a->job->comm = comm;
    a->job->done = true;
  }
  a->job->cv.notify_all();
  return nullptr;
}
}  // namespace

extern "C" int gnx_comm_init_rccl(gnx_state* h, const uint8_t* id128, int32_t rank, int32_t world) {
  GNXCHK(rccl_load());
  HIPCHK(hipSetDevice(h->cfg.device));
  if (world < 1 || rank < 0 || rank >= world) {
    gnx_set_error("gnx_comm_init_rccl: rank %d of %d", rank, world);
    return 1;
  }
  Comm* c = nullptr;
  GNXCHK(comm_new(h, &c, world));
  c->kind = world > 1 ? COMM_RCCL : COMM_SINGLE;
  c->rank = rank;
  c->world = world;
  // (a one-rank communicator too: bench.py --gpus 1 goes through the same calls)
  {
    // (job and args outlive a thread that is left behind past the deadline: never freed then)
    InitJob* job = new InitJob();
    InitArgs* args = new InitArgs();
    args->job = job;
    args->device = h->cfg.device;
    args->world = world;
    args->rank = rank;
    memcpy(&args->id, id128, sizeof(args->id));
    pthread_t th;
    if (pthread_create(&th, nullptr, init_thread, args) != 0) {
      (void)gnx_comm_free(h);
      gnx_set_error("gnx_comm_init_rccl: could not start the thread that joins the communicator");
      return 1;
    }
    const char* ts = getenv("GNX_COMM_INIT_TIMEOUT_S");
    const double limit = ts && atof(ts) > 0 ? atof(ts) : 180.0;
    std::unique_lock<std::mutex> lk(job->mu);
    const bool ok = job->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return job->done; });
    if (!ok) {
      fprintf(stderr,
              "geonomics_amd: rank %d of %d waited %.0f s inside ncclCommInitRank for the other ranks "
              "(GNX_COMM_INIT_TIMEOUT_S); a rank must have failed before it got there. Giving up.\n",
              rank, world, limit);
      // (a library call that ends the process: at least Python's and C's buffered output and
      // the collectors' open files reach the disk - INTEGRATION.md: exit code 86 = init timeout,
      // the launcher starts a fresh child process, it never retries in-process)
      fflush(NULL);
      _exit(86);
    }
    lk.unlock();
    (void)pthread_join(th, nullptr);
    const ncclResult_t r = job->res;
    c->nccl = job->comm;
    delete job;
    delete args;
    if (r != ncclSuccess) {
      c->nccl = nullptr;
      gnx_set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
      (void)gnx_comm_free(h);
      return 1;
    }
  }
  const char* f = getenv("GNX_COMM_FORCE_RCCL");
  c->forced = world == 1 && f && f[0] == '1';
  if (world > 1 || c->forced) c->kind = COMM_RCCL;
  return 0;
}

extern "C" int gnx_comm_init_single(gnx_state* h) {
  Comm* c = nullptr;
  GNXCHK(comm_new(h, &c, 1));
  return 0;
}

extern "C" int gnx_comm_local_create(int32_t world, void** group) {
  *group = new LocalGroup(world);
  return 0;
}

extern "C" int gnx_comm_local_join(gnx_state* h, void* group, int32_t rank) {
  LocalGroup* g = (LocalGroup*)group;
  if (rank < 0 || rank >= g->world) {
    gnx_set_error("gnx_comm_local_join: rank %d of %d", rank, g->world);
    return 1;
  }
  Comm* c = nullptr;
  GNXCHK(comm_new(h, &c, g->world));
  c->kind = g->world > 1 ? COMM_LOCAL : COMM_SINGLE;
  c->rank = rank;
  c->world = g->world;
  c->grp = g;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipEventCreateWithFlags(&g->posted[rank], hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming));
  return 0;
}

extern "C" int gnx_comm_local_abort(void* group) {
  ((LocalGroup*)group)->abort();
  return 0;
}

extern "C" int gnx_comm_local_destroy(void* group) {
  LocalGroup* g = (LocalGroup*)group;
  for (hipEvent_t e : g->posted)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : g->done)
    if (e) (void)hipEventDestroy(e);
  delete g;
  return 0;
}

extern "C" int gnx_comm_free(gnx_state* h) {
  Comm* c = comm_of(h);
  if (!c) return 0;
  (void)hipStreamSynchronize(h->stream);
  for (int k = 0; k < RB_COUNT; ++k)
    if (c->rbuf[k]) (void)hipFree(c->rbuf[k]);
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  if (c->pin) (void)hipHostFree(c->pin);
  delete c;
  h->tile_comm = nullptr;
  return 0;
}

// Known words through every operation of the transport - the gather of host and device words
// (and its device copy), a ragged two-part exchange with every rank including itself, the
// in-place sum - checked on the host.  TiledStepper runs it once when the ranks have joined:
// a transport that does not deliver fails here, loudly, not as a wrong population later.
extern "C" int gnx_comm_selftest(gnx_state* h) {
  Comm* c = comm_of(h);
  if (!c) {
    gnx_set_error("gnx_comm_selftest: the handle has no communicator");
    return 1;
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  struct DevBuf {                                   // (freed on every way out)
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
  } b_tail, b_32, b_8, b_sum, b_geno;
  const int w = c->world, me = c->rank;
  auto fail = [&](const char* what, long long got, long long want) {
    gnx_set_error("gnx_comm_selftest (rank %d of %d): %s: got %lld, expected %lld", me, w, what,
                  got, want);
    return 1;
  };
  // -- gather: three host words and two words a kernel left on the device
  HIPCHK(hipMalloc(&b_tail.p, 64));
  int32_t* d_tail = (int32_t*)b_tail.p;
  const int32_t tail[2] = {100 + me, 200 + 3 * me};
  HIPCHK(hipMemcpyAsync(d_tail, tail, 8, hipMemcpyHostToDevice, h->stream));
  const int64_t vec[5] = {me, 7LL * me + 1, -(int64_t)me - (1LL << 40), 0, 0};
  std::vector<int64_t> all((size_t)w * 5);
  int rc = host_allgather(h, vec, 5, all.data(), d_tail, 2);
  if (rc) return rc;
  for (int r = 0; r < w; ++r) {
    const int64_t want[5] = {r, 7LL * r + 1, -(int64_t)r - (1LL << 40), 100 + r, 200 + 3 * r};
    for (int k = 0; k < 5; ++k)
      if (all[(size_t)r * 5 + k] != want[k]) return fail("gathered word", all[(size_t)r * 5 + k], want[k]);
  }
  if (w > 1 || c->forced) {
    std::vector<int64_t> dev((size_t)w * 5);
    HIPCHK(hipMemcpy(dev.data(), c->rbuf[RB_VEC_RECV], dev.size() * 8, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < dev.size(); ++k)
      if (dev[k] != all[k]) return fail("gathered word (device copy)", dev[k], all[k]);
  }
  // -- exchange: src sends 1 + (3 src + 5 dst) % 4 elements to dst, as int32 and as bytes
  auto cnt = [](int s, int d) { return (int64_t)(1 + (3 * s + 5 * d) % 4); };
  std::vector<int64_t> mat((size_t)w * w);
  for (int s = 0; s < w; ++s)
    for (int d = 0; d < w; ++d) mat[(size_t)s * w + d] = cnt(s, d);
  std::vector<int32_t> s32;
  std::vector<uint8_t> s8;
  for (int d = 0; d < w; ++d)
    for (int64_t k = 0; k < cnt(me, d); ++k) {
      s32.push_back(me * 100000 + d * 100 + (int32_t)k);
      s8.push_back((uint8_t)(me * 31 + d * 7 + k));
    }
  HIPCHK(hipMalloc(&b_32.p, s32.size() * 4));
  HIPCHK(hipMalloc(&b_8.p, s8.size() + 16));
  void *d32 = b_32.p, *d8 = b_8.p;
  HIPCHK(hipMemcpyAsync(d32, s32.data(), s32.size() * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d8, s8.data(), s8.size(), hipMemcpyHostToDevice, h->stream));
  const Part parts[2] = {{d32, 4, RB_KEYS}, {d8, 1, RB_PAD}};
  const int64_t sent0 = c->bytes_sent;
  rc = exchange(h, parts, 2, mat.data());
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = 1;
  c->bytes_sent = sent0;                              // (not the population's bytes)
  std::vector<int32_t> r32;
  std::vector<uint8_t> r8;
  int64_t n_in = 0;
  for (int s = 0; s < w; ++s) n_in += cnt(s, me);
  r32.resize((size_t)n_in);
  r8.resize((size_t)n_in);
  if (!rc) {
    HIPCHK(hipMemcpy(r32.data(), c->rbuf[RB_KEYS], r32.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(r8.data(), c->rbuf[RB_PAD], r8.size(), hipMemcpyDeviceToHost));
  }
  if (rc) {
    gnx_set_error("gnx_comm_selftest (rank %d of %d): the exchange failed", me, w);
    return 1;
  }
  size_t at = 0;
  for (int s = 0; s < w; ++s)
    for (int64_t k = 0; k < cnt(s, me); ++k, ++at) {
      if (r32[at] != s * 100000 + me * 100 + (int32_t)k)
        return fail("exchanged int32", r32[at], s * 100000 + me * 100 + k);
      if (r8[at] != (uint8_t)(s * 31 + me * 7 + k))
        return fail("exchanged byte", r8[at], (uint8_t)(s * 31 + me * 7 + k));
    }
  // -- ... and elements the size of a migrant's genome (16 x W64 bytes: 25 KB at L = 10^5), the
  //    largest unit the step sends: the same ragged counts, every byte checked
  {
    const size_t unit = (size_t)std::max(16 * h->W64, 64);
    std::vector<uint8_t> sg;
    for (int d = 0; d < w; ++d)
      for (int64_t k = 0; k < cnt(me, d); ++k)
        for (size_t j = 0; j < unit; ++j)
          sg.push_back((uint8_t)(me * 131 + d * 17 + k * 5 + j * 3 + (j >> 8)));
    HIPCHK(hipMalloc(&b_geno.p, sg.size() + 16));
    HIPCHK(hipMemcpyAsync(b_geno.p, sg.data(), sg.size(), hipMemcpyHostToDevice, h->stream));
    const Part pg[1] = {{b_geno.p, unit, RB_MIG_GENO}};
    const int64_t sent1 = c->bytes_sent;
    rc = exchange(h, pg, 1, mat.data());
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = 1;
    c->bytes_sent = sent1;
    if (rc) {
      gnx_set_error("gnx_comm_selftest (rank %d of %d): the exchange of genome-sized elements failed", me, w);
      return 1;
    }
    std::vector<uint8_t> rg((size_t)n_in * unit);
    HIPCHK(hipMemcpy(rg.data(), c->rbuf[RB_MIG_GENO], rg.size(), hipMemcpyDeviceToHost));
    size_t pos = 0;
    for (int sr = 0; sr < w; ++sr)
      for (int64_t k = 0; k < cnt(sr, me); ++k)
        for (size_t j = 0; j < unit; ++j, ++pos) {
          const uint8_t want = (uint8_t)(sr * 131 + me * 17 + k * 5 + j * 3 + (j >> 8));
          if (rg[pos] != want) return fail("exchanged genome-sized element, byte", rg[pos], want);
        }
  }
  // -- sum in place
  HIPCHK(hipMalloc(&b_sum.p, 8 * 4));
  int32_t* d_sum = (int32_t*)b_sum.p;
  int32_t mine[8], got[8];
  for (int k = 0; k < 8; ++k) mine[k] = me * 10 + k - 3;
  HIPCHK(hipMemcpyAsync(d_sum, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
  rc = allreduce_i32(h, d_sum, 8);
  if (!rc && hipMemcpyAsync(got, d_sum, sizeof(got), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = 1;
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = 1;
  if (rc) {
    gnx_set_error("gnx_comm_selftest (rank %d of %d): the all-reduce failed", me, w);
    return 1;
  }
  for (int k = 0; k < 8; ++k) {
    const long long want = 10LL * w * (w - 1) / 2 + (long long)w * (k - 3);
    if (got[k] != want) return fail("summed word", got[k], want);
  }
  // (fault injection for the tests of what the callers do with a failure: this rank only,
  // after every collective of the test has run)
  const char* inj = getenv("GNX_COMM_SELFTEST_FAIL");
  if (inj && inj[0] && atoi(inj) == me) return fail("injected failure (GNX_COMM_SELFTEST_FAIL)", 0, 1);
  return 0;
}

extern "C" int64_t gnx_comm_bytes_sent(gnx_state* h) {
  Comm* c = comm_of(h);
  return c ? c->bytes_sent : 0;
}

// ---------------------------------------------------------------- the step
// [req counts | pair count] behind the 64 virtual-tile counts of h->vt_count: one device vector
// for the second count exchange
__global__ void k_tail_assemble(int T, const int32_t* __restrict__ req, int have_req,
                                const int32_t* __restrict__ P_src, int P_host,
                                int32_t* __restrict__ dst) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < T) dst[k] = have_req ? req[k] : 0;
  // (P_host >= 0: the host has waited for this tile's pair count - Poisson births - and the
  // device word may be stale: an empty tile's mate search returns before it writes it)
  if (k == T) dst[T] = P_host >= 0 ? P_host : *P_src;
}

// One time step of this tile (gnx_tile_set) and, through the communicator, of the whole
// tiled landscape, in two calls (gnx_tile_step = both):
//   gnx_tile_step_begin: age + movement, routing of emigrants and ghosts, ONE batch of neighbour
//     sends, import, cell sort + mate search + pairs, global offspring offsets, births, gamete
//     service for ghost mates - every offspring of the step has its record, its alleles at the
//     selected loci and its phenotype when it returns; what the host does to the newborns (the
//     reference's mutation and pedigree recording, ops/mutation.py:169-206, structs/species.py:
//     692-736, sit between the births and the deaths of _do_pop_dynamics) runs between the calls;
//   gnx_tile_step_end: ONE all-reduce of both density fields and the counters, densities, death
//     probabilities, mortality.  out[3]: exact != 0 -> the global (N after the step, births,
//     deaths) at the price of one more KB-sized collective; else the counts that rode on the
//     step's own all-reduce: (N at the START of the step, births, deaths of the PREVIOUS step).
// Host waits per step on several tiles: the two count exchanges (the pair count rides on the
// second one when every pair has the same number of births) and the survivor count.
extern "C" int gnx_tile_step_begin(gnx_state* h, int32_t burn) {
  Comm* c = comm_of(h);
  if (!c) {
    gnx_set_error("gnx_tile_step: no communicator (gnx_comm_init_rccl / _single / gnx_comm_local_join)");
    return 1;
  }
  if (c->mid) {
    gnx_set_error("gnx_tile_step_begin: the previous step was not finished (gnx_tile_step_end)");
    return 1;
  }
  if (!h->have_sp) {
    gnx_set_error("species parameters not set");
    return 1;
  }
  const int w = c->world, me = c->rank, T = h->tile_R * h->tile_C;
  if (T != w) {
    gnx_set_error("gnx_tile_step: %d tiles but %d ranks", T, w);
    return 1;
  }
  if (h->cfg.W % 8 || h->cfg.H % 8 || 8 % h->tile_R || 8 % h->tile_C) {
    gnx_set_error("gnx_tile_step: tile-major offspring ids need landscape dimensions divisible by 8 "
                  "and a tile grid that divides 8 x 8 (got %d x %d tiles of a %d x %d landscape)",
                  h->tile_R, h->tile_C, h->cfg.W, h->cfg.H);
    return 3;
  }
  // offspring ids virtual tile by virtual tile (gnx_set_id_order 1, gnx_kernels_pop.hip): a tile
  // owns whole virtual tiles, the rank of a pair inside its virtual tile is a local matter, and
  // the 64 birth counts per virtual tile ride on the count exchange
  if (h->id_order != 1) GNXCHK(gnx_set_id_order(h, 1));
  const int nt = h->cfg.n_traits, W64 = h->W64;
  const bool geno = h->genomes_assigned && h->cfg.L > 0;
  const bool fixed = h->sp.n_births_fixed != 0;
  // (gnx_totals: THIS tile's own individuals, births and deaths, as gnx_step counts them)
  h->tot[0] += 1;
  h->tot[1] += h->N - h->n_ghost;
  std::vector<int64_t> cnt(2 * T + 4), mats((size_t)w * (2 * T + 4));
  PhaseClock clk(c);
  // 1. age + movement, the routing's counting pass; everybody's counts in ONE exchange (this
  //    tile's own among them: the wait for them is the exchange's), then the pass that fills the
  //    staging buffers and ONE batch of sends: migrants and ghosts
  void* d_cnt = nullptr;
  GNXCHK(gnx_tile2_route_begin(h, 1, &d_cnt));
  if (w > 1) {
    GNXCHK(host_allgather(h, nullptr, 2 * T, mats.data(), (const int32_t*)d_cnt, 2 * T));
    for (int k = 0; k < 2 * T; ++k) cnt[k] = mats[(size_t)me * 2 * T + k];
    clk.mark(0);
    GNXCHK(gnx_tile2_route_finish(h, cnt.data()));
    std::vector<int64_t> m_mig((size_t)w * w), m_gh((size_t)w * w);
    for (int s = 0; s < w; ++s)
      for (int d = 0; d < w; ++d) {
        m_mig[(size_t)s * w + d] = mats[(size_t)s * 2 * T + d];
        m_gh[(size_t)s * w + d] = mats[(size_t)s * 2 * T + T + d];
      }
    void *p_rec, *p_z, *p_g, *p_gh;
    GNXCHK(gnx_tile2_route_ptrs(h, &p_rec, &p_z, &p_g, &p_gh));
    Part parts[3];
    int np = 0;
    parts[np++] = Part{p_rec, 32, RB_MIG_REC};
    if (nt) parts[np++] = Part{p_z, (size_t)4 * nt, RB_MIG_Z};
    if (geno) parts[np++] = Part{p_g, (size_t)16 * W64, RB_MIG_GENO};
    GNXCHK(exchange(h, parts, np, m_mig.data()));
    Part gh{p_gh, 32, RB_GHOST};
    GNXCHK(exchange(h, &gh, 1, m_gh.data()));
    int64_t in_mig = 0, in_gh = 0;
    for (int s = 0; s < w; ++s) {
      in_mig += m_mig[(size_t)s * w + me];
      in_gh += m_gh[(size_t)s * w + me];
    }
    GNXCHK(gnx_tile2_import(h, in_mig, c->rbuf[RB_MIG_REC], nt ? c->rbuf[RB_MIG_Z] : nullptr,
                            geno ? c->rbuf[RB_MIG_GENO] : nullptr, in_gh, c->rbuf[RB_GHOST]));
    clk.mark(1);
  } else {
    clk.mark(0);
  }
  // 2. pairs; the gamete-request counts, the virtual tiles' birth counts and - a fixed number of
  //    births per pair - the pair count itself stay on the device and travel with ONE count
  //    exchange (Poisson births, one tile: the pair count is waited for)
  std::vector<int64_t> pc(2 + T);
  void* d_req = nullptr;
  const bool nowait = w > 1 && fixed;
  GNXCHK(gnx_tile2_requests_dev(h, w > 1 ? 1 : 0, &d_req));
  GNXCHK(gnx_tile2_pairs_mode(h, nowait ? 1 : 0));
  const int rc_pairs = gnx_tile2_pairs(h, burn, pc.data());
  GNXCHK(gnx_tile2_pairs_mode(h, 0));
  GNXCHK(gnx_tile2_requests_dev(h, 0, nullptr));      // (the Python-driven protocol waits for them)
  GNXCHK(rc_pairs);
  int64_t P = pc[0], B = pc[1];
  int64_t total_births = B, total_pairs = P;
  std::vector<int64_t> m_req((size_t)w * w, 0);
  // (P < 0: not known on the host yet - the classification was k_pair_compact's, by slots)
  GNXCHK(gnx_l_pair_cls(h, nowait ? -1 : P, w == 1));
  if (w > 1) {
    // [B | 64 virtual-tile counts | T request counts | P]: all but the first from device memory
    const bool have_req = h->n_req_known == -2;
    hipLaunchKernelGGL(k_tail_assemble, dim3((T + 64) / 64), dim3(64), 0, h->stream, T,
                       (const int32_t*)d_req, have_req ? 1 : 0, (const int32_t*)h->cnt_dev,
                       nowait ? -1 : (int)P, h->vt_count + 64);
    const int cs = 1 + 64 + T + 1;
    std::vector<int64_t> g2((size_t)w * cs);
    const int64_t mine_b = B;
    GNXCHK(host_allgather(h, &mine_b, cs, g2.data(), (const int32_t*)h->vt_count, 64 + T + 1));
    if (nowait) {
      // this tile's own pair count has arrived with everybody's: the bookkeeping gnx_tile2_pairs
      // left open
      P = g2[(size_t)me * cs + 1 + 64 + T];
      GNXCHK(gnx_tile2_pairs_settle(h, burn, P, &B));
    }
    total_births = total_pairs = 0;
    std::vector<int64_t> vt(64, 0);
    for (int r = 0; r < w; ++r) {
      const int64_t Pr = g2[(size_t)r * cs + 1 + 64 + T];
      total_pairs += Pr;
      total_births += fixed ? Pr * (int64_t)h->sp.n_births_lambda : g2[(size_t)r * cs];
      for (int q = 0; q < 64; ++q) vt[q] += g2[(size_t)r * cs + 1 + q];
      for (int d = 0; d < w; ++d) m_req[(size_t)r * w + d] = g2[(size_t)r * cs + 1 + 64 + d];
    }
    GNXCHK(gnx_tile2_set_requests(h, &m_req[(size_t)me * w]));
    // the virtual tiles' base offsets, in births: an exclusive scan of 64 numbers, the same on
    // every rank; to the device through pinned memory
    int64_t run = 0;
    for (int q = 0; q < 64; ++q) {
      c->pin[q] = run * h->vt_mul;
      run += vt[q];
    }
    hipLaunchKernelGGL(k_copy_i64, dim3(1), dim3(64), 0, h->stream, 64, (const int64_t*)c->pin_dev,
                       h->vt_base);
    HIPCHK(hipGetLastError());
  }
  clk.mark(2);
  void* p_req = nullptr;
  const int64_t id_base = h->max_id + 1;        // (the global maximum: every rank keeps it)
  GNXCHK(gnx_tile2_offspring(h, burn, id_base, nullptr, &p_req));
  GNXCHK(gnx_set_max_id(h, id_base - 1 + total_births));
  // gametes of ghost mates: requests out, gametes back
  if (w > 1 && !burn && geno) {
    int64_t n_req = 0, m = 0;
    for (int d = 0; d < w; ++d) n_req += m_req[(size_t)me * w + d];
    for (int s = 0; s < w; ++s) m += m_req[(size_t)s * w + me];
    Part rq{p_req, 24, RB_REQ};
    GNXCHK(exchange(h, &rq, 1, m_req.data()));
    void* p_out = nullptr;
    GNXCHK(gnx_tile2_serve(h, m, m ? c->rbuf[RB_REQ] : nullptr, &p_out));
    std::vector<int64_t> m_back((size_t)w * w);
    for (int s = 0; s < w; ++s)
      for (int d = 0; d < w; ++d) m_back[(size_t)s * w + d] = m_req[(size_t)d * w + s];
    Part gm{p_out, (size_t)8 * W64, RB_GAMETES};
    GNXCHK(exchange(h, &gm, 1, m_back.data()));
    if (n_req) GNXCHK(gnx_tile2_put(h, n_req, c->rbuf[RB_GAMETES]));
  }
  // the offspring that took a remote gamete: their alleles at the selected loci and their
  // phenotype from their finished rows (everybody else's came with the births)
  GNXCHK(gnx_tile2_settle_births(h, burn));
  clk.mark(3);
  c->mid = true;
  c->mid_pairs = total_pairs;
  c->mid_births = total_births;
  return 0;
}

// first id and number of the offspring of ALL tiles in the step between _begin and _end
extern "C" int gnx_tile_step_births(gnx_state* h, int64_t* first_id, int64_t* total) {
  Comm* c = comm_of(h);
  if (!c || !c->mid) {
    gnx_set_error("gnx_tile_step_births: between gnx_tile_step_begin and gnx_tile_step_end");
    return 1;
  }
  *total = c->mid_births;
  *first_id = h->max_id - c->mid_births + 1;
  return 0;
}

extern "C" int gnx_tile_step_end(gnx_state* h, int32_t burn, int32_t with_selection, int32_t exact,
                                 int64_t* out) {
  Comm* c = comm_of(h);
  if (!c || !c->mid) {
    gnx_set_error("gnx_tile_step_end: no step was begun (gnx_tile_step_begin)");
    return 1;
  }
  c->mid = false;
  const int w = c->world;
  PhaseClock clk(c);
  void* red = nullptr;
  int64_t n_words = 0;
  GNXCHK(gnx_tile2_finish_births(h, burn, &red, &n_words));
  // ONE all-reduce: both density fields and the counters
  GNXCHK(allreduce_i32(h, (int32_t*)red, n_words));
  // 3. densities, death probabilities, mortality (the survivor count is waited for)
  int64_t tot[3] = {0, 0, 0};
  GNXCHK(gnx_tile2_die(h, burn, with_selection, c->mid_pairs > 0 ? 1 : 0, tot));
  h->tot[2] += h->last_births;
  h->tot[3] += h->last_deaths;
  if (!burn) h->tot[4] += h->last_xo_births;
  h->step += 1;
  c->steps += 1;
  clk.mark(4);
  if (exact) {
    int64_t mine[3];
    GNXCHK(gnx_counts(h, &mine[0], &mine[1], &mine[2]));
    if (w > 1) {
      std::vector<int64_t> g((size_t)w * 3);
      GNXCHK(host_allgather(h, mine, 3, g.data()));
      mine[0] = mine[1] = mine[2] = 0;
      for (int r = 0; r < w; ++r)
        for (int k = 0; k < 3; ++k) mine[k] += g[(size_t)r * 3 + k];
    }
    out[0] = mine[0];
    out[1] = mine[1];
    out[2] = mine[2];
    c->pre = -1;
    return 0;
  }
  // (N at the start of this step = N before the previous step's deaths - those deaths)
  const int64_t n_pre = tot[0], b_glob = tot[1], d_prev = tot[2];
  out[0] = c->pre >= 0 ? c->pre - d_prev : n_pre - b_glob;
  out[1] = b_glob;
  out[2] = d_prev;
  c->pre = n_pre;
  return 0;
}

// T tiled steps in one call, nothing between them - the tiles' gnx_walk.  Between two of its
// steps the mortality leaves the dead in place (no compaction: gnx_internal.h, holes): the next
// step's movement and routing skip them, its imports go behind the uncompacted stretch and its
// cell sort - which already removes the emigrants by a key behind every cell - removes the dead
// by one more.  The last step compacts, so the population is dense whenever anybody can look.
// out[5]: (N at the start, births, deaths of the step before) of the LAST step as
// gnx_tile_step(exact = 0) reports them - or, exact != 0, the global (N, births, deaths) after
// it - then the SUM over the T steps of the global N at the start and of the global births.
// Reference: Model.walk -> _do_timestep T times (sim/model.py:966-1161) on every rank.
extern "C" int gnx_tile_walk(gnx_state* h, int64_t T, int32_t burn, int32_t with_selection,
                             int32_t exact, int64_t* out) {
  for (int k = 0; k < 5; ++k) out[k] = 0;
  int rc = 0;
  for (int64_t t = 0; t < T && !rc; ++t) {
    int64_t o[3] = {0, 0, 0};
    h->tile_lazy_ok = t + 1 < T;
    // (the per-step counts that ride on the step's own all-reduce: N at the start, births)
    rc = gnx_tile_step(h, burn, with_selection, 0, o);
    h->tile_lazy_ok = false;
    if (rc) break;
    out[0] = o[0];
    out[1] = o[1];
    out[2] = o[2];
    out[3] += o[0];
    out[4] += o[1];
  }
  if (!rc && exact && T > 0) {
    // the global (N, births, deaths) after the last step: one more KB-sized collective
    Comm* c = comm_of(h);
    int64_t mine[3];
    GNXCHK(gnx_counts(h, &mine[0], &mine[1], &mine[2]));
    if (c && c->world > 1) {
      std::vector<int64_t> g((size_t)c->world * 3);
      GNXCHK(host_allgather(h, mine, 3, g.data()));
      mine[0] = mine[1] = mine[2] = 0;
      for (int r = 0; r < c->world; ++r)
        for (int k = 0; k < 3; ++k) mine[k] += g[(size_t)r * 3 + k];
    }
    out[0] = mine[0];
    out[1] = mine[1];
    out[2] = mine[2];
    if (c) c->pre = -1;
  }
  return rc;
}

// A step that was begun and cannot be ended on every rank (the host's work on the newborns
// failed somewhere): the handle leaves the "between _begin and _end" state WITHOUT the density
// all-reduce and the mortality, so that the ranks can raise together instead of one of them
// waiting inside the all-reduce for ever.  The population holds this step's offspring and no
// deaths: the run is over, the handle can still be read and freed.
extern "C" int gnx_tile_step_abort(gnx_state* h) {
  Comm* c = comm_of(h);
  if (c) c->mid = false;
  return 0;
}

// What the communicator says about itself - for a bench line that certifies what it ran on:
// out[0] transport (0 one rank, no RCCL calls; 1 RCCL; 2 local = handles of one process),
// [1] rank, [2] world, [3] ncclCommCount, [4] ncclCommUserRank, [5] ncclCommCuDevice (each -1
// without an RCCL communicator), [6] the HIP device ordinal of the handle, [7] tiled steps taken,
// [8..12] host wall time of the step's five phases summed over those steps, microseconds
// (Comm::phase_s), [13] bytes this rank has sent, [14] collections of the genome blocks, [15] 0.
extern "C" int gnx_comm_info(gnx_state* h, int64_t* out) {
  for (int k = 0; k < 16; ++k) out[k] = 0;
  out[3] = out[4] = out[5] = -1;
  out[6] = h->cfg.device;
  out[14] = h->gc_runs;
  Comm* c = comm_of(h);
  if (!c) return 0;
  out[0] = c->kind;
  out[1] = c->rank;
  out[2] = c->world;
  if (c->nccl) {
    int v = -1;
    NCCLCHK(g_rccl.CommCount(c->nccl, &v));
    out[3] = v;
    NCCLCHK(g_rccl.CommUserRank(c->nccl, &v));
    out[4] = v;
    NCCLCHK(g_rccl.CommCuDevice(c->nccl, &v));
    out[5] = v;
  }
  out[7] = c->steps;
  for (int k = 0; k < 5; ++k) out[8 + k] = (int64_t)(c->phase_s[k] * 1e6);
  out[13] = c->bytes_sent;
  return 0;
}

extern "C" int gnx_tile_step(gnx_state* h, int32_t burn, int32_t with_selection, int32_t exact,
                             int64_t* out) {
  GNXCHK(gnx_tile_step_begin(h, burn));
  return gnx_tile_step_end(h, burn, with_selection, exact, out);
}
