// Genome kernels: bitmask crossover (the dominant byte mover of the step; kernels in
// gnx_xo.h), the compact table of alleles at the selected loci, phenotypes,
// starting genomes, point mutations, genome gathers.
//
// Genome layout in HBM: G[row][hom][W64] u64, little-endian bits, bit l of
// homologue h == Individual.g[l, h] (structs/individual.py:103-104).  W64 is
// padded to a multiple of 16 words so every homologue starts on a 128-byte line
// and splits into whole 16-byte chunks.  Individuals reference LOGICAL rows through
// GnxSoA.grow; where a row's two homologues live is the second indirection of
// gnx_half.h (hmap: logical block -> physical block, shared between a parent and the
// children that inherit the homologue unrecombined).  Half-rows are never moved.
#include "gnx_internal.h"
#include "gnx_rng.h"
#include "gnx_xo.h"
#include "gnx_half.h"
#include "gnx_tb.h"
#include "gnx_compact.h"

// ---------------------------------------------------------------- crossover jobs
// Every birth of the step at once: child k takes the k-th row from the top of the free
// stack; a gamete whose parent is a ghost (tiled run: the mate lives on a neighbour
// tile, its gamete arrives from there) gets prow = -1 and is skipped by the kernel.
__global__ void __launch_bounds__(256)
k_xo_jobs_all(int64_t B, int64_t first, int32_t* __restrict__ grow,
              const int32_t* __restrict__ off_parent, const int32_t* __restrict__ off_keys,
              const uint8_t* __restrict__ off_start, const int32_t* __restrict__ free_rows,
              int64_t n_free, GnxHalves H, const int32_t* __restrict__ bp_off,
              const int32_t* __restrict__ bp_loci, GnxXoJob* __restrict__ jobs,
              int32_t* __restrict__ n_jobs) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = k < B;
  int32_t row = -1;
  if (act) {
    row = free_rows[n_free - 1 - k];
    grow[first + k] = row;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int32_t prow = act ? grow[off_parent[2 * k + p]] : -1;
    gnx_xo_gamete(H, act, row, p, prow, act ? off_keys[2 * k + p] : 0,
                  act ? off_start[2 * k + p] : 0, bp_off, bp_loci, jobs, n_jobs);
  }
}

// (the job builder of the deferred mode, k_xo_jobs_surv, sits with the mortality kernels in
// gnx_kernels_demog.hip: it shares their block-rank compaction)
void gnx_launch_xo_jobs_surv(gnx_state* h, int64_t first_slot, const int32_t* d_alive,
                             const int32_t* d_blk_off, int buf);

template <int U>
static void xo_launch_sparse(gnx_state* h, hipStream_t st, int grid, int buf, bool nt, int lo,
                             int hi, unsigned long long* acc) {
  const int W16 = (h->BW > 0 ? h->BW : h->W64 / h->NB) / 2;       // chunks per block
  static const bool inline_env = !(getenv("GNX_XO_INLINE_BP") && atoi(getenv("GNX_XO_INLINE_BP")) == 0);
  const GnxJobBp* ib = (h->jobs_inline[buf] && inline_env) ? (const GnxJobBp*)h->jobs_bp[buf] : nullptr;
  // two jobs per wave and iteration (gnx_xo.h: k_xo_sparse_pair) when something runs beside
  // the crossover - the step's normal state: 0.625 against 0.636 ms/step, the launch 0.181
  // against 0.188 ms.  With the chip to itself (gnx_set_crossover_overlap(2)) one job per
  // iteration is the faster kernel (0.142 against 0.165 ms): GNX_XO_PAIR=0 / 2 force one.
  static const int pair_env = getenv("GNX_XO_PAIR") ? atoi(getenv("GNX_XO_PAIR")) : 1;
  if (U == 1 && W16 <= 64 && ib && (pair_env == 2 || (pair_env == 1 && h->xo_wait_at != 2))) {
    static const int kj = getenv("GNX_XO_GROUP") ? atoi(getenv("GNX_XO_GROUP")) : 2;
#define GNX_XO_PAIR_LAUNCH(NT, KK)                                                                \
  hipLaunchKernelGGL((k_xo_sparse_pair<NT, KK>), dim3(grid), dim3(256), 0, st, h->n_jobs_dev[buf], \
                     W16, (const u64x2*)h->G, (u64x2*)h->G, (const GnxXoJob*)h->jobs[buf],        \
                     h->bp_off, h->bp_loci, lo, hi, acc, ib)
    if (nt) {
      if (kj == 3) GNX_XO_PAIR_LAUNCH(true, 3);
      else if (kj == 4) GNX_XO_PAIR_LAUNCH(true, 4);
      else GNX_XO_PAIR_LAUNCH(true, 2);
    } else {
      GNX_XO_PAIR_LAUNCH(false, 2);
    }
#undef GNX_XO_PAIR_LAUNCH
    return;
  }
  if (nt)
    hipLaunchKernelGGL((k_xo_sparse<U, true>), dim3(grid), dim3(256), 0, st, h->n_jobs_dev[buf],
                       W16, (const u64x2*)h->G, (u64x2*)h->G, (const GnxXoJob*)h->jobs[buf],
                       h->bp_off, h->bp_loci, lo, hi, acc, ib);
  else
    hipLaunchKernelGGL((k_xo_sparse<U, false>), dim3(grid), dim3(256), 0, st, h->n_jobs_dev[buf],
                       W16, (const u64x2*)h->G, (u64x2*)h->G, (const GnxXoJob*)h->jobs[buf],
                       h->bp_off, h->bp_loci, lo, hi, acc, ib);
}

// the crossover of job buffer `buf` on stream `st` (the share [lo, hi) / 1024 of its jobs);
// max_jobs bounds the grid.  narrow: few workgroups per CU, for a launch that shares the
// chip with the step's small kernels.
static int xo_launch(gnx_state* h, hipStream_t st, int buf, int64_t max_jobs, bool narrow,
                     int lo = 0, int hi = 1024) {
  // one wave per gamete, 4 waves per block, job-strided beyond 32 blocks per CU
  // (measured: profiles/r02b_xo_lab_*.txt - time is flat in the grid size from 16 to 64
  // blocks per CU and in the unroll from 4 to 8; non-temporal loads +2 %); beside a whole
  // step of small kernels (GNX_XO_SORT_WAIT=0) 2 blocks per CU, 6 chunks in flight
  static const int bpc_env = getenv("GNX_XO_BPC") ? atoi(getenv("GNX_XO_BPC")) : 0;
  static const int unroll_env = getenv("GNX_XO_UNROLL") ? atoi(getenv("GNX_XO_UNROLL")) : 0;
  static const int nt = getenv("GNX_XO_NT") ? atoi(getenv("GNX_XO_NT")) : 1;
  // beside the whole next step: 2 workgroups per CU while a job was a whole homologue; with
  // half-homologue blocks 4 .. 16 measure alike (1.23-1.25 ms/step) and 2 loses 15 %
  static const int tail_bpc_env = getenv("GNX_XO_TAIL_BPC") ? atoi(getenv("GNX_XO_TAIL_BPC")) : 0;
  const int tail_bpc = tail_bpc_env ? tail_bpc_env : (h->NB > 1 ? 8 : 2);
  static const int tail_unroll = getenv("GNX_XO_TAIL_UNROLL") ? atoi(getenv("GNX_XO_TAIL_UNROLL")) : 6;
  const int bpc = narrow ? tail_bpc : (bpc_env ? bpc_env : 32);
  // the narrow share of a split launch is accounted for on its own
  unsigned long long* acc = h->xo_jobs_acc ? h->xo_jobs_acc + ((narrow && lo > 0) ? 1 : 0) : nullptr;
  const int W16 = (h->BW > 0 ? h->BW : h->W64 / h->NB) / 2;       // chunks per block
  const int grid = gnx_grid(max_jobs * h->NB, 4, 256 * bpc);
  if (h->sparse_paths) {
    // (blocks shorter than a homologue: as many loads in flight as the block has chunks)
    const int U = (narrow && h->NB == 1) ? tail_unroll
                                          : (unroll_env ? unroll_env : gnx_xo_pick_unroll(W16));
    switch (U) {
      case 1: xo_launch_sparse<1>(h, st, grid, buf, nt, lo, hi, acc); break;
      case 2: xo_launch_sparse<2>(h, st, grid, buf, nt, lo, hi, acc); break;
      case 3: xo_launch_sparse<3>(h, st, grid, buf, nt, lo, hi, acc); break;
      case 4: xo_launch_sparse<4>(h, st, grid, buf, nt, lo, hi, acc); break;
      case 5: xo_launch_sparse<5>(h, st, grid, buf, nt, lo, hi, acc); break;
      case 6: xo_launch_sparse<6>(h, st, grid, buf, nt, lo, hi, acc); break;
      case 7: xo_launch_sparse<7>(h, st, grid, buf, nt, lo, hi, acc); break;
      default: xo_launch_sparse<8>(h, st, grid, buf, nt, lo, hi, acc); break;
    }
  } else {
    // dense masks: genome chunks stream past once (non-temporal), the path table is
    // re-read by every gamete that drew the key and stays in L2 / the Infinity Cache
    if (nt)
      hipLaunchKernelGGL((k_xo_dense<4, true>), dim3(grid), dim3(256), 0, st, h->n_jobs_dev[buf],
                         W16, (const u64x2*)h->G, (u64x2*)h->G, (const GnxXoJob*)h->jobs[buf],
                         (const u64x2*)h->paths, h->W64 / 2, lo, hi, acc);
    else
      hipLaunchKernelGGL((k_xo_dense<4, false>), dim3(grid), dim3(256), 0, st, h->n_jobs_dev[buf],
                         W16, (const u64x2*)h->G, (u64x2*)h->G, (const GnxXoJob*)h->jobs[buf],
                         (const u64x2*)h->paths, h->W64 / 2, lo, hi, acc);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// device-driven step (gnx_dd.hip): the crossover of job buffer `buf`, job count on the device
int gnx_dd_l_crossover(gnx_state* h, int buf, hipStream_t st) {
  // (at most every second individual is a parent of a kept pair: half the capacity in births,
  // two gametes each; the kernel is job-strided, the bound only sizes the grid)
  return xo_launch(h, st, buf, h->cfg.cap_inds / 4, false);
}

// algorithmic bytes per birth whose two gametes are both copied.  Dense masks:
// 4 parental homologues + 2 masks read, 2 homologues written = 8 * L/8 = L bytes.  Sparse
// paths: each gamete chunk copies ONE parental homologue (the other is never needed, the
// mask comes from a handful of breakpoints): 2 reads + 2 writes = 4 * L/8 = L/2 bytes per
// birth (padded row width W64*8 is what actually moves).  A gamete without a switch point
// moves nothing (gnx_half.h); the kernels count the gametes they copy (xo_jobs_acc).
double gnx_xo_bytes_per_birth(const gnx_state* h) {
  return (h->sparse_paths ? 4.0 : 8.0) * (double)h->W64 * 8.0;
}

// stream `st` waits until no crossover reads or writes job buffer `buf` any more
static int xo_wait_buf(gnx_state* h, hipStream_t st, int buf) {
  if (h->xo_inflight[buf]) {
    HIPCHK(hipStreamWaitEvent(st, h->ev_xo_done[buf], 0));
    h->xo_inflight[buf] = false;
    h->xo_wide_inflight[buf] = false;
  }
  return 0;
}

int gnx_l_crossover_all(gnx_state* h, int64_t first_slot, int64_t B) {
  if (B == 0) return 0;
  GNXCHK(gnx_xo_launch_pending(h));
  // rows written here may be parents' rows of a crossover still running on stream2
  GNXCHK(xo_wait_buf(h, h->stream, 0));
  GNXCHK(xo_wait_buf(h, h->stream, 1));
  const int buf = h->jobs_cur;
  GnxSoA s = h->soa[h->cur];
  GNXCHK(gnx_half_reserve(h, 2 * (int64_t)h->NB * B));
  HIPCHK(hipMemsetAsync(h->n_jobs_dev[buf], 0, sizeof(int32_t), h->stream));
  h->jobs_inline[buf] = false;
  hipLaunchKernelGGL(k_xo_jobs_all, dim3(gnx_grid(B, 256)), dim3(256), 0, h->stream, B, first_slot,
                     s.grow, h->off_parent, h->off_keys, h->off_start, h->free_rows, h->n_free,
                     gnx_halves(h), gnx_alias_bp(h), gnx_alias_loci(h), (GnxXoJob*)h->jobs[buf],
                     h->n_jobs_dev[buf]);
  if (h->stream2) {
    // tiled runs serve the neighbours' gamete requests on stream2 meanwhile: the rows
    // handed out above must be visible there
    HIPCHK(hipEventRecord(h->ev_jobs, h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_jobs, 0));
  }
  gnx_time_begin(h, GNX_K_CROSSOVER);
  GNXCHK(xo_launch(h, h->stream, buf, 2 * B, false));
  gnx_time_end(h, GNX_K_CROSSOVER, 0.0);   // bytes: the kernels count the gametes they copy
  h->n_free -= B;
  h->last_xo_births = B;
  return 0;
}

// Tiled runs: the offspring whose mate is a ghost (one gamete request each, req_k) get
// their row and the crossover of their LOCAL gamete at once - the remote gamete is put
// next to it and their alleles at the selected loci are read from the finished row -
// while every other offspring of the step waits for the death draws like on one GPU.
// (One atomic per wave for the fresh blocks and one for the jobs, every table entry loaded before
// the first is stored: the generic per-block helpers - gnx_xo_gamete, gnx_half_new - take two
// atomic round trips per block, 2 NB = 40 blocks per offspring one after the other: 55 us on the
// tile step's chain for ~2 000 offspring.)
__global__ void __launch_bounds__(256)
k_xo_jobs_req(int n_req, int64_t first, int32_t* __restrict__ grow,
              const int32_t* __restrict__ req_k, const int32_t* __restrict__ off_parent,
              const int32_t* __restrict__ off_keys, const uint8_t* __restrict__ off_start,
              const int32_t* __restrict__ free_rows, int64_t n_free, GnxHalves H,
              const int32_t* __restrict__ bp_off, const int32_t* __restrict__ bp_loci,
              GnxXoJob* __restrict__ jobs, int32_t* __restrict__ n_jobs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool act = q < n_req;
  const int NB = H.NB;
  const int64_t k = act ? req_k[q] : 0;
  int32_t row = -1, prow = -1;
  int key = 0, st = 0;
  if (act) {
    row = free_rows[n_free - 1 - q];
    grow[first + k] = row;
    prow = grow[off_parent[2 * k]];
    key = off_keys[2 * k];
    st = off_start[2 * k];
  }
  // the local parent's gamete (homologue 0): the blocks that hold a switch point are cut, the
  // others refer to the parent's; empty blocks for the gamete that arrives from the neighbour tile
  const bool local = act && prow >= 0;
  const unsigned int all = (NB >= 32) ? ~0u : ((1u << NB) - 1u);
  unsigned int mixed = all, sel = 0u;
  if (local && bp_off)
    gnx_block_masks(bp_loci + bp_off[key], bp_off[key + 1] - bp_off[key], st, NB, H.BW, mixed, sel);
  mixed &= all;
  const int nf0 = act ? __popc(mixed) : 0;
  const int nf = act ? nf0 + NB : 0, nj = local ? nf0 : 0;
  int xf = nf, xj = nj;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int yf = __shfl_up(xf, d), yj = __shfl_up(xj, d);
    if (lane >= d) {
      xf += yf;
      xj += yj;
    }
  }
  const int tf = __shfl(xf, 63), tj = __shfl(xj, 63);
  int top = 0, jb = 0;
  if (lane == 63) {
    top = tf ? atomicSub(H.top, tf) : 0;
    jb = tj ? atomicAdd(n_jobs, tj) : 0;
  }
  top = __shfl(top, 63);
  jb = __shfl(jb, 63);
  if (!act) return;
  const int pop = top - 1 - (xf - nf);          // my fresh blocks: stack[pop], stack[pop - 1], ...
  int job = jb + (xj - nj);
  // everything this thread reads, before it writes anything
  int32_t pe0[GNX_MAX_NB], pe1[GNX_MAX_NB], fb[2 * GNX_MAX_NB];
#pragma unroll
  for (int b = 0; b < GNX_MAX_NB; ++b) {
    pe0[b] = (local && b < NB) ? H.hmap[((int64_t)prow * 2) * NB + b] : 0;
    pe1[b] = (local && b < NB) ? H.hmap[((int64_t)prow * 2 + 1) * NB + b] : 0;
  }
#pragma unroll
  for (int r = 0; r < 2 * GNX_MAX_NB; ++r) fb[r] = r < nf ? H.stack[pop - r] : 0;
  int fr = 0;
#pragma unroll
  for (int b = 0; b < GNX_MAX_NB; ++b) {
    const int64_t lb = ((int64_t)row * 2) * NB + b;
    if (b >= NB) {
      // (blocks the layout does not have)
    } else if ((mixed >> b) & 1u) {
      int32_t dst = fb[0];
#pragma unroll
      for (int r = 1; r < GNX_MAX_NB; ++r) dst = fr == r ? fb[r] : dst;
      ++fr;
      H.hmap[lb] = (int32_t)((uint32_t)dst | GNX_OWN);
      if (local) {
        GnxXoJob jrec;
        jrec.ph0 = GNX_BLK(pe0[b]);
        jrec.ph1 = GNX_BLK(pe1[b]);
        jrec.dst = dst;
        jrec.ks = (key * 2 + st) | (b << 24);
        jobs[job++] = jrec;
      }
    } else {
      const int hsel = (sel >> b) & 1u;
      const int32_t e = hsel ? pe1[b] : pe0[b];
      H.hmap[lb] = GNX_BLK(e);
      if (e < 0) H.hmap[((int64_t)prow * 2 + hsel) * NB + b] = GNX_BLK(e);   // shared from now on
    }
  }
  for (int b = 0; b < NB; ++b)
    H.hmap[((int64_t)row * 2 + 1) * NB + b] = (int32_t)((uint32_t)H.stack[pop - nf0 - b] | GNX_OWN);
}

int gnx_l_crossover_requests(gnx_state* h, int64_t first_slot, int64_t n_req) {
  if (n_req == 0) return 0;
  GNXCHK(gnx_xo_launch_pending(h));
  GNXCHK(xo_wait_buf(h, h->stream, 0));
  GNXCHK(xo_wait_buf(h, h->stream, 1));
  const int buf = h->jobs_cur;
  GNXCHK(gnx_half_reserve(h, 2 * (int64_t)h->NB * n_req));
  HIPCHK(hipMemsetAsync(h->n_jobs_dev[buf], 0, sizeof(int32_t), h->stream));
  h->jobs_inline[buf] = false;
  hipLaunchKernelGGL(k_xo_jobs_req, dim3(gnx_grid(n_req, 256)), dim3(256), 0, h->stream, (int)n_req,
                     first_slot, h->soa[h->cur].grow, h->req_k, h->off_parent, h->off_keys,
                     h->off_start, h->free_rows, h->n_free, gnx_halves(h), gnx_alias_bp(h),
                     gnx_alias_loci(h), (GnxXoJob*)h->jobs[buf], h->n_jobs_dev[buf]);
  if (h->stream2) {      // the rows handed out above must be visible to the gamete puts
    HIPCHK(hipEventRecord(h->ev_jobs, h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_jobs, 0));
  }
  gnx_time_begin(h, GNX_K_CROSSOVER);
  GNXCHK(xo_launch(h, h->stream, buf, n_req, false));
  gnx_time_end(h, GNX_K_CROSSOVER, 0.0);
  h->n_free -= n_req;
  return 0;
}

// Offspring [first, first + B) that have no genome row yet get one, and their crossover,
// now, on `stream` (somebody needs the genomes before the step's death draws).
__global__ void k_pending_flags(int64_t N, int64_t first, const int32_t* grow, int32_t* alive,
                                int32_t* cnt3) {
  __shared__ int lds[16];
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool fx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    fx[r] = i < N && i >= first && grow[i] < 0;
    if (i < N) alive[i] = fx[r] ? 2 : 0;
  }
  int rank[4], tot;
  gnx_block_ranks(fx, rank, tot, lds);
  if (threadIdx.x == 0) cnt3[blockIdx.x] = tot;
}

int gnx_l_crossover_pending(gnx_state* h, int64_t first_slot, int64_t B) {
  if (B == 0) return 0;
  GNXCHK(gnx_xo_launch_pending(h));
  GNXCHK(xo_wait_buf(h, h->stream, 0));
  GNXCHK(xo_wait_buf(h, h->stream, 1));
  const int64_t N = h->N;
  const int nb = (int)((N + GNX_CB - 1) / GNX_CB);
  const int buf = h->jobs_cur;
  int32_t* cnt3 = h->blk_cnt + 2 * h->blk_stride;
  int32_t* off3 = h->blk_off + 2 * h->blk_stride;
  hipLaunchKernelGGL(k_pending_flags, dim3(nb), dim3(256), 0, h->stream, N, first_slot,
                     h->soa[h->cur].grow, h->flag, cnt3);
  // totals land in cnt_dev[2] (what k_xo_jobs_surv reads) and in pinned memory
  GNXCHK(gnx_block_scan(h, 1, N, cnt3, off3, h->cnt_dev + 2, h->h_pin_dev + 8));
  GNXCHK(gnx_half_reserve(h, 2 * (int64_t)h->NB * B));
  HIPCHK(hipMemsetAsync(h->n_jobs_dev[buf], 0, sizeof(int32_t), h->stream));
  gnx_launch_xo_jobs_surv(h, first_slot, h->flag, h->blk_off, buf);
  gnx_time_begin(h, GNX_K_CROSSOVER);
  GNXCHK(xo_launch(h, h->stream, buf, 2 * B, false));
  HIPCHK(hipStreamSynchronize(h->stream));
  const int64_t S = h->h_pin[8];
  gnx_time_end(h, GNX_K_CROSSOVER, 0.0);
  h->n_free -= S;
  h->last_xo_births = S;
  return 0;
}

int gnx_xo_launch_pending(gnx_state* h) {
  const int buf = h->xo_ready_buf;
  if (buf < 0) return 0;
  h->xo_ready_buf = -1;
  // behind everything `stream` has been given so far (the jobs, and with launch policy
  // 1 / 2 the sort that is meant to run alone)
  HIPCHK(hipEventRecord(h->ev_jobs, h->stream));
  HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_jobs, 0));
  hipStream_t main = h->stream;
  h->stream = h->stream2;                        // the timer events go where the kernel goes
  // Split launch (xo_split / 1024 of the jobs): the first share at full width, alone on the
  // chip but for the compaction and the next step's movement - the cell sort waits for it;
  // the rest narrow, beside the sort, the mate search, the density and the death draws,
  // which are latency-bound chains and leave the memory system idle.
  const int split = (h->xo_sort_waits && h->xo_split > 0 && h->xo_split < 1024) ? h->xo_split : 0;
  h->xo_last_split = split;
  gnx_time_begin(h, GNX_K_CROSSOVER);
  int rc = xo_launch(h, h->stream2, buf, h->xo_ready_jobs, !h->xo_sort_waits, 0,
                     split ? split : 1024);
  gnx_time_end(h, GNX_K_CROSSOVER, 0.0);
  if (split && !rc) {
    HIPCHK(hipEventRecord(h->ev_xo_wide[buf], h->stream2));
    h->xo_wide_inflight[buf] = true;
    gnx_time_begin(h, GNX_K_CROSSOVER);
    rc = xo_launch(h, h->stream2, buf, h->xo_ready_jobs, true, split, 1024);
    gnx_time_end(h, GNX_K_CROSSOVER_TAIL, 0.0);
  }
  h->stream = main;
  GNXCHK(rc);
  HIPCHK(hipEventRecord(h->ev_xo_done[buf], h->stream2));
  h->xo_inflight[buf] = true;
  h->xo_running = true;
  return 0;
}

int gnx_xo_wait_inflight(gnx_state* h) {
  GNXCHK(xo_wait_buf(h, h->stream, 0));
  GNXCHK(xo_wait_buf(h, h->stream, 1));
  return 0;
}

// what the cell sort waits for: the full-width share of a split launch (its narrow share
// runs on beside the sort), or the whole crossover
int gnx_xo_wait_wide(gnx_state* h) {
  for (int buf = 0; buf < 2; ++buf) {
    if (h->xo_wide_inflight[buf]) {
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_xo_wide[buf], 0));
      h->xo_wide_inflight[buf] = false;
    } else if (!h->xo_last_split) {
      GNXCHK(xo_wait_buf(h, h->stream, buf));
    }
  }
  return 0;
}

// before the death draws of a step with a deferred crossover: the job buffer that
// gnx_l_crossover_survivors will fill is free, *zero = its counter (k_alive clears it)
int gnx_xo_prepare_jobs(gnx_state* h, int32_t** zero) {
  GNXCHK(gnx_xo_launch_pending(h));              // at most one set of jobs waits
  const int buf = h->jobs_cur;
  GNXCHK(xo_wait_buf(h, h->stream, buf));       // the crossover two steps back read this buffer
  *zero = h->n_jobs_dev[buf];
  return 0;
}

int gnx_l_crossover_survivors(gnx_state* h, int64_t first_slot, int64_t B, const int32_t* d_alive,
                              const int32_t* d_scan) {
  if (B == 0) return 0;
  const int buf = h->jobs_cur;                   // prepared by gnx_xo_prepare_jobs
  gnx_launch_xo_jobs_surv(h, first_slot, d_alive, d_scan, buf);
  HIPCHK(hipGetLastError());
  // (gnx_xo_launch_pending records the event the crossover's stream waits for - one record,
  // not two: every event operation costs the recording stream 3 - 6 us)
  h->xo_ready_buf = buf;
  h->xo_ready_jobs = 2 * B;
  h->jobs_cur ^= 1;
  // (tiles: the next step's routing reads the migrants' genome rows first thing - gnx_xo_join - so a
  // crossover held back would only be waited for there: at once)
  if (h->xo_launch_policy == 0 || h->tiled || h->tile2_mode) GNXCHK(gnx_xo_launch_pending(h));
  return 0;
}

int gnx_xo_flush_deferred(gnx_state* h) {
  if (h->xo_deferred) {
    GNXCHK(gnx_xo_launch_pending(h));
    // somebody needs this step's genomes before its death draws: every pending offspring
    // gets its row and its crossover now (same genomes as the deferred path would give
    // the survivors; the dead's rows return to the free stack with the other deaths)
    h->xo_deferred = false;
    GNXCHK(gnx_l_crossover_pending(h, h->xo_first, h->xo_B));
  }
  return 0;
}

int gnx_xo_join(gnx_state* h) {
  GNXCHK(gnx_xo_flush_deferred(h));
  GNXCHK(gnx_xo_launch_pending(h));
  GNXCHK(xo_wait_buf(h, h->stream, 0));
  GNXCHK(xo_wait_buf(h, h->stream, 1));
  return 0;
}

// ---------------------------------------------------------------- selected-locus table
// path_sel[k] bit e = homologue path k is on at selected locus e
__global__ void k_path_sel(int n_paths, int TW, int n_sel, int W64, const int32_t* sel_loci,
                           const u64* paths, u64* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_paths * TW) return;
  const int k = idx / TW, w = idx - k * TW;
  u64 v = 0;
  for (int e = w * 64; e < min(n_sel, w * 64 + 64); ++e) {
    const int l = sel_loci[e];
    v |= ((paths[(int64_t)k * W64 + (l >> 6)] >> (l & 63)) & 1ull) << (e & 63);
  }
  out[idx] = v;
}

int gnx_l_path_sel(gnx_state* h) {
  (void)hipFree(h->path_sel);
  h->path_sel = nullptr;
  if (h->n_paths == 0 || h->TW == 0) return 0;
  HIPCHK(hipMalloc((void**)&h->path_sel, (size_t)h->n_paths * h->TW * 8));
  const int n = h->n_paths * h->TW;
  hipLaunchKernelGGL(k_path_sel, dim3(gnx_grid(n, 128)), dim3(128), 0, h->stream, h->n_paths, h->TW,
                     h->n_sel, h->W64, h->sel_loci, (const u64*)h->paths, (u64*)h->path_sel);
  HIPCHK(hipGetLastError());
  return 0;
}

// tb of a slot from its genome row: one WAVE per (slot, homologue, 64-locus word of the table),
// a lane per selected locus - two dependent loads per lane (block table, genome word) and a
// ballot that IS the word.  (Round 4: one thread per (slot, homologue) walking its n_sel loci one
// after the other, 2 x n_sel dependent round trips: 76 us for the 2 000 offspring of a tile
// step that took a neighbour's gamete, 0.14 ms per tile and step - profiles/r05_ab_runs.txt.)
__global__ void __launch_bounds__(256)
k_tb_from_rows(int64_t first, int64_t n, const int32_t* list, const int64_t* slots,
               int TW, int n_sel, int W64, const int32_t* sel_loci, const u64* G,
               const int32_t* grow, GnxHalves H, u64* tb) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wv >= 2 * n * TW) return;                       // (wave-uniform)
  const int w = (int)(wv % TW);
  const int64_t t = wv / TW;
  const int64_t q = t >> 1;
  const int hom = (int)(t & 1);
  const int64_t slot = slots ? slots[q] : first + (list ? list[q] : q);
  const int32_t row = grow[slot];
  if (row < 0) return;
  const int64_t lh = (int64_t)row * 2 + hom;
  const int e = w * 64 + lane;
  bool bit = false;
  if (e < n_sel) {
    const int l = sel_loci[e];
    bit = ((G[gnx_word_at(H, lh, l >> 6)] >> (l & 63)) & 1ull) != 0ull;
  }
  const u64 v = __ballot(bit);
  if (lane == 0) tb[(slot * 2 + hom) * TW + w] = v;
}

int gnx_l_tb_from_rows(gnx_state* h, int64_t first, int64_t n, const int32_t* d_list,
                       const int64_t* d_slots, bool join) {
  if (n == 0 || h->TW == 0 || !h->genomes_assigned) return 0;
  if (join) GNXCHK(gnx_xo_join(h));
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL(k_tb_from_rows, dim3(gnx_grid(2 * n * h->TW * 64, 256, 1 << 30)), dim3(256), 0,
                     h->stream, first, n, d_list, d_slots, h->TW, h->n_sel, h->W64, h->sel_loci,
                     (const u64*)h->G, s.grow, gnx_halves(h), (u64*)s.tb);
  HIPCHK(hipGetLastError());
  return 0;
}

// tb of offspring k, homologue p = the gamete of parent p (gnx_gamete_tb).  A ghost
// parent's gamete (tiled run) is filled in once it has arrived.
__global__ void k_newborn_tb(int64_t B, int64_t first, int TW, const int32_t* off_parent,
                             const int32_t* off_keys, const uint8_t* off_start,
                             const uint8_t* ghost, const u64* path_sel, u64* tb) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * B) return;
  const int64_t ps = off_parent[t];
  if (ghost[ps]) return;
  const int64_t k = t >> 1;
  const int p = (int)(t & 1);
  gnx_gamete_tb(TW, (const uint64_t*)tb + ps * 2 * TW,
                (const uint64_t*)path_sel + (int64_t)off_keys[t] * TW, off_start[t] != 0,
                (uint64_t*)tb + ((first + k) * 2 + p) * TW);
}

int gnx_l_newborn_tb(gnx_state* h, int64_t first_slot, int64_t B) {
  if (B == 0 || h->TW == 0) return 0;
  GnxSoA s = h->soa[h->cur];
  hipLaunchKernelGGL(k_newborn_tb, dim3(gnx_grid(2 * B, 256)), dim3(256), 0, h->stream, B,
                     first_slot, h->TW, h->off_parent, h->off_keys, h->off_start, s.ghost,
                     (const u64*)h->path_sel, (u64*)s.tb);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- phenotype
// (gnx_phenotype_tb) of slots [first, first + n)
__global__ void k_phenotype(int64_t first, int64_t n, int64_t cap, int TW, const u64* tb,
                            GnxTraitTab T, const uint8_t* dom, float* z, const int32_t* list) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int64_t slot = first + (list ? list[k] : k);
  const uint64_t* t0 = (const uint64_t*)tb + (slot * 2 + 0) * TW;
  gnx_phenotype_tb(t0, t0 + TW, T, dom, cap, slot, z);
}

// d_list: the n slots first_slot + d_list[k] instead of n slots in a row
int gnx_l_phenotype(gnx_state* h, int64_t first_slot, int64_t n, const int32_t* d_list) {
  if (n == 0 || h->cfg.n_traits == 0) return 0;
  GnxSoA s = h->soa[h->cur];
  gnx_time_begin(h);
  hipLaunchKernelGGL(k_phenotype, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, first_slot, n,
                     h->cfg.cap_inds, h->TW, (const u64*)s.tb, gnx_trait_tab(h), h->dom, s.z, d_list);
  gnx_time_end(h, GNX_K_PHENOTYPE, (double)n * (16.0 * h->TW + 4.0 * h->cfg.n_traits));
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
k_assign_genomes(int64_t N, int L, int W64, u64* G, const int32_t* grow, GnxHalves H,
                 const int32_t* n_per_site, unsigned long long site_seed,
                 const int32_t* __restrict__ order) {
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
    // homologue q belongs to the (q / 2)-th individual in ID order (the order the reference
    // walks its individuals in, structs/genome.py:1132-1133): slot order means nothing
    const int64_t slot = order ? order[q >> 1] : (q >> 1);
    if (lane == 0) G[gnx_word_at(H, (int64_t)grow[slot] * 2 + (q & 1), (int)word)] = w;
  }
}

__global__ void k_iota32(int64_t N, int32_t* v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) v[i] = (int32_t)i;
}

__global__ void k_assign_rows(int64_t N, int64_t cap_rows, int spread, int32_t* grow,
                              int32_t* free_rows, GnxHalves H) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    const int32_t row = (int32_t)(i * spread);
    grow[i] = row;
    // every individual starts with its own blocks, at its logical row's address
    for (int q = 0; q < 2 * H.NB; ++q) {
      H.hmap[(int64_t)row * 2 * H.NB + q] = (int32_t)((uint32_t)(row * 2 * H.NB + q) | GNX_OWN);
    }
  }
  // free stacks: rows N..cap_rows-1 (row numbers are physical: x spread) and their blocks,
  // popped from the top (highest first)
  if (i < cap_rows - N) {
    const int32_t row = (int32_t)((cap_rows - 1 - i) * spread);
    free_rows[i] = row;
    for (int q = 0; q < 2 * H.NB; ++q)
      H.stack[2 * H.NB * i + q] = row * 2 * H.NB + (2 * H.NB - 1 - q);
  }
  if (i == 0) *H.top = (int32_t)(2 * H.NB * (cap_rows - N));
}

int gnx_l_assign_genomes(gnx_state* h, const int32_t* d_n_per_site) {
  const gnx_config& c = h->cfg;
  int64_t N = h->N;
  if (N > c.cap_rows) {
    gnx_set_error("cap_rows %lld < N %lld", (long long)c.cap_rows, (long long)N);
    return 2;
  }
  GNXCHK(gnx_xo_join(h));
  GnxSoA s = h->soa[h->cur];
  int64_t m = N > c.cap_rows - N ? N : c.cap_rows - N;
  hipLaunchKernelGGL(k_assign_rows, dim3(gnx_grid(m, 256)), dim3(256), 0, h->stream, N, c.cap_rows,
                     h->row_spread, s.grow, h->free_rows, gnx_halves(h));
  h->n_free = c.cap_rows - N;
  h->half_free_est = 2 * (int64_t)h->NB * (c.cap_rows - N);
  if (N > 0 && d_n_per_site) {
    // the slots in id order (a tile keeps its slot order: its stepper assigns by global rank)
    const int32_t* order = nullptr;
    if (!h->tiled) {
      int idbits = 1;
      while (idbits < 40 && (h->max_id >> idbits) != 0) ++idbits;
      hipLaunchKernelGGL(k_iota32, dim3(gnx_grid(N, 256)), dim3(256), 0, h->stream, N, h->perm[0]);
      GNXCHK(gnx_prim_sort64_bits(h->sort64_tmp, h->sort64_tmp_bytes, (const uint64_t*)s.id,
                                  h->key64[0], h->perm[0], h->os_vtmp, (size_t)N, idbits,
                                  h->stream, true));
      order = h->os_vtmp;
    }
    int64_t threads = (int64_t)h->W64 * 64;
    hipLaunchKernelGGL(k_assign_genomes, dim3(gnx_grid(threads, 256)), dim3(256), 0, h->stream, N,
                       c.L, h->W64, (u64*)h->G, s.grow, gnx_halves(h), d_n_per_site,
                       gnx_site_seed(c.seed), order);
  }
  HIPCHK(hipGetLastError());
  h->genomes_assigned = true;
  return 0;
}

// ---------------------------------------------------------------- mutation
// ops/mutation.py:62-131: set allele 1 at (locus, homologue) of the chosen
// offspring.
// A block cut for the mutated individual and never shared takes the bit in place; one that
// is or was shared (an unrecombined stretch of a parent's, or a block a child refers to)
// goes on a list, and one workgroup walks that list in order, copying first.  (The copy is
// the individual's own from then on; what it replaced is the collector's business.)
__global__ void k_mutate(int n, u64* G, const int32_t* grow, GnxHalves H, const int64_t* slot,
                         const int32_t* locus, const uint8_t* hom, int32_t* list,
                         int32_t* n_list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool shared = false;
  if (i < n) {
    const int l = locus[i];
    const int w = l >> 6, b = w / H.BW;
    const int32_t e = H.hmap[((int64_t)grow[slot[i]] * 2 + hom[i]) * H.NB + b];
    shared = e >= 0;
    if (!shared) atomicOr(G + (int64_t)GNX_BLK(e) * H.BW + (w - b * H.BW), 1ull << (l & 63));
  }
  const int32_t idx = gnx_wave_append(n_list, shared);
  if (shared) list[idx] = i;
}

__global__ void __launch_bounds__(256)
k_mutate_shared(const int32_t* list, const int32_t* n_list, u64* G, const int32_t* grow,
                GnxHalves H, const int64_t* slot, const int32_t* locus, const uint8_t* hom) {
  __shared__ int32_t s_old, s_new;
  const int n = *n_list;
  for (int t = 0; t < n; ++t) {
    const int i = list[t];
    const int l = locus[i];
    const int w = l >> 6, b = w / H.BW;
    if (threadIdx.x == 0) {
      const int64_t lb = ((int64_t)grow[slot[i]] * 2 + hom[i]) * H.NB + b;
      const int32_t e = H.hmap[lb];
      s_old = s_new = GNX_BLK(e);
      if (e >= 0) {                // (an earlier mutation of this list may have copied it already)
        const int32_t q = H.stack[atomicSub(H.top, 1) - 1];
        H.hmap[lb] = (int32_t)((uint32_t)q | GNX_OWN);
        s_new = q;
      }
    }
    __syncthreads();
    const int32_t po = s_old, pn = s_new;
    if (pn != po)
      for (int x = threadIdx.x; x < H.BW; x += 256)
        G[(int64_t)pn * H.BW + x] = G[(int64_t)po * H.BW + x];
    __syncthreads();
    if (threadIdx.x == 0) G[(int64_t)pn * H.BW + (w - b * H.BW)] |= 1ull << (l & 63);
    __syncthreads();
  }
}

int gnx_l_mutate(gnx_state* h, int n, const int64_t* d_slot, const int32_t* d_locus,
                 const uint8_t* d_hom) {
  if (n == 0) return 0;
  GNXCHK(gnx_xo_join(h));
  int32_t* list = nullptr;
  HIPCHK(hipMalloc((void**)&list, ((size_t)n + 1) * sizeof(int32_t)));
  int32_t* n_list = list + n;
  HIPCHK(hipMemsetAsync(n_list, 0, sizeof(int32_t), h->stream));
  hipLaunchKernelGGL(k_mutate, dim3(gnx_grid(n, 256)), dim3(256), 0, h->stream, n, (u64*)h->G,
                     h->soa[h->cur].grow, gnx_halves(h), d_slot, d_locus, d_hom, list, n_list);
  // the mutations that hit a shared block need a copy each at most
  int32_t n_shared = 0;
  int rc = gnx_d2h(h, &n_shared, n_list, sizeof(n_shared));
  if (!rc && n_shared > 0) rc = gnx_half_reserve(h, n_shared);
  if (rc) {
    (void)hipFree(list);
    return rc;
  }
  if (n_shared > 0)
    hipLaunchKernelGGL(k_mutate_shared, dim3(1), dim3(256), 0, h->stream, (const int32_t*)list,
                       (const int32_t*)n_list, (u64*)h->G, h->soa[h->cur].grow, gnx_halves(h),
                       d_slot, d_locus, d_hom);
  HIPCHK(hipStreamSynchronize(h->stream));
  (void)hipFree(list);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- genome gather
__global__ void k_gather_genomes(int64_t n, int W16, const u64x2* G, const int32_t* grow,
                                 GnxHalves H, const int64_t* slots, u64x2* out) {
  const int64_t total = n * 2 * (int64_t)W16;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    int64_t k = g / (2 * W16);
    int64_t c = g - k * 2 * W16;
    int64_t slot = slots ? slots[k] : k;
    const int hh = c >= W16 ? 1 : 0;
    out[g] = G[gnx_chunk_at(H, (int64_t)grow[slot] * 2 + hh, (int)(c - hh * W16))];
  }
}

__global__ void k_scatter_genomes(int64_t n, int W16, const u64x2* in, u64x2* G,
                                  const int32_t* grow, GnxHalves H, int64_t first_slot) {
  const int64_t total = n * 2 * (int64_t)W16;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    int64_t k = g / (2 * W16);
    int64_t c = g - k * 2 * W16;
    const int hh = c >= W16 ? 1 : 0;
    G[gnx_chunk_at(H, (int64_t)grow[first_slot + k] * 2 + hh, (int)(c - hh * W16))] = in[g];
  }
}

int gnx_l_scatter_genomes(gnx_state* h, int64_t n, const uint64_t* d_in, int64_t first_slot) {
  if (n == 0) return 0;
  GNXCHK(gnx_xo_join(h));
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_scatter_genomes, dim3(gnx_grid(n * 2 * W16, 256, 256 * 32)), dim3(256), 0,
                     h->stream, n, W16, (const u64x2*)d_in, (u64x2*)h->G, h->soa[h->cur].grow,
                     gnx_halves(h), first_slot);
  HIPCHK(hipGetLastError());
  return 0;
}

int gnx_l_gather_genomes(gnx_state* h, int64_t n, const int64_t* d_slots, uint64_t* d_out) {
  if (n == 0) return 0;
  GNXCHK(gnx_xo_join(h));
  const int W16 = h->W64 / 2;
  hipLaunchKernelGGL(k_gather_genomes, dim3(gnx_grid(n * 2 * W16, 256, 256 * 32)), dim3(256), 0,
                     h->stream, n, W16, (const u64x2*)h->G, h->soa[h->cur].grow, gnx_halves(h),
                     d_slots, (u64x2*)d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- block collector
// gnx_gc: the blocks some living individual's table points at are marked, everything else
// goes (back) on the free stack.  Individuals whose crossover is still to come have no row
// yet; the dead of earlier steps are gone from the slots; a crossover in flight on stream2
// writes blocks of the living and reads blocks that, at worst, go on the stack now and are
// written again by a later crossover behind it on the same stream.
__global__ void k_gc_mark(int64_t N, const int32_t* __restrict__ grow, GnxHalves H,
                          uint8_t* __restrict__ mark, const GnxDD* __restrict__ dd) {
  N = gnx_dd_n(dd, N);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int per = 2 * H.NB;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * per; t += stride) {
    const int64_t i = t / per;
    const int32_t row = grow[i];
    if (row >= 0) mark[GNX_BLK(H.hmap[(int64_t)row * per + (t - i * per)])] = 1;
  }
}

// the sweep is an order-preserving compaction of the unmarked block numbers (gnx_compact.h:
// count per 1024 items, scan of the counts by one workgroup, write) - one atomic per wave
// on the stack height took 1.4 ms for the 8 x 10^6 blocks of the metric workload
__device__ __forceinline__ int32_t gc_block_id(int64_t t, int per, int spread) {
  const int64_t r = t / per;
  return (int32_t)(r * spread * per + (t - r * per));
}

__global__ void __launch_bounds__(256)
k_gc_count(int64_t n, int per, int spread, const uint8_t* __restrict__ mark,
           int32_t* __restrict__ cnt) {
  __shared__ int lds[16];
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool f[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t t = base + r * 256 + threadIdx.x;
    f[r] = t < n && mark[gc_block_id(t, per, spread)] == 0;
  }
  int rank[4], tot;
  gnx_block_ranks(f, rank, tot, lds);
  if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256)
k_gc_write(int64_t n, int per, int spread, uint8_t* __restrict__ mark,
           const int32_t* __restrict__ off, int nb, int32_t* __restrict__ stack,
           int32_t* __restrict__ top) {
  __shared__ int lds[16];
  const int64_t base = (int64_t)blockIdx.x * GNX_CB;
  bool f[4];
  int32_t id[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t t = base + r * 256 + threadIdx.x;
    id[r] = t < n ? gc_block_id(t, per, spread) : 0;
    f[r] = t < n && mark[id[r]] == 0;
    if (t < n) mark[id[r]] = 0;
  }
  int rank[4], tot;
  gnx_block_ranks(f, rank, tot, lds);
  const int32_t o = off[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (f[r]) stack[o + rank[r]] = id[r];
  if (blockIdx.x == 0 && threadIdx.x == 0) *top = off[nb];
}

int gnx_gc(gnx_state* h) {
  if (h->cfg.L == 0 || !h->genomes_assigned) return 0;
  const int per = 2 * h->NB;
  const int64_t n = (int64_t)h->cfg.cap_rows * per;
  const int nb = (int)((n + GNX_CB - 1) / GNX_CB);
  // device-driven step: between two steps, behind everything enqueued so far; the population's
  // size is read on the device and nothing is read back (the next step's record carries the
  // stack's new height)
  const bool ddm = h->dd_active;
  // (an uncompacted population - gnx_internal.h: holes - is marked over the whole stretch it is
  // spread over: the dead's slots own no row any more, k_dead_rows)
  if (h->N > 0 || ddm)
    hipLaunchKernelGGL(k_gc_mark, dim3(2048), dim3(256), 0, h->stream, gnx_extent(h), h->soa[h->cur].grow,
                       gnx_halves(h), h->half_mark, ddm ? (const GnxDD*)h->dd : nullptr);
  hipLaunchKernelGGL(k_gc_count, dim3(nb), dim3(256), 0, h->stream, n, per, h->row_spread,
                     (const uint8_t*)h->half_mark, h->gc_cnt);
  GNXCHK(gnx_block_scan(h, 1, n, h->gc_cnt, h->gc_off, nullptr, nullptr));
  hipLaunchKernelGGL(k_gc_write, dim3(nb), dim3(256), 0, h->stream, n, per, h->row_spread,
                     h->half_mark, (const int32_t*)h->gc_off, nb, h->half_free, h->half_top);
  h->gc_runs += 1;
  if (ddm) {
    HIPCHK(hipGetLastError());
    return 0;
  }
  int32_t top = 0;
  GNXCHK(gnx_d2h(h, &top, h->half_top, sizeof(top)));
  h->half_free_est = top;
  return 0;
}

int gnx_half_reserve(gnx_state* h, int64_t blocks) {
  if (h->cfg.L == 0 || !h->genomes_assigned || blocks <= 0) return 0;
  if (h->half_free_est < blocks) {
    GNXCHK(gnx_gc(h));
    if (h->half_free_est < blocks) {
      gnx_set_error("genome blocks exhausted: %lld needed, %lld free after a collection",
                    (long long)blocks, (long long)h->half_free_est);
      return 2;
    }
  }
  h->half_free_est -= blocks;       // what the kernel may take at most
  return 0;
}

// ---------------------------------------------------------------- block bookkeeping check
// After a collection: out[0] logical blocks of the individuals that have a genome row / 2,
// out[1] broken references (no physical block), out[2] 0, out[3] physical blocks in use
// (= marked by the collector: distinct blocks the living refer to), out[4] free blocks,
// out[5] blocks in all.  Consistent iff out[1] == 0 and out[3] + out[4] == out[5];
// 2 * out[0] - out[3] blocks are shared.
__global__ void k_half_check(int64_t N, const int32_t* grow, GnxHalves H, int64_t n_halves,
                             const uint8_t* mark, unsigned long long* out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < N; i += stride) {
    const int32_t row = grow[i];
    if (row < 0) continue;
    atomicAdd(&out[0], 1ull);
    for (int q = 0; q < 2 * H.NB; ++q) {
      const int32_t e = H.hmap[(int64_t)row * 2 * H.NB + q];
      if (e == -1 || GNX_BLK(e) >= n_halves) atomicAdd(&out[1], 1ull);
    }
  }
  for (int64_t q = t0; q < n_halves; q += stride)
    if (mark[q]) atomicAdd(&out[3], 1ull);
}

extern "C" int gnx_debug_halves(gnx_state* h, int64_t* out) {
  if (h->cfg.L == 0 || !h->genomes_assigned) {
    gnx_set_error("gnx_debug_halves: genomes not assigned");
    return 1;
  }
  GNXCHK(gnx_xo_join(h));
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 5 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(d, 0, 5 * sizeof(unsigned long long), h->stream));
  const int64_t n_halves = (int64_t)h->cfg.cap_rows * h->row_spread * 2 * h->NB;
  // marks as the collector would set them (its sweep clears them again)
  if (h->N > 0)
    hipLaunchKernelGGL(k_gc_mark, dim3(2048), dim3(256), 0, h->stream, h->N, h->soa[h->cur].grow,
                       gnx_halves(h), h->half_mark, (const GnxDD*)nullptr);
  hipLaunchKernelGGL(k_half_check, dim3(1024), dim3(256), 0, h->stream, h->N, h->soa[h->cur].grow,
                     gnx_halves(h), n_halves, (const uint8_t*)h->half_mark, d);
  unsigned long long host[5];
  int rc = gnx_d2h(h, host, d, sizeof(host));
  (void)hipFree(d);
  GNXCHK(rc);
  GNXCHK(gnx_gc(h));                 // collects, clears the marks, counts the free blocks
  for (int k = 0; k < 5; ++k) out[k] = (int64_t)host[k];
  out[0] *= h->NB;
  out[2] = h->gc_runs;
  out[4] = h->half_free_est;
  out[5] = 2 * h->cfg.cap_rows * h->NB;
  return 0;
}

extern "C" int gnx_genome_info(gnx_state* h, int64_t* out) {
  out[0] = h->NB;
  out[1] = h->BW > 0 ? h->BW : (h->NB > 0 ? h->W64 / h->NB : 0);
  out[2] = h->gc_runs;
  out[3] = h->row_spread;
  out[4] = h->sparse_paths ? 1 : 0;
  out[5] = h->half_free_est;
  out[6] = h->n_free;
  out[7] = h->xo_deferred ? 1 : 0;
  return 0;
}
