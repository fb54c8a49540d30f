// Plumbing only: stable radix sort of (cell key, slot) pairs and exclusive scans
// from rocPRIM.  Everything domain-specific is hand-written in the other files.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "gnx_internal.h"
#include "gnx_compact.h"

// rocPRIM's radix_sort switches to a merge sort below 2^20 items: ~22 launches of a few
// microseconds for the 2 x 10^5 pairs of a step, where Onesweep over the ~24 significant
// id bits takes 5.  Merge sort only for inputs that fit a few blocks.
using gnx_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                   rocprim::default_config, 8192>;

// wider digits, fewer passes: 10 bits per Onesweep pass (the rank table must fit a block of
// 1024 threads) sort the 40..50-bit (cell, id) keys in 4..5 passes instead of 5..7
template <int IPT>
using gnx_onesweep10_t = rocprim::radix_sort_onesweep_config<
    rocprim::kernel_config<1024, IPT>, rocprim::kernel_config<1024, IPT>, 10,
    rocprim::block_radix_rank_algorithm::match>;
template <int IPT>
using gnx_sort_config10_t = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                       gnx_onesweep10_t<IPT>, 8192>;
using gnx_sort_config10 = gnx_sort_config10_t<6>;

int gnx_prim_sort_bytes(size_t n, int bits, size_t* bytes) {
  *bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                   (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, bits,
                                   (hipStream_t)0));
  return 0;
}

int gnx_prim_sort(void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout, const int32_t* vin,
                  int32_t* vout, size_t n, int bits, hipStream_t s) {
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0, bits, s));
  return 0;
}

int gnx_prim_scan_bytes(size_t n, size_t* bytes) {
  *bytes = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, *bytes, (const int32_t*)nullptr, (int32_t*)nullptr, 0, n,
                                 rocprim::plus<int32_t>(), (hipStream_t)0));
  return 0;
}

int gnx_prim_scan(void* tmp, size_t bytes, const int32_t* in, int32_t* out, size_t n,
                  hipStream_t s) {
  HIPCHK(rocprim::exclusive_scan(tmp, bytes, in, out, 0, n, rocprim::plus<int32_t>(), s));
  return 0;
}

int gnx_prim_sort64_bytes(size_t n, size_t* bytes) {
  *bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                   (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, 64,
                                   (hipStream_t)0));
  size_t b10 = 0;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(nullptr, b10, (const uint64_t*)nullptr,
                                                           (uint64_t*)nullptr, (const int32_t*)nullptr,
                                                           (int32_t*)nullptr, n, 0, 64, (hipStream_t)0));
  if (b10 > *bytes) *bytes = b10;
  size_t b32 = 0;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(nullptr, b32, (const uint32_t*)nullptr,
                                                      (uint32_t*)nullptr, (const int32_t*)nullptr,
                                                      (int32_t*)nullptr, n, 0, 32, (hipStream_t)0));
  if (b32 > *bytes) *bytes = b32;
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(nullptr, b32, (const uint32_t*)nullptr,
                                                    (uint32_t*)nullptr, (const int32_t*)nullptr,
                                                    (int32_t*)nullptr, n, 0, 32, (hipStream_t)0));
  if (b32 > *bytes) *bytes = b32;
  return 0;
}

int gnx_prim_sort64(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                    const int32_t* vin, int32_t* vout, size_t n, hipStream_t s) {
  HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0, 64, s));
  return 0;
}

// only the low `end_bit` bits of the keys are significant (fewer radix passes)
int gnx_prim_sort64_bits(void* tmp, size_t bytes, const uint64_t* kin, uint64_t* kout,
                         const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, bool alone) {
  // alone on the chip: 10-bit digits, 6 keys per thread: 0.138 ms for the 1.25 x 10^6 40-bit keys of the metric
  // workload (8-bit default config 0.187; 2 / 3 / 4 / 8 / 12 keys per thread 0.191 / 0.166 /
  // 0.154 / 0.151 / 0.167).  Beside a crossover (whole-step overlap) its 1024-thread blocks
  // wait longer for a CU than the default's and the step loses 5-9 %: default digits there.
  // GNX_SORT_BITS=8 / 10 forces one or the other.
  static const int forced = getenv("GNX_SORT_BITS") ? atoi(getenv("GNX_SORT_BITS")) : 0;
  const int digit = forced ? forced : (alone ? 10 : 8);
  if (digit == 10)
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                        end_bit, s));
  else
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                      end_bit, s));
  return 0;
}

// 32-bit keys (hash cells of the id-ordered index), same digit choice
int gnx_prim_sort32_bits(void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout,
                         const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, bool alone) {
  static const int forced = getenv("GNX_SORT_BITS") ? atoi(getenv("GNX_SORT_BITS")) : 0;
  const int digit = forced ? forced : (alone ? 10 : 8);
  if (digit == 10)
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config10>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                        end_bit, s));
  else
    HIPCHK(rocprim::radix_sort_pairs<gnx_sort_config>(tmp, bytes, kin, kout, vin, vout, n, 0,
                                                      end_bit, s));
  return 0;
}

// ---------------------------------------------------------------- the step's cell sort
// Stable LSD radix sort of (key, 32-bit value) pairs by the low `end_bit` key bits, hand-written
// for gfx950 (round 6: rounds 3-5 launched rocPRIM's private Onesweep device functions here).
// One pass per digit place, each pass ONE kernel (k_pass) in the Onesweep manner:
//   * a workgroup takes tiles in ticket order (atomic counter), so every tile before its own is
//     already running or done;
//   * ranks inside the tile come from wave-wide matching: RB ballots tell a lane which lanes of
//     its wave hold the same digit (64-wide wavefronts: one ballot is one scalar mask), the
//     lowest of them bumps the wave's own 16-bit counter table in LDS, no LDS atomics;
//   * the tile's digit counts are published (status | count in one word per tile and digit) and
//     the counts of the tiles before it are summed by DECOUPLED LOOK-BACK, four states per round
//     trip instead of one (the walk is what a pass waits for: ~1 us per dependent L2 read);
//   * keys and values are put in digit order in LDS first, so a digit's keys leave as one
//     contiguous run (64-byte runs at 16 keys per digit and tile instead of 4-byte scatters).
// The scratch (global digit counts of every place | tile counters | look-back states) is ZERO on
// entry and the caller wipes it afterwards (k_permute does, on the way: no fill kernels on the
// step's chain).  The global counts come from whoever writes the keys: k_keys_hist here, or - in
// gnx_step - the movement kernel itself (k_move), in which case the first pass also gathers its
// keys through the id-ordered index (GATHER) and the sort's front leaves the chain altogether.
namespace gnx_os {
constexpr uint32_t ST_PART = 1u << 30, ST_INCL = 2u << 30, ST_VAL = (1u << 30) - 1u;

__device__ __forceinline__ uint32_t ld_state(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive scan of one value per thread over the workgroup; wsum: LDS [BS / 64]; *total optional
template <int BS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
  constexpr int NW = BS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  __syncthreads();                 // (wsum may still be read from an earlier scan)
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t s = wsum[w];
    woff += w < wave ? s : 0u;
    tot += s;
  }
  if (total) *total = tot;
  return woff + x - v;
}

template <typename K, int RB, int BS, int IPT>
struct PassLds {
  static constexpr int R = 1 << RB, NW = BS / 64, TILE = BS * IPT;
  static constexpr size_t whist = 0;                                  // u16 [NW][R]
  static constexpr size_t dofs = whist + (size_t)NW * R * 2;          // u32 [R]
  static constexpr size_t lstart = dofs + (size_t)R * 4;              // u32 [R]
  static constexpr size_t skey = (lstart + (size_t)R * 4 + 15) & ~(size_t)15;   // K [TILE]
  static constexpr size_t sval = skey + (size_t)TILE * sizeof(K);     // i32 [TILE]
  static constexpr size_t wsum = sval + (size_t)TILE * 4;             // u32 [NW]
  static constexpr size_t bid = wsum + (size_t)NW * 4;                // u32
  static constexpr size_t bytes = bid + 16;
};

// One pass.  kin / vin: the pairs in the order the place before left them (GATHER: key k is
// cell32[k < ord_n ? ord[k] : k] and value k - the id-ordered index of gnx_internal.h); ghist
// [R]: how many keys carry each digit of this place (raw counts); state [tiles][R], counter:
// zero on entry.
template <typename K, int RB, int BS, int IPT, bool GATHER>
__global__ void __launch_bounds__(BS)
k_pass(const K* __restrict__ kin, K* __restrict__ kout, const int32_t* __restrict__ vin,
       int32_t* __restrict__ vout, unsigned int n, const uint32_t* __restrict__ ghist,
       uint32_t* __restrict__ state, uint32_t* __restrict__ counter, int shift,
       const int32_t* __restrict__ ord, long long ord_n, const uint32_t* __restrict__ cell32) {
  using L = PassLds<K, RB, BS, IPT>;
  constexpr int R = L::R, NW = L::NW, TILE = L::TILE;
  constexpr int PER = R > BS ? R / BS : 1;          // digits a thread owns (consecutive)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint16_t* whist = (uint16_t*)(lds + L::whist);
  uint32_t* dofs = (uint32_t*)(lds + L::dofs);
  uint32_t* lstart = (uint32_t*)(lds + L::lstart);
  K* skey = (K*)(lds + L::skey);
  int32_t* sval = (int32_t*)(lds + L::sval);
  uint32_t* wsum = (uint32_t*)(lds + L::wsum);
  uint32_t* s_bid = (uint32_t*)(lds + L::bid);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) *s_bid = atomicAdd(counter, 1u);
  for (int q = tid; q < NW * R / 2; q += BS) ((uint32_t*)whist)[q] = 0u;
  // where each digit's keys start in the output: exclusive scan of the global counts (every
  // workgroup works it out for itself - R words from L2 - instead of a scan kernel on the chain)
  const int d0 = tid * PER;
  uint32_t gex[PER];
  {
    uint32_t g[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      g[j] = (d0 + j < R) ? ghist[d0 + j] : 0u;
      sum += g[j];
    }
    uint32_t run = block_excl_scan<BS>(sum, wsum, nullptr);      // (its barriers also publish s_bid and whist)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      gex[j] = run;
      run += g[j];
    }
  }
  const unsigned int bid = *s_bid;
  const unsigned int base = bid * (unsigned int)TILE;
  const unsigned int nvalid = n - base < (unsigned int)TILE ? n - base : (unsigned int)TILE;
  // ---- load: wave w takes the tile's keys [w * 64 * IPT, (w + 1) * 64 * IPT), item r of lane l
  // is key r * 64 + l of that stretch (coalesced; memory order = (wave, item, lane))
  K key[IPT];
  int32_t val[IPT];
  const unsigned int wbase = (unsigned int)wave * 64u * IPT + (unsigned int)lane;
  if (GATHER) {
    int32_t slot[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
      const unsigned int q = wbase + r * 64u;
      const long long k = (long long)base + q;
      slot[r] = q < nvalid ? (k < ord_n ? ord[k] : (int32_t)k) : -1;
      val[r] = (int32_t)k;
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) key[r] = slot[r] >= 0 ? (K)cell32[slot[r]] : (K)0;
  } else {
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
      const unsigned int q = wbase + r * 64u;
      key[r] = q < nvalid ? kin[base + q] : (K)0;
      val[r] = q < nvalid ? vin[base + q] : 0;
    }
  }
  // ---- rank inside the wave, item by item (stable: items in order, lanes in order)
  uint32_t dig[IPT], rank[IPT];
  uint16_t* wh = whist + (size_t)wave * R;
#pragma unroll
  for (int r = 0; r < IPT; ++r) {
    const bool valid = wbase + r * 64u < nvalid;
    const uint32_t d = (uint32_t)(key[r] >> shift) & (uint32_t)(R - 1);
    dig[r] = d;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const unsigned long long bb = __ballot((d >> b) & 1u);
      m &= ((d >> b) & 1u) ? bb : ~bb;
    }
    const int leader = __ffsll((long long)m) - 1;
    const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t old = 0;
    if (valid && lane == leader) {
      old = wh[d];
      wh[d] = (uint16_t)(old + (uint32_t)__popcll(m));
    }
    old = __shfl(old, leader < 0 ? 0 : leader);
    rank[r] = old + below;
  }
  __syncthreads();
  // ---- the tile's count of every digit; the waves' counters become their exclusive offsets
  uint32_t tot[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    uint32_t acc = 0;
    if (d0 + j < R) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint32_t c = whist[(size_t)w * R + d0 + j];
        whist[(size_t)w * R + d0 + j] = (uint16_t)acc;
        acc += c;
      }
      // published at once: the tiles behind this one wait for nothing else
      st_state(&state[(size_t)bid * R + d0 + j], (bid == 0 ? ST_INCL : ST_PART) | acc);
    }
    tot[j] = acc;
  }
  // ---- where each digit starts inside the tile
  {
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) sum += tot[j];
    uint32_t run = block_excl_scan<BS>(sum, wsum, nullptr);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (d0 + j < R) lstart[d0 + j] = run;
      run += tot[j];
    }
  }
  __syncthreads();
  // ---- keys and values in digit order in LDS
#pragma unroll
  for (int r = 0; r < IPT; ++r) {
    if (wbase + r * 64u < nvalid) {
      const uint32_t lp = lstart[dig[r]] + wh[dig[r]] + rank[r];
      skey[lp] = key[r];
      sval[lp] = val[r];
    }
  }
  // ---- decoupled look-back: the counts of the tiles before this one, digit by digit
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (d0 + j >= R) continue;
    const int d = d0 + j;
    uint32_t prefix = 0;
    int b = (int)bid - 1;
    while (b >= 0) {
      uint32_t v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        v[q] = b - q >= 0 ? ld_state(&state[(size_t)(b - q) * R + d]) : ST_INCL;
      bool done = false;
      int used = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (done || used < q) continue;
        if (v[q] == 0u) continue;                    // not published yet: the round is cut here
        prefix += v[q] & ST_VAL;
        used = q + 1;
        if (v[q] & ST_INCL) done = true;
      }
      if (done) break;
      b -= used;
      if (used == 0) __builtin_amdgcn_s_sleep(1);
    }
    if (bid > 0) st_state(&state[(size_t)bid * R + d], ST_INCL | (prefix + tot[j]));
    dofs[d] = gex[j] + prefix - lstart[d];
  }
  __syncthreads();
  // ---- out: position l of the tile's digit order goes to (global start of its digit) + (l -
  // local start of its digit); consecutive l of one digit are consecutive addresses
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const unsigned int l = (unsigned int)k * BS + (unsigned int)tid;
    if (l < nvalid) {
      const K kk = skey[l];
      const uint32_t d = (uint32_t)(kk >> shift) & (uint32_t)(R - 1);
      const uint32_t pos = dofs[d] + l;
      kout[pos] = kk;
      vout[pos] = sval[l];
    }
  }
}

// digit places and bits per place for keys of `end_bit` significant bits: as few places as
// GNX_OS_MAX_RB-bit digits allow, the bits spread evenly over them, never fewer than 8 bits per
// place (the kernels are instantiated for 8 .. 11)
// (a tile of 4 096 keys wants its digits few enough that each holds several keys: 9 bits at most -
// 11-bit digits with tiles of 2 048 keys made the look-back the pass: two tiles on one GPU 2.07
// against 1.68 ms a step)
#define GNX_OS_MAX_RB 9
static void digits(int end_bit, int* places, int* rb) {
  int p = (end_bit + GNX_OS_MAX_RB - 1) / GNX_OS_MAX_RB;
  if (p < 1) p = 1;
  int b = (end_bit + p - 1) / p;
  if (b < 8) b = 8;
  *places = p;
  *rb = b;
}

// scratch layout in 32-bit words: [places][R] global counts | 16 words (tile counters) |
// [places][tiles][R] look-back states
struct Layout {
  int places, rb, R;
  unsigned int tiles;
  size_t hist, counters, states, words;
};
static Layout layout(size_t n, int end_bit, int tile) {
  Layout l;
  digits(end_bit, &l.places, &l.rb);
  l.R = 1 << l.rb;
  l.tiles = (unsigned int)((n + tile - 1) / tile);
  if (l.tiles < 1) l.tiles = 1;
  l.hist = 0;
  l.counters = (size_t)l.places * l.R;
  l.states = l.counters + 16;
  l.words = l.states + (size_t)l.places * l.tiles * l.R;
  return l;
}

template <typename K, int RB, int BS, int IPT, bool GATHER>
static int launch_pass(const K* kin, K* kout, const int32_t* vin, int32_t* vout, unsigned int n,
                       const uint32_t* ghist, uint32_t* state, uint32_t* counter, int shift,
                       const int32_t* ord, long long ord_n, const uint32_t* cell32, hipStream_t s) {
  using L = PassLds<K, RB, BS, IPT>;
  auto kern = k_pass<K, RB, BS, IPT, GATHER>;
  static bool attr_set = false;          // (more LDS than the 64-KB default needs asking for)
  if (!attr_set) {
    if (L::bytes > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)L::bytes));
    attr_set = true;
  }
  const unsigned int tiles = (n + L::TILE - 1) / L::TILE;
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(BS), L::bytes, s, kin, kout, vin, vout, n, ghist, state,
                     counter, shift, ord, ord_n, cell32);
  return 0;
}

template <typename K, int BS, int IPT, bool GATHER>
static int launch_pass_rb(int rb, const K* kin, K* kout, const int32_t* vin, int32_t* vout,
                          unsigned int n, const uint32_t* ghist, uint32_t* state, uint32_t* counter,
                          int shift, const int32_t* ord, long long ord_n, const uint32_t* cell32,
                          hipStream_t s) {
  switch (rb) {
    case 8: return launch_pass<K, 8, BS, IPT, GATHER>(kin, kout, vin, vout, n, ghist, state, counter, shift, ord, ord_n, cell32, s);
    case 9: return launch_pass<K, 9, BS, IPT, GATHER>(kin, kout, vin, vout, n, ghist, state, counter, shift, ord, ord_n, cell32, s);
    case 10: return launch_pass<K, 10, BS, IPT, GATHER>(kin, kout, vin, vout, n, ghist, state, counter, shift, ord, ord_n, cell32, s);
    case 11: return launch_pass<K, 11, BS, IPT, GATHER>(kin, kout, vin, vout, n, ghist, state, counter, shift, ord, ord_n, cell32, s);
  }
  gnx_set_error("radix sort: no kernel for %d-bit digits", rb);
  return 1;
}

// the passes: global counts already in scratch (k_keys_hist / k_hist / k_move), the rest zero.
// gather: the first pass takes its keys through the id-ordered index (kin / vin unused there).
template <typename K, int BS, int IPT>
static int passes(void* scratch, K* ktmp, int32_t* vtmp, const K* kin, K* kout, const int32_t* vin,
                  int32_t* vout, size_t n, int end_bit, hipStream_t s, bool gather,
                  const int32_t* ord, long long ord_n, const uint32_t* cell32) {
  const Layout l = layout(n, end_bit, BS * IPT);
  uint32_t* w = (uint32_t*)scratch;
  bool to_output = (l.places - 1) % 2 == 0;
  const K* ki = kin;
  const int32_t* vi = vin;
  for (int place = 0; place < l.places; ++place) {
    K* ko = to_output ? kout : ktmp;
    int32_t* vo = to_output ? vout : vtmp;
    uint32_t* st = w + l.states + (size_t)place * l.tiles * l.R;
    if (place == 0 && gather) {
      if constexpr (sizeof(K) == 4) {
        GNXCHK((launch_pass_rb<K, BS, IPT, true>(l.rb, ki, ko, vi, vo, (unsigned int)n,
                                                  w + l.hist + (size_t)place * l.R, st,
                                                  w + l.counters + place, place * l.rb, ord, ord_n,
                                                  cell32, s)));
      } else {
        gnx_set_error("radix sort: the gathering first pass takes 32-bit keys");
        return 1;
      }
    } else {
      GNXCHK((launch_pass_rb<K, BS, IPT, false>(l.rb, ki, ko, vi, vo, (unsigned int)n,
                                                 w + l.hist + (size_t)place * l.R, st,
                                                 w + l.counters + place, place * l.rb, nullptr, 0,
                                                 nullptr, s)));
    }
    ki = ko;
    vi = vo;
    to_output = !to_output;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// global digit counts of every place in one pass over the keys (LDS tables, one flush)
template <typename K>
__global__ void __launch_bounds__(512)
k_hist(const K* __restrict__ keys, unsigned int n, uint32_t* __restrict__ ghist, int places, int rb) {
  extern __shared__ uint32_t lh[];                  // [places][R]
  const int R = 1 << rb;
  for (int q = threadIdx.x; q < places * R; q += 512) lh[q] = 0u;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) {
    const K k = keys[i];
    for (int pl = 0; pl < places; ++pl) atomicAdd(&lh[pl * R + ((uint32_t)(k >> (pl * rb)) & (R - 1))], 1u);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < places * R; q += 512) {
    const uint32_t v = lh[q];
    if (v) atomicAdd(&ghist[q], v);
  }
}

// Keys of the id-ordered sequence (entry k is slot ord[k] for k < ord_n, else slot k itself:
// offspring appended since the index was last compacted) written out as (key, value = k) pairs,
// and their global digit counts - the sort's front when nobody else has counted (the
// device-driven step, tiles, operator calls).  ghist: [places][R], zero on entry.
template <unsigned IPT>
__global__ void __launch_bounds__(1024)
k_keys_hist(long long N, long long ord_n, const int32_t* __restrict__ ord,
            const uint32_t* __restrict__ cell32, uint32_t* __restrict__ key,
            int32_t* __restrict__ val, uint32_t* __restrict__ ghist, int places, int rb,
            long long n_fixed, unsigned int sentinel, const GnxDD* __restrict__ dd) {
  // device-driven step: the population's size lives on the device and the sort runs over a
  // fixed n_fixed >= N entries; those behind the population carry the largest key, so the
  // stable sort leaves them behind everybody
  if (dd) {
    N = dd->N;
    ord_n = dd->ord_n;
  }
  extern __shared__ uint32_t lh[];                  // [places][R]
  const int R = 1 << rb;
  const unsigned tid = threadIdx.x;
  for (int q = tid; q < places * R; q += 1024) lh[q] = 0u;
  __syncthreads();
  const long long base = (long long)blockIdx.x * (1024 * IPT);
  int32_t slot[IPT];
  uint32_t c[IPT];
#pragma unroll
  for (unsigned r = 0; r < IPT; ++r) {
    const long long k = base + r * 1024 + tid;
    slot[r] = k < N ? (k < ord_n ? ord[k] : (int32_t)k) : -1;
  }
#pragma unroll
  for (unsigned r = 0; r < IPT; ++r) c[r] = slot[r] >= 0 ? cell32[slot[r]] : sentinel;
#pragma unroll
  for (unsigned r = 0; r < IPT; ++r) {
    const long long k = base + r * 1024 + tid;
    if (slot[r] < 0 && k >= n_fixed) continue;
    key[k] = c[r];
    val[k] = (int32_t)k;
    for (int pl = 0; pl < places; ++pl) atomicAdd(&lh[pl * R + ((c[r] >> (pl * rb)) & (R - 1u))], 1u);
  }
  __syncthreads();
  for (int q = tid; q < places * R; q += 1024) {
    const uint32_t v = lh[q];
    if (v) atomicAdd(&ghist[q], v);
  }
}
}  // namespace gnx_os

// tile geometries: 0 = 512 threads x 8 keys (4 096 keys a tile), 1 = 256 x 4 (1 024 keys a tile:
// four times the workgroups for a sort that is all latency at 10^5 keys); 64-bit keys: 512 x 8
#define GNX_OS_BIG_BS 512
#define GNX_OS_BIG_IPT 8
#define GNX_OS_SMALL_BS 256
#define GNX_OS_SMALL_IPT 4
#define GNX_OS_64_BS 512
#define GNX_OS_64_IPT 8

void gnx_os_digits(int end_bit, int* places, int* rb) { gnx_os::digits(end_bit, places, rb); }

size_t gnx_os_words_used(size_t n, int end_bit, int geometry) {
  return gnx_os::layout(n, end_bit, geometry == 1 ? GNX_OS_SMALL_BS * GNX_OS_SMALL_IPT
                                                   : GNX_OS_BIG_BS * GNX_OS_BIG_IPT).words;
}
int gnx_os_keys_hist(void* scratch, unsigned int* ticket, int64_t N, int64_t ord_n,
                     const int32_t* ord, const uint32_t* cell32, uint32_t* key, int32_t* val,
                     int end_bit, hipStream_t s, const GnxDD* dd, int geometry) {
  (void)ticket;
  int places, rb;
  gnx_os::digits(end_bit, &places, &rb);
  const size_t lds = (size_t)places * (1u << rb) * sizeof(uint32_t);
  if (geometry == 1) {
    // small populations: 2 048 keys per workgroup instead of 6 144 - three times the workgroups
    const unsigned int blocks = (unsigned int)((N + 2047) / 2048);
    hipLaunchKernelGGL((gnx_os::k_keys_hist<2>), dim3(blocks), dim3(1024), lds, s, (long long)N,
                       (long long)ord_n, ord, cell32, key, val, (uint32_t*)scratch, places, rb,
                       (long long)N, (1u << end_bit) - 1u, dd);
  } else {
    // dd: N is the fixed number of entries the sort runs over, the population's own size is
    // read on the device
    const unsigned int blocks = (unsigned int)((N + 6143) / 6144);
    hipLaunchKernelGGL((gnx_os::k_keys_hist<6>), dim3(blocks), dim3(1024), lds, s, (long long)N,
                       (long long)ord_n, ord, cell32, key, val, (uint32_t*)scratch, places, rb,
                       (long long)N, (1u << end_bit) - 1u, dd);
  }
  HIPCHK(hipGetLastError());
  return 0;
}
int gnx_os_sort32_ranked(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin,
                         uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                         hipStream_t s, int geometry) {
  if (geometry == 1)
    return gnx_os::passes<uint32_t, GNX_OS_SMALL_BS, GNX_OS_SMALL_IPT>(
        scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s, false, nullptr, 0, nullptr);
  return gnx_os::passes<uint32_t, GNX_OS_BIG_BS, GNX_OS_BIG_IPT>(
      scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s, false, nullptr, 0, nullptr);
}
// the cell sort of gnx_step when the movement kernel has counted the digits (k_move): the first
// pass gathers cell32 through the id-ordered index itself - no kernel in front of the passes
int gnx_os_sort32_gather(void* scratch, uint32_t* ktmp, int32_t* vtmp, uint32_t* kout, int32_t* vout,
                         size_t n, int end_bit, const int32_t* ord, int64_t ord_n,
                         const uint32_t* cell32, hipStream_t s) {
  return gnx_os::passes<uint32_t, GNX_OS_BIG_BS, GNX_OS_BIG_IPT>(
      scratch, ktmp, vtmp, (const uint32_t*)nullptr, kout, (const int32_t*)nullptr, vout, n, end_bit,
      s, true, ord, (long long)ord_n, cell32);
}

// 64-bit keys; scratch zero on entry, gnx_os_words_used64 words to wipe afterwards; tmp (>= 12 n
// bytes) holds the keys and values between the passes
size_t gnx_os_words_used64(size_t n, int end_bit) {
  return gnx_os::layout(n, end_bit, GNX_OS_64_BS * GNX_OS_64_IPT).words;
}
int gnx_os_sort64_clean(void* scratch, void* tmp, const uint64_t* kin, uint64_t* kout,
                        const int32_t* vin, int32_t* vout, size_t n, int end_bit, hipStream_t s) {
  uint64_t* ktmp = (uint64_t*)tmp;
  int32_t* vtmp = (int32_t*)(ktmp + n);
  int places, rb;
  gnx_os::digits(end_bit, &places, &rb);
  const unsigned int blocks = (unsigned int)std::min<size_t>((n + 4095) / 4096, 1024);
  hipLaunchKernelGGL((gnx_os::k_hist<uint64_t>), dim3(blocks ? blocks : 1), dim3(512),
                     (size_t)places * (1u << rb) * 4, s, kin, (unsigned int)n, (uint32_t*)scratch,
                     places, rb);
  return gnx_os::passes<uint64_t, GNX_OS_64_BS, GNX_OS_64_IPT>(scratch, ktmp, vtmp, kin, kout, vin,
                                                                vout, n, end_bit, s, false, nullptr,
                                                                0, nullptr);
}

size_t gnx_os_scratch_bytes(size_t n, int end_bit) {
  size_t w = std::max({gnx_os_words_used(n, end_bit, 0),
                       // (the small geometry: only ever used below GNX_DD_MAX_CAP slots)
                       gnx_os_words_used(std::min<size_t>(n, 1u << 21), end_bit, 1)});
  return w * sizeof(unsigned int);
}

// generic entry (keys given, nothing counted yet): counts, then the passes; the caller has the
// scratch zero and wipes gnx_os_scratch_bytes of it afterwards
int gnx_os_sort32(void* scratch, uint32_t* ktmp, int32_t* vtmp, const uint32_t* kin,
                  uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit,
                  hipStream_t s, int variant) {
  (void)variant;
  int places, rb;
  gnx_os::digits(end_bit, &places, &rb);
  const unsigned int blocks = (unsigned int)std::min<size_t>((n + 4095) / 4096, 1024);
  hipLaunchKernelGGL((gnx_os::k_hist<uint32_t>), dim3(blocks ? blocks : 1), dim3(512),
                     (size_t)places * (1u << rb) * 4, s, kin, (unsigned int)n, (uint32_t*)scratch,
                     places, rb);
  return gnx_os::passes<uint32_t, GNX_OS_BIG_BS, GNX_OS_BIG_IPT>(
      scratch, ktmp, vtmp, kin, kout, vin, vout, n, end_bit, s, false, nullptr, 0, nullptr);
}

// ---------------------------------------------------------------- block-count scan
// Exclusive scan of K (<= 3) count arrays (cnt[k * stride + b], b < nb) into
// off[k * stride + b], off[k * stride + nb] = total.  One workgroup of 1024 threads.
// out[k] = total of array k; `host` (pinned, device-visible) receives the same numbers.
__global__ void __launch_bounds__(1024)
k_block_scan(int K, int nb, int stride, const int32_t* __restrict__ cnt, int32_t* __restrict__ off,
             int32_t* __restrict__ out, int64_t* __restrict__ host, long long seq,
             const int32_t* __restrict__ extra) {
  // all K (<= 3) arrays in one sweep: two barriers per 1024 blocks instead of three per array
  __shared__ int wsum[3][16];
  __shared__ int carry_s[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 3) carry_s[tid] = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + tid;
    int v[3], x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] = (k < K && b < nb) ? cnt[k * stride + b] : 0;
      x[k] = v[k];
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int y = __shfl_up(x[k], d);
        if (lane >= d) x[k] += y;
      }
    }
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 3; ++k) wsum[k][wave] = x[k];
    }
    __syncthreads();
    int tot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int woff = 0, t = 0;
      for (int w = 0; w < 16; ++w) {
        const int s = wsum[k][w];
        woff += w < wave ? s : 0;
        t += s;
      }
      tot[k] = t;
      if (k < K && b < nb) off[k * stride + b] = carry_s[k] + woff + x[k] - v[k];
    }
    __syncthreads();
    if (tid < 3) carry_s[tid] += tot[tid];
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = 0; k < 3; ++k) {
      const int total = k < K ? carry_s[k] : 0;
      if (k < K) off[k * stride + nb] = total;
      if (out) out[k] = total;
      if (host) host[k] = total;
    }
    if (host && extra) host[12] = (int64_t)*extra;    // one more word for the host
    // a host that polls host[3] for `seq` (gnx_wait_published) sees the totals first
    if (host && seq)
      __hip_atomic_store(&host[3], (int64_t)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int gnx_block_scan(gnx_state* h, int K, int64_t n_items, const int32_t* cnt, int32_t* off,
                   int32_t* out, int64_t* host, int64_t seq, hipStream_t st, const int32_t* extra) {
  const int nb = (int)((n_items + GNX_CB - 1) / GNX_CB);
  hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, st ? st : h->stream, K, nb, h->blk_stride, cnt, off,
                     out, host, (long long)seq, extra);
  HIPCHK(hipGetLastError());
  return 0;
}
