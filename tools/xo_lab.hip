// Kernel lab for the crossover (the step's dominant byte mover): times the PRODUCT
// kernels of geonomics_amd/csrc/gnx_xo.h on a synthetic job list shaped like the metric
// workload (L = 1e5, 1.6 M genome rows = 40 GB, ~2 x 10^5 births per launch), against
// the round-1 kernel and a bare copy, over unroll / grid / cache-policy / job-order
// variants.  Every variant is launched `reps` times, variants interleaved round-robin
// so that drift of the box (clocks, thermals) hits all alike; min / median / mean of the
// per-launch HIP-event times are reported.  All variants must produce the same
// checksum of the child rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/xo_lab tools/xo_lab.hip
//   ./tools/xo_lab [reps=30] [births=218405] [set=all|quick|dense]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <random>
#include <functional>
#include "../geonomics_amd/csrc/gnx_xo.h"

// how the lab describes a gamete (and the format of a replay file dumped by an earlier
// build): logical parent row, child half-row, path, start homologue; converted to the
// product's GnxXoJob (physical half-rows, csrc/gnx_half.h) on upload
struct LabJob {
  int32_t prow, dst, key, start;
};
static inline GnxXoJob lab_to_job(const LabJob& j) {
  return GnxXoJob{j.prow * 2, j.prow * 2 + 1, j.dst, j.key * 2 + j.start};
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// ---- the round-1 kernel (k_crossover_stream<8> of round 1), for reference ----------
template <int XO_UNROLL>
__global__ void __launch_bounds__(256)
k_r1_stream(int64_t B, int W16, const u64x2* __restrict__ G, u64x2* __restrict__ Gout,
            const int32_t* __restrict__ grow, int64_t first_slot,
            const int32_t* __restrict__ off_parent, const int32_t* __restrict__ off_keys,
            const uint8_t* __restrict__ off_start, const int32_t* __restrict__ bp_off,
            const int32_t* __restrict__ bp_loci) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t gam = wave0; gam < 2 * B; gam += n_waves) {
    const int64_t k = gam >> 1;
    const int p = (int)(gam & 1);
    const int prow = __builtin_amdgcn_readfirstlane(grow[off_parent[2 * k + p]]);
    const int key = __builtin_amdgcn_readfirstlane(off_keys[2 * k + p]);
    const u64 s = __builtin_amdgcn_readfirstlane((int)off_start[2 * k + p]) ? ~0ull : 0ull;
    const int crow = __builtin_amdgcn_readfirstlane(grow[first_slot + k]);
    const u64x2* h0 = G + ((int64_t)prow * 2 + 0) * W16;
    const u64x2* h1 = G + ((int64_t)prow * 2 + 1) * W16;
    u64x2* dst = Gout + ((int64_t)crow * 2 + p) * W16;
    const int bp0 = __builtin_amdgcn_readfirstlane(bp_off[key]);
    const int nbp = __builtin_amdgcn_readfirstlane(bp_off[key + 1]) - bp0;
    const int mybp = lane < nbp ? bp_loci[bp0 + lane] : 0x7fffffff;
    for (int c0 = lane; c0 < W16; c0 += 64 * XO_UNROLL) {
      u64x2 m[XO_UNROLL], v[XO_UNROLL];
      bool mixed = false;
#pragma unroll
      for (int u = 0; u < XO_UNROLL; ++u) {
        const int c = min(c0 + u * 64, W16 - 1);
        m[u] = xo_mask_lanes(c, s, mybp, nbp);
        const bool one = (m[u].a & m[u].b) == ~0ull;
        mixed |= !one && (m[u].a | m[u].b) != 0ull;
        v[u] = (one ? h1 : h0)[c];
      }
      if (__builtin_expect(mixed, 0)) {
#pragma unroll
        for (int u = 0; u < XO_UNROLL; ++u) {
          const int c = min(c0 + u * 64, W16 - 1);
          if ((m[u].a & m[u].b) != ~0ull && (m[u].a | m[u].b) != 0ull) {
            const u64x2 b = h1[c];
            v[u].a = (v[u].a & ~m[u].a) | (b.a & m[u].a);
            v[u].b = (v[u].b & ~m[u].b) | (b.b & m[u].b);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < XO_UNROLL; ++u) {
        const int c = c0 + u * 64;
        if (c < W16) xo_store(dst + c, v[u]);
      }
    }
  }
}

// bare copy of the jobs' half-rows (homologue `start` of the parent), same job walk
template <int U, bool NT_LD>
__global__ void __launch_bounds__(256)
k_copy_jobs(const int32_t* __restrict__ n_jobs_p, int W16, const u64x2* __restrict__ G,
            u64x2* __restrict__ Gout, const GnxXoJob* __restrict__ jobs) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_jobs = *n_jobs_p;
  const int n_waves = (int)gridDim.x * 4;
  for (int j = (int)blockIdx.x * 4 + wv; j < n_jobs; j += n_waves) {
    const GnxXoJob jb = jobs[j];
    const int ks_ = __builtin_amdgcn_readfirstlane(jb.ks);
    const int dsth = __builtin_amdgcn_readfirstlane(jb.dst);
    const int ph = __builtin_amdgcn_readfirstlane((ks_ & 1) ? jb.ph1 : jb.ph0);
    const u64x2* src = G + (int64_t)ph * W16;
    u64x2* dst = Gout + (int64_t)dsth * W16;
    for (int c0 = lane; c0 < W16; c0 += 64 * U) {
      u64x2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = xo_load<NT_LD>(src + min(c0 + u * 64, W16 - 1));
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (c0 + u * 64 < W16) xo_store(dst + c0 + u * 64, v[u]);
    }
  }
}

// plain streaming copy (float4 per lane), the guide's "achievable" reference
__global__ void __launch_bounds__(256) k_stream_copy(int64_t n16, const u64x2* __restrict__ a,
                                                     u64x2* __restrict__ b) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
    b[i] = a[i];
}

// random genotypes, zero padding beyond locus L (as the product keeps it)
__global__ void k_fill(int64_t n, u64* g, int W64, int L) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u64 z = (u64)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    const int w = (int)(i % W64);
    const int left = L - w * 64;
    if (left <= 0) z = 0;
    else if (left < 64) z &= (1ull << left) - 1ull;
    g[i] = z;
  }
}

// order-independent checksum of the jobs' destination half-rows
__global__ void k_checksum(int n_jobs, int W16, const u64x2* G, const GnxXoJob* jobs, u64* out) {
  const int64_t total = (int64_t)n_jobs * W16;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  u64 acc = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int j = (int)(t / W16), c = (int)(t - (int64_t)j * W16);
    const u64x2 v = G[(int64_t)jobs[j].dst * W16 + c];
    acc += (v.a * 0x9E3779B97F4A7C15ull) ^ (v.b + (u64)c * 1315423911ull + (u64)jobs[j].dst);
  }
  atomicAdd(out, acc);
}

struct Variant {
  std::string name;
  std::function<void()> launch;
  double bytes;
  std::vector<float> ms;
  u64 sum = 0;
  bool check = true;
};

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  const int B = argc > 2 ? atoi(argv[2]) : 218405;
  const std::string set = argc > 3 ? argv[3] : "all";
  // optional: a job list dumped from a product run (tools/xo_trend.py <steps> <file>)
  // replaces job kind 0 ("nat/rnd-dst"); kinds 1..3 are derived from it
  const char* jobs_file = argc > 4 ? argv[4] : nullptr;
  const int L = 100000;
  const int W64 = ((L + 63) / 64 + 15) / 16 * 16, W16 = W64 / 2;       // 1568, 784
  const int64_t rows = 1601024 + 1024, live = 1200000;
  const int n_paths = 10000;
  const int n_jobs = 2 * B;
  printf("xo_lab: L=%d W16=%d rows=%lld births=%d jobs=%d reps=%d set=%s\n", L, W16,
         (long long)rows, B, n_jobs, reps, set.c_str());
  // XO_LAB_SPREAD=k: logical row r lives at physical row r * k of a k times larger table
  // (is the achieved bandwidth a matter of WHERE in HBM the rows sit?)
  const int spread = getenv("XO_LAB_SPREAD") ? atoi(getenv("XO_LAB_SPREAD")) : 1;
  u64x2* G;
  CHK(hipMalloc((void**)&G, (size_t)rows * spread * 2 * W16 * 16));
  hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, rows * spread * 2 * (int64_t)W64, (u64*)G, W64, L);
  CHK(hipDeviceSynchronize());

  std::mt19937_64 rng(12345);
  // which rows are live parents / free: a random subset, as after many generations
  // XO_LAB_LO / XO_LAB_HI restrict the rows the synthetic jobs use to [lo, hi)
  const int64_t lo_row = getenv("XO_LAB_LO") ? atoll(getenv("XO_LAB_LO")) : 0;
  const int64_t hi_row = getenv("XO_LAB_HI") ? atoll(getenv("XO_LAB_HI")) : rows;
  std::vector<int32_t> perm;
  for (int64_t i = lo_row; i < hi_row; ++i) perm.push_back((int32_t)i);
  std::shuffle(perm.begin(), perm.end(), rng);
  const int64_t n_live = (int64_t)perm.size() * 3 / 4;
  std::vector<int32_t> live_rows(perm.begin(), perm.begin() + n_live);
  std::vector<int32_t> free_rows(perm.begin() + n_live, perm.end());      // random order
  if ((int64_t)free_rows.size() < B) { printf("too many births\n"); return 1; }
  // recombination paths, rate 1/L
  std::vector<int32_t> bp_off(n_paths + 1, 0), bp_loci;
  std::binomial_distribution<int> nb(L - 1, 1.0 / L);
  for (int k = 0; k < n_paths; ++k) {
    int n = std::min(nb(rng), 24);
    std::vector<int32_t> b;
    for (int q = 0; q < n; ++q) b.push_back(1 + (int32_t)(rng() % (L - 1)));
    std::sort(b.begin(), b.end());
    b.erase(std::unique(b.begin(), b.end()), b.end());
    bp_loci.insert(bp_loci.end(), b.begin(), b.end());
    bp_off[k + 1] = (int32_t)bp_loci.size();
  }
  // dense masks of the same paths (for k_xo_dense)
  std::vector<u64> paths((size_t)n_paths * W64, 0);
  for (int k = 0; k < n_paths; ++k) {
    int par = 0, q = bp_off[k];
    for (int l = 0; l < L; ++l) {
      while (q < bp_off[k + 1] && bp_loci[q] == l) { par ^= 1; ++q; }
      if (par) paths[(size_t)k * W64 + (l >> 6)] |= 1ull << (l & 63);
    }
  }
  // births: parents, keys, start homologues
  std::vector<int32_t> par(2 * (size_t)B), key(2 * (size_t)B);
  std::vector<uint8_t> st(2 * (size_t)B);
  for (int64_t q = 0; q < 2 * (int64_t)B; ++q) {
    par[q] = live_rows[rng() % live_rows.size()];
    key[q] = (int32_t)(rng() % n_paths);
    st[q] = (uint8_t)(rng() & 1);
  }
  // job lists: [0] natural order + random child rows; [1] natural order + ascending child
  // rows; [2] births sorted by gamete-0 parent row + ascending child rows;
  // [3] all gametes sorted by parent row (child rows ascending by birth)
  auto make_jobs = [&](int kind) {
    std::vector<int32_t> crow(free_rows.begin(), free_rows.begin() + B);
    std::vector<int32_t> order(B);
    for (int k = 0; k < B; ++k) order[k] = k;
    if (kind >= 1) std::sort(crow.begin(), crow.end());
    if (kind == 2)
      std::sort(order.begin(), order.end(), [&](int a, int b) { return par[2 * a] < par[2 * b]; });
    std::vector<LabJob> jobs(n_jobs);
    for (int q = 0; q < B; ++q) {
      const int k = order[q];
      for (int p = 0; p < 2; ++p)
        jobs[2 * q + p] = LabJob{par[2 * k + p], crow[q] * 2 + p, key[2 * k + p], st[2 * k + p]};
    }
    if (kind == 3)
      std::sort(jobs.begin(), jobs.end(), [](const LabJob& a, const LabJob& b) { return a.prow < b.prow; });
    return jobs;
  };
  const int NK = 4;
  GnxXoJob* d_jobs[NK];
  std::vector<LabJob> file_jobs;
  if (jobs_file) {
    FILE* f = fopen(jobs_file, "rb");
    if (!f) { printf("cannot open %s\n", jobs_file); return 1; }
    file_jobs.resize(n_jobs);
    size_t got = fread(file_jobs.data(), sizeof(LabJob), n_jobs, f);
    fclose(f);
    if ((int)got != n_jobs) { printf("job file holds %zu jobs, births*2 = %d\n", got, n_jobs); return 1; }
    for (auto& j : file_jobs)
      if (j.prow < 0 || j.prow >= rows || j.dst < 0 || j.dst >= 2 * rows || j.key < 0 ||
          j.key >= n_paths) { printf("job out of range\n"); return 1; }
    printf("replaying %d product jobs from %s\n", n_jobs, jobs_file);
  }
  for (int kind = 0; kind < NK; ++kind) {
    auto jobs = make_jobs(kind);
    if (jobs_file) {
      // replay: [0] as dumped; [1] the product's parents, synthetic child rows; [2] synthetic
      // parents, the product's child rows; [3] as dumped, shuffled
      jobs = file_jobs;
      std::vector<char> used(rows, 0);
      for (auto& j : file_jobs) { used[j.prow] = 1; used[j.dst >> 1] = 1; }
      std::vector<int32_t> unused;
      for (int64_t r = 0; r < rows; ++r) if (!used[r]) unused.push_back((int32_t)r);
      std::shuffle(unused.begin(), unused.end(), rng);
      if (kind == 1)
        for (size_t q = 0; q + 1 < jobs.size(); q += 2) {
          const int32_t r = unused[(q / 2) % unused.size()];
          jobs[q].dst = r * 2;
          jobs[q + 1].dst = r * 2 + 1;
        }
      const char* alt = getenv("XO_LAB_ALT");
      if (kind == 2 && !alt)
        for (size_t q = 0; q < jobs.size(); ++q) jobs[q].prow = unused[(unused.size() - 1 - q % unused.size())];
      if (kind == 3 && !alt) std::shuffle(jobs.begin(), jobs.end(), rng);
      if (alt && kind == 2) {
        // synthetic children drawn only from unused rows BELOW the product's highest row
        std::vector<int32_t> low;
        for (int32_t r : unused) if (r < 1400000) low.push_back(r);
        for (size_t q = 0; q + 1 < jobs.size(); q += 2) {
          const int32_t r = low[(q / 2) % low.size()];
          jobs[q].dst = r * 2;
          jobs[q + 1].dst = r * 2 + 1;
        }
      }
      if (alt && kind == 3) {
        // the product's children moved to the nearest unused row above them
        std::vector<char> taken(rows, 0);
        for (auto& j : file_jobs) taken[j.prow] = 1;
        for (size_t q = 0; q + 1 < jobs.size(); q += 2) {
          int64_t r = (jobs[q].dst >> 1) + 1;
          while (r < rows && (used[r] || taken[r])) ++r;
          if (r >= rows) r = jobs[q].dst >> 1;
          taken[r] = 1;
          jobs[q].dst = (int32_t)r * 2;
          jobs[q + 1].dst = (int32_t)r * 2 + 1;
        }
      }
    }
    for (auto& j : jobs) { j.prow *= spread; j.dst = ((j.dst >> 1) * spread) * 2 + (j.dst & 1); }
    std::vector<GnxXoJob> up(jobs.size());
    for (size_t q = 0; q < jobs.size(); ++q) up[q] = lab_to_job(jobs[q]);
    CHK(hipMalloc((void**)&d_jobs[kind], (size_t)n_jobs * sizeof(GnxXoJob)));
    CHK(hipMemcpy(d_jobs[kind], up.data(), (size_t)n_jobs * sizeof(GnxXoJob), hipMemcpyHostToDevice));
  }
  // round-1 metadata (job kind 0): slots = rows; grow identity
  int32_t *d_grow, *d_offp, *d_offk, *d_bpoff, *d_bploci, *d_njobs;
  uint8_t* d_offs;
  u64 *d_paths, *d_sum;
  {
    std::vector<int32_t> grow(rows + B);
    for (int64_t i = 0; i < rows; ++i) grow[i] = (int32_t)(i * spread);
    for (int k = 0; k < B; ++k) grow[rows + k] = free_rows[k] * spread;
    CHK(hipMalloc((void**)&d_grow, grow.size() * 4));
    CHK(hipMemcpy(d_grow, grow.data(), grow.size() * 4, hipMemcpyHostToDevice));
  }
  CHK(hipMalloc((void**)&d_offp, par.size() * 4));
  CHK(hipMalloc((void**)&d_offk, key.size() * 4));
  CHK(hipMalloc((void**)&d_offs, st.size()));
  CHK(hipMalloc((void**)&d_bpoff, bp_off.size() * 4));
  CHK(hipMalloc((void**)&d_bploci, std::max<size_t>(bp_loci.size(), 1) * 4));
  CHK(hipMalloc((void**)&d_njobs, 4));
  CHK(hipMalloc((void**)&d_paths, paths.size() * 8));
  CHK(hipMalloc((void**)&d_sum, 8));
  CHK(hipMemcpy(d_offp, par.data(), par.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_offk, key.data(), key.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_offs, st.data(), st.size(), hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_bpoff, bp_off.data(), bp_off.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_bploci, bp_loci.data(), bp_loci.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_njobs, &n_jobs, 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_paths, paths.data(), paths.size() * 8, hipMemcpyHostToDevice));

  const double sparse_bytes = (double)n_jobs * W16 * 16 * 2;   // 1 read + 1 write per chunk
  const double dense_bytes = (double)n_jobs * W16 * 16 * 4;    // 2 hom + mask read, 1 write
  std::vector<Variant> V;
  auto add = [&](std::string n, std::function<void()> f, double bytes, bool check = true) {
    Variant v;
    v.name = n;
    v.launch = f;
    v.bytes = bytes;
    v.check = check;
    V.push_back(v);
  };
  const char* kn[NK] = {"nat/rnd-dst", "nat/asc-dst", "par0-sorted/asc-dst", "prow-sorted"};
#define SPARSE(U, NT, BPC, KIND)                                                                  \
  add(std::string("xo_sparse U=" #U " nt=" #NT " bpc=" #BPC " ") + kn[KIND], [=]() {             \
    hipLaunchKernelGGL((k_xo_sparse<U, NT>), dim3(256 * BPC), dim3(256), 0, 0, d_njobs, W16, G, G, \
                       d_jobs[KIND], d_bpoff, d_bploci, 0, 1024, nullptr, (const GnxJobBp*)nullptr); }, sparse_bytes)
#define DENSE(U, NT, BPC, KIND)                                                                   \
  add(std::string("xo_dense U=" #U " nt=" #NT " bpc=" #BPC " ") + kn[KIND], [=]() {              \
    hipLaunchKernelGGL((k_xo_dense<U, NT>), dim3(256 * BPC), dim3(256), 0, 0, d_njobs, W16, G, G,  \
                       d_jobs[KIND], (const u64x2*)d_paths, W16, 0, 1024, nullptr); }, dense_bytes)
#define COPYJ(U, NT, BPC, KIND)                                                                   \
  add(std::string("copy_jobs U=" #U " nt=" #NT " bpc=" #BPC " ") + kn[KIND], [=]() {             \
    hipLaunchKernelGGL((k_copy_jobs<U, NT>), dim3(256 * BPC), dim3(256), 0, 0, d_njobs, W16, G, G, \
                       d_jobs[KIND]); }, sparse_bytes, false)
#define R1(U, BPC)                                                                                \
  add("r1_stream U=" #U " bpc=" #BPC " nat/rnd-dst", [=]() {                                     \
    hipLaunchKernelGGL((k_r1_stream<U>), dim3(256 * BPC), dim3(256), 0, 0, (int64_t)B, W16, G, G,  \
                       d_grow, (int64_t)rows, d_offp, d_offk, d_offs, d_bpoff, d_bploci); },     \
      sparse_bytes)

  if (set == "dense") {
    DENSE(4, false, 32, 0); DENSE(4, true, 32, 0); DENSE(2, false, 32, 0); DENSE(2, true, 32, 0);
    DENSE(4, false, 16, 0); DENSE(4, true, 64, 0); DENSE(4, true, 32, 1); DENSE(3, true, 32, 1);
  } else {
    // the round-1 sweep, repeated properly
    R1(8, 32); R1(8, 16); R1(8, 64); R1(4, 32); R1(4, 64);
    // product kernel
    SPARSE(7, false, 32, 0); SPARSE(7, true, 32, 0); SPARSE(8, false, 32, 0); SPARSE(8, true, 32, 0);
    SPARSE(4, true, 32, 0); SPARSE(5, true, 32, 0); SPARSE(6, true, 32, 0);
    SPARSE(7, true, 32, 1); SPARSE(7, true, 32, 2); SPARSE(7, true, 32, 3);
    SPARSE(7, false, 32, 1); SPARSE(7, false, 32, 3);
    if (set == "all") {
      SPARSE(7, true, 8, 1); SPARSE(7, true, 16, 1); SPARSE(7, true, 64, 1); SPARSE(7, true, 128, 1);
      SPARSE(4, true, 64, 1); SPARSE(8, true, 64, 1); SPARSE(8, true, 16, 1);
      COPYJ(7, false, 32, 0); COPYJ(7, true, 32, 0); COPYJ(7, true, 32, 1); COPYJ(7, true, 32, 3);
      COPYJ(4, true, 32, 1); COPYJ(7, true, 64, 1);
      DENSE(4, false, 32, 0); DENSE(4, true, 32, 0); DENSE(4, true, 32, 1); DENSE(2, true, 32, 1);
    }
  }
  // plain streaming copy of a 5.5-GB span (the guide's achievable HBM figure)
  const int64_t n16 = (int64_t)n_jobs * W16;
  u64x2 *cp_a, *cp_b;
  CHK(hipMalloc((void**)&cp_a, (size_t)n16 * 16));
  CHK(hipMalloc((void**)&cp_b, (size_t)n16 * 16));
  CHK(hipMemset(cp_a, 1, (size_t)n16 * 16));
  add("stream_copy float4 grid=256*32", [=]() {
    hipLaunchKernelGGL(k_stream_copy, dim3(256 * 32), dim3(256), 0, 0, n16, cp_a, cp_b); },
      (double)n16 * 32, false);

  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  // warm-up + checksum of every variant
  for (auto& v : V) {
    v.launch();
    CHK(hipDeviceSynchronize());
    CHK(hipGetLastError());
    if (v.check) {
      CHK(hipMemset(d_sum, 0, 8));
      hipLaunchKernelGGL(k_checksum, dim3(2048), dim3(256), 0, 0, n_jobs, W16, G, d_jobs[0], d_sum);
      CHK(hipMemcpy(&v.sum, d_sum, 8, hipMemcpyDeviceToHost));
    }
  }
  for (int r = 0; r < reps; ++r)
    for (auto& v : V) {
      CHK(hipEventRecord(e0));
      v.launch();
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms);
    }
  printf("%-52s %8s %8s %8s %8s  %s\n", "variant", "min ms", "med ms", "mean ms", "max ms", "TB/s(med)");
  for (auto& v : V) {
    std::vector<float> s = v.ms;
    std::sort(s.begin(), s.end());
    double mean = 0;
    for (float x : s) mean += x;
    mean /= s.size();
    const float med = s[s.size() / 2];
    printf("%-52s %8.3f %8.3f %8.3f %8.3f  %6.2f", v.name.c_str(), s.front(), med, mean, s.back(),
           v.bytes / (med * 1e-3) / 1e12);
    if (v.check) printf("  sum=%016llx", v.sum);
    printf("\n");
  }
  // job kinds 0 vs 1..3 write different child rows; the checksum walks kind 0's rows, so
  // only kind-0 variants of one kernel family are comparable: report mismatches there
  u64 ref = 0;
  bool have = false, ok = true;
  for (auto& v : V)
    if (v.check && v.name.find("nat/rnd-dst") != std::string::npos) {
      if (!have) { ref = v.sum; have = true; }
      else if (v.sum != ref) { ok = false; printf("CHECKSUM MISMATCH: %s\n", v.name.c_str()); }
    }
  printf(ok ? "checksums of the nat/rnd-dst variants agree\n" : "CHECKSUMS DIFFER\n");
  return ok ? 0 : 2;
}
