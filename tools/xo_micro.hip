// Ceiling study for the crossover access pattern: copy 12.5-KB half-rows of a
// large table (rows of 2 x W16 16-byte chunks), one wave per half-row.
//   hipcc --offload-arch=gfx950 -O3 -o xo_micro tools/xo_micro.hip && ./xo_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
struct alignas(16) u64x2 { u64 a, b; };

template <int U, bool NT_LD, bool NT_ST>
__global__ void __launch_bounds__(256)
k_copy(int64_t n, int W16, const u64x2* __restrict__ G, u64x2* __restrict__ Gout,
       const int32_t* __restrict__ src, const int32_t* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t g = wave0; g < n; g += n_waves) {
    const int s = __builtin_amdgcn_readfirstlane(src[g]);
    const int d = __builtin_amdgcn_readfirstlane(dst[g]);
    const u64x2* in = G + (int64_t)s * W16;
    u64x2* out = Gout + (int64_t)d * W16;
    for (int c0 = lane; c0 < W16; c0 += 64 * U) {
      u64x2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = min(c0 + u * 64, W16 - 1);
        if (NT_LD) {
          v[u].a = __builtin_nontemporal_load(&in[c].a);
          v[u].b = __builtin_nontemporal_load(&in[c].b);
        } else {
          v[u] = in[c];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64;
        if (c < W16) {
          if (NT_ST) {
            __builtin_nontemporal_store(v[u].a, &out[c].a);
            __builtin_nontemporal_store(v[u].b, &out[c].b);
          } else {
            out[c] = v[u];
          }
        }
      }
    }
  }
}

template <int U, bool NL, bool NS>
static double run(const char* name, int grid, int64_t n, int W16, u64x2* G, const int32_t* s,
                  const int32_t* d, int reps) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_copy<U, NL, NS>), dim3(grid), dim3(256), 0, 0, n, W16, G, G, s, d);
  CHK(hipDeviceSynchronize());
  float best = 1e9, tot = 0;
  for (int r = 0; r < reps; ++r) {
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((k_copy<U, NL, NS>), dim3(grid), dim3(256), 0, 0, n, W16, G, G, s, d);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms);
    tot += ms;
  }
  double bytes = (double)n * W16 * 16 * 2;
  printf("%-44s U=%d nt_ld=%d nt_st=%d grid=%6d  avg %.3f ms  best %.3f ms  %.2f TB/s (avg)\n", name, U,
         (int)NL, (int)NS, grid, tot / reps, best, bytes / (tot / reps * 1e-3) / 1e12);
  return tot / reps;
}

int main(int argc, char** argv) {
  const int W16 = 784;                       // L = 100000 -> 1568 words -> 784 chunks / homologue
  const int64_t rows = 1600000;              // individuals' rows; half-rows = 2 * rows (40 GB)
  const int64_t n = 410000;                  // gametes per step at the metric workload
  u64x2* G;
  CHK(hipMalloc((void**)&G, (size_t)rows * 2 * W16 * 16));
  CHK(hipMemset(G, 1, (size_t)rows * 2 * W16 * 16));
  std::mt19937_64 rng(1);
  std::vector<int32_t> seq_s(n), seq_d(n), rnd_s(n), rnd_d(n), live_s(n), live_d(n);
  // sequential: src half-rows 0..n-1, dst n..2n-1
  for (int64_t i = 0; i < n; ++i) { seq_s[i] = (int32_t)i; seq_d[i] = (int32_t)(n + i); }
  // random: src = random half-row among the first 1.2M rows; dst = both homologues of random rows above
  std::vector<int32_t> perm(rows);
  for (int64_t i = 0; i < rows; ++i) perm[i] = (int32_t)i;
  std::shuffle(perm.begin(), perm.end(), rng);
  for (int64_t i = 0; i < n; ++i) {
    rnd_s[i] = (int32_t)((rng() % 1200000) * 2 + (rng() & 1));
    rnd_d[i] = (int32_t)(perm[i / 2] * 2 + (i & 1));        // child row: two gametes side by side
  }
  // "live" layout: like rnd but destination rows sequential (fresh rows)
  for (int64_t i = 0; i < n; ++i) { live_s[i] = rnd_s[i]; live_d[i] = (int32_t)(1200000 * 2 + i); }
  int32_t *ds, *dd;
  CHK(hipMalloc((void**)&ds, n * 4));
  CHK(hipMalloc((void**)&dd, n * 4));
  auto up = [&](std::vector<int32_t>& s, std::vector<int32_t>& d) {
    CHK(hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dd, d.data(), n * 4, hipMemcpyHostToDevice));
  };
  const int reps = 8;
  int grids[] = {256 * 8, 256 * 32, 256 * 64, (int)((n + 3) / 4)};
  up(seq_s, seq_d);
  for (int g : grids) run<4, false, true>("sequential src, sequential dst", g, n, W16, G, ds, dd, reps);
  run<8, false, true>("sequential src, sequential dst", 256 * 32, n, W16, G, ds, dd, reps);
  run<4, false, false>("sequential src, sequential dst", 256 * 32, n, W16, G, ds, dd, reps);
  run<4, true, true>("sequential src, sequential dst", 256 * 32, n, W16, G, ds, dd, reps);
  up(live_s, live_d);
  for (int g : grids) run<4, false, true>("random src, sequential dst", g, n, W16, G, ds, dd, reps);
  run<4, true, true>("random src, sequential dst", 256 * 32, n, W16, G, ds, dd, reps);
  up(rnd_s, rnd_d);
  for (int g : grids) run<4, false, true>("random src, random dst rows", g, n, W16, G, ds, dd, reps);
  run<8, false, true>("random src, random dst rows", 256 * 32, n, W16, G, ds, dd, reps);
  run<2, false, true>("random src, random dst rows", 256 * 32, n, W16, G, ds, dd, reps);
  run<4, false, false>("random src, random dst rows", 256 * 32, n, W16, G, ds, dd, reps);
  run<4, true, true>("random src, random dst rows", 256 * 32, n, W16, G, ds, dd, reps);
  run<4, true, false>("random src, random dst rows", 256 * 32, n, W16, G, ds, dd, reps);
  return 0;
}
