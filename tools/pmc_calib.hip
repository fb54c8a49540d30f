// Calibration of the L2's memory-side request counters on the crossover's access pattern
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").
//
// A wave copies one 640-byte block (40 lanes x 16 bytes, 128-byte aligned) from a random place of
// a pool to another random place - k_xo_sparse_pair without its records and its blend - so the
// bytes are known: 5 lines read, 5 lines written per block.  Variants:
//   pool of 1 / 32 / 200 GiB  (does the address translation of a large pool add requests?  The
//                              metric workload's genome table spans 200 GB of addresses)
//   + one lane's 16 bytes from a third random block   (the switch point's chunk: one request?)
//   + one lane's 16 bytes from a line of the block the wave is loading anyway, by a second
//     instruction - a plain load (EXTRA 2) or a non-temporal one (3): in the crossover the lanes
//     behind the switch point read the SAME line of the other homologue as the blending lane
//     does; is a line that a non-temporal load brought in still there for the second request?
// Run under  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace
// and       rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace ;
// tools/pmc_calib_summary.py divides the counters by the block count printed here.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                     \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct alignas(16) u64x2 {
  unsigned long long a, b;
};
struct Job {
  uint32_t src, dst, other, lane;
};

// POOL tags the kernel's name (1 / 32 GiB), EXTRA = the single-lane load from a third block
template <int POOL, int EXTRA>
__global__ void __launch_bounds__(256)
k_calib(int n_jobs, const u64x2* __restrict__ G, u64x2* __restrict__ Gout,
        const Job* __restrict__ jobs) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int n_waves = (int)gridDim.x * 4;
  const int c = min(lane, 39);
  for (int j = (int)blockIdx.x * 4 + wv; j < n_jobs; j += n_waves) {
    const Job jb = jobs[j];
    const u64x2* sp = G + (int64_t)jb.src * 40 + c;
    u64x2 v;                                    // (as xo_load<true> / xo_store, csrc/gnx_xo.h)
    v.a = __builtin_nontemporal_load(&sp->a);
    v.b = __builtin_nontemporal_load(&sp->b);
    if (EXTRA == 1 && lane == (int)jb.lane) {
      const u64x2 x = G[(int64_t)jb.other * 40 + c];
      v.a ^= x.a & 1ull;
      v.b ^= x.b & 1ull;
    }
    if (EXTRA >= 2 && lane == (int)jb.lane) {
      const u64x2* xp = G + (int64_t)jb.src * 40 + (c ^ 1);
      u64x2 x;
      if (EXTRA == 2) {
        x = *xp;
      } else {
        x.a = __builtin_nontemporal_load(&xp->a);
        x.b = __builtin_nontemporal_load(&xp->b);
      }
      v.a ^= x.a & 1ull;
      v.b ^= x.b & 1ull;
    }
    if (lane < 40) {
      u64x2* dp = Gout + (int64_t)jb.dst * 40 + lane;
      __builtin_nontemporal_store(v.a, &dp->a);
      __builtin_nontemporal_store(v.b, &dp->b);
    }
  }
}

template <int POOL, int EXTRA>
static void run(int n_jobs, u64x2* G, Job* jobs, int reps) {
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((k_calib<POOL, EXTRA>), dim3(2048), dim3(256), 0, 0, n_jobs, G, G, jobs);
  CHK(hipDeviceSynchronize());
}

int main() {
  const int n_jobs = 350000;
  const size_t huge = (size_t)200 << 30, big = (size_t)32 << 30, small = (size_t)1 << 30;
  u64x2* G;
  CHK(hipMalloc(&G, huge));
  CHK(hipMemset(G, 0, huge));
  std::vector<Job> hj(n_jobs);
  Job* jobs[3];
  for (int k = 0; k < 3; ++k) {
    const uint64_t nblk = (k == 2 ? huge : k ? big : small) / 640;
    uint64_t s = 0x9E3779B97F4A7C15ull + k;
    auto next = [&]() {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      return (s >> 24);
    };
    for (int j = 0; j < n_jobs; ++j) {
      // sources anywhere, destinations spread over the pool without repeats
      hj[j].src = (uint32_t)(next() % nblk);
      hj[j].other = (uint32_t)(next() % nblk);
      hj[j].dst = (uint32_t)(((uint64_t)j * (nblk / n_jobs)) + (next() % (nblk / n_jobs)));
      hj[j].lane = (uint32_t)(next() % 40);
    }
    CHK(hipMalloc(&jobs[k], n_jobs * sizeof(Job)));
    CHK(hipMemcpy(jobs[k], hj.data(), n_jobs * sizeof(Job), hipMemcpyHostToDevice));
  }
  const int reps = 4;
  run<1, 0>(n_jobs, G, jobs[0], reps);
  run<32, 0>(n_jobs, G, jobs[1], reps);
  run<1, 1>(n_jobs, G, jobs[0], reps);
  run<32, 1>(n_jobs, G, jobs[1], reps);
  run<32, 2>(n_jobs, G, jobs[1], reps);
  run<32, 3>(n_jobs, G, jobs[1], reps);
  run<200, 0>(n_jobs, G, jobs[2], reps);
  run<200, 1>(n_jobs, G, jobs[2], reps);
  printf("blocks_per_launch %d  block_bytes 640  launches_per_variant %d\n", n_jobs, reps);
  return 0;
}
