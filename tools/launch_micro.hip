// What a kernel launch, an event hand-over and a graph replay cost the HOST and the GPU on
// this box: the numbers the small-population path is designed against.
//   hipcc --offload-arch=gfx950 -O2 tools/launch_micro.hip -o tools/launch_micro && tools/launch_micro
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

struct Big { char b[1024]; };
__global__ void k_small(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += 1; }
__global__ void k_big(Big a, int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += a.b[5]; }

__global__ void k_work(int* p, int iters) {
  int v = p[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1664525 + 1013904223;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

// persistent kernel: `phases` grid-wide barriers (counter + poll), to price a barrier
__global__ void k_barriers(unsigned int* ctr, int phases, int* p) {
  unsigned int target = 0;
  for (int ph = 0; ph < phases; ++ph) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += 1;
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
        __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  int* d = nullptr;
  CHK(hipMalloc(&d, 64));
  CHK(hipMemset(d, 0, 64));
  hipStream_t s, s2;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int R = 200, K = 25;
  Big big{};
  for (int rep = 0; rep < 2; ++rep) {
    // (a) K small launches per round on one stream
    CHK(hipStreamSynchronize(s));
    double t0 = now();
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_small, dim3(512), dim3(256), 0, s, d);
    double t1 = now();
    CHK(hipStreamSynchronize(s));
    double t2 = now();
    if (rep) printf("small launches, one stream : host %.2f us per launch, end-to-end %.2f us per kernel\n",
                    1e6 * (t1 - t0) / (R * K), 1e6 * (t2 - t0) / (R * K));
    // (b) 1-KB arguments
    t0 = now();
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, s, big, d);
    t1 = now();
    CHK(hipStreamSynchronize(s));
    t2 = now();
    if (rep) printf("1-KB arguments             : host %.2f us per launch, end-to-end %.2f us per kernel\n",
                    1e6 * (t1 - t0) / (R * K), 1e6 * (t2 - t0) / (R * K));
    // (c) event hand-over to a second stream and back, every 5th launch
    hipEvent_t e1, e2;
    CHK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    t0 = now();
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < K; ++k) {
        hipLaunchKernelGGL(k_small, dim3(512), dim3(256), 0, s, d);
        if (k % 5 == 4) {
          CHK(hipEventRecord(e1, s));
          CHK(hipStreamWaitEvent(s2, e1, 0));
          hipLaunchKernelGGL(k_small, dim3(512), dim3(256), 0, s2, (int*)nullptr);
          CHK(hipEventRecord(e2, s2));
          CHK(hipStreamWaitEvent(s, e2, 0));
        }
      }
    t1 = now();
    CHK(hipStreamSynchronize(s));
    t2 = now();
    if (rep) printf("+ 5 hand-overs per 25      : host %.2f us per round of 25, end-to-end %.2f us per round\n",
                    1e6 * (t1 - t0) / R, 1e6 * (t2 - t0) / R);
    // (d) graph of K kernel nodes on one stream
    hipGraph_t g;
    hipGraphExec_t ex;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_small, dim3(512), dim3(256), 0, s, d);
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ex, s));
    CHK(hipStreamSynchronize(s));
    t0 = now();
    for (int r = 0; r < R; ++r) CHK(hipGraphLaunch(ex, s));
    t1 = now();
    CHK(hipStreamSynchronize(s));
    t2 = now();
    if (rep) printf("graph of 25 kernel nodes   : host %.2f us per replay, end-to-end %.2f us per replay\n",
                    1e6 * (t1 - t0) / R, 1e6 * (t2 - t0) / R);
    CHK(hipGraphExecDestroy(ex));
    CHK(hipGraphDestroy(g));
  }
  // (f) S streams side by side, each K kernels on `wgs` workgroups: do they overlap?
  for (int shape = 0; shape < 3; ++shape) {
    const int wgs = shape == 0 ? 64 : 1024;
    const int iters = shape == 2 ? 4000 : 250;
    printf("-- kernels of %d workgroups x 256 threads, %d iterations each\n", wgs, iters);
    const int S = 8;
    hipStream_t st[S];
    hipGraphExec_t gx[S];
    int* buf = nullptr;
    CHK(hipMalloc(&buf, (size_t)S * 1024 * 256 * sizeof(int)));
    for (int q = 0; q < S; ++q) CHK(hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking));
    for (int q = 0; q < S; ++q) {
      hipGraph_t g;
      CHK(hipStreamBeginCapture(st[q], hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < K; ++k)
        hipLaunchKernelGGL(k_work, dim3(wgs), dim3(256), 0, st[q], buf + (size_t)q * 1024 * 256, iters);
      CHK(hipStreamEndCapture(st[q], &g));
      CHK(hipGraphInstantiate(&gx[q], g, nullptr, nullptr, 0));
      CHK(hipGraphDestroy(g));
    }
    for (int use = 1; use <= S; use *= 2) {
      for (int mode = 0; mode < 2; ++mode) {
        for (int q = 0; q < use; ++q) CHK(hipStreamSynchronize(st[q]));
        double t0 = now();
        for (int r = 0; r < 40; ++r)
          for (int q = 0; q < use; ++q) {
            if (mode == 0) {
              for (int k = 0; k < K; ++k)
                hipLaunchKernelGGL(k_work, dim3(wgs), dim3(256), 0, st[q], buf + (size_t)q * 1024 * 256, iters);
            } else {
              CHK(hipGraphLaunch(gx[q], st[q]));
            }
          }
        double t1 = now();
        for (int q = 0; q < use; ++q) CHK(hipStreamSynchronize(st[q]));
        double t2 = now();
        printf("%d stream(s), %s: host %.1f us, end-to-end %.1f us per round of 25 kernels and stream\n",
               use, mode ? "graph replays " : "direct launches", 1e6 * (t1 - t0) / (40 * use),
               1e6 * (t2 - t0) / (40 * use));
      }
    }
    for (int q = 0; q < S; ++q) {
      CHK(hipGraphExecDestroy(gx[q]));
      CHK(hipStreamDestroy(st[q]));
    }
    CHK(hipFree(buf));
  }
  // (e) a persistent kernel with 25 grid barriers, 256 and 64 workgroups of 256 / 1024 threads
  unsigned int* ctr = nullptr;
  CHK(hipMalloc(&ctr, 4));
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int blocks = cfg < 2 ? 256 : 64, threads = (cfg & 1) ? 1024 : 256;
    double best = 1e9;
    for (int r = 0; r < 20; ++r) {
      CHK(hipMemsetAsync(ctr, 0, 4, s));
      CHK(hipStreamSynchronize(s));
      double t0 = now();
      hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(threads), 0, s, ctr, K, d);
      CHK(hipStreamSynchronize(s));
      best = std::min(best, now() - t0);
    }
    printf("persistent kernel, 25 grid barriers, %3d x %4d threads: %.2f us in all (launch + sync included)\n",
           blocks, threads, 1e6 * best);
  }
  return 0;
}
