#!/usr/bin/env python3
"""The metric workload through the Geonomics API itself: make_model -> walk('burn') ->
walk(T, 'main'), timed as SURVEY 8(d) defines the metric (sum of N_t over wall time of
mod.walk).  Compare with bench.py, which drives the fused C-ABI step directly."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
import geonomics_amd as gnx                            # noqa: E402
from geonomics_amd.sim import params as P              # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c4_metric'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = bench.WORKLOADS[name]
W, H = cfg['W'], cfg['H']
d = bench.model_api_params(cfg, name, T)
# GNX_MODEL_BENCH_MATE=nearest|inverse: the non-default mate choices at the clumped equilibrium
mode = os.environ.get('GNX_MODEL_BENCH_MATE', 'uniform')
mating = d['comm']['species']['spp_0']['mating']
mating['choose_nearest_mate'] = mode == 'nearest'
mating['inverse_dist_mating'] = mode == 'inverse'
os.environ.setdefault('GNX_CAP_FACTOR', '2.0')
t0 = time.time()
mod = gnx.make_model(d)
t1 = time.time()
mod.walk(10000, 'burn', verbose=False)
spp = mod.comm[0]
t2 = time.time()
print('make_model %.1f s, burn-in %d steps %.1f s (incl. genome assignment), N=%d' % (
    t1 - t0, len(spp.Nt), t2 - t1, len(spp)), flush=True)
mod.walk(5, 'main', verbose=False)
spp._dev.synchronize()
n0 = len(spp.Nt)
prof = bool(os.environ.get('GNX_MODEL_BENCH_PROFILE'))
if prof:
    spp._dev.profiling(True)
    import cProfile
    pr = cProfile.Profile()
    pr.enable()
t3 = time.perf_counter()
mod.walk(T, 'main', verbose=False)
spp._dev.synchronize()
dt = time.perf_counter() - t3
if prof:
    pr.disable()
    import pstats
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
    kt = spp._dev.kernel_times()
    for k, v in kt.items():
        print('  %-11s %8.3f ms/step' % (k, v['ms'] / T))
    print('  kernel sum  %8.3f ms/step' % (sum(v['ms'] for v in kt.values()) / T))
ind_steps = sum(spp.Nt[n0 - 1:n0 - 1 + T])
print('Model.walk: %.3f ms/step, %.3e individual-timesteps/s (N=%d, births/step=%.0f)' % (
    1e3 * dt / T, ind_steps / dt, len(spp), np.mean(spp.n_births[-T:])))
if os.environ.get('GNX_MODEL_BENCH_CLUMP'):
    xy = mod.get_coords()
    cs = 10.0
    Hc, _, _ = np.histogram2d(xy[:, 1], xy[:, 0], bins=[int(H / cs) + 1, int(W / cs) + 1],
                              range=[[0, (int(H / cs) + 1) * cs], [0, (int(W / cs) + 1) * cs]])
    from scipy.ndimage import uniform_filter
    S9 = uniform_filter(Hc, 3, mode='constant') * 9
    print('cells %d  max per cell %d  mean %.2f  E[n^2]/E[n]^2 %.1f' % (
        Hc.size, Hc.max(), Hc.mean(), (Hc ** 2).mean() / Hc.mean() ** 2))
    print('candidates scanned per individual (3x3 cells): mean %.0f  (uniform would be %.0f)' % (
        (Hc * S9).sum() / Hc.sum(), 9 * Hc.mean()))
