#!/usr/bin/env python3
"""Per-kernel HIP-event times of the hot path without genomes (burn-mode steps)
or with them (--genomes): a quick loop for kernel work.  GPU only."""
import argparse
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geonomics_amd import _native as nat

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='c4_metric')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--genomes', action='store_true')
ap.add_argument('--no-traits', action='store_true')
ap.add_argument('--no-profile', action='store_true')
ap.add_argument('--tile-step', action='store_true', help='gnx_tile_step on a 1 x 1 grid (one rank, no transport): the tile protocol alone')
ap.add_argument('--walk', action='store_true', help='gnx_walk: counts on the device, one graph launch per step')
a = ap.parse_args()
cfg = dict(bench.WORKLOADS[a.workload])
if not a.genomes:
    cfg['L'] = 0
    cfg['n_traits'] = 0
if a.no_traits:
    cfg['n_traits'] = 0
dev, _, _ = bench.build_device(cfg, 42, 0)
for _ in range(3):
    dev.step(True, False)
if a.genomes:
    bench.setup_genomes(dev, cfg, 42)
for _ in range(2):
    dev.step(not a.genomes, a.genomes)
if a.tile_step:
    dev.tile_set(1, 1, 0, 0)
    dev.comm_init_single()
    dev.set_max_id(int(dev.download(nat.F_ID).max()))
    for _ in range(3):
        dev.tile_step(not a.genomes, a.genomes, False)
if a.walk:
    dev.walk(4, not a.genomes, a.genomes)      # (graph capture: a fixed cost of the first call)
dev.profiling(not a.no_profile)
import time
dev.synchronize()
t0 = time.perf_counter()
dev.reset_totals()
if a.walk:
    dev.walk(a.steps, not a.genomes, a.genomes)
elif a.tile_step:
    for _ in range(a.steps):
        dev.tile_step(not a.genomes, a.genomes, False)
else:
    for _ in range(a.steps):
        dev.step(not a.genomes, a.genomes)
dev.synchronize()
dt = time.perf_counter() - t0
tot = dev.totals()
n = tot['ind_steps']
kt = dev.kernel_times()
print('N=%d  ms/step=%.3f  ind-steps/s=%.3e  dd_steps=%d' % (dev.N, 1e3 * dt / a.steps, n / dt, tot['dd_steps']))
for k, v in kt.items():
    gbs = v['bytes'] / (v['ms'] * 1e-3) / 1e9 if v['ms'] > 0 else 0.0
    print('  %-14s %8.3f ms/step  (%d launches)  %8.1f MB/launch  %7.0f GB/s' % (
        k, v['ms'] / a.steps, v['launches'], v['bytes'] / max(v['launches'], 1) / 1e6, gbs))
print('  sum         %8.3f' % (sum(v['ms'] for v in kt.values()) / a.steps))
