#!/usr/bin/env python3
"""ms/step of the metric workload at its steady state: walk <warm> steps, then time <steps> through gnx_walk.
    python tools/steady_ab.py [warm] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = bench.WORKLOADS[os.environ.get('AB_WORKLOAD', 'c4_metric')]
dev, _, _ = bench.build_device(cfg, 42, 0)
for _ in range(3):
    dev.step(True, False)
bench.setup_genomes(dev, cfg, 42)
dev.walk(warm, False, True)
dev.synchronize()
dev.reset_totals()
t0 = time.perf_counter()
dev.walk(steps, False, True)
dev.synchronize()
dt = time.perf_counter() - t0
tot = dev.totals()
print('N=%d  ms/step=%.4f  ind-steps/s=%.3e' % (dev.N, 1e3 * dt / steps, tot['ind_steps'] / dt))
