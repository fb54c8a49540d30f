#!/usr/bin/env python3
"""Per-step crossover rate over a long run of the metric workload (does the kernel's
achieved bandwidth depend on how scrambled the genome rows have become?).  GPU only."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(bench.WORKLOADS['c4_metric'])
dev, _, _ = bench.build_device(cfg, 42, 0)
for _ in range(3):
    dev.step(True, False)
bench.setup_genomes(dev, cfg, 42)
dev.profiling(2)
print('step      N   births  with_genome  xo_ms   TB/s')
for t in range(steps):
    n0 = dev.N
    dev.step(False, True)
    b = dev.counts()[1]
    kt = dev.kernel_times()['crossover']
    if t % 10 == 0 or t < 5:
        print('%4d %8d %7d %9d  %6.3f  %5.2f' % (t, n0, b, dev.last_crossover_births, kt['ms'],
                                                 kt['bytes'] / max(kt['ms'], 1e-9) / 1e9), flush=True)
jobs = dev.last_crossover_jobs()
if len(sys.argv) > 2:
    jobs.tofile(sys.argv[2])
    print('wrote %d jobs to %s' % (len(jobs), sys.argv[2]))
rows = dev.download(5 + 3)     # F_GROW
print('rows: min %d max %d; mean |row[i+1]-row[i]| in slot order %.0f' % (
    rows.min(), rows.max(), np.abs(np.diff(rows.astype(np.int64))).mean()))
