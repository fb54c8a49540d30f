#!/usr/bin/env python3
"""Device-memory and throughput stability of a long run (GPU only): free device memory and
ms/step at the start and at the end of many steps of a mid-size model, single tile and two
in-process tiles."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402

cfg = dict(bench.WORKLOADS['small'])
dev, _, _ = bench.build_device(cfg, 42, 0)
for _ in range(3):
    dev.step(True, False)
bench.setup_genomes(dev, cfg, 42)
for _ in range(20):
    dev.step(False, True)
dev.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
for _ in range(200):
    dev.step(False, True)
dev.synchronize()
t_first = (time.perf_counter() - t0) / 200
for _ in range(2600):
    dev.step(False, True)
dev.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    dev.step(False, True)
dev.synchronize()
t_last = (time.perf_counter() - t0) / 200
free1 = torch.cuda.mem_get_info()[0]
print('steps 3000  N=%d  free memory change %+.1f MB  ms/step first %.3f last %.3f' % (
    dev.N, (free1 - free0) / 1e6, 1e3 * t_first, 1e3 * t_last))
