#!/usr/bin/env python3
"""Host-side cost of each tile entry point at world = 1 (GPU only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geonomics_amd import _native as nat
cfg = dict(bench.WORKLOADS['c4_metric'])
dev, _, _ = bench.build_device(cfg, 42, 0)
for _ in range(3):
    dev.step(True, False)
bench.setup_genomes(dev, cfg, 42)
dev.step(False, True)
dev.tile_set(1, 1, 0, 0)
T = {}
def tm(name, fn):
    dev.synchronize(); t = time.perf_counter(); r = fn(); dev.synchronize()
    T.setdefault(name, []).append(1e3 * (time.perf_counter() - t)); return r
max_id = None
import torch
for it in range(6):
    tm('age', dev.age); tm('move', dev.move)
    tm('export_migrants', lambda: dev.tile_export_migrants(True))
    tm('export_halo', lambda: dev.tile_export_halo())
    P, B = tm('tile_pairs', lambda: dev.tile_pairs(False))
    ids, nb = tm('pair_info', dev.tile_pair_info)
    goff = tm('searchsorted_cpu', lambda: (torch.searchsorted(torch.from_numpy(ids), torch.from_numpy(ids))).numpy().astype(np.int64))
    if max_id is None:
        max_id = int(dev.download(nat.F_ID).max())
    tm('offspring', lambda: dev.tile_offspring(False, max_id + 1, goff))
    max_id += B; dev.set_max_id(max_id)
    tm('finish_births', lambda: dev.tile_finish_births(False))
    b0 = tm('get_bins', lambda: dev.get_bins(0)); tm('set_bins', lambda: dev.set_bins(0, b0))
    tm('die', lambda: dev.tile_die(False, True, True))
    dev.step_index = dev.step_index + 1
for k, v in T.items():
    print('%-18s %8.3f ms (min %.3f)' % (k, np.mean(v[1:]), np.min(v)))
print('N', dev.N, 'P', P)
