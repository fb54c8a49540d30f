#!/usr/bin/env python3
"""Long run of the metric workload with the bookkeeping checked along the way: the
tables of the shared genome blocks are sound after a collection (gnx_debug_halves), blocks in use
level off (no leak), the population stays at its carrying capacity.
    python tools/soak.py [steps] [check every]
GNX_SOAK_WALK=1: the steps between two checks in one gnx_walk call each (the device-driven step
where the handle takes it: GNX_SOAK_WORKLOAD=c2 / c3).  GNX_SOAK_TILE=1: every step through
gnx_tile_step on a 1 x 1 grid (the tile protocol with tile-major offspring ids, one rank)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = bench.WORKLOADS[os.environ.get('GNX_SOAK_WORKLOAD', 'c4_metric')]
dev, _, _ = bench.build_device(cfg, seed=42, device=0)
for _ in range(3):
    dev.step(True, False)
bench.setup_genomes(dev, cfg, 42)
t0 = time.time()
import numpy as np                                      # noqa: E402
n_mut = int(os.environ.get('GNX_SOAK_MUTATE', '0'))    # random mutations per step (copy-on-write)
rng = np.random.RandomState(1)
walk = bool(os.environ.get('GNX_SOAK_WALK'))
tile = bool(os.environ.get('GNX_SOAK_TILE'))
if tile:
    from geonomics_amd import _native as nat
    dev.tile_set(1, 1, 0, 0)
    dev.comm_init_single()
    dev.comm_selftest()
    dev.set_max_id(int(dev.download(nat.F_ID).max()))
t = 0
t_walk = 0.0            # time inside the steps alone (the checks between the pieces excluded)
seg_open = False
while t < steps:
    if not seg_open:                 # a timed stretch runs from one check to the next
        dev.synchronize()
        tw0 = time.perf_counter()
        seg_open = True
    if walk:
        k = min(every - (t % every), steps - t)
        dev.walk(k, False, True)
        t += k
    elif tile:
        dev.tile_step(False, True, False)
        t += 1
    else:
        dev.step(False, True)
        t += 1
    if t % every == 0 or t == steps:
        dev.synchronize()
        t_walk += time.perf_counter() - tw0
        seg_open = False
    if n_mut:
        dev.mutate(rng.randint(0, dev.N, n_mut).astype(np.int64),
                   rng.randint(0, cfg['L'], n_mut).astype(np.int32),
                   rng.randint(0, 2, n_mut).astype(np.uint8))
    if t % every == 0 or t == steps:
        rows, broken, refs, used, free, total = (int(v) for v in dev.debug_halves())
        n, b, d = dev.counts()
        ok = broken == 0 and used + free == total and used <= 2 * rows
        print('step %5d  N=%d births=%d deaths=%d  blocks: logical %d  physical in use %d '
              '(%.1f %% shared)  free %d  %s  %.1f s%s' % (
                  t, n, b, d, 2 * rows, used, 100.0 * (1 - used / max(2 * rows, 1)), free,
                  ('ok' if ok else 'INCONSISTENT') + '  %.4f ms/step so far' % (1e3 * t_walk / t),
                  time.time() - t0,
                  '  (device-driven steps so far: %d)' % dev.totals()['dd_steps'] if walk else ''),
              flush=True)
        if not ok:
            sys.exit(1)
dev.close()
