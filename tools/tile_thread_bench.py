#!/usr/bin/env python3
"""Device-resident tile transport at the metric workload on ONE GPU: `world`
tiles as threads of one process (tests/_local_comm.py instead of RCCL), phase
times per step.  Kernels of the tiles share the GPU, so the compute phases are
about `world` times slower than on a real node; the exchange phases show the
host-side cost of the device path (grouping, counts, address hand-over)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('GNX_TILE_PROFILE', '1')
import torch                                         # noqa: E402
import bench                                         # noqa: E402
from _local_comm import Hub, LocalComm               # noqa: E402
from geonomics_amd.parallel import DeviceShard, TiledStepper, tile_grid   # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
workload = sys.argv[3] if len(sys.argv) > 3 else 'c4_metric'
cfg = bench.WORKLOADS[workload]
grid = tile_grid(world)
hub = Hub(world, library_group=os.environ.get('GNX_TILE_V3', '1') != '0')
out = [None] * world


def body(rank):
    try:
        torch.cuda.set_device(0)
        dev, _, _ = bench.build_device(cfg, seed=42, device=0, grid=grid, rank=rank,
                                         cap_factor=float(os.environ.get("GNX_TILE_CAP", "1.6")))
        shard = DeviceShard(dev)
        st = TiledStepper(shard, LocalComm(hub, rank), cfg['W'] * grid[1], cfg['H'] * grid[0],
                          10.0, move=True, max_id=cfg['N'] * world - 1, grid=grid, fixed_births=1)
        for _ in range(3):
            st.step(True, False)
        bench.setup_genomes(dev, cfg, 42)
        shard.has_genomes = True
        for _ in range(2):
            st.step(False, True)
        st.phase_s.clear()
        st.bytes_sent = 0
        dev.synchronize()
        hub.barrier.wait()
        t0 = time.perf_counter()
        check = int(os.environ.get('GNX_TILE_CHECK', '0'))      # block bookkeeping every k steps
        prof = None
        if rank == 0 and os.environ.get('GNX_TT_CPROFILE'):      # where rank 0's host time goes
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        # GNX_TT_WALK=0: one call per step (a compaction in every step); default: the timed steps in
        # one call of TiledStepper.walk (gnx_tile_walk: no compaction between them)
        use_walk = os.environ.get('GNX_TT_WALK', '1') != '0' and not check
        if use_walk:
            n, _, _ = st.walk(steps, False, True, exact=False)
        for k in range(0 if use_walk else steps):
            n = st.step(False, True, exact=False) if st.v2 else st.step(False, True)
            if check and (k + 1) % check == 0:
                rows, broken, refs, used, free, total = (int(v) for v in dev.debug_halves())
                assert broken == 0 and used + free == total and used <= 2 * rows, (
                    rank, k, rows, broken, refs, used, free, total)
                if rank == 0:
                    print('step %d: N=%s, blocks consistent on every tile (rank 0: %d logical, '
                          '%d physical)' % (k + 1, n, 2 * rows, used), flush=True)
        dev.synchronize()
        dt = time.perf_counter() - t0
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats('tottime').print_stats(28)
        out[rank] = (dt / steps * 1e3, {k: v / steps * 1e3 for k, v in st.phase_s.items()},
                     st.bytes_sent / steps, n)
        dev.close()
    except BaseException:
        hub.abort()
        raise


ths = [threading.Thread(target=body, args=(r,)) for r in range(world)]
[t.start() for t in ths]
[t.join() for t in ths]
for r, o in enumerate(out):
    if o:
        print('rank %d: %.2f ms/step, %.1f MB sent/step, (N, births, deaths) = %s' % (
            r, o[0], o[2] / 1e6, o[3]))
        print('   ' + '  '.join('%s %.2f' % kv for kv in o[1].items()))
