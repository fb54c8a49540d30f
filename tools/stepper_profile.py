#!/usr/bin/env python3
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
if os.environ.get("INIT_TORCH_CUDA"):
    torch.cuda.set_device(0); _ = torch.zeros(1, device="cuda")
from geonomics_amd.parallel import Comm, DeviceShard, TiledStepper
cfg = dict(bench.WORKLOADS['c4_metric'])
dev, _, _ = bench.build_device(cfg, 42, 0)
sh = DeviceShard(dev)
st = TiledStepper(sh, Comm(None), cfg['W'], cfg['H'], 10.0, move=True, max_id=cfg['N'] - 1, fixed_births=1)
for _ in range(3):
    st.step(True, False)
bench.setup_genomes(dev, cfg, 42); sh.has_genomes = True
st.step(False, True)
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for _ in range(5):
    st.step(False, True)
dev.synchronize()
print('ms/step', 1e3 * (time.perf_counter() - t) / 5)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
if st.profile:
    n = sum(1 for _ in range(1))
    tot = sum(st.phase_s.values())
    for k, v in st.phase_s.items():
        print('  %-22s %8.3f ms total' % (k, 1e3 * v))
