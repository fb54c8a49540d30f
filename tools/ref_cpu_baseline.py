#!/usr/bin/env python3
"""the REFERENCE itself (erthward/geonomics, /root/reference), timed in the
build container on one core - it is single-threaded and never travels to the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_cpu_baseline.py [N] [L] [T]

Harness shape of the reference's own tests/runtime/runtime_test.py:155-164: make_model,
burn in, then the mean wall time of T main steps; metric = sum_t N_t / wall.  The model is
C2-shaped (BASELINE.json configs[1]): one species on a 2-layer landscape, neutral loci,
r = 1/L, mating radius scaled to ~30 candidates per individual, template defaults
elsewhere.  The largest size that finishes in well under 30 minutes here is the default.
Writes profiles/ref_cpu_baseline.json (bench.py quotes it as cpu_baseline.reference)."""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from _ref_import import import_reference      # noqa: E402

gnx = import_reference()
import make_golden as MG                       # noqa: E402  (base_params: template defaults)
import scipy                                   # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    side = int(round(np.sqrt(N)))              # density 1 individual per cell, sum(K) = N
    p = MG.base_params(dim=(side, side), N=N, L=L, traits=False, r_alpha=None, r_beta=None,
                       n_recomb=1000, seed=42, mating_radius=3, K_factor=1.0)
    p['model']['T'] = T
    t0 = time.time()
    mod = gnx.make_model(gnx.make_params_dict(p, 'ref_cpu'))
    t_make = time.time() - t0
    t0 = time.time()
    mod.walk(T=100000, mode='burn', verbose=False)
    t_burn = time.time() - t0
    spp = mod.comm[0]
    n_burn = len(spp.Nt)
    mod.walk(2, 'main', verbose=False)          # warm-up
    n0 = len(spp.Nt)
    t0 = time.perf_counter()
    mod.walk(T, 'main', verbose=False)
    dt = time.perf_counter() - t0
    ind_steps = float(sum(spp.Nt[n0 - 1:n0 - 1 + T]))
    cpu = ''
    try:
        cpu = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo')
               if l.startswith('model name')][0]
    except Exception:
        pass
    out = {'value': ind_steps / dt, 'unit': 'individual-timesteps/s', 'cores': 1,
           'kind': 'reference', 'where': 'build container (the reference never reaches the GPU box)',
           'sample': 'erthward/geonomics 1.4.9, Model.walk(%d, "main") after %d burn-in steps: '
                     '%dx%d 2-layer landscape, N~%d, L=%d neutral loci, r=1/L, n_recomb_sims=1000, '
                     'mating_radius=3 (~30 candidates), use_tskit=False' % (
                         T, n_burn, side, side, int(ind_steps / T), L),
           's_per_step': dt / T, 'make_model_s': round(t_make, 1), 'burn_in_s': round(t_burn, 1),
           'cpu': cpu, 'host_cores': os.cpu_count(), 'python': platform.python_version(),
           'numpy': np.__version__, 'scipy': scipy.__version__}
    path = os.path.join(ROOT, 'profiles', 'ref_cpu_baseline.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
