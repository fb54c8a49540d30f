#!/usr/bin/env python3
"""Iterations of one model side by side on one GPU (GNX_CONCURRENT_ITS=K; the reference
runs them in turn, sim/model.py:866-953, TODO at :924-925).

    python tools/its_bench.py [workload] [--its 8] [--T 100]

runs the workload through the Geonomics API twice - n_its iterations one after another
(1 lane) and side by side (its lanes) - and reports individual-timesteps/s of
the MAIN phases (sum of N_t over all iterations / wall time from the first main phase's
start to the last one's end), plus whether every iteration ended the same in both runs."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
import geonomics_amd as gnx                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('workload', nargs='?', default='c2')
ap.add_argument('--its', type=int, default=8)
ap.add_argument('--T', type=int, default=100)
ap.add_argument('--rand-comm', action='store_true', help='a new community (and burn-in) per iteration')
ap.add_argument('--lanes', type=int, default=0, help='lanes of the concurrent run (default: its)')
a = ap.parse_args()
cfg = bench.WORKLOADS[a.workload]
os.environ.setdefault('GNX_CAP_FACTOR', '2.0')


def run(k):
    d = bench.model_api_params(cfg, a.workload, a.T)
    d['model']['its']['n_its'] = a.its
    d['model']['its']['rand_comm'] = bool(a.rand_comm)
    t0 = time.time()
    mod = gnx.make_model(d)
    t1 = time.time()
    os.environ['GNX_CONCURRENT_ITS'] = str(k)
    mod.run(verbose=False)
    for spp in mod.comm.values():
        spp._dev.synchronize()
    t2 = time.time()
    tm = mod.iteration_times
    T = a.T
    ind_steps = sum(sum(v['Nt'][-T - 1:-1]) for it in mod.iteration_log.values()
                    for v in it.values())
    # sequential: the main phases' own time (what lies between them - the restore of the
    # burned-in community, 250 MB of genomes at C2 - is not part of the metric); side by
    # side: from the first main phase's start to the last one's end, restores included
    span = (sum(e - s for s, e in tm.values()) if k == 1 else
            max(e for _, e in tm.values()) - min(s for s, _ in tm.values()))
    n_burn = len(next(iter(mod.iteration_log[0].values()))['Nt']) - T
    print('concurrent=%d: make_model %.1f s, run %.1f s (burn-in %d steps), main phases: '
          '%d x %d steps in %.3f s = %.3f ms per step and iteration, %.3e '
          'individual-timesteps/s' % (k, t1 - t0, t2 - t1, n_burn, a.its, T, span,
                                      1e3 * span / (a.its * T), ind_steps / span), flush=True)
    print('   per iteration, ms per main step: ' + ' '.join(
        '%.3f' % (1e3 * (tm[i][1] - tm[i][0]) / T) for i in sorted(tm)), flush=True)
    return mod.iteration_log, ind_steps / span


log1, r1 = run(1)
logk, rk = run(a.lanes or a.its)
print('iterations identical in both runs: %s' % (log1 == logk))
print('speed-up of the main phases: x%.2f' % (rk / r1))
