#!/usr/bin/env python3
"""Steps per second of a reference-scale model (the template's default: 20x20, N=250,
L=100) through Model.walk - the regime most Geonomics scripts run in; launch- and
host-bound rather than bandwidth-bound."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geonomics_amd as gnx                            # noqa: E402
from geonomics_amd.sim import params as P              # noqa: E402

d = P.default_params_dict(1, 1)
d['model']['T'] = 100000
d['comm']['species']['spp_0']['gen_arch']['use_tskit'] = False
mod = gnx.make_model(gnx.make_params_dict(d, 'small'))
mod.walk(10000, 'burn', verbose=False)
mod.walk(200, 'main', verbose=False)
T = 3000
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
mod.walk(T, 'main', verbose=False)
pr.disable()
dt = time.perf_counter() - t0
print('N=%d  %.3f ms/step  %.0f steps/s' % (len(mod.comm[0]), 1e3 * dt / T, T / dt))
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
