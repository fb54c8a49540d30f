#!/usr/bin/env python3
"""K independent device states of one workload stepped on one GPU, no Model API on top:
one after another (gnx_step each), interleaved in thirds from one thread (gnx_step_many), or
by K host threads.  Prints ms per step and handle for each way."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geonomics_amd import _native as nat

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
only = sys.argv[4] if len(sys.argv) > 4 else None
cfg = dict(bench.WORKLOADS[name])
devs = []
for k in range(K):
    dev, _, _ = bench.build_device(cfg, 42 + k, 0)
    for _ in range(3):
        dev.step(True, False)
    bench.setup_genomes(dev, cfg, 42 + k)
    for _ in range(3):
        dev.step(False, True)
    devs.append(dev)


def sync():
    for d in devs:
        d.synchronize()


def timed(label, fn):
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    dt = time.perf_counter() - t0
    print('%-28s %.3f ms per step and handle  (%d handles, %d steps)' % (
        label, 1e3 * dt / (K * steps), K, steps), flush=True)


def one_after_another():
    for d in devs:
        for _ in range(steps):
            d.step(False, True)


def round_robin():
    for _ in range(steps):
        for d in devs:
            d.step(False, True)


def many():
    for _ in range(steps):
        nat.step_many(devs, False, True)


def threads():
    def loop(d):
        for _ in range(steps):
            d.step(False, True)
    ts = [threading.Thread(target=loop, args=(d,)) for d in devs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()


def walk_each():
    for d in devs:
        d.walk(steps, False, True)


def walk_many():
    nat.walk_many(devs, steps, False, True)


modes = [('one after another', one_after_another), ('gnx_walk, one after another', walk_each),
         ('gnx_walk_many', walk_many), ('round robin (gnx_step)', round_robin),
         ('gnx_step_many', many), ('host threads', threads)]
for label, fn in modes:
    if only is None or only == fn.__name__:
        timed(label, fn)
for d in devs:
    d.close()
