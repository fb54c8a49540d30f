#!/usr/bin/env python3
"""bench.py - individual-timesteps/s of the Geonomics hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one pass of the per-generation loop (age, movement, mate search,
mating with bitmask crossover, density, selection, mortality) over the whole
resident population.  Metric = sum_t N_t / wall time over K timed steps
(SURVEY 8d), inputs resident in HBM before the timed region.

Workload `c4_metric` (default) is the configuration BASELINE.json's metric is
quoted on: 2048x2048 two-layer landscape, 10^6 individuals, 10^5-locus genomes,
4 traits x 10 loci selected on the second layer, ConductanceSurface movement,
mating_radius 10, b 0.2, one birth per pair, recombination rate 1/L.

For --gpus N > 1 the driver launches one rank per GPU with torch.distributed
(RCCL), one landscape tile per rank, stepped by geonomics_amd.parallel.TiledStepper:
migrants (with genomes), halo ghosts, pair lists, gametes of cross-border mates and the
density bins are exchanged every step (csrc/gnx_tile.hip).  `--gpus 8` is BASELINE.json's
configs[4] (C5) exactly: a 4096x4096 landscape tiled 2 x 4 (tiles 1024 wide, 2048 tall),
10^7 individuals, L = 10^5 (31 GB of genomes per GPU); `--gpus 4` / `--gpus 2` are the
2 x 2 / 1 x 2 sub-grids of the same tiles (2048x4096 with 5x10^6, 2048x2048 with
2.5x10^6 individuals), so per-GPU work is fixed ("weak").  `--scaling weak2048` instead
grows the landscape by one 2048x2048 tile of the metric workload per GPU.
`python bench.py --gpus N` without a launcher starts the N ranks itself.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (W, H, N, L, n_traits, loci_per_trait, move_surf, n_paths)
    'c4_metric': dict(W=2048, H=2048, N=1_000_000, L=100_000, n_traits=4,
                      loci_per_trait=10, move_surf=True, n_paths=10_000),
    # the same with the parameters-file template's recombination default
    # (r_distr_alpha = 0.5, sim/params.py:445: free recombination): dense masks, the
    # crossover reads both parental homologues and the 125-MB path table
    'c4_dense': dict(W=2048, H=2048, N=1_000_000, L=100_000, n_traits=4,
                     loci_per_trait=10, move_surf=True, n_paths=10_000, paths='dense'),
    # one tile of BASELINE.json's configs[4] (C5: 4096x4096 tiled 2 x 4, 10^7 individuals)
    'c5_tile': dict(W=1024, H=2048, N=1_250_000, L=100_000, n_traits=4,
                    loci_per_trait=10, move_surf=True, n_paths=10_000),
    'c3': dict(W=1024, H=1024, N=100_000, L=100_000, n_traits=4,
               loci_per_trait=10, move_surf=False, n_paths=10_000),
    'c2': dict(W=1024, H=1024, N=100_000, L=10_000, n_traits=0,
               loci_per_trait=0, move_surf=False, n_paths=10_000),
    'small': dict(W=256, H=256, N=20_000, L=10_000, n_traits=4,
                  loci_per_trait=10, move_surf=True, n_paths=1_000),
}


def smooth_field(W, H, seed):
    """Seeded smooth random field in [0,1] (conductance / selection layer)."""
    rng = np.random.RandomState(seed)
    k = 16
    coarse = rng.rand(H // k + 3, W // k + 3)
    from scipy.ndimage import zoom
    f = zoom(coarse, k, order=3)[:H, :W]
    f = (f - f.min()) / (f.max() - f.min())
    return f.astype(np.float32)


def _set_bit_range(row, lo, hi):
    """set bits [lo, hi) of a little-endian u64 word array"""
    if hi <= lo:
        return
    w0, w1 = lo >> 6, (hi - 1) >> 6
    full = np.uint64(0xFFFFFFFFFFFFFFFF)
    m0 = full << np.uint64(lo & 63)
    m1 = full >> np.uint64(63 - ((hi - 1) & 63))
    if w0 == w1:
        row[w0] |= m0 & m1
    else:
        row[w0] |= m0
        row[w0 + 1:w1] = full
        row[w1] |= m1


def sparse_paths(n, L, seed, W64):
    """n recombination paths for per-locus rate 1/L (r_0 = 0): the path switches
    homologue at Binomial(L-1, 1/L) distinct loci (structs/genome.py:173-221);
    bit l = homologue the path is on at locus l."""
    rng = np.random.RandomState(seed)
    out = np.zeros((n, W64), dtype=np.uint64)
    ks = rng.binomial(L - 1, 1.0 / L, n)
    for i in range(n):
        if ks[i]:
            bp = np.unique(rng.randint(1, L, ks[i])).tolist() + [L]
            for j in range(0, len(bp) - 1, 2):
                _set_bit_range(out[i], bp[j], bp[j + 1])
    return out


def dense_paths(n, L, seed, W64):
    """n recombination paths for per-locus rate 0.5 (free recombination): every locus
    switches homologue with probability 1/2, i.e. the path bits are fair coins"""
    rng = np.random.RandomState(seed)
    out = rng.randint(0, 2 ** 63, (n, W64), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    out |= rng.randint(0, 2, (n, W64)).astype(np.uint64)
    out[:, 0] &= ~np.uint64(1)                       # r_0 = 0 (structs/genome.py:183)
    return out


def build_device(cfg, seed, device, grid=(1, 1), rank=0, host_init=None, cap_factor=2.0):
    from geonomics_amd import _native as nat
    R, C = grid
    W, H, L = cfg['W'] * C, cfg['H'] * R, cfg['L']
    N = cfg['N'] * R * C
    lyr0 = smooth_field(W, H, 1) * 0.5 + 0.5          # K / conductance layer
    lyr1 = np.tile(np.linspace(0, 1, W, dtype=np.float32), (H, 1))
    rasts = np.stack([lyr0, lyr1])
    K_factor = N / float(lyr0.sum())                  # sum(K) = N
    n_own = N if host_init and R * C == 1 else cfg['N']
    cap_rows = int(n_own * cap_factor) + 1024
    cap = cap_rows
    dev = nat.Device(W, H, 2, L=L, n_traits=cfg['n_traits'], cap_inds=cap,
                     cap_rows=cap_rows, seed=seed, device=device)
    dev.upload_rasters(rasts)
    sp = nat.default_species_params(
        mating_radius=10.0, K_layer=0, K_factor=K_factor,
        move_surf=nat.SURF_MIXTURE if cfg['move_surf'] else nat.SURF_NONE,
        move_surf_layer=0, move_surf_kappa=12.0)
    dev.set_species_params(sp)
    rng = np.random.RandomState(seed)
    if cfg['n_traits']:
        loci = np.sort(rng.choice(L, cfg['n_traits'] * cfg['loci_per_trait'],
                                  replace=False)).reshape(cfg['n_traits'], -1)
        for t in range(cfg['n_traits']):
            n = cfg['loci_per_trait']
            alpha = 0.1 * np.array([1 - (i % 2) * 2 for i in range(n)], float)
            dev.set_trait(t, np.sort(loci[t]), alpha, 1, 0.05, 1.0, False)
    if host_init is None:
        host_init = R * C > 1
    if not host_init:
        dev.init_population(N)
    else:
        # the same uniform initial population on every rank (same generator), of which
        # each rank keeps the individuals of its own tile; ids = index in the common list
        prng = np.random.RandomState(seed + 12345)
        x = np.minimum(prng.rand(N) * W, W - 0.001).astype(np.float32)
        y = np.minimum(prng.rand(N) * H, H - 0.001).astype(np.float32)
        sex = (prng.rand(N) < 0.5).astype(np.uint8)
        r, c = divmod(rank, C)
        mine = np.nonzero((x // (W // C) == c) & (y // (H // R) == r))[0]
        dev.upload_population(x[mine], y[mine], np.zeros(mine.size, np.int32), sex[mine],
                              mine.astype(np.int64))
        dev.set_max_id(N - 1)
    return dev, rasts, K_factor


def setup_genomes(dev, cfg, seed):
    L = cfg['L']
    mk = dense_paths if cfg.get('paths') == 'dense' else sparse_paths
    dev.set_recomb_paths(mk(cfg['n_paths'], L, seed + 1, dev.W64))
    # start_p_fixed = 0.5: n_l = round(2N * 0.5) = N ones per site
    dev.assign_genomes(np.full(L, dev.N, dtype=np.int32))


def _port_state(W, H, N, L, n_traits, n_paths=256):
    """the numpy oracle's state of a workload: same densities, genome length and traits as the
    device's (oracle/gnx_step.py), random genomes"""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import gnx_step as S
    import gnx_oracle as O
    rng = np.random.RandomState(3)
    lyr0 = smooth_field(W, H, 1) * 0.5 + 0.5
    lyr1 = np.tile(np.linspace(0, 1, W, dtype=np.float32), (H, 1))
    traits = []
    if n_traits:
        loci = np.sort(rng.choice(L, n_traits * 10, replace=False)).reshape(n_traits, 10)
        traits = [dict(loci=loci[t],
                       alpha=0.1 * np.array([1 - (i % 2) * 2 for i in range(10)], float),
                       layer=1, phi=0.05, gamma=1.0, univ_adv=False) for t in range(n_traits)]
    W64 = O.words_per_hom(L)
    paths = sparse_paths(n_paths, L, 5, W64)
    st = S.State(np.stack([lyr0, lyr1]),
                 S.Params(mating_radius=10.0, K_factor=N / float(lyr0.sum())),
                 7, L=L, traits=traits, paths_packed=paths)
    st.init_population(N)
    S.step(st, burn=True)
    g = np.random.default_rng(3).integers(0, 2 ** 64, (st.N, 2, W64), dtype=np.uint64,
                                          endpoint=False)
    st.set_genomes(g)
    return S, st


def cpu_baseline(budget_s=20.0, full_scale=True):
    """Oracle ("port": numpy, 1 thread).  `value`: a bounded sample of the METRIC workload
    (same densities, genome length and traits, 1/64 of the landscape: the full 10^6 x 10^5 bits
    are 25 GB of numpy arrays and minutes per step).  BASELINE configs[1] and [2] (C2, C3:
    10^5 individuals) run at FULL scale, a few steps each (SURVEY 8d ii), as flat fields."""
    W = H = 256
    L = 100_000
    N = int(1_000_000 * (W * H) / (2048 * 2048))
    S, st = _port_state(W, H, N, L, 4)
    S.step(st, burn=False)                      # warm-up
    t0 = time.time()
    done = 0
    steps = 0
    while time.time() - t0 < budget_s and steps < 50:
        done += st.N
        S.step(st, burn=False)
        steps += 1
    dt = time.time() - t0
    out = dict(value=done / dt, unit='individual-timesteps/s', cores=1, kind='port',
               sample='c4_metric at 1/64 of its area: %d steps of a %dx%d landscape, N~%d, L=%d, '
                      '4 traits, numpy oracle (oracle/gnx_step.py), 1 thread of %d host cores'
                      % (steps, W, H, N, L, os.cpu_count()))
    if full_scale:
        for name, n_steps in (('c2', 3), ('c3', 2)):
            cfg = WORKLOADS[name]
            try:
                t1 = time.time()
                S, st = _port_state(cfg['W'], cfg['H'], cfg['N'], cfg['L'], cfg['n_traits'])
                setup = time.time() - t1
                t1 = time.time()
                done = 0
                for _ in range(n_steps):
                    done += st.N
                    S.step(st, burn=False)
                dt = time.time() - t1
                out['%s_full_value' % name] = done / dt
                out['%s_full_sample' % name] = (
                    '%s at FULL scale: %d steps of %dx%d, N~%d, L=%d, %d traits, %.2f s/step '
                    '(+ %.0f s set-up), numpy oracle, 1 thread' % (
                        name, n_steps, cfg['W'], cfg['H'], st.N, cfg['L'], cfg['n_traits'],
                        dt / n_steps, setup))
                del st
            except Exception as e:          # the contract line must still be printed
                out['%s_full_value' % name] = None
                out['%s_full_sample' % name] = 'failed: %s: %s' % (type(e).__name__, e)
    return out


def model_api_params(cfg, name, T):
    """the workload as a Geonomics parameters dict (what a user's script would hold)"""
    import geonomics_amd as gnx
    from geonomics_amd.sim import params as P
    W, H, L, N = cfg['W'], cfg['H'], cfg['L'], cfg['N']
    sp = {'genomes': True, 'n_traits': cfg['n_traits'],
          'movement_surface': bool(cfg['move_surf'])}
    d = P.default_params_dict(layers=[{'type': 'defined'}, {'type': 'defined'}], species=[sp])
    lyr0 = smooth_field(W, H, 1) * 0.5 + 0.5
    lyr1 = np.tile(np.linspace(0, 1, W), (H, 1))
    d['landscape']['main']['dim'] = (W, H)
    d['landscape']['layers']['lyr_0']['init']['defined']['rast'] = lyr0.astype(np.float64)
    d['landscape']['layers']['lyr_1']['init']['defined']['rast'] = lyr1
    s = d['comm']['species']['spp_0']
    s['init'].update({'N': N, 'K_layer': 'lyr_0', 'K_factor': N / float(lyr0.sum())})
    s['mating'].update({'mating_radius': 10, 'b': 0.2, 'n_births_fixed': True,
                        'n_births_distr_lambda': 1})
    if cfg['move_surf']:
        s['movement']['move_surf'].update({'layer': 'lyr_0', 'mixture': True,
                                           'vm_distr_kappa': 12})
    s['gen_arch'].update({'L': L, 'r_distr_alpha': None, 'r_distr_beta': None,
                          'n_recomb_sims': cfg['n_paths'], 'use_tskit': False, 'mu_neut': 0,
                          'mu_delet': 0})
    for t in range(cfg['n_traits']):
        s['gen_arch']['traits']['trait_%i' % t].update(
            {'layer': 'lyr_1', 'n_loci': cfg['loci_per_trait'], 'alpha_distr_sigma': 0})
    d['model'].update({'T': T, 'burn_T': 30, 'seed': {'num': 42}})
    return gnx.make_params_dict(d, name)


def model_api_measure(cfg, name, steps, warmup=5):
    """SURVEY 8(d)'s definition of the metric, literally: sum of N_t over the wall time of
    Model.walk(T, 'main') after make_model and the burn-in, i.e. at the model's
    equilibrium and through the Geonomics API (structs/species.py on the C-ABI)."""
    import geonomics_amd as gnx
    os.environ.setdefault('GNX_CAP_FACTOR', '2.0')
    t0 = time.time()
    mod = gnx.make_model(model_api_params(cfg, name, steps))
    mod.walk(10000, 'burn', verbose=False)
    spp = mod.comm[0]
    n_burn = len(spp.Nt)
    mod.walk(warmup, 'main', verbose=False)
    spp._dev.synchronize()
    setup = time.time() - t0
    n0 = len(spp.Nt)
    t1 = time.perf_counter()
    mod.walk(steps, 'main', verbose=False)
    spp._dev.synchronize()
    dt = time.perf_counter() - t1
    ind_steps = float(sum(spp.Nt[n0 - 1:n0 - 1 + steps]))
    out = {'value': ind_steps / dt, 'unit': 'individual-timesteps/s',
           'ms_per_step': 1e3 * dt / steps, 'steps': steps,
           'state': 'equilibrium: make_model, %d burn-in steps, %d main steps, then timed '
                    'Model.walk' % (n_burn, warmup),
           'mean_N': ind_steps / steps, 'births_per_step': float(np.mean(spp.n_births[-steps:])),
           'setup_s': round(setup, 1)}
    for s_ in mod.comm.values():
        s_._dev.close()
    return out, mod


def measured_copy_bandwidth(nbytes=8 << 30, reps=5):
    """read + write bytes / time of the library's hand-written copy kernel (16 bytes per lane
    and access, non-temporal, csrc/gnx_stats.hip: gnx_measure_copy) over two 8-GiB buffers;
    the buffers are large on purpose: on this part the rate of a streaming kernel grows with
    the span of HBM it touches (profiles/r02_xo_lab_footprint.txt)"""
    try:
        from geonomics_amd import _native as nat
        return nat.measure_copy(nbytes, reps)
    except Exception:
        return None


def kernel_profile(dev, do_step, n_steps=10):
    """per-kernel-family HIP-event time and algorithmic bytes per step (the
    library's own accounting) over n_steps extra steps OUTSIDE the timed region"""
    dev.profiling(1)
    for _ in range(n_steps):
        do_step(False)
    dev.synchronize()
    kt = dev.kernel_times()
    dev.profiling(False)
    fam = {k: {'ms_per_step': v['ms'] / n_steps, 'bytes_per_step': v['bytes'] / n_steps,
               'GBps': (v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['ms'] > 0 else 0.0}
           for k, v in kt.items() if v['launches'] > 0}
    return fam


def measure_other_workload(name, steps=30, warmup=5, steady_warm=1500):
    """a short run of another BASELINE configuration on the same box (N = 1): ms/step,
    individual-timesteps/s, the dominant kernel family and its rate"""
    cfg = WORKLOADS[name]
    t0 = time.time()
    dev, _, _ = build_device(cfg, seed=42, device=0)
    for _ in range(3):
        dev.step(True, False)
    setup_genomes(dev, cfg, seed=42)
    for _ in range(warmup):
        dev.step(False, True)
    dev.synchronize()
    setup = time.time() - t0
    # the same number of steps the host-driven way first (gnx_step: two count read-backs and
    # ~45 runtime calls per step) ...
    t2 = time.perf_counter()
    for _ in range(steps):
        dev.step(False, True)
    dev.synchronize()
    dt_host = time.perf_counter() - t2
    dev.walk(4, False, True)         # (untimed: the walk's one-off set-up - pinned ring, graphs)
    dev.synchronize()
    dev.reset_totals()
    t1 = time.perf_counter()
    # ... then gnx_walk: `steps` time steps in one call; the library takes the device-driven path
    # (counts on the device, one graph launch per step) for populations of this size
    dev.walk(steps, False, True)
    dev.synchronize()
    dt = time.perf_counter() - t1
    tot = dev.totals()               # accumulated inside the library: nothing read per step
    n, births = tot['ind_steps'], tot['births']
    young = {'ms_per_step': 1e3 * dt / steps, 'value': n / dt, 'mean_N': n / steps,
             'births_per_step': births / steps,
             'steps_since_genome_assignment': [warmup + steps + 4 + 1, warmup + 2 * steps + 4]}
    # ... and the same at the model's STEADY state: offspring land next to their parents, the
    # population clumps over the first ~1000 steps and the mate search's candidate lists grow
    # (C2: 14 individuals share the average individual's hash cell at step 1, ~300 at the steady
    # state) - the figure a long run sees (reference tests/runtime/runtime_test.py:155-164 times
    # T = 250 steps after the burn-in).
    done = warmup + 2 * steps + 4
    if steady_warm > 0:
        dev.walk(steady_warm, False, True)
        dev.synchronize()
        done += steady_warm
        dev.reset_totals()
        t1 = time.perf_counter()
        dev.walk(steps, False, True)
        dev.synchronize()
        dt = time.perf_counter() - t1
        tot = dev.totals()
        n, births = tot['ind_steps'], tot['births']
    fam = kernel_profile(dev, lambda burn: dev.step(burn, not burn), 10)
    dev.close()
    dom = max(fam, key=lambda k: fam[k]['ms_per_step'])
    dense = cfg.get('paths') == 'dense'
    out = {'workload': '%s: %dx%d, N0=%d, L=%d, %d traits, move_surf=%s, %s' % (
               name, cfg['W'], cfg['H'], cfg['N'], cfg['L'], cfg['n_traits'], cfg['move_surf'],
               'r=0.5 (dense masks)' if dense else 'r=1/L'),
           'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * dt / steps,
           'value': n / dt, 'unit': 'individual-timesteps/s', 'mean_N': n / steps,
           'births_per_step': births / steps, 'setup_s': round(setup, 1),
           # ms_per_step / value above: the STEADY state (timed steps start this many steps
           # after the genomes were assigned); `young`: the first steps after a uniform start
           'state': ('steady' if steady_warm > 0 else 'young'),
           'steps_since_genome_assignment': [done + 1, done + steps],
           'young': young,
           'dominant_kernel': dom, 'dominant_ms_per_step': fam[dom]['ms_per_step'],
           'dominant_GBps': fam[dom]['GBps'], 'dominant_frac': fam[dom]['GBps'] / 8000.0,
           'bytes_per_step': sum(v['bytes_per_step'] for v in fam.values()),
           'kernel_ms_per_step': {k: round(v['ms_per_step'], 4) for k, v in fam.items()},
           # which way gnx_walk took the steps, and the same steps through gnx_step
           'path': ('gnx_walk, device-driven: counts on the device, one HIP graph launch per '
                    'step, no read-back' if tot['dd_steps'] > 0 else
                    'gnx_walk, host-driven (gnx_step per step)'),
           'ms_per_step_gnx_step': 1e3 * dt_host / steps}
    out['step_frac'] = out['bytes_per_step'] / (out['ms_per_step'] * 1e-3) / 1e9 / 8000.0
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child
    processes (before this process touches the GPU), relay rank 0's line and the first
    non-zero exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else None))
    rc = 0
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    try:
        pending = list(procs)
        while pending:
            for p_ in list(pending):
                code = p_.poll()
                if code is None:
                    continue
                pending.remove(p_)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:          # a dead rank leaves its peers in a collective
                        q.kill()
            time.sleep(0.2)
        reader.join(timeout=10)
        if rc == 0 and out0:
            sys.stdout.write(out0[0].decode())
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
                p_.wait()
    raise SystemExit(rc)


def ref_cpu_number():
    """the reference itself, timed in the build container by tools/ref_cpu_baseline.py
    (it never travels to the GPU box): quoted next to the same-run port number"""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'ref_cpu_baseline.json')))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None)
    ap.add_argument('--steps', type=int, default=100)       # T >= 100
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS))
    ap.add_argument('--scaling', default='c5', choices=['c5', 'weak2048'],
                    help='N > 1: tiles of BASELINE C5 (default) or 2048x2048 metric tiles')
    ap.add_argument('--steady-warmup', type=int, default=1500,
                    help='N = 1: steps walked before the second, steady-state measurement '
                         '(c4_metric_steady in the line; 0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-model-api', action='store_true',
                    help='skip the second measurement through Model.walk (N = 1 only)')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the short runs of BASELINE configs[1], [2] and the dense-mask '
                         'variant (N = 1, default workload only)')
    args = ap.parse_args()

    env_world = os.environ.get('WORLD_SIZE')
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)                                   # does not return
    world = int(env_world or '1')
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    import torch
    torch.set_num_threads(1)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device is visible '
                         '(the product has no CPU path)')
    # rehearsal on a 1-GPU box: GNX_BENCH_BACKEND=gloo GNX_BENCH_SINGLE_DEVICE=1
    backend = os.environ.get('GNX_BENCH_BACKEND', 'nccl')
    single_dev = bool(os.environ.get('GNX_BENCH_SINGLE_DEVICE'))
    if single_dev:
        local_rank = 0
    # (a launcher that shows every rank ONE device of its own - HIP_VISIBLE_DEVICES per rank - has
    # LOCAL_RANK beyond the visible ordinals: the rank's device is then ordinal 0)
    dev_ordinal = local_rank if local_rank < torch.cuda.device_count() else 0
    torch.cuda.set_device(dev_ordinal)
    dist = None
    devs_ident = None            # every rank's GPU by PCI address / UUID (world > 1)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_ordinal))
        else:
            # (gloo announces its connections on stdout; this program's stdout is one JSON line)
            from geonomics_amd.parallel import _StdoutToStderr
            with _StdoutToStderr():
                dist.init_process_group(backend)
        # every rank on its own GPU (a launcher that put two ranks on one device would
        # give a number that is not an N-GPU number)
        # (physical identity - PCI address or UUID - not the ordinal: ordinals repeat when every rank
        # is shown one device)
        pr = torch.cuda.get_device_properties(dev_ordinal)
        ident = None
        if all(hasattr(pr, a) for a in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')):
            ident = 'pci %04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        elif getattr(pr, 'uuid', None) is not None:
            ident = 'uuid %s' % pr.uuid
        else:
            ident = 'ordinal %d' % dev_ordinal
        devs = [None] * world
        dist.all_gather_object(devs, ident)
        assert dist.get_world_size() == world
        if not single_dev and len(set(devs)) != world:
            raise SystemExit('bench.py: the ranks do not sit on distinct GPUs: %s' % devs)
        devs_ident = devs

    if args.workload is None:
        args.workload = 'c4_metric' if world == 1 or args.scaling == 'weak2048' else 'c5_tile'
    cfg = WORKLOADS[args.workload]
    t_setup = time.time()
    stepper = None
    grid = (1, 1)
    force_stepper = bool(os.environ.get('GNX_BENCH_FORCE_STEPPER'))   # overhead study at N=1
    if world > 1 or force_stepper:
        from geonomics_amd.parallel import Comm, DeviceShard, TiledStepper, tile_grid
        grid = tile_grid(world)
    dev, rasts, K_factor = build_device(cfg, seed=42, device=dev_ordinal, grid=grid, rank=rank)
    if world > 1 or force_stepper:
        shard = DeviceShard(dev)
        stepper = TiledStepper(shard, Comm(dist), cfg['W'] * grid[1], cfg['H'] * grid[0], 10.0,
                               move=True, max_id=cfg['N'] * world - 1, grid=grid,
                               fixed_births=1)
        stepper.profile = False

    v2 = stepper is not None and stepper.v2

    def do_step(burn):
        """-> (population at the START of the step - global on tiles -, births)"""
        if stepper is None:
            n0 = dev.N
            dev.step(burn, not burn)
            return n0, dev.counts()[1]
        if v2:       # the counts that ride on the step's own all-reduce: no extra collective
            n, b, _ = stepper.step(burn, not burn, exact=False)
            return n, b
        n, b, _ = stepper.step(burn, not burn)
        return n, b

    # a short burn-in brings the uniform initial population to its density-
    # regulated spatial distribution before genomes are assigned
    for _ in range(3):
        do_step(True)
    setup_genomes(dev, cfg, seed=42)
    if stepper is not None:
        stepper.shard.has_genomes = True
    dev.synchronize()
    t_setup = time.time() - t_setup

    n_glob = None
    for _ in range(args.warmup):
        n_glob, _ = do_step(False)
    if stepper is not None and n_glob is None and not v2:
        n_glob = int(stepper.comm.allreduce_sum(np.array([dev.N], np.int64))[0])

    def barrier():
        torch.cuda.synchronize()
        dev.synchronize()
        if dist is not None:
            dist.barrier()

    if stepper is None:
        # (untimed: gnx_walk's one-off set-up - its pinned ring and graphs - where it takes the
        # device-driven path; two more warm-up steps otherwise)
        dev.walk(2, False, True)
    # inside the timed region only the dominant kernel is bracketed with HIP events
    dev.profiling(2 if os.environ.get('GNX_BENCH_PROFILE_ALL') is None else 1)
    barrier()
    t0 = time.perf_counter()
    ind_steps = 0
    births = 0
    xo_births = 0
    gc_before = dev.genome_info()['gc_runs']
    if stepper is None:
        # one C call per step and nothing else: N at the start of every step, births and the
        # births that got a genome are summed inside the library (gnx_totals) and read once,
        # behind the synchronisation that closes the timed region
        dev.reset_totals()
        dev.walk(args.steps, False, True)       # (10^6 individuals: gnx_step per step inside)
    else:
        for _ in range(args.steps):
            if v2:
                n0, b = do_step(False)      # global population at the start of the step
                ind_steps += n0
            else:
                ind_steps += n_glob         # global population at the start of the step
                n_glob, b = do_step(False)
            births += b
            xo_births += dev.last_crossover_births
    dev.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if stepper is None:
        tot = dev.totals()
        ind_steps, births, xo_births = tot['ind_steps'], tot['births'], tot['xo_births']
    barrier()
    gc_in_timed = dev.genome_info()['gc_runs'] - gc_before
    kt = dev.kernel_times()
    dev.profiling(False)

    mx = torch.tensor([elapsed], device='cuda')
    if dist is not None:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    total_ind_steps = float(ind_steps)      # tiled: already the global count
    max_elapsed = float(mx.item())

    # the other ways to share the GPU between the deferred crossover and the next step
    # (gnx_set_crossover_overlap), reported next to the contract's numbers and measured the
    # same way right after them: 2 = nothing runs beside the crossover (the kernel's own
    # rate), 1 = a narrow crossover beside the WHOLE next step
    alt = None
    alone = None
    if stepper is None and not os.environ.get('GNX_BENCH_NO_ALT'):
        def other_mode(mode, label):
            dev.set_crossover_overlap(mode)
            for _ in range(5):
                do_step(False)
            dev.profiling(2)
            dev.synchronize()
            t1 = time.perf_counter()
            n_alt = 0
            k_alt = min(args.steps, 50)
            for _ in range(k_alt):
                n0, _b = do_step(False)
                n_alt += n0
            dev.synchronize()
            dt = time.perf_counter() - t1
            kx = dev.kernel_times()['crossover']
            dev.profiling(False)
            dev.set_crossover_overlap(0)
            a_gbps = (kx['bytes'] / (kx['ms'] * 1e-3)) / 1e9 if kx['ms'] > 0 else 0.0
            return {'mode': label, 'value': n_alt / dt, 'ms_per_step': 1e3 * dt / k_alt,
                    'steps': k_alt, 'crossover_avg_launch_ms': kx['ms'] / max(kx['launches'], 1),
                    'crossover_achieved_GBps': a_gbps, 'crossover_frac': a_gbps / 8000.0}
        alone = other_mode(2, 'nothing runs beside the crossover (the compactions and the next '
                              'movement wait for it); one job per wave and iteration '
                              '(k_xo_sparse<1>) - the timed region runs k_xo_sparse_pair, '
                              'two jobs per iteration, which is the slower kernel alone')
        alt = other_mode(1, 'crossover (8 workgroups per CU) beside the whole next step: nothing waits '
                             'for it but the next crossover')
    fam = None
    steady = None
    if stepper is None:
        fam = kernel_profile(dev, do_step, 10)
        if args.steady_warmup > 0:
            # the metric workload at ITS steady state: the conductance surface piles the
            # population onto ridges and offspring land beside their parents, the population
            # grows from N0 to ~1.33 N0 and the mate search's candidate lists lengthen over the
            # first ~1000 steps.  Same call as the timed region (gnx_walk), after a long walk.
            dev.walk(args.steady_warmup, False, True)
            dev.synchronize()
            dev.reset_totals()
            gc0 = dev.genome_info()['gc_runs']
            ts = time.perf_counter()
            dev.walk(args.steps, False, True)
            dev.synchronize()
            dts = time.perf_counter() - ts
            tot_s = dev.totals()
            steady = {'ms_per_step': 1e3 * dts / args.steps, 'value': tot_s['ind_steps'] / dts,
                      'unit': 'individual-timesteps/s', 'steps': args.steps,
                      'mean_N': tot_s['ind_steps'] / args.steps,
                      'births_per_step': tot_s['births'] / args.steps,
                      'warmup_steps_before': args.steady_warmup,
                      'gc_runs_in_timed_region': dev.genome_info()['gc_runs'] - gc0}
    phases = None
    if stepper is not None and not os.environ.get('GNX_BENCH_NO_PHASES'):
        # per-phase host wall time of the tile protocol, from a few extra steps with a
        # device synchronisation at every phase mark (outside the timed region)
        stepper.profile = True
        stepper.phase_s = {}
        n_prof = 5
        for _ in range(n_prof):
            do_step(False)
        phases = {k: 1e3 * v / n_prof for k, v in stepper.phase_s.items()}
        stepper.profile = False

    # what the library's communicator says about itself on EVERY rank (gnx_comm_info: ncclCommCount,
    # ncclCommUserRank, ncclCommCuDevice, the handle's HIP device, host ms per phase of the
    # tile step): the line certifies the ranks it ran on
    comm_infos = None
    cert_errors = []
    if stepper is not None and stepper.v3:
        mine_info = dev.comm_info()
        mine_info['local_rank'] = local_rank
        mine_info['pid'] = os.getpid()
        if dist is not None:
            comm_infos = [None] * world
            dist.all_gather_object(comm_infos, mine_info)
        else:
            comm_infos = [mine_info]
        # (what does not add up is REPORTED in the line - config.rccl.certified false, the reasons
        # beside it - rather than raised: the timed region is over, the number stands or falls with
        # the evidence printed next to it)
        if world > 1 and backend == 'nccl' and not single_dev:
            for r_, ci in enumerate(comm_infos):
                if not (ci['transport'] == 'rccl' and ci['nccl_comm_count'] == world and
                        ci['nccl_comm_user_rank'] == r_):
                    cert_errors.append('rank %d is not RCCL rank %d of %d: transport %s, count %s, user rank %s'
                                       % (r_, r_, world, ci['transport'], ci['nccl_comm_count'],
                                          ci['nccl_comm_user_rank']))

    if rank == 0:
        xo = kt['crossover']
        ach = (xo['bytes'] / (xo['ms'] * 1e-3)) / 1e9 if xo['ms'] > 0 else 0.0
        peak = 8000.0
        # HBM bytes per launch from the PMC passes of the latest committed profile
        # (tools/profile_round.sh; separate --pmc runs, gfx950 corrections applied there)
        # The PMC runs are shorter (fewer births per launch than here), so the measured
        # bytes are scaled by the launches' algorithmic bytes: traffic = this run's
        # algorithmic bytes x (PMC bytes / algorithmic bytes of the PMC run).
        traffic = traffic_src = None
        import glob
        for pmc in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_crossover.json')))[::-1]:
            try:
                j = json.load(open(pmc))
                if j.get('workload') == args.workload and j.get('algorithmic_bytes_per_launch'):
                    ratio = j['hbm_bytes_per_launch'] / j['algorithmic_bytes_per_launch']
                    traffic = ratio * xo['bytes'] / max(xo['launches'], 1)
                    traffic_src = {'file': 'profiles/' + os.path.basename(pmc),
                                   'hbm_bytes_per_launch': j['hbm_bytes_per_launch'],
                                   'algorithmic_bytes_per_launch':
                                       j['algorithmic_bytes_per_launch'],
                                   'ratio': ratio}
                    break
            except Exception:
                continue
        copy_gbps = measured_copy_bandwidth()
        dense = cfg.get('paths') == 'dense'
        per_gamete = (4.0 if dense else 2.0) * dev.W64 * 8.0
        # the crossover's algorithmic bytes per copied block: the block read and written, plus -
        # sparse paths - the 128-byte line around the switch point from the other homologue
        # (gnx_kernel_time); `copied` = the share that is block copies, for the gamete figures
        blk = dev.W64 * 8.0 / max(dev.blocks_per_hom, 1)
        copied = 1.0 if dense else 2.0 * blk / (2.0 * blk + 128.0)
        out = {
            'metric': 'individual-timesteps/sec', 'value': total_ind_steps / max_elapsed,
            'unit': 'individual-timesteps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * max_elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u64', 'data': 'synthetic',
            'config': {
                'workload': '%s: %dx%d 2-layer landscape per GPU, N0=%d per GPU, L=%d, '
                            '%d traits x %d loci, move_surf=%s, mating_radius=10, b=0.2, '
                            'lambda=1 fixed, %s, n_recomb_sims=%d' % (
                                args.workload, cfg['W'], cfg['H'], cfg['N'], cfg['L'],
                                cfg['n_traits'], cfg['loci_per_trait'], cfg['move_surf'],
                                'r=0.5 (free recombination, dense masks)' if dense else 'r=1/L',
                                cfg['n_paths']),
                'parallelism': ('1 tile' if world == 1 else
                                'tiles %dx%d: migrants+halo+gametes p2p, pair lists / density '
                                'bins collectives (RCCL), payloads %s' % (
                                    grid + ('device-resident' if stepper.dev_transport
                                            else 'staged through the host',))),
                # which code drives a tiled step: the library's own communicator (one C call per
                # step, RCCL issued by libgnxhip.so) or TiledStepper._step_v2 over torch.distributed
                'tile_protocol': (None if stepper is None else
                                  'gnx_tile_step' if stepper.v3 else 'torch.distributed'),
                'landscape': '%dx%d' % (cfg['W'] * grid[1], cfg['H'] * grid[0]),
                'N0_global': cfg['N'] * world,
                'mean_N_global': ind_steps / args.steps, 'births_per_step_global': births / args.steps,
                'births_with_genome_per_step_rank0': xo_births / args.steps,
                'setup_s': round(t_setup, 2),
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'k_xo_dense<4>' if dense else 'k_xo_sparse',
                'achieved': ach, 'peak': peak,
                'unit': 'GB/s', 'frac': ach / peak, 'traffic': traffic,
                # `traffic` = this run's algorithmic bytes x the HBM / algorithmic ratio of the
                # committed PMC passes named in traffic_pmc: counters are not read in this run
                'traffic_measured_in_this_run': False,
                'traffic_pmc': traffic_src,
                'launches': xo['launches'],
                'avg_launch_ms': xo['ms'] / max(xo['launches'], 1),
                # blocks the kernel copied (counted on the device) x (2 x block bytes + the switch
                # point's 128-byte line from the other homologue; dense masks: 4 x block bytes):
                # of the two gametes of every offspring that survives its first
                # death draw, the blocks that hold a switch point - the others refer to the
                # parent's block and move nothing (csrc/gnx_half.h)
                'algorithmic_bytes_per_launch': xo['bytes'] / max(xo['launches'], 1),
                # (rounds 1-4 counted the block copies only: this figure x `copies_share`)
                'copies_share': copied,
                # in whole gametes (a gamete = one homologue of L/8 bytes read and written)
                'gamete_equivalents_copied_per_launch':
                    copied * xo['bytes'] / max(xo['launches'], 1) / per_gamete,
                'share_of_the_survivors_gametes_not_copied':
                    1.0 - (copied * xo['bytes'] / per_gamete) / max(2.0 * xo_births, 1.0),
                # SURVEY 8(d)'s figure for the same launch: every birth cut in full, L bytes
                # each (4 homologue reads + 2 masks + 2 writes) - what the kernel would move
                # without the deferral behind the death draws and without shared blocks
                'survey_bytes_per_launch': births / args.steps * cfg['L'],
                'work_avoided_factor': (births / args.steps * cfg['L']) /
                                       max(xo['bytes'] / max(xo['launches'], 1), 1.0),
                # also quote a device-to-device copy measured on this box (the
                # library's own 16-byte-per-lane copy kernel, two 8-GiB buffers)
                'measured_copy_GBps': copy_gbps,
                'frac_of_measured_copy': (ach / copy_gbps) if copy_gbps else None,
            },
            # only the dominant kernel is bracketed with events inside the timed region
            # (GNX_BENCH_PROFILE_ALL=1 times every kernel family, with more event overhead)
            'kernel_ms_per_step': {k: v['ms'] / args.steps for k, v in kt.items()
                                   if v['launches'] > 0},
            # the collector of the shared genome blocks fires every 15-20 steps: a short timed
            # region holds zero or one run of it
            'gc_runs_in_timed_region': gc_in_timed,
            'steps_since_genome_assignment': [args.warmup + (2 if stepper is None else 0) + 1,
                                              args.warmup + (2 if stepper is None else 0) + args.steps],
        }
        if steady is not None:
            out['c4_metric_steady' if args.workload == 'c4_metric' else 'steady'] = steady
        if fam is not None:
            # the whole step against the roofline: every kernel family's algorithmic bytes
            # (the crossover's as moved) over the timed region's ms per step
            step_bytes = sum(v['bytes_per_step'] for v in fam.values())
            out['roofline']['step'] = {
                'algorithmic_bytes_per_step': step_bytes,
                'achieved': step_bytes / (out['ms_per_step'] * 1e-3) / 1e9,
                'peak': peak, 'unit': 'GB/s',
                'frac': step_bytes / (out['ms_per_step'] * 1e-3) / 1e9 / peak,
                'kernel_ms_sum': sum(v['ms_per_step'] for v in fam.values()),
                'families': {k: {'ms': round(v['ms_per_step'], 4),
                                 'MB': round(v['bytes_per_step'] / 1e6, 2)}
                             for k, v in fam.items()},
                'note': 'bytes and per-family times from 10 extra steps with every kernel '
                        'family bracketed by HIP events, outside the timed region'}
            # the mortality compaction moves only the survivors of the tail into the holes of
            # the dead (k_fill); a stable copy of every survivor - rounds 1-2, and
            # what VERDICT r2's step target was quoted against - is 24 + 2 x record bytes per
            # individual.  Both accountings, like the crossover's work_avoided_factor.
            n_sel = cfg['n_traits'] * cfg['loci_per_trait']
            rec = 34.0 + 4.0 * 2 + 4.0 * cfg['n_traits'] + 16.0 * ((n_sel + 63) // 64)
            stable = ((ind_steps + births) / args.steps / world) * (24.0 + 2.0 * rec)   # N at the death draws
            if os.environ.get('GNX_COMPACT_FILL') == '0':
                stable = fam['compact']['bytes_per_step']
            with_stable = step_bytes - fam['compact']['bytes_per_step'] + stable
            out['roofline']['step']['compaction'] = {
                'moved_MB': round(fam['compact']['bytes_per_step'] / 1e6, 2),
                'stable_copy_MB': round(stable / 1e6, 2),
                'frac_counting_a_stable_copy': with_stable / (out['ms_per_step'] * 1e-3) / 1e9 / peak}
            # what shares HBM with the crossover while it runs: the mortality compaction (it
            # runs entirely inside the launch) and the next step's movement (it starts ~60 us
            # into the launch and outlasts it by a few tens of microseconds, so the combined
            # rate below is an upper bound of the window's traffic)
            beside = sum(fam[k]['bytes_per_step'] for k in ('compact', 'move') if k in fam)
            lms = xo['ms'] / max(xo['launches'], 1)
            if lms > 0:
                out['roofline']['beside_the_crossover'] = {
                    'families': ['compact', 'move'], 'bytes_per_step': beside,
                    'crossover_plus_beside_GBps': (xo['bytes'] / max(xo['launches'], 1) + beside)
                    / (lms * 1e-3) / 1e9,
                    'frac': (xo['bytes'] / max(xo['launches'], 1) + beside) / (lms * 1e-3) / 1e9 / peak}
        if phases is not None:
            out['tile_phase_ms_per_step'] = phases
        if comm_infos is not None:
            out['config']['rccl'] = {
                'certified': bool(world > 1 and not cert_errors and
                                  all(ci['transport'] == 'rccl' and ci['nccl_comm_count'] == world
                                      for ci in comm_infos)),
                'certification_errors': cert_errors,
                'devices': devs_ident,
                'nccl_comm_count': [ci['nccl_comm_count'] for ci in comm_infos],
                'nccl_comm_user_rank': [ci['nccl_comm_user_rank'] for ci in comm_infos],
                'nccl_comm_device': [ci['nccl_comm_device'] for ci in comm_infos],
                'hip_device': [ci['hip_device'] for ci in comm_infos],
                'transport': [ci['transport'] for ci in comm_infos],
                'tile_steps': [ci['tile_steps'] for ci in comm_infos],
                'MB_sent_per_step': [round(ci['bytes_sent'] / max(ci['tile_steps'], 1) / 1e6, 3)
                                     for ci in comm_infos]}
            # host wall ms per phase of gnx_tile_step, per rank, over ALL the steps the handle
            # took (burn-in, warm-up and timed): where a rank's step time goes on a real node
            out['tile_step_host_ms_per_phase'] = [
                {k: round(v, 4) for k, v in ci['phase_ms_per_step'].items()} for ci in comm_infos]
        if alone is not None:
            out['roofline']['kernel_alone'] = alone
        if alt is not None:
            out['whole_step_overlap'] = alt
        if world == 1 and not args.no_cpu_baseline:     # rank 0 at N = 1 only
            out['cpu_baseline'] = cpu_baseline()
            ref = ref_cpu_number()
            if ref is not None:
                # flat, small fields (a parser that keeps scalars only keeps them)
                out['cpu_baseline']['reference_value'] = ref.get('value')
                out['cpu_baseline']['reference_sample'] = (
                    '%s; 1 core of %s (%s); measured in the build container by '
                    'tools/ref_cpu_baseline.py - the reference never reaches the GPU box' % (
                        ref.get('sample'), ref.get('host_cores'), ref.get('cpu')))
        model_api = None
        if world == 1 and not args.no_model_api:
            # the same workload through the drop-in API, at the model's own equilibrium.  Run
            # BEFORE the other workloads: after their allocate / free cycles of 100-GB tables in
            # this process the Model's genome table lands where its crossover runs 10 % slower
            # (0.59 against 0.65-0.71 ms per step on the same box; DESIGN 4.1 iv)
            if dev is not None:
                dev.close()
            dev = None
            try:
                model_api, _ = model_api_measure(cfg, args.workload, min(args.steps, 40))
            except Exception as e:      # the contract line must still be printed
                model_api = {'error': '%s: %s' % (type(e).__name__, e)}
        if (world == 1 and args.workload == 'c4_metric' and not args.no_other_workloads
                and stepper is None):
            # BASELINE configs[1], [2] and the template's recombination default, short runs
            if dev is not None:
                dev.close()
            dev = None
            out['other_workloads'] = {}
            for name in ('c2', 'c3', 'c4_dense'):
                try:
                    # (the small ones run 0.2 ms a step: 200 steps, or the figure is the box's jitter)
                    out['other_workloads'][name] = measure_other_workload(
                        name, steps=30 if name == 'c4_dense' else 200,
                        warmup=5 if name == 'c4_dense' else 20,
                        steady_warm=300 if name == 'c4_dense' else 1500)
                except Exception as e:
                    out['other_workloads'][name] = {'error': '%s: %s' % (type(e).__name__, e)}
        if model_api is not None:
            out['model_api'] = model_api
        # last in the line (a reader that keeps only the tail of stdout still sees them)
        if 'other_workloads' in out:
            out['summary'] = {k: ({'ms_per_step': round(v['ms_per_step'], 4), 'value': v['value'],
                                   'state': v.get('state'),
                                   'ms_per_step_young': round(v['young']['ms_per_step'], 4),
                                   'step_frac': round(v['step_frac'], 4)}
                                  if 'ms_per_step' in v else v)
                              for k, v in out['other_workloads'].items()}
            if 'c4_metric_steady' in out:
                out['summary']['c4_metric_steady'] = {
                    'ms_per_step': round(out['c4_metric_steady']['ms_per_step'], 4),
                    'value': out['c4_metric_steady']['value'],
                    'mean_N': out['c4_metric_steady']['mean_N']}
            out['summary']['c4_metric'] = {'ms_per_step': round(out['ms_per_step'], 4),
                                           'value': out['value'],
                                           'step_frac': (round(out['roofline']['step']['frac'], 4)
                                                         if 'step' in out['roofline'] else None)}
        print(json.dumps(out))
    if dev is not None:
        dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
