"""TEST INFRASTRUCTURE ONLY.  One whole time step of the hot path on the CPU,
assembled from the oracle operators (gnx_oracle.py) with the SAME random
streams the HIP kernels use (gnx_draws.py).  Used (i) by bench.py as the
`cpu_baseline` ("port": numpy, single thread) and (ii) by tests to cross-check
whole-step behaviour of the device path.  Follows the reference's fn-queue
(sim/model.py:603-667): age -> movement -> pop dynamics (ops/demography.py:183).
"""
import numpy as np
from scipy.spatial import cKDTree

import gnx_oracle as O
import gnx_draws as D
import philox as P

F = np.float32


class Params:
    """Species parameters with the parameters-file template defaults."""

    def __init__(self, **kw):
        self.b = 0.2
        self.R = 0.5
        self.n_births_lambda = 1
        self.n_births_fixed = True
        self.sexed = False
        self.p_male = 0.5
        self.mating_radius = 10.0
        self.mate_mode = 'uniform'
        self.repro_age = (0, 0)
        self.max_age = None
        self.d_min = 0.0
        self.d_max = 1.0
        self.window_width = None
        self.move = True
        self.dir_mu = 0.0
        self.dir_kappa = 0.0
        self.move_distr = 'lognormal'
        self.move_p1 = 0.01
        self.move_p2 = 0.5
        self.disp_distr = 'lognormal'
        self.disp_p1 = -1.0
        self.disp_p2 = 0.05
        self.K_layer = 0
        self.K_factor = 1.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class State:
    def __init__(self, rasts, params, seed, L=0, traits=(), paths_packed=None):
        self.rasts = np.asarray(rasts, dtype=F)
        self.n_layers, self.H, self.W = self.rasts.shape
        self.p = params
        self.seed = int(seed)
        self.L = L
        self.traits = list(traits)       # dicts: loci, alpha, layer, phi, gamma, univ_adv
        self.paths = paths_packed
        self.lat = O.DensityLattice((self.W, self.H), params.window_width)
        self.step = 0
        self.max_id = -1
        self.geno = None
        r = params.mating_radius
        cs = r * (1.0 + 1e-9) if (r is not None and r > 0) else 8.0
        if r is not None and r > 0:          # (the device's grid: oracle/gnx_oracle.py hash_grid)
            cs /= 8.0 if params.mate_mode == 'nearest' else float(O.cell_div())
        cs = max(cs, max(self.W, self.H) / 2048.0)
        self.cs = cs
        self.ncx = max(1, int(np.ceil(self.W / cs)))
        self.ncy = max(1, int(np.ceil(self.H / cs)))
        self.Nt, self.n_births, self.n_deaths = [], [], []

    @property
    def N(self):
        return self.x.size

    def init_population(self, n):
        self.x, self.y, self.sex = D.init_positions(self.seed, n, self.W, self.H)
        self.age = np.zeros(n, np.int32)
        self.id = np.arange(n, dtype=np.int64)
        self.max_id = n - 1
        self.fit = np.ones(n, F)
        self.e = O.gather_e(list(self.rasts), self.x, self.y).astype(F)
        self.z = np.zeros((n, len(self.traits)), F)

    def set_population(self, x, y, age, sex, ids):
        self.x = np.asarray(x, F)
        self.y = np.asarray(y, F)
        self.age = np.asarray(age, np.int32)
        self.sex = np.asarray(sex, np.uint8)
        self.id = np.asarray(ids, np.int64)
        self.max_id = int(self.id.max()) if self.id.size else -1
        self.fit = np.zeros(self.x.size, F)
        self.e = O.gather_e(list(self.rasts), self.x, self.y).astype(F)
        self.z = np.zeros((self.x.size, len(self.traits)), F)

    def set_genomes(self, geno):
        self.geno = np.ascontiguousarray(geno, dtype=np.uint64)
        self._set_z(np.arange(self.N))

    def assign_genomes(self, n_per_site):
        # the k-th genome drawn goes to the individual with the k-th smallest id: the order
        # the reference walks its individuals in (a dict in insertion = id order,
        # structs/genome.py:1132-1133), whatever order the slots are in
        g = O.starting_genomes(self.N, self.L, n_per_site, self.seed)
        out = np.empty_like(g)
        out[np.argsort(self.id, kind='stable')] = g
        self.set_genomes(out)

    def _set_z(self, idx):
        for t, tr in enumerate(self.traits):
            self.z[idx, t] = O.phenotype_packed(self.geno, idx, tr['loci'],
                                                tr['alpha']).astype(F)


def _permute(s, order):
    for name in ('x', 'y', 'age', 'sex', 'id', 'fit', 'e', 'z'):
        setattr(s, name, getattr(s, name)[order])
    if s.geno is not None:
        s.geno = s.geno[order]


def move(s, inc_age=True):
    p = s.p
    if inc_age:
        s.age = s.age + 1
    if not p.move or s.N == 0:
        return
    th, ds = D.move_draws(s.seed, s.id, s.step, p.move_distr, p.move_p1, p.move_p2,
                          p.dir_mu, p.dir_kappa)
    s.x, s.y = O.move_transform(s.x, s.y, th, ds, (s.W, s.H), dtype=F)
    s.e = O.gather_e(list(s.rasts), s.x, s.y).astype(F)


def sort_by_cell(s):
    inv_cs = 1.0 / s.cs                 # (gnx_cell_of multiplies by the reciprocal)
    cx = np.minimum(s.ncx - 1, (s.x.astype(np.float64) * inv_cs).astype(np.int64))
    cy = np.minimum(s.ncy - 1, (s.y.astype(np.float64) * inv_cs).astype(np.int64))
    order = np.argsort(cy * s.ncx + cx, kind='stable')
    _permute(s, order)


def choose_mates_fast(s, focal=None):
    """the build's uniform mate choice (index sampling over the canonical candidate
    list, oracle/gnx_oracle.py: choose_mates_uniform)"""
    return O.choose_mates_uniform(s.x, s.y, s.id, s.p.mating_radius, s.seed, s.step,
                                  (s.W, s.H), focal=focal)


def find_pairs(s):
    p = s.p
    n = s.N
    if n == 0:
        return np.zeros((0, 2), np.int64)
    keep = D.keep_draws(s.seed, s.id, s.step, p.b)
    if p.mating_radius is None or p.mating_radius < 0:
        f, m = D.panmixia_draws(s.seed, s.id, s.step, n)
        ok = keep & (f != m)
        ok &= (s.age[f] >= p.repro_age[0]) & (s.age[m] >= p.repro_age[1])
        if p.sexed:
            ok &= (s.sex[f] == 0) & (s.sex[m] == 1)
        s.pair_trial = np.nonzero(ok)[0]          # the trial's slot orders the pair
        return np.stack([f[ok], m[ok]], 1)
    if p.mate_mode == 'uniform':
        mate = choose_mates_fast(s)
    else:
        mate = O.choose_mates(s.x, s.y, s.id, p.mating_radius, s.seed, s.step,
                              mode=p.mate_mode)
    has = (mate >= 0) & keep
    m0 = np.maximum(mate, 0)
    has &= (s.age >= p.repro_age[0]) & (s.age[m0] >= p.repro_age[1])
    if p.sexed:
        has &= (s.sex == 0) & (s.sex[m0] == 1)
        i = np.nonzero(has)[0]
        return np.stack([i, mate[i]], 1)
    return O.pairs_from_mates(np.where(has, mate, -1), has, s.id)


def mate(s, pairs, burn):
    p = s.p
    P_ = len(pairs)
    if P_ == 0:
        return 0
    if p.n_births_fixed:
        nb = np.full(P_, int(p.n_births_lambda), np.int64)
    else:
        nb = D.births_draws(s.seed, s.id[pairs[:, 0]], s.step, p.n_births_lambda)
    B = int(nb.sum())
    if B == 0:
        return 0
    par = np.repeat(pairs, nb, axis=0)
    oid = s.max_id + 1 + np.arange(B, dtype=np.int64)
    mx = (s.x[par[:, 0]] + s.x[par[:, 1]]) / F(2.0)
    my = (s.y[par[:, 0]] + s.y[par[:, 1]]) / F(2.0)
    th, ds = D.dispersal_draws(s.seed, oid, s.step, p.disp_distr, p.disp_p1, p.disp_p2)
    ox, oy, _ = O.dispersal(mx, my, th, ds, (s.W, s.H), dtype=F)
    genomes = (not burn) and s.geno is not None
    n_paths = s.paths.shape[0] if s.paths is not None else 1
    start, keys, sex = D.offspring_draws(s.seed, oid, s.step, n_paths, p.sexed, p.p_male)
    n0 = s.N
    s.x = np.concatenate([s.x, ox.astype(F)])
    s.y = np.concatenate([s.y, oy.astype(F)])
    s.age = np.concatenate([s.age, np.zeros(B, np.int32)])
    s.sex = np.concatenate([s.sex, sex])
    s.id = np.concatenate([s.id, oid])
    s.fit = np.concatenate([s.fit, np.ones(B, F)])
    s.e = np.concatenate([s.e, O.gather_e(list(s.rasts), ox, oy).astype(F)])
    s.z = np.concatenate([s.z, np.zeros((B, s.z.shape[1]), F)])
    if genomes:
        child = O.crossover(s.geno, s.paths, par, keys, start)
        s.geno = np.concatenate([s.geno, child])
        s._set_z(np.arange(n0, n0 + B))
    s.max_id += B
    return B


def death_probs(s, with_selection, VN, VP):
    p = s.p
    lat = s.lat
    cx = s.x.astype(np.int64)
    cy = s.y.astype(np.int64)
    cN = O.spline_coeffs(lat, VN)
    # N.max() over every cell (ops/demography.py:116)
    Nr = O.spline_raster(lat, VN)
    nmax = Nr.max()
    Nc = np.clip(O.spline_at(lat, VN, cx + 0.5, cy + 0.5, cN), 0, None)
    if VP is not None:
        Pc = np.clip(O.spline_at(lat, VP, cx + 0.5, cy + 0.5), 0, None)
    else:
        Pc = np.zeros_like(Nc)
    K = s.rasts[p.K_layer][cy, cx].astype(np.float64) * p.K_factor
    with np.errstate(divide='ignore', invalid='ignore'):
        dNdt = p.R * (1 - (Nc / K)) * Nc
        dNdt = np.clip(dNdt, -nmax, None)
        dNdt[np.isnan(dNdt)] = -nmax
        dNdt[np.isinf(dNdt)] = -nmax
        N_d = p.b * p.n_births_lambda * Pc - dNdt
        d = N_d / Nc
    d[np.isnan(d)] = 0
    d = np.clip(d, p.d_min, p.d_max)
    pd_ = d.copy()
    if with_selection and s.traits:
        w = O.fitness_traits(s.e.astype(np.float64), s.z.astype(np.float64),
                             [t['layer'] for t in s.traits], [t['phi'] for t in s.traits],
                             [t['gamma'] for t in s.traits],
                             [t['univ_adv'] for t in s.traits])
        s.fit = w.astype(F)
        pd_ = O.prob_death(d, w)
    if p.max_age is not None:
        pd_[s.age > p.max_age] = 1.0
    return pd_, d


def pop_dynamics(s, burn=False, with_selection=True):
    sort_by_cell(s)
    s.pair_trial = None
    pairs = find_pairs(s)
    # offspring ids follow the canonical (hash cell, id) order of the pairs' focal
    # individuals (of the trial's slot under panmixia): tiling-independent
    if len(pairs):
        who = pairs[:, 0] if s.pair_trial is None else s.pair_trial
        k = O.pair_order_keys(s.x[who], s.y[who], s.id[who], (s.W, s.H), s.p.mating_radius,
                              s.p.mate_mode)
        pairs = pairs[np.argsort(k, kind='stable')]
    VP = None
    if len(pairs):
        mx = (s.x[pairs[:, 0]] + s.x[pairs[:, 1]]) / F(2.0)
        my = (s.y[pairs[:, 0]] + s.y[pairs[:, 1]]) / F(2.0)
        VP = s.lat.node_density(mx, my)
    B = mate(s, pairs, burn)
    VN = s.lat.node_density(s.x, s.y)
    pd_, _ = death_probs(s, with_selection and not burn, VN, VP)
    u = D.death_draws(s.seed, s.id, s.step).astype(np.float64)
    dead = u < pd_
    # (kept for the tests that explain where a device run and this one part ways)
    s.last_death = dict(ids=s.id.copy(), u=u, p=pd_.copy())
    keep = ~dead
    _permute(s, np.nonzero(keep)[0])
    s.n_births.append(B)
    s.n_deaths.append(int(dead.sum()))
    return len(pairs), B, int(dead.sum())


def step(s, burn=False, with_selection=True):
    s.Nt.append(s.N)
    move(s, inc_age=True)
    out = pop_dynamics(s, burn, with_selection)
    s.step += 1
    return out
