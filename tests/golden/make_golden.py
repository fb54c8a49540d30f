#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own operators.

Runs ONLY in the build container (needs /root/reference, read-only); writes
small .npz fixtures next to this file.  Each fixture holds the inputs handed to
one reference operator, the random draws it consumed (recovered by replaying the
seeded legacy numpy stream in the same call order) and the reference's outputs.
Nothing of the reference's source is stored - only data.

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import copy
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference  # noqa: E402

gnx = import_reference()
from geonomics.ops import mating as ref_mating            # noqa: E402
from geonomics.ops import movement as ref_movement        # noqa: E402
from geonomics.ops import selection as ref_selection      # noqa: E402
from geonomics.ops import demography as ref_demography    # noqa: E402
from geonomics.structs import genome as ref_genome        # noqa: E402
from geonomics.utils import spatial as ref_spatial        # noqa: E402
import scipy                                               # noqa: E402

META = dict(reference='erthward/geonomics 1.4.9', numpy=np.__version__,
            scipy=scipy.__version__)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, meta=str(META), **arrs)
    print('wrote %-28s %7.1f KB' % (name + '.npz', os.path.getsize(path) / 1e3))


def base_params(dim=(24, 24), N=150, L=160, traits=True, r_alpha=0.05,
                r_beta=None, n_recomb=40, dom=False, sex=False, seed=7,
                mating_radius=3, K_factor=0.6, move_surf=False,
                dist_distr='lognormal', n_births_fixed=True, lam=1):
    W, H = dim
    lyr0 = np.ones((H, W))
    lyr1 = np.tile(np.linspace(0, 1, W), (H, 1))
    gen_arch = {
        'gen_arch_file': None, 'L': L, 'start_p_fixed': 0.5,
        'start_neut_zero': False, 'mu_neut': 0, 'mu_delet': 0,
        'delet_alpha_distr_shape': 0.2, 'delet_alpha_distr_scale': 0.2,
        'r_distr_alpha': r_alpha, 'r_distr_beta': r_beta, 'dom': dom,
        'pleiotropy': False, 'recomb_rate_custom_fn': None,
        'n_recomb_paths_mem': int(1e4), 'n_recomb_paths_tot': int(1e5),
        'n_recomb_sims': n_recomb, 'allow_ad_hoc_recomb': False,
        'jitter_breakpoints': False, 'mut_log': False, 'use_tskit': False,
        'tskit_simp_interval': 100,
    }
    if traits:
        gen_arch['traits'] = {
            'trait_0': {'layer': 'lyr_1', 'phi': 0.05, 'n_loci': 4, 'mu': 0,
                        'alpha_distr_mu': 0.1, 'alpha_distr_sigma': 0,
                        'max_alpha_mag': None, 'gamma': 1, 'univ_adv': False},
            'trait_1': {'layer': 'lyr_0', 'phi': 0.1, 'n_loci': 1, 'mu': 0,
                        'alpha_distr_mu': 0.1, 'alpha_distr_sigma': 0,
                        'max_alpha_mag': None, 'gamma': 2, 'univ_adv': True},
            'trait_2': {'layer': 'lyr_1', 'phi': 0.2, 'n_loci': 7, 'mu': 0,
                        'alpha_distr_mu': 0.25, 'alpha_distr_sigma': 0.1,
                        'max_alpha_mag': None, 'gamma': 1.5, 'univ_adv': False},
        }
    movement = {
        'move': True, 'direction_distr_mu': 0, 'direction_distr_kappa': 0,
        'movement_distance_distr_param1': 0.01,
        'movement_distance_distr_param2': 0.5,
        'movement_distance_distr': dist_distr,
        'dispersal_distance_distr_param1': -1,
        'dispersal_distance_distr_param2': 0.05,
        'dispersal_distance_distr': 'lognormal',
    }
    if move_surf:
        movement['move_surf'] = {'layer': 'lyr_1', 'mixture': True,
                                 'vm_distr_kappa': 12, 'approx_len': 300}
    p = {
        'landscape': {
            'main': {'dim': dim, 'res': (1, 1), 'ulc': (0, 0), 'prj': None},
            'layers': {
                'lyr_0': {'init': {'defined': {'rast': lyr0, 'pts': None,
                                               'vals': None,
                                               'interp_method': None}}},
                'lyr_1': {'init': {'defined': {'rast': lyr1, 'pts': None,
                                               'vals': None,
                                               'interp_method': None}}},
            }},
        'comm': {'species': {'spp_0': {
            'init': {'N': N, 'K_layer': 'lyr_0', 'K_factor': K_factor},
            'mating': {'repro_age': 0, 'sex': sex, 'sex_ratio': 1 / 1,
                       'R': 0.5, 'b': 0.2, 'n_births_distr_lambda': lam,
                       'n_births_fixed': n_births_fixed,
                       'mating_radius': mating_radius,
                       'choose_nearest_mate': False,
                       'inverse_dist_mating': False},
            'mortality': {'max_age': None, 'd_min': 0, 'd_max': 1,
                          'density_grid_window_width': None},
            'movement': movement,
            'gen_arch': gen_arch,
        }}},
        'model': {'T': 20, 'burn_T': 30, 'num': None,
                  'seed': {'num': seed},
                  'its': {'n_its': 1, 'rand_landscape': False,
                          'rand_comm': False, 'rand_genarch': True,
                          'repeat_burn': False}},
    }
    return p


def make_ref_model(**kw):
    p = gnx.make_params_dict(base_params(**kw), 'golden')
    mod = gnx.make_model(p)
    return mod


def assign_genomes(mod):
    """Skip burn-in: mark burned and assign genomes exactly as
    sim/model.py:712-734 does at the end of burn-in."""
    spp = mod.comm[0]
    spp.burned = True
    spp.n_births.append(10)           # _calc_estimated_total_mutations reads it
    spp._set_genomes_and_tables(mod.burn_T, mod.T)
    mod.comm.burned = True
    return spp


def stack_g(spp):
    return np.stack([np.int8(ind.g) for ind in spp.values()])


# ---------------------------------------------------------------- G2 / A9
def g2_recomb_paths():
    out = {}
    for tag, (L, n, alpha, beta) in {
            'sparse': (300, 25, 0.01, None),
            'free': (130, 12, 0.5, None),
            'beta': (257, 16, 2.0, 30.0),
            'homog': (190, 10, None, None)}.items():
        np.random.seed(11)
        rec = ref_genome.Recombinations(L, None, n, alpha, beta, None, False)
        rates = np.array(rec._rates, dtype=np.float64)
        beta_draws = None
        if alpha is not None and beta is not None:
            np.random.seed(11)
            beta_draws = np.random.beta(a=alpha, b=beta, size=L)
        np.random.seed(12)
        rec._set_events(True, np.array([]), False)
        subs = np.array([[*rec._subsetters[k]] for k in range(n)], dtype=np.uint8)
        np.random.seed(12)
        cross = np.array([np.random.binomial(1, rates) for _ in range(n)],
                         dtype=np.uint8)
        out[tag + '_rates'] = rates
        out[tag + '_crossovers'] = cross
        out[tag + '_subsetters'] = subs
        out[tag + '_args'] = np.array([L, n, -1 if alpha is None else alpha,
                                       -1 if beta is None else beta], float)
        if beta_draws is not None:
            out[tag + '_beta_draws'] = beta_draws
    save('g2_recomb_paths', **out)


# ---------------------------------------------------------------- G1 / A10
def g1_crossover():
    out = {}
    for tag, kw in {'sparse': dict(r_alpha=0.02, L=160, n_recomb=40),
                    'free': dict(r_alpha=0.5, L=131, n_recomb=30)}.items():
        mod = make_ref_model(**kw)
        spp = assign_genomes(mod)
        ids = np.array([*spp])
        rng = np.random.RandomState(5)
        n_pairs = 37
        pairs = np.stack([rng.choice(ids, n_pairs), rng.choice(ids, n_pairs)], 1)
        n_births = rng.randint(1, 4, n_pairs)
        recomb_keys = [*rng.randint(0, spp.gen_arch.recombinations._n,
                                    2 * n_births.sum())]
        np.random.seed(99)
        res = ref_mating._do_mating(spp, pairs, n_births, [*recomb_keys])
        child = np.stack([np.int8(off[0]) for pr in res for off in pr])
        np.random.seed(99)
        B = int(n_births.sum())
        start_homs = np.array([np.random.binomial(1, 0.5, 2) for _ in range(B)])
        subs = np.array([[*spp.gen_arch.recombinations._subsetters[k]]
                         for k in range(spp.gen_arch.recombinations._n)],
                        dtype=np.uint8)
        out[tag + '_parents_g'] = stack_g(spp)
        out[tag + '_parent_ids'] = ids
        out[tag + '_subsetters'] = subs
        out[tag + '_pairs'] = pairs
        out[tag + '_n_births'] = n_births
        out[tag + '_recomb_keys'] = np.array(recomb_keys)
        out[tag + '_start_homs'] = start_homs
        out[tag + '_child_g'] = child
    save('g1_crossover', **out)


# ---------------------------------------------------------------- G3/G6
def g3_phenotype_fitness():
    out = {}
    for tag, kw in {'codom': dict(dom=False), 'dom': dict(dom=True)}.items():
        mod = make_ref_model(L=180, **kw)
        spp = assign_genomes(mod)
        ga = spp.gen_arch
        g = stack_g(spp)
        z = np.array([ind.z for ind in spp.values()], dtype=np.float64)
        e = spp._get_e()
        w = spp._calc_fitness()
        rng = np.random.RandomState(3)
        d_rast = rng.rand(*mod.land[0].rast.shape) * 0.6
        d_at = d_rast[spp._cells[:, 1], spp._cells[:, 0]]
        pd_ = ref_selection._calc_prob_death(spp, d_at.copy())
        out[tag + '_g'] = g
        out[tag + '_z'] = z
        out[tag + '_e'] = e
        out[tag + '_w'] = w
        out[tag + '_d_at'] = d_at
        out[tag + '_p_death'] = pd_
        out[tag + '_dom'] = np.asarray(ga.dom)
        out[tag + '_x'] = spp._get_x()
        out[tag + '_y'] = spp._get_y()
        for t, trt in ga.traits.items():
            out['%s_t%i_loci' % (tag, t)] = np.asarray(trt.loci, dtype=np.int64)
            out['%s_t%i_alpha' % (tag, t)] = np.asarray(trt.alpha, dtype=float)
            out['%s_t%i_par' % (tag, t)] = np.array(
                [trt.lyr_num, trt.phi, trt.gamma, float(trt.univ_adv)])
        out[tag + '_rasts'] = np.stack([lyr.rast for lyr in mod.land.values()])
    # deleterious-locus fitness (ops/selection.py:78-94)
    mod = make_ref_model(L=120, traits=False)
    spp = assign_genomes(mod)
    spp.gen_arch.delet_loci = np.array([3, 17, 64, 100])
    spp.gen_arch.delet_loci_s = np.array([0.1, 0.02, 0.3, 0.07])
    out['delet_g'] = stack_g(spp)
    out['delet_loci'] = spp.gen_arch.delet_loci
    out['delet_s'] = spp.gen_arch.delet_loci_s
    out['delet_w'] = ref_selection._calc_fitness_deleterious_mutations(spp)
    save('g3_phenotype_fitness', **out)


# ---------------------------------------------------------------- G4 / A13
class _Land:
    def __init__(self, dim):
        self.dim = dim
        self.res = (1, 1)
        self._dim_om = max(len(str(d)) for d in dim)


def g4_density():
    out = {}
    rng = np.random.RandomState(21)
    cases = {'a': ((50, 50), None, 2400, 'unif'),
             'b': ((64, 64), 8, 3000, 'clump'),
             'c': ((40, 40), 7, 900, 'unif'),
             'd': ((100, 100), None, 9000, 'grad')}
    for tag, (dim, ww, n, kind) in cases.items():
        if kind == 'unif':
            x = rng.rand(n) * dim[0]
            y = rng.rand(n) * dim[1]
        elif kind == 'clump':
            cx = rng.rand(6) * dim[0]
            cy = rng.rand(6) * dim[1]
            k = rng.randint(0, 6, n)
            x = cx[k] + rng.randn(n) * 5
            y = cy[k] + rng.randn(n) * 5
        else:
            x = dim[0] * rng.beta(2, 1, n)
            y = rng.rand(n) * dim[1]
        x = np.clip(x, 0, dim[0] - 0.001)
        y = np.clip(y, 0, dim[1] - 0.001)
        st = ref_spatial._DensityGridStack(_Land(dim), ww)
        dens = st._calc_density(x, y)
        # node coordinates / per-node densities of the four grids
        pts = np.vstack([st.grids[k].grid_coords for k in range(4)])
        vals = np.hstack([st.grids[k]._calc_density(x, y).flatten()
                          for k in range(4)])
        areas = np.hstack([st.grids[k].areas.flatten() for k in range(4)])
        out[tag + '_dim'] = np.array(dim)
        out[tag + '_ww'] = np.array([st.window_width], dtype=float)
        out[tag + '_x'] = x
        out[tag + '_y'] = y
        out[tag + '_dens'] = dens
        out[tag + '_node_pts'] = pts           # (i=y, j=x) per node
        out[tag + '_node_vals'] = vals
        out[tag + '_node_areas'] = areas
    save('g4_density', **out)


# ---------------------------------------------------------------- G5 / A14
def g5_demography():
    rng = np.random.RandomState(8)
    H, W = 30, 36
    N = rng.rand(H, W) * 3
    N[rng.rand(H, W) < 0.1] = 0
    K = rng.rand(H, W) * 2
    K[rng.rand(H, W) < 0.1] = 0
    K[0, 0] = 1e-9
    n_pairs = rng.rand(H, W) * 0.4
    n_pairs[rng.rand(H, W) < 0.3] = 0
    out = dict(N=N, K=K, n_pairs=n_pairs)
    for tag, (R, b, lam, dmin, dmax) in {'a': (0.5, 0.2, 1, 0, 1),
                                         'b': (1.3, 0.6, 3, 0.05, 0.9)}.items():
        dNdt = ref_demography._calc_dNdt(R=R, N=N, K=K)
        N_b = ref_demography._calc_N_b(b=b, n_births_distr_lambda=lam,
                                       n_pairs=n_pairs)
        N_d = ref_demography._calc_Nd(N_b=N_b, dNdt=dNdt)
        d = ref_demography._calc_d(N_d=N_d.copy(), N=N, d_min=dmin, d_max=dmax)
        out[tag + '_par'] = np.array([R, b, lam, dmin, dmax], float)
        out[tag + '_dNdt'] = dNdt
        out[tag + '_N_b'] = N_b
        out[tag + '_N_d'] = N_d
        out[tag + '_d'] = d
    save('g5_demography', **out)


# ---------------------------------------------------------------- G7 / A2
def g7_movement():
    out = {}
    for tag, (distr, p1, p2, mu, kappa) in {
            'lognormal': ('lognormal', 0.5, 0.5, 0, 0),
            'wald': ('wald', 1.5, 2.0, 1.0, 2.5),
            'levy': ('levy', 0.0, 0.3, 0, 0)}.items():
        mod = make_ref_model(traits=False, L=10, N=200)
        spp = mod.comm[0]
        spp._pv.movement_distance_distr = distr
        spp._pv.movement_distance_distr_param1 = p1
        spp._pv.movement_distance_distr_param2 = p2
        spp._pv.direction_distr_mu = mu
        spp._pv.direction_distr_kappa = kappa
        x0 = spp._get_x().copy()
        y0 = spp._get_y().copy()
        np.random.seed(31)
        ref_movement._do_movement(spp)
        x1 = spp._get_x().copy()
        y1 = spp._get_y().copy()
        np.random.seed(31)
        theta = np.random.vonmises(mu, kappa, size=len(x0))
        if distr == 'lognormal':
            dist = np.random.lognormal(mean=p1, sigma=p2, size=len(x0))
        elif distr == 'wald':
            dist = np.random.wald(mean=p1, scale=p2, size=len(x0))
        else:
            from scipy.stats import levy
            dist = levy.rvs(loc=p1, scale=p2, size=len(x0))
        out[tag + '_x0'] = x0
        out[tag + '_y0'] = y0
        out[tag + '_theta'] = theta
        out[tag + '_dist'] = dist
        out[tag + '_x1'] = x1
        out[tag + '_y1'] = y1
        out[tag + '_dim'] = np.array(spp._land_dim)
        out[tag + '_par'] = np.array([p1, p2, mu, kappa], float)
    # _set_e gather after movement (structs/species.py:913-922)
    spp._set_e(mod.land)
    out['e_after'] = spp._get_e()
    out['e_rasts'] = np.stack([lyr.rast for lyr in mod.land.values()])
    out['e_x'] = spp._get_x()
    out['e_y'] = spp._get_y()

    # dispersal with the retry loop (ops/movement.py:98-141)
    mod = make_ref_model(traits=False, L=10, N=50)
    spp = mod.comm[0]
    p1, p2 = 0.2, 0.6
    spp._pv.dispersal_distance_distr_param1 = p1
    spp._pv.dispersal_distance_distr_param2 = p2
    rng = np.random.RandomState(4)
    B = 300
    mx = np.where(rng.rand(B) < 0.5, rng.rand(B) * 1.5,
                  rng.rand(B) * spp._land_dim[0])
    my = np.where(rng.rand(B) < 0.3, spp._land_dim[1] - rng.rand(B) * 1.5,
                  rng.rand(B) * spp._land_dim[1])
    ox = np.zeros(B)
    oy = np.zeros(B)
    A = 12
    th = np.zeros((A, B))
    ds = np.ones((A, B))
    used = np.zeros(B, dtype=np.int32)
    for k in range(B):
        np.random.seed(1000 + k)
        ox[k], oy[k] = ref_movement._do_dispersal(spp, mx[k], my[k], p1, p2)
        # replay: each attempt draws vonmises(0,0) then lognormal(p1,p2)
        np.random.seed(1000 + k)
        for a in range(A):
            t = np.random.vonmises(0, 0)
            d = np.random.lognormal(mean=p1, sigma=p2)
            th[a, k] = t
            ds[a, k] = d
            nx = np.clip(mx[k] + np.cos(t) * d, 0, spp._land_dim[0] - 0.001)
            ny = np.clip(my[k] + np.sin(t) * d, 0, spp._land_dim[1] - 0.001)
            if (nx > 0 and nx < spp._land_dim[0] and ny > 0
                    and ny < spp._land_dim[1]):
                used[k] = a
                break
        else:
            raise RuntimeError('increase A')
    out['disp_mx'] = mx
    out['disp_my'] = my
    out['disp_theta'] = th
    out['disp_dist'] = ds
    out['disp_x'] = ox
    out['disp_y'] = oy
    out['disp_used'] = used
    out['disp_dim'] = np.array(spp._land_dim)
    save('g7_movement', **out)


# ---------------------------------------------------------------- G8 / A6-A8
def g8_pairing():
    out = {}
    # (a) _find_mates filters with injected pairs
    for tag, (sex, repro_age) in {'asex': (False, 0), 'asex_age': (False, 2),
                                  'sex': (True, 0),
                                  'sex_age': (True, (1, 3))}.items():
        mod = make_ref_model(traits=False, L=10, N=120, sex=sex)
        spp = mod.comm[0]
        rng = np.random.RandomState(17)
        n = len(spp)
        ages = rng.randint(0, 6, n)
        sexes = rng.randint(0, 2, n)
        for k, ind in enumerate(spp.values()):
            ind.age = int(ages[k])
            ind.sex = int(sexes[k])
        # remap ids so that ordinal index != id
        focal = rng.choice(n, 70, replace=False)
        mate = (focal + rng.randint(1, n, 70)) % n
        pairs = np.stack([focal, mate], 1)
        # make some reciprocal duplicates
        pairs = np.vstack([pairs, pairs[:15, ::-1]])
        spp._get_mating_pairs = (lambda choose_nearest=False,
                                 inverse_dist_mating=False, _p=pairs: _p.copy())
        mates = ref_mating._find_mates(spp, sex=sex, repro_age=repro_age)
        out[tag + '_pairs_in'] = pairs
        out[tag + '_ages'] = ages
        out[tag + '_sexes'] = sexes
        out[tag + '_ids'] = np.array([*spp])
        out[tag + '_repro_age'] = np.atleast_1d(np.array(repro_age))
        out[tag + '_mates_out'] = np.asarray(mates).reshape(-1, 2)

    # (b) neighbour sets + nearest-mate pairs from the KD-tree
    mod = make_ref_model(traits=False, L=10, N=400, mating_radius=2.5)
    spp = mod.comm[0]
    coords = spp._coords.copy()
    spp._set_kd_tree()
    nb = spp._kd_tree.tree.query_ball_point(coords, 2.5)
    out['kd_coords'] = coords
    out['kd_radius'] = np.array([2.5])
    out['kd_nb_counts'] = np.array([len(l) - 1 for l in nb])
    flat = [sorted(set(l) - {i}) for i, l in enumerate(nb)]
    out['kd_nb_flat'] = np.array([j for l in flat for j in l], dtype=np.int64)
    near = spp._kd_tree._get_mating_pairs(coords, 2.5, choose_nearest=True)
    out['kd_nearest_pairs'] = np.asarray(near)
    # uniform mode: every chosen mate must come from the candidate set
    np.random.seed(3)
    uni = spp._kd_tree._get_mating_pairs(coords, 2.5)
    out['kd_uniform_pairs'] = np.asarray(uni)
    np.random.seed(3)
    inv = spp._kd_tree._get_mating_pairs(coords, 2.5, inverse_dist_mating=True)
    out['kd_inverse_pairs'] = np.asarray(inv)

    # (c) Bernoulli(b) thinning (structs/species.py:2210-2214)
    np.random.seed(40)
    spp._pv.b = 0.35
    pairs = spp._get_mating_pairs()
    np.random.seed(40)
    spp._set_kd_tree()
    raw = spp._kd_tree._get_mating_pairs(coords=spp._coords, dist=2.5)
    keep = np.random.binomial(n=1, p=0.35, size=raw.shape[0])
    out['thin_raw'] = np.asarray(raw)
    out['thin_keep'] = keep
    out['thin_pairs'] = np.asarray(pairs)

    # (d) panmixia (structs/species.py:2178-2194)
    spp._pv.mating_radius = None
    spp._pv.b = 0.4
    np.random.seed(41)
    pp = spp._get_mating_pairs()
    np.random.seed(41)
    n_mates = np.random.binomial(n=len(spp), p=0.4)
    draws = np.random.choice(spp._kd_tree.tree.indices, replace=True,
                             size=n_mates * 2)
    out['pan_n_mates'] = np.array([n_mates])
    out['pan_draws'] = draws
    out['pan_pairs'] = np.asarray(pp)

    # (e) births (ops/mating.py:120-126)
    np.random.seed(42)
    nb_ = ref_mating._draw_n_births(500, 0.7)
    np.random.seed(42)
    out['births_poisson'] = np.random.poisson(0.7, 500)
    out['births_out'] = nb_
    save('g8_pairing', **out)


# ---------------------------------------------------------------- G9 / G
def g9_starting_genomes():
    out = {}
    mod = make_ref_model(L=150, N=101, traits=True)
    spp = mod.comm[0]
    rng = np.random.RandomState(6)
    p = rng.rand(150)
    p[:6] = [0, 1, 1e-6, 1 - 1e-6, 0.5, 0.2475]
    spp.gen_arch.p = p
    np.random.seed(50)
    spp = assign_genomes(mod)
    g = stack_g(spp)
    out['p'] = p
    out['N'] = np.array([len(spp)])
    out['site_counts'] = g.sum(axis=(0, 2))
    out['z'] = np.array([ind.z for ind in spp.values()], dtype=float)
    out['g'] = g
    for t, trt in spp.gen_arch.traits.items():
        out['t%i_loci' % t] = np.asarray(trt.loci, dtype=np.int64)
        out['t%i_alpha' % t] = np.asarray(trt.alpha, dtype=float)
    save('g9_starting_genomes', **out)


# ---------------------------------------------------------------- G11 / A3
def g11_conductance():
    out = {}
    rng = np.random.RandomState(12)
    rast = rng.rand(7, 9)
    rast[2, 3] = 0
    rast[0:2, 0:2] = 0          # corner cell (0,0) has an all-zero neighbourhood
    rast[4, 4:6] = 1.0          # ties for the unimodal arg-max
    rast[3, 5] = 1.0
    for tag, mix in {'mix': True, 'uni': False}.items():
        np.random.seed(60)
        surf = ref_spatial._make_conductance_surface(rast, mixture=mix,
                                                     approx_len=4000,
                                                     vm_distr_kappa=12)
        s = np.float64(surf)
        out[tag + '_mean_cos'] = np.cos(s).mean(axis=2)
        out[tag + '_mean_sin'] = np.sin(s).mean(axis=2)
        out[tag + '_mean_cos2'] = np.cos(2 * s).mean(axis=2)
        out[tag + '_mean_sin2'] = np.sin(2 * s).mean(axis=2)
    out['rast'] = rast
    out['kappa'] = np.array([12.0])
    out['approx_len'] = np.array([4000])
    save('g11_conductance', **out)


# ---------------------------------------------------------------- G10
def g10_envelopes():
    """Whole-model envelopes from the reference: neutral 2-layer model, real
    burn-in (ADF stubbed to pass; paired t-tests active), 100 main steps,
    24 seeds (each with its own randomly drawn trait architecture; 8 in round 2: the
    tolerances of the whole-model tests follow the sampling error of the seed means)."""
    out = {}
    for s in range(1, 25):
        mod = make_ref_model(dim=(30, 30), N=300, L=60, traits=True,
                             K_factor=0.5, r_alpha=0.5, n_recomb=60, seed=s,
                             mating_radius=4)
        mod.walk(T=400, mode='burn', verbose=False)
        assert mod.comm.burned
        spp = mod.comm[0]
        nburn = len(spp.Nt)
        mod.walk(T=100, mode='main', verbose=False)
        out['s%i_Nt' % s] = np.array(spp.Nt)
        out['s%i_births' % s] = np.array(spp.n_births)
        out['s%i_deaths' % s] = np.array(spp.n_deaths)
        out['s%i_nburn' % s] = np.array([nburn])
        g = stack_g(spp)
        out['s%i_freq' % s] = g.mean(axis=(0, 2))
        out['s%i_mean_fit' % s] = np.array([np.mean(spp._get_fit())])
        out['s%i_K_sum' % s] = np.array([spp.K.sum()])
        for t, trt in spp.gen_arch.traits.items():
            out['s%i_t%i_loci' % (s, t)] = np.asarray(trt.loci, dtype=np.int64)
            out['s%i_t%i_alpha' % (s, t)] = np.asarray(trt.alpha, dtype=float)
            out['s%i_t%i_par' % (s, t)] = np.array(
                [trt.lyr_num, trt.phi, trt.gamma, float(trt.univ_adv)])
    out['n_seeds'] = np.array([24])
    save('g10_envelopes', **out)


def g12_stats():
    """sim/stats.py calculators on a reference species + the text the
    reference's writers (utils/io.py:126-168) produce for given values."""
    import tempfile
    from geonomics.sim import stats as ref_stats
    from geonomics.utils import io as ref_io
    out = {}
    mod = make_ref_model(L=48, N=90, traits=True)
    spp = assign_genomes(mod)
    # some drift so that frequencies differ from the 0.5 start and LD exists
    rng = np.random.RandomState(5)
    for ind in spp.values():
        flip = rng.rand(*ind.g.shape) < 0.25
        ind.g = np.where(flip, 0, ind.g).astype(ind.g.dtype)
    out['g'] = stack_g(spp)
    out['het'] = ref_stats._calc_het(spp)
    out['het_mean'] = np.array(ref_stats._calc_het(spp, mean=True))
    out['maf'] = ref_stats._calc_maf(spp)
    with np.errstate(all='ignore'):
        out['ld'] = ref_stats._calc_ld(spp)
    d = tempfile.mkdtemp()
    vals_a = np.array([0.25, 0.5, 1 / 3.0, 0.0])
    p1 = os.path.join(d, 'a.csv')
    ref_io._append_row_to_csv(p1, vals_a, 0)
    ref_io._append_row_to_csv(p1, vals_a[::-1], 5)
    ref_io._append_row_to_csv(p1, 0.125, 10)
    out['row_in'] = vals_a
    out['row_txt'] = np.array(open(p1).read())
    p2 = os.path.join(d, 'b.txt')
    m = out['ld'][:4, :4]
    ref_io._append_array2d_to_array_stack(p2, m)
    ref_io._append_array2d_to_array_stack(p2, m * 2)
    out['stack_txt'] = np.array(open(p2).read())
    p3 = os.path.join(d, 'c.csv')
    ref_io._write_dict_to_csv(p3, {'Nt': [100, np.nan, 104, 99],
                                   'mean_fit': [0.9912345, np.nan, np.nan, 1.0]})
    out['other_float_txt'] = np.array(open(p3).read())
    ref_io._write_dict_to_csv(p3, {'Nt': [100, 101, 104, 99]})
    out['other_int_txt'] = np.array(open(p3).read())
    save('g12_stats', **out)


def g13_change():
    """ops/change.py: layer series, demographic size series, and the K / layer
    trajectory of a whole reference model with landscape + demographic +
    life-history change events."""
    from geonomics.ops import change as ref_change
    out = {}
    rng = np.random.RandomState(11)

    class Lyr:
        pass
    lyr = Lyr()
    lyr.rast = rng.rand(5, 6)
    lyr.dim = (6, 5)
    lyr._scale_min, lyr._scale_max = 0, 1
    lyr.ulc, lyr.res, lyr.prj, lyr.idx = (0, 0), (1, 1), None, 0
    end = rng.rand(5, 6)
    series = ref_change._make_lyr_series(lyr, end, start_t=3, end_t=11, n_steps=4)[0]
    out['lyr_start'], out['lyr_end'] = lyr.rast, end
    out['lyr_t'] = np.array([t for t, _ in series])
    out['lyr_rasts'] = np.stack([r for _, r in series])

    class Spp:
        pass

    class Ch:
        base_K = None

        def _set_base_K(self, spp):
            self.base_K = spp.K

    def run(T, **kw):
        spp, ch = Spp(), Ch()
        spp.K = np.full((2, 2), 2.0)
        fns = list(ref_change._get_dem_change_fns(spp, **kw))
        Ks, ts = [], [t for t, _ in fns]
        for t in range(T):
            spp.t = t
            for tt, fn in fns:
                if tt == t:
                    fn(ch, spp)
            Ks.append(spp.K[0, 0])
        return np.array(Ks), np.array(ts)

    out['mono_K'], out['mono_t'] = run(20, kind='monotonic', start_t=5, end_t=12, rate=0.98)
    out['cyc_K'], out['cyc_t'] = run(60, kind='cyclical', start_t=5, end_t=45, n_cycles=4,
                                     size_range=(0.5, 1.5))
    out['cyc2_K'], out['cyc2_t'] = run(40, kind='cyclical', start_t=2, end_t=30, n_cycles=3,
                                       min_size=0.25, max_size=2.0, increase_first=False)
    out['cust_K'], out['cust_t'] = run(30, kind='custom', timesteps=[4, 9, 20],
                                       sizes=[2, 5, 0.5])
    np.random.seed(3)
    out['stoch_u_K'], out['stoch_u_t'] = run(30, kind='stochastic', start_t=3, end_t=23,
                                             interval=4, size_range=(0.5, 1.5),
                                             distr='uniform')
    np.random.seed(3)
    out['stoch_n_K'], out['stoch_n_t'] = run(30, kind='stochastic', start_t=3, end_t=23,
                                             interval=None, size_range=(0.5, 1.5),
                                             distr='normal')
    # whole model
    p = base_params(dim=(20, 20), N=120, L=40, traits=True, seed=5)
    lyrs = p['landscape']['layers']
    k0 = [*lyrs][0]
    end_rast = np.linspace(0.2, 1.0, 400).reshape(20, 20)
    lyrs[k0]['change'] = {0: {'change_rast': end_rast, 'start_t': 4, 'end_t': 12,
                              'n_steps': 3}}
    sp = p['comm']['species'][[*p['comm']['species']][0]]
    sp['change'] = {'dem': {0: {'kind': 'monotonic', 'start_t': 2, 'end_t': 6, 'rate': 0.9,
                                'interval': None, 'distr': None, 'n_cycles': None,
                                'size_range': None, 'timesteps': None, 'sizes': None},
                            1: {'kind': 'custom', 'start_t': None, 'end_t': None,
                                'rate': None, 'interval': None, 'distr': None,
                                'n_cycles': None, 'size_range': None,
                                'timesteps': [14, 16], 'sizes': [0.5, 1.5]}},
                    'life_hist': {'b': {'timesteps': [3, 10], 'vals': [0.5, 0.1]}}}
    p['model']['T'] = 20
    mod = gnx.make_model(gnx.make_params_dict(p, 'golden'))
    spp = mod.comm[0]
    mod.walk(T=400, mode='burn', verbose=False)
    assert mod.comm.burned
    Ksum, Lsum, bs = [], [], []
    for t in range(20):
        mod.walk(1, 'main', verbose=False)
        Ksum.append(spp.K.sum())
        Lsum.append(mod.land[spp.K_layer].rast.sum())
        bs.append(spp.b)
    out['model_K_layer'] = np.array([spp.K_layer])
    out['model_start_rast'] = p['landscape']['layers'][k0]['init']['defined']['rast']
    out['model_end_rast'] = end_rast
    out['model_K_factor'] = np.array([spp.K_factor])
    out['model_Ksum'] = np.array(Ksum)
    out['model_Lsum'] = np.array(Lsum)
    out['model_b'] = np.array(bs)
    out['model_K_final'] = spp.K
    save('g13_change', **out)


def g14_data():
    """sim/data.py formatters on a reference sample: VCF and FASTA text, the
    attribute strings the CSV writer emits, and the sampling schedule."""
    import re
    from geonomics.sim import data as ref_data
    out = {}
    mod = make_ref_model(L=24, N=60, traits=True, dim=(12, 12))
    spp = assign_genomes(mod)
    rng = np.random.RandomState(2)
    for ind in spp.values():          # drift + a few fixed sites (0 and 1)
        flip = rng.rand(*ind.g.shape) < 0.3
        g = np.where(flip, 0, ind.g).astype(ind.g.dtype)
        g[3, :] = 0
        g[7, :] = 1
        ind.g = g
    ids = sorted([*spp])[:7]
    sample = {i: spp[i] for i in ids}
    genotypes = {i: np.int8(spp[i].g) for i in ids}
    out['ids'] = np.array(ids)
    out['g'] = np.stack([genotypes[i] for i in ids])
    for att in ('x', 'y', 'age', 'sex'):
        out[att] = np.array([getattr(sample[i], att) for i in ids], dtype=float)
    out['z'] = np.array([np.asarray(sample[i].z, dtype=float) for i in ids])
    out['e'] = np.array([np.asarray(sample[i].e, dtype=float) for i in ids])
    vcf = ref_data._format_vcf(sample, genotypes, spp.gen_arch, include_fixed_sites=False)
    vcf_all = ref_data._format_vcf(sample, genotypes, spp.gen_arch, include_fixed_sites=True)
    out['vcf'] = np.array(re.sub(r'##fileDate=\d+', '##fileDate=DATE', vcf))
    out['vcf_fixed'] = np.array(re.sub(r'##fileDate=\d+', '##fileDate=DATE', vcf_all))
    out['fasta'] = np.array(ref_data._format_fasta(sample, genotypes))
    out['csv_str_z'] = np.array([str(sample[i].z) for i in ids])
    out['csv_str_e'] = np.array([str(sample[i].e) for i in ids])
    out['transect'] = np.array(ref_data._get_transect_points([(1.5, 2.0), (9.0, 11.0)], 5))
    # sampling schedules: (T, when) -> list of timesteps
    for tag, (T, when) in {'w0': (20, 0), 'wNone': (20, None), 'w6': (20, 6),
                           'wlist': (20, [2, 5, 19]), 'wlist2': (20, [2, 5])}.items():
        class P(dict):
            __getattr__ = dict.__getitem__
        params = P(model=P(T=T, data=P(
            sampling=P(scheme='all', when=when, include_landscape=False),
            format=P(gen_format=['vcf', 'fasta'], geo_vect_format='csv',
                     geo_rast_format='txt', nonneut_loc_format=None))))
        dc = ref_data._DataCollector('m', params)
        ts = [dc.next_t] + list(dc.when)
        out['when_' + tag] = np.array(ts)
    save('g14_data', **out)


def g15_wf():
    """The reference's Wright-Fisher validation (tests/validation/wf/wf_test.py) in small:
    positions re-drawn uniformly every step, walk one main step, until every locus is
    fixed; persistence times of the loci and the harmonic-mean population size."""
    out = {}
    for s in range(1, 5):
        mod = make_ref_model(dim=(10, 10), N=100, L=60, traits=False, K_factor=1.0,
                             r_alpha=0.5, n_recomb=80, seed=s, mating_radius=20)
        mod.walk(T=400, mode='burn', verbose=False)
        assert mod.comm.burned
        spp = mod.comm[0]
        t0 = len(spp.Nt)
        freqs = []
        for t in range(2500):
            g = stack_g(spp)
            f = g.mean(axis=(0, 2))
            freqs.append(f)
            if ((f == 0) | (f == 1)).all():
                break
            n = len(spp)
            new_x = np.random.uniform(0, 10, n)
            new_y = np.random.uniform(0, 10, n)
            for k, ind in enumerate(spp.values()):
                ind.x = new_x[k]
                ind.y = new_y[k]
            spp._set_coords_and_cells()
            mod.walk(1, 'main', verbose=False)
        F = np.array(freqs)
        fixed = (F == 0) | (F == 1)
        persist = np.where(fixed.any(axis=0), fixed.argmax(axis=0), len(F))
        Nt = np.array(spp.Nt[t0:], dtype=float)
        out['s%i_persist' % s] = persist
        out['s%i_Nharm' % s] = np.array([1.0 / np.mean(1.0 / Nt)])
        out['s%i_steps' % s] = np.array([len(F)])
        print('seed', s, 'steps', len(F), 'mean persistence', persist.mean(),
              'N_harm', out['s%i_Nharm' % s][0], flush=True)
    save('g15_wf', **out)


def g16_spatial_tester():
    """SpatialTester.update (sim/burnin.py:44-59) over the burn-in steps of a reference
    model: positions handed to it at every update, and the mean / std of the per-cell
    count differences it records (the burn-in stationarity tests run on these series)."""
    from geonomics.sim import burnin as ref_burnin
    out = {}
    for s, dim in ((1, (24, 24)), (2, (17, 17))):
        mod = make_ref_model(dim=dim, N=200, L=16, traits=False, seed=s)
        spp = mod.comm[0]
        tester = ref_burnin.SpatialTester(spp)
        xs = [np.array(spp._get_x())]
        ys = [np.array(spp._get_y())]
        for t in range(7):
            mod.walk(1, 'burn', verbose=False)
            xs.append(np.array(spp._get_x()))
            ys.append(np.array(spp._get_y()))
            tester.update(spp)
        out['s%i_dim' % s] = np.array(dim)
        out['s%i_n' % s] = np.array([len(a) for a in xs])
        out['s%i_x' % s] = np.concatenate(xs)
        out['s%i_y' % s] = np.concatenate(ys)
        out['s%i_mean' % s] = np.array(tester.stats[np.mean], dtype=float)
        out['s%i_std' % s] = np.array(tester.stats[np.std], dtype=float)
        out['s%i_counts' % s] = np.array(tester.counts)
    save('g16_spatial_tester', **out)


def g17_pedigree_segments():
    """The tree-sequence edge rows of a gamete (use_tskit=True): Recombinations._set_seg_info
    and _get_seg_info (structs/genome.py:209-281) called as ops/mating.py:141-148 calls
    them - (start homologue, event key, the parent's two node ids) -> (parent node, left,
    right) per segment.  Pure numpy in the reference; tskit only stores the rows
    (structs/species.py:731-736).  Breakpoint jitter off (it adds uniform noise)."""
    out = {}
    for tag, (L, n, alpha) in {'sparse': (300, 25, 0.01), 'homog': (190, 10, None),
                               'free': (64, 8, 0.5)}.items():
        np.random.seed(11)
        rec = ref_genome.Recombinations(L, None, n, alpha, None, None, False)
        np.random.seed(12)
        rec._set_events(True, np.array([], dtype=int), True)       # use_tskit: breakpoints kept
        bps = [np.asarray(rec._breakpoints[k], dtype=np.int64) for k in range(n)]
        rng = np.random.RandomState(3)
        keys, starts, nodes0 = [], [], []
        seg_node, seg_left, seg_right, seg_n = [], [], [], []
        for k in range(n):
            for st in (0, 1):
                pid = int(rng.randint(0, 500))
                node_ids = np.array([2 * pid, 2 * pid + 1])
                segs = [*rec._get_seg_info(start_homologue=st, event_key=k, node_ids=node_ids)]
                keys.append(k)
                starts.append(st)
                nodes0.append(2 * pid)
                seg_n.append(len(segs))
                seg_node += [int(sg[0]) for sg in segs]
                seg_left += [float(sg[1]) for sg in segs]
                seg_right += [float(sg[2]) for sg in segs]
        out[tag + '_L'] = np.array([L])
        out[tag + '_bp_off'] = np.concatenate([[0], np.cumsum([b.size for b in bps])]).astype(np.int64)
        out[tag + '_bp_loci'] = (np.concatenate(bps) if bps else np.zeros(0)).astype(np.int64)
        out[tag + '_keys'] = np.array(keys)
        out[tag + '_starts'] = np.array(starts)
        out[tag + '_parent_node0'] = np.array(nodes0)
        out[tag + '_seg_n'] = np.array(seg_n)
        out[tag + '_seg_node'] = np.array(seg_node)
        out[tag + '_seg_left'] = np.array(seg_left)
        out[tag + '_seg_right'] = np.array(seg_right)
        print('g17', tag, 'gametes', len(keys), 'segments', len(seg_node), flush=True)
    save('g17_pedigree_segments', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g7', 'g8', 'g9',
                             'g10', 'g11', 'g12', 'g13', 'g14', 'g15', 'g16', 'g17']
    fns = {'g1': g1_crossover, 'g2': g2_recomb_paths,
           'g3': g3_phenotype_fitness, 'g4': g4_density,
           'g5': g5_demography, 'g7': g7_movement, 'g8': g8_pairing,
           'g9': g9_starting_genomes, 'g10': g10_envelopes,
           'g11': g11_conductance, 'g12': g12_stats, 'g13': g13_change,
           'g14': g14_data, 'g15': g15_wf, 'g16': g16_spatial_tester,
           'g17': g17_pedigree_segments}
    for w in which:
        fns[w]()
