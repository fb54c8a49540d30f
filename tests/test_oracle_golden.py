"""Pin the oracle (oracle/gnx_oracle.py) against golden vectors captured from the
REFERENCE's own operators (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import gnx_oracle as O
from conftest import load_golden


# ------------------------------------------------------------------ A9
@pytest.mark.parametrize('tag', ['sparse', 'free', 'beta', 'homog'])
def test_recomb_rates_and_paths(tag):
    g = load_golden('g2_recomb_paths')
    L, n, alpha, beta = g[tag + '_args']
    L, n = int(L), int(n)
    alpha = None if alpha < 0 else alpha
    beta = None if beta < 0 else beta
    bd = g[tag + '_beta_draws'] if (tag + '_beta_draws') in g.files else None
    rates = O.recomb_rates(L, alpha, beta, bd)
    np.testing.assert_array_equal(rates, g[tag + '_rates'])
    assert rates[0] == 0
    paths = O.recomb_paths(g[tag + '_crossovers'])
    subs = np.stack([O.subsetter_from_path(p) for p in paths])
    np.testing.assert_array_equal(subs, g[tag + '_subsetters'])
    # packed round trip + breakpoint CSR
    pk = O.pack_bits(paths)
    np.testing.assert_array_equal(O.unpack_bits(pk, L), paths)
    offs, loci = O.breakpoints_from_paths(paths)
    for k in range(n):
        bp = loci[offs[k]:offs[k + 1]]
        rebuilt = np.zeros(L, dtype=np.int64)
        for b in bp:
            rebuilt[b:] ^= 1
        np.testing.assert_array_equal(rebuilt, paths[k])
        np.testing.assert_array_equal(bp, np.nonzero(g[tag + '_crossovers'][k])[0])


# ------------------------------------------------------------------ A10
@pytest.mark.parametrize('tag', ['sparse', 'free'])
def test_crossover_bit_exact(tag):
    g = load_golden('g1_crossover')
    pg = g[tag + '_parents_g']
    N, L, _ = pg.shape
    ids = g[tag + '_parent_ids']
    subs = g[tag + '_subsetters']
    paths = subs[:, 1::2]                       # '01' <=> path bit 1
    np.testing.assert_array_equal(subs[:, 0::2], 1 - paths)
    pairs = g[tag + '_pairs']
    nb = g[tag + '_n_births']
    keys = O.reference_key_layout(nb, g[tag + '_recomb_keys'])
    row_of = {int(i): k for k, i in enumerate(ids)}
    prow = np.array([[row_of[int(a)], row_of[int(b)]] for a, b in pairs])
    parent_rows = np.repeat(prow, nb, axis=0)
    geno = O.pack_genomes(pg)
    np.testing.assert_array_equal(O.unpack_genomes(geno, L), pg)
    child = O.crossover(geno, O.pack_bits(paths), parent_rows, keys,
                        g[tag + '_start_homs'])
    np.testing.assert_array_equal(O.unpack_genomes(child, L), g[tag + '_child_g'])


# ------------------------------------------------------------------ A12/A15
@pytest.mark.parametrize('tag', ['codom', 'dom'])
def test_phenotype_fitness_death(tag):
    g = load_golden('g3_phenotype_fitness')
    G = g[tag + '_g']
    dom = g[tag + '_dom']
    z_ref = g[tag + '_z']
    e = g[tag + '_e']
    n_trt = z_ref.shape[1]
    z = np.zeros_like(z_ref)
    lyr, phi, gamma, ua = [], [], [], []
    geno = O.pack_genomes(G)
    for t in range(n_trt):
        loci = g['%s_t%i_loci' % (tag, t)]
        alpha = g['%s_t%i_alpha' % (tag, t)]
        z[:, t] = O.phenotype(G, loci, alpha, dom)
        zp = O.phenotype_packed(geno, np.arange(G.shape[0]), loci, alpha, dom)
        np.testing.assert_array_equal(zp, z[:, t])
        par = g['%s_t%i_par' % (tag, t)]
        lyr.append(int(par[0]))
        phi.append(par[1])
        gamma.append(par[2])
        ua.append(bool(par[3]))
    np.testing.assert_allclose(z, z_ref, rtol=0, atol=1e-15)
    # environment gather
    e2 = O.gather_e(list(g[tag + '_rasts']), g[tag + '_x'], g[tag + '_y'])
    np.testing.assert_array_equal(e2, e)
    w = O.fitness_traits(e, z, lyr, phi, gamma, ua)
    np.testing.assert_allclose(w, g[tag + '_w'], rtol=1e-14)
    pd_ = O.prob_death(g[tag + '_d_at'], w)
    np.testing.assert_allclose(pd_, g[tag + '_p_death'], rtol=1e-14)


def test_deleterious_fitness():
    g = load_golden('g3_phenotype_fitness')
    w = O.fitness_deleterious(g['delet_g'], g['delet_loci'], g['delet_s'])
    np.testing.assert_allclose(w, g['delet_w'], rtol=1e-14)


# ------------------------------------------------------------------ A13
# (mean |diff| / mean density, max |diff| / peak density) against the reference's raster
DENSITY_BOUNDS = {'a': (0.0125, 0.055), 'b': (0.010, 0.015), 'c': (0.011, 0.040), 'd': (0.006, 0.025)}


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_density_nodes_exact_and_raster_tolerance(tag):
    g = load_golden('g4_density')
    dim = tuple(int(v) for v in g[tag + '_dim'])
    ww = g[tag + '_ww'][0]
    x, y = g[tag + '_x'], g[tag + '_y']
    lat = O.DensityLattice(dim, ww)
    V = lat.node_density(x, y)
    # node values: the four reference grids are exactly the lattice nodes
    pts = g[tag + '_node_pts']
    ii = np.rint(pts[:, 0] / lat.hww).astype(int)
    jj = np.rint(pts[:, 1] / lat.hww).astype(int)
    assert len(pts) == lat.J[0] * lat.J[1]
    assert len(set(zip(ii.tolist(), jj.tolist()))) == len(pts)
    np.testing.assert_allclose(lat.areas[ii, jj], g[tag + '_node_areas'],
                               rtol=1e-12)
    np.testing.assert_allclose(V[ii, jj], g[tag + '_node_vals'], rtol=1e-12)
    # raster: natural bicubic spline vs the reference's Clough-Tocher griddata.
    # Stated tolerance (density), per fixture: (mean |diff| / mean density,
    # max |diff| / peak density) measured a 1.06 % / 4.9 %, b 0.81 % / 1.2 %, c 0.91 % / 3.5 %,
    # d 0.46 % / 2.1 % - the rougher the node field (a: 24 individuals per window, b: 47,
    # d: 90), the more the two interpolants differ between the nodes.
    ref = np.clip(g[tag + '_dens'], 0, None)
    mine = O.density_raster(lat, x, y)
    assert mine.shape == ref.shape
    assert not np.isnan(ref).any()
    diff = np.abs(mine - ref)
    b_mean, b_max = DENSITY_BOUNDS[tag]
    assert diff.mean() <= b_mean * ref.mean(), (diff.mean(), ref.mean())
    assert diff.max() <= b_max * ref.max(), (diff.max(), ref.max())


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_density_tolerance_is_the_references_own_ambiguity(tag):
    """Why A13's raster tolerance is what it is.  The reference interpolates the
    lattice nodes with scipy.interpolate.griddata(method='cubic') (utils/spatial.py:132-146): a
    Clough-Tocher scheme on qhull's Delaunay triangulation.  The nodes form a REGULAR lattice -
    every square has two valid diagonals, the triangulation is degenerate, and which diagonal
    qhull picks depends on the order the points are handed in.  So the reference's raster is one
    member of a family: its own algorithm, given the same nodes in another order, lands 3-9 % of
    the peak density away from the committed raster - FARTHER than the build's natural bicubic
    spline through the same nodes does (a 4.9 %, b 1.2 %, c 3.5 %, d 2.1 %).  No smooth
    interpolant of the nodes can agree with "the reference" better than the reference agrees
    with itself; the nodes (exact to 1e-12, test above) are what is pinned."""
    from scipy.interpolate import griddata
    g = load_golden('g4_density')
    dim = tuple(int(v) for v in g[tag + '_dim'])
    x, y = g[tag + '_x'], g[tag + '_y']
    lat = O.DensityLattice(dim, g[tag + '_ww'][0])
    V = lat.node_density(x, y)
    ref = np.clip(g[tag + '_dens'], 0, None)
    H, W = ref.shape
    GX, GY = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    PX, PY = np.meshgrid(lat.cx, lat.cy)
    pts = np.column_stack([PX.ravel(), PY.ravel()])
    vals = V.ravel()
    mine = np.abs(O.density_raster(lat, x, y) - ref)
    ct_max, ct_mean, rasters = [], [], []
    for seed in range(1, 8):
        perm = np.random.RandomState(seed).permutation(len(pts))
        ct = np.clip(np.nan_to_num(griddata(pts[perm], vals[perm], (GX, GY), method='cubic')),
                     0, None)
        rasters.append(ct)
        d = np.abs(ct - ref)
        ct_max.append(d.max() / ref.max())
        ct_mean.append(d.mean() / ref.mean())
    # the reference's algorithm is not one raster: members of its family differ among themselves
    # by more than the build differs from the committed member
    spread = (np.max(rasters, axis=0) - np.min(rasters, axis=0)).max() / ref.max()
    assert mine.max() / ref.max() <= spread, (mine.max() / ref.max(), spread)
    # ... and the build is as close to the committed raster as a typical other member is
    assert mine.max() / ref.max() <= np.median(ct_max), (mine.max() / ref.max(), ct_max)
    assert mine.mean() / ref.mean() <= 1.15 * np.median(ct_mean), (mine.mean() / ref.mean(), ct_mean)


# ------------------------------------------------------------------ A14
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_demography_algebra(tag):
    g = load_golden('g5_demography')
    R, b, lam, dmin, dmax = g[tag + '_par']
    dNdt, N_b, N_d, d = O.calc_d(g['N'], g['K'], g['n_pairs'], R, b, lam,
                                 dmin, dmax)
    np.testing.assert_array_equal(dNdt, g[tag + '_dNdt'])
    np.testing.assert_array_equal(N_b, g[tag + '_N_b'])
    np.testing.assert_array_equal(N_d, g[tag + '_N_d'])
    np.testing.assert_array_equal(d, g[tag + '_d'])


# ------------------------------------------------------------------ A2
@pytest.mark.parametrize('tag', ['lognormal', 'wald', 'levy'])
def test_movement_transform(tag):
    g = load_golden('g7_movement')
    nx, ny = O.move_transform(g[tag + '_x0'], g[tag + '_y0'], g[tag + '_theta'],
                              g[tag + '_dist'], g[tag + '_dim'])
    np.testing.assert_array_equal(nx, g[tag + '_x1'])
    np.testing.assert_array_equal(ny, g[tag + '_y1'])


def test_set_e_gather():
    g = load_golden('g7_movement')
    e = O.gather_e(list(g['e_rasts']), g['e_x'], g['e_y'])
    np.testing.assert_array_equal(e, g['e_after'])


def test_dispersal_retry_loop():
    g = load_golden('g7_movement')
    ox, oy, used = O.dispersal(g['disp_mx'], g['disp_my'], g['disp_theta'],
                               g['disp_dist'], g['disp_dim'])
    np.testing.assert_array_equal(used, g['disp_used'])
    np.testing.assert_array_equal(ox, g['disp_x'])
    np.testing.assert_array_equal(oy, g['disp_y'])
    assert (g['disp_used'] > 0).sum() >= 10      # the retry path is exercised
    assert (ox > 0).all() and (oy > 0).all()


# ------------------------------------------------------------------ A6-A8
@pytest.mark.parametrize('tag', ['asex', 'asex_age', 'sex', 'sex_age'])
def test_find_mates_filters(tag):
    g = load_golden('g8_pairing')
    pairs_in = g[tag + '_pairs_in']
    ages, sexes, ids = g[tag + '_ages'], g[tag + '_sexes'], g[tag + '_ids']
    ra = g[tag + '_repro_age']
    sexed = tag.startswith('sex')
    repro_age = tuple(ra) if len(ra) == 2 else ra[0]
    # express the injected pair list as (mate, keep) per focal: the reference
    # pair list has one row per focal, plus reciprocal duplicates
    n = len(ids)
    if sexed:
        ok = (sexes[pairs_in[:, 0]] == 0) & (sexes[pairs_in[:, 1]] == 1)
        pr = pairs_in[ok]
    else:
        seen = {}
        for a, b in pairs_in:
            seen.setdefault(frozenset((int(a), int(b))), (a, b))
        pr = np.array(list(seen.values()))
    pr = O.repro_age_filter(pr, ages, repro_age, sexed)
    mine = {frozenset((int(ids[a]), int(ids[b]))) for a, b in pr}
    ref = {frozenset((int(a), int(b))) for a, b in g[tag + '_mates_out']}
    assert mine == ref
    if sexed:   # orientation: female first
        id2k = {int(i): k for k, i in enumerate(ids)}
        for a, b in g[tag + '_mates_out']:
            assert sexes[id2k[int(a)]] == 0 and sexes[id2k[int(b)]] == 1


def test_dedup_rule_equals_frozenset_dedup():
    rng = np.random.RandomState(0)
    n = 300
    mate = rng.randint(-1, n, n)
    mate[mate == np.arange(n)] = -1
    # force reciprocal choices
    for i in range(0, 60, 2):
        mate[i], mate[i + 1] = i + 1, i
    keep = rng.rand(n) < 0.6
    pr = O.pairs_from_mates(mate, keep)
    mine = [frozenset((int(a), int(b))) for a, b in pr]
    assert len(mine) == len(set(mine))
    ref = {frozenset((i, int(mate[i]))) for i in range(n) if mate[i] >= 0 and keep[i]}
    assert set(mine) == ref


def test_neighbour_sets_match_kdtree():
    g = load_golden('g8_pairing')
    c = g['kd_coords']
    r = g['kd_radius'][0]
    nbs = O.neighbour_lists(c[:, 0], c[:, 1], r, dtype=np.float64)
    np.testing.assert_array_equal([len(l) for l in nbs], g['kd_nb_counts'])
    np.testing.assert_array_equal(np.concatenate(nbs), g['kd_nb_flat'])
    # reference uniform / inverse-distance choices always come from these sets
    for key in ('kd_uniform_pairs', 'kd_inverse_pairs'):
        pr = g[key]
        assert len(pr) == (g['kd_nb_counts'] > 0).sum()
        for i, m in pr:
            assert m in nbs[i]


def test_nearest_mate_matches_kdtree():
    g = load_golden('g8_pairing')
    c = g['kd_coords']
    r = g['kd_radius'][0]
    mate = O.choose_mates(c[:, 0], c[:, 1], np.arange(len(c)), r, 1, 0,
                          mode='nearest', dtype=np.float64)
    ref = g['kd_nearest_pairs']
    got = np.stack([np.nonzero(mate >= 0)[0], mate[mate >= 0]], 1)
    np.testing.assert_array_equal(got, ref)


def test_uniform_mate_choice_is_uniform():
    # chi-square of the index-sampled choice over candidate positions
    rng = np.random.RandomState(2)
    n = 600
    x = rng.rand(n) * 20
    y = rng.rand(n) * 20
    counts = np.zeros(4)
    tot = 0
    for step in range(30):
        mate = O.choose_mates(x, y, np.arange(n) + 1000, 1.2, 77, step, dim=(20, 20))
        nbs = O.neighbour_lists(x, y, 1.2)
        for i in range(n):
            if len(nbs[i]) == 4:
                counts[list(nbs[i]).index(mate[i])] += 1
                tot += 1
    exp = tot / 4
    chi2 = ((counts - exp) ** 2 / exp).sum()
    assert tot > 1000 and chi2 < 16.3      # p = 0.001, 3 dof


def test_bernoulli_thinning_and_panmixia():
    g = load_golden('g8_pairing')
    raw, keep = g['thin_raw'], g['thin_keep']
    np.testing.assert_array_equal(raw[keep.astype(bool)], g['thin_pairs'])
    pp = O.panmictic_pairs(g['pan_draws'], int(g['pan_n_mates'][0]))
    ref = {frozenset((int(a), int(b))) for a, b in g['pan_pairs']}
    assert [frozenset((int(a), int(b))) for a, b in pp] == \
        [frozenset((int(a), int(b))) for a, b in g['pan_pairs']] or \
        {frozenset((int(a), int(b))) for a, b in pp} == ref
    assert len(pp) == len(g['pan_pairs'])


def test_births():
    g = load_golden('g8_pairing')
    nb = O.n_births(500, 0.7, False, g['births_poisson'])
    np.testing.assert_array_equal(nb, g['births_out'])
    assert (O.n_births(7, 2, True) == 2).all()


def test_poisson_knuth_distribution():
    from philox import philox4x32, u01
    n = 40000
    u = np.concatenate([u01(philox4x32(5, np.arange(n, dtype=np.uint64), b))
                        for b in range(8)], axis=1)
    for lam in (0.7, 2.0):
        k = O.poisson_knuth(lam, u)
        assert abs(k.mean() - lam) < 0.03
        assert abs(k.var() - lam) < 0.08


# ------------------------------------------------------------------ G
def test_starting_mutation_counts_and_genomes():
    g = load_golden('g9_starting_genomes')
    N = int(g['N'][0])
    n = O.starting_mutation_counts(N, g['p'])
    np.testing.assert_array_equal(n, g['site_counts'])
    geno = O.starting_genomes(N, len(n), n, seed=9)
    G = O.unpack_genomes(geno, len(n))
    np.testing.assert_array_equal(G.sum(axis=(0, 2)), n)
    # homologue occupancy is uniform: every homologue carries ~ sum(p) ones
    per_hom = G.sum(axis=1).ravel()
    assert abs(per_hom.mean() - n.sum() / (2 * N)) < 1e-9
    assert per_hom.std() < 3 * np.sqrt((g['p'] * (1 - g['p'])).sum()) / 1.5
    # phenotypes of the reference's own starting genomes
    for t in range(3):
        z = O.phenotype(g['g'], g['t%i_loci' % t], g['t%i_alpha' % t])
        np.testing.assert_allclose(z, g['z'][:, t], atol=1e-15)


# ------------------------------------------------------------------ A3
def test_conductance_distribution_moments():
    g = load_golden('g11_conductance')
    rast = g['rast']
    kappa = g['kappa'][0]
    H, W = rast.shape
    cy, cx = np.mgrid[0:H, 0:W]
    cy, cx = cy.ravel(), cx.ravel()
    from scipy.special import i0, i1, iv
    a1 = i1(kappa) / i0(kappa)            # E cos(theta - loc) of von Mises
    a2 = iv(2, kappa) / i0(kappa)
    p = O.conductance_weights(rast, cy, cx)
    tol = 4.5 / np.sqrt(g['approx_len'][0])
    # mixture: E cos = A1 * sum_k p_k cos(dir_k)  (likewise sin, 2nd harmonic)
    mc = (p * np.cos(O.QUEEN_DIRS)).sum(1) * a1
    ms = (p * np.sin(O.QUEEN_DIRS)).sum(1) * a1
    mc2 = (p * np.cos(2 * O.QUEEN_DIRS)).sum(1) * a2
    ms2 = (p * np.sin(2 * O.QUEEN_DIRS)).sum(1) * a2
    assert np.abs(mc - g['mix_mean_cos'].ravel()).max() < tol
    assert np.abs(ms - g['mix_mean_sin'].ravel()).max() < tol
    assert np.abs(mc2 - g['mix_mean_cos2'].ravel()).max() < tol
    assert np.abs(ms2 - g['mix_mean_sin2'].ravel()).max() < tol
    assert np.allclose(p[0], 0.125)       # all-zero neighbourhood -> uniform
    loc = O.conductance_unimodal_loc(rast, cy, cx)
    assert np.abs(a1 * np.cos(loc) - g['uni_mean_cos'].ravel()).max() < tol
    assert np.abs(a1 * np.sin(loc) - g['uni_mean_sin'].ravel()).max() < tol


# ------------------------------------------------------------------ G10
def _g10_traits(g, s):
    traits = []
    for t in range(3):
        par = g['s%i_t%i_par' % (s, t)]
        traits.append(dict(loci=g['s%i_t%i_loci' % (s, t)], alpha=g['s%i_t%i_alpha' % (s, t)],
                           layer=int(par[0]), phi=par[1], gamma=par[2], univ_adv=bool(par[3])))
    return traits


def test_whole_model_envelopes_vs_reference():
    """Whole-loop statistics of the oracle step (same operators, device random
    streams; 48 runs) against 24 reference runs of the same model (30x30, 2 layers, N0 300,
    K_factor 0.5, radius 4, L 60, r 0.5, 3 traits; each run's own trait
    architecture).  Trajectories cannot match stream for stream; the means must."""
    import gnx_step as S
    g = load_golden('g10_envelopes')
    W = H = 30
    L = 60
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))])
    rng = np.random.RandomState(0)
    paths = O.pack_bits(O.recomb_paths((rng.rand(60, L) < 0.5).astype(np.uint8)
                                       * (np.arange(L) > 0)))
    ref = dict(burn=[], first=[], main=[], br=[])
    mine = dict(burn=[], first=[], main=[], br=[])
    drift_ref, drift_mine = [], []
    n_seeds = int(g['n_seeds'][0])
    for s in range(1, n_seeds + 1):
        nb = int(g['s%i_nburn' % s][0])
        R = g['s%i_Nt' % s]
        ref['burn'].append(R[10:nb].mean())
        ref['first'].append(R[nb:nb + 20].mean())
        ref['main'].append(R[-50:].mean())
        Rprev = np.concatenate([[300], R[:-1]])
        ref['br'].append((g['s%i_births' % s][10:nb] / Rprev[10:nb]).mean())
        sel = np.concatenate([tr['loci'] for tr in _g10_traits(g, s)])
        neutral = np.setdiff1d(np.arange(L), sel)
        drift_ref.append(g['s%i_freq' % s][neutral] - 0.5)
        # two oracle runs per reference run (seeds 100 + s and 200 + s): the oracle's side of
        # the drift ratio below then carries half the variance of the reference's
        for base in (100, 200):
            st = S.State(rasts, S.Params(mating_radius=4.0, K_factor=0.5), base + s, L=L,
                         traits=_g10_traits(g, s), paths_packed=paths)
            st.init_population(300)
            for _ in range(60):
                S.step(st, burn=True)
            st.assign_genomes(O.starting_mutation_counts(st.N, np.full(L, 0.5)))
            for _ in range(100):
                S.step(st, burn=False)
            Nt = np.array(st.Nt[1:] + [st.N])
            mine['burn'].append(Nt[10:60].mean())
            mine['first'].append(Nt[60:80].mean())
            mine['main'].append(Nt[-50:].mean())
            mine['br'].append((np.array(st.n_births[10:60]) / np.array(st.Nt[10:60])).mean())
            # genetic drift of the neutral loci over the 100 main steps (start 0.5)
            drift_mine.append(O.unpack_genomes(st.geno, L).mean(axis=(0, 2))[neutral] - 0.5)
    v_ref = np.mean(np.concatenate(drift_ref) ** 2)
    v_mine = np.mean(np.concatenate(drift_mine) ** 2)
    # The loci of one run drift together (one pedigree): the runs are the samples.  The mean
    # squared drift of a run has a coefficient of variation of 0.42 over the 24 reference runs
    # and 0.20 - 0.31 over the oracle's; a bootstrap over runs puts the standard error of the
    # ratio at 0.08, most of it the reference's 24 runs.  Measured: 0.896 (0.816 from seeds
    # 100 + s alone, 0.976 from 200 + s): the window is +- 2 standard errors.
    assert 0.85 < v_mine / v_ref < 1.18, (v_mine, v_ref)
    m = {k: (np.mean(ref[k]), np.mean(mine[k])) for k in ref}
    # the mean of N over the last 50 steps carries 2.5 % sampling error over the 24 reference
    # runs (sd 23 of 185 per run) and 1.8 % over the oracle's 48; measured: burn -1.4 %,
    # births / N -1.0 %, first 20 main steps -1.2 %, last 50 main steps +3.8 % (+7.8 % from
    # seeds 100 + s alone, -0.3 % from 200 + s, +0.2 % from a third set 300 + s)
    assert abs(m['burn'][1] / m['burn'][0] - 1) < 0.03, m
    assert abs(m['br'][1] / m['br'][0] - 1) < 0.03, m
    assert abs(m['first'][1] / m['first'][0] - 1) < 0.05, m
    assert abs(m['main'][1] / m['main'][0] - 1) < 0.05, m


# ---- g12: statistics (reference sim/stats.py) ---------------------------------------
def test_stats_oracle_vs_reference():
    d = load_golden("g12_stats")
    g = d['g']
    np.testing.assert_array_equal(O.stats_het(g), d['het'])
    assert float(np.mean(O.stats_het(g))) == float(d['het_mean'])
    np.testing.assert_array_equal(O.stats_maf(g), d['maf'])
    np.testing.assert_allclose(O.stats_ld(g), d['ld'], rtol=1e-10, atol=1e-14, equal_nan=True)


def test_uniform_mate_choice_properties():
    """index sampling: a mate iff a neighbour exists, always within the radius, the same
    (by id) however the individuals are ordered, and the exact fallback when almost all
    candidates of the 3x3 cells are out of reach"""
    rng = np.random.RandomState(8)
    n, W, H, r = 1500, 40, 30, 1.7
    x = (rng.rand(n) * W).astype(np.float32)
    y = (rng.rand(n) * H).astype(np.float32)
    # a dense clump next to a loner whose only neighbour is 1 of ~400 candidates
    x[:400] = 20.0 + rng.rand(400) * 0.8
    y[:400] = 12.0 + rng.rand(400) * 0.8
    x[400], y[400] = 18.35, 13.9
    x[401], y[401] = 18.40, 13.95
    ids = rng.permutation(10**5)[:n]
    mate = O.choose_mates(x, y, ids, r, 5, 3, dim=(W, H))
    nbs = O.neighbour_lists(x, y, r)
    has = np.array([len(v) > 0 for v in nbs])
    np.testing.assert_array_equal(mate >= 0, has)
    for i in np.nonzero(has)[0]:
        assert mate[i] in nbs[i]
    perm = rng.permutation(n)
    mate_p = O.choose_mates(x[perm], y[perm], ids[perm], r, 5, 3, dim=(W, H))
    np.testing.assert_array_equal(np.where(mate_p >= 0, ids[perm][np.maximum(mate_p, 0)], -1),
                                  np.where(mate >= 0, ids[np.maximum(mate, 0)], -1)[perm])
    focal = rng.rand(n) < 0.3
    mate_f = O.choose_mates(x, y, ids, r, 5, 3, dim=(W, H), focal=focal)
    np.testing.assert_array_equal(mate_f[focal], mate[focal])
    assert (mate_f[~focal] == -1).all()
    # different steps give different picks
    assert (O.choose_mates(x, y, ids, r, 5, 4, dim=(W, H)) != mate).mean() > 0.3


def test_spatial_tester_vs_reference():
    """G16: the reference's SpatialTester.update series (sim/burnin.py:44-59) over the
    first burn-in steps of two reference models"""
    g = load_golden('g16_spatial_tester')
    for s in (1, 2):
        dim = tuple(int(v) for v in g['s%i_dim' % s])
        off = np.concatenate([[0], np.cumsum(g['s%i_n' % s])])
        counts = np.zeros((dim[1], dim[0]))
        for t in range(len(off) - 1):
            x = g['s%i_x' % s][off[t]:off[t + 1]]
            y = g['s%i_y' % s][off[t]:off[t + 1]]
            counts, m, sd = O.spatial_diff_stats(counts, x, y, dim)
            assert abs(m - g['s%i_mean' % s][t]) < 1e-12
            assert abs(sd - g['s%i_std' % s][t]) < 1e-12
        np.testing.assert_array_equal(counts, g['s%i_counts' % s])


def test_inverse_distance_choice_follows_the_weights():
    """the build's inverse-distance mate choice (index sampling + acceptance u * r < r - d,
    exact weighted fallback) draws neighbour j of a focal individual with probability
    (r - d_j) / sum_k (r - d_k) (utils/spatial.py:222-227), never a coincident one"""
    rng = np.random.RandomState(2)
    r = 3.0
    # one focal individual (id 0) in the middle of a fixed ring of neighbours
    ang = np.linspace(0, 2 * np.pi, 13)[:-1]
    dist = np.array([0.3, 0.6, 0.9, 1.2, 1.5, 1.8, 2.1, 2.4, 2.7, 2.95, 3.2, 0.0])
    x = np.concatenate([[10.0], 10.0 + dist * np.cos(ang)]).astype(np.float32)
    y = np.concatenate([[10.0], 10.0 + dist * np.sin(ang)]).astype(np.float32)
    ids = np.arange(x.size)
    counts = np.zeros(x.size)
    T = 6000
    for step in range(T):
        m = O.choose_mates(x, y, ids, r, 11, step, mode='inverse', dim=(20, 20),
                           focal=np.arange(x.size) == 0)
        counts[m[0]] += 1
    dx, dy = x - x[0], y - y[0]
    d = np.sqrt(dx * dx + dy * dy)
    w = np.where((d > 0) & (d <= r), r - d, 0.0)
    w[0] = 0
    p = w / w.sum()
    assert counts[w == 0].sum() == 0                     # out of range / coincident: never
    assert np.abs(counts / T - p).max() < 4.5 * np.sqrt(p.max() / T)
    # the fallback path (few blocks, many rejections) agrees with the weights too
    xs = np.concatenate([[10.0], 10.0 + rng.rand(400) * 0.02 + 2.93]).astype(np.float32)
    ys = np.full(401, 10.0, np.float32)
    got = np.array([O.choose_mates(xs, ys, np.arange(401), r, 3, s_, mode='inverse',
                                   dim=(20, 20), focal=np.arange(401) == 0)[0]
                    for s_ in range(300)])
    assert (got > 0).all()
