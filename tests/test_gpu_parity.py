"""Parity of the HIP path (through the C-ABI, geonomics_amd/_native.py) with the
oracle and with golden vectors captured from the reference.  Needs an MI355X."""
import numpy as np
import pytest

import gnx_oracle as O
import gnx_draws as D
import philox as P
from conftest import load_golden

pytestmark = pytest.mark.gpu


def native():
    from geonomics_amd import _native
    return _native


def make_dev(W, H, rasts=None, L=0, n_traits=0, cap=4096, seed=11, **sp_kw):
    nat = native()
    if rasts is None:
        rasts = np.ones((1, H, W), np.float32)
    rasts = np.asarray(rasts, np.float32)
    dev = nat.Device(W, H, rasts.shape[0], L=L, n_traits=n_traits, cap_inds=cap,
                     cap_rows=cap, seed=seed)
    dev.upload_rasters(rasts)
    dev.set_species_params(nat.default_species_params(**sp_kw))
    return dev


def upload_simple(dev, x, y, ids=None, age=None, sex=None):
    n = len(x)
    ids = np.arange(n) if ids is None else ids
    age = np.zeros(n, np.int32) if age is None else age
    sex = np.zeros(n, np.uint8) if sex is None else sex
    dev.upload_population(x, y, age, sex, ids)


# ------------------------------------------------------------------ RNG
def test_device_draws_match_oracle_streams():
    nat = native()
    n = 5000
    for distr, p1, p2, mu, kappa in [('lognormal', 0.01, 0.5, 0.0, 0.0),
                                     ('wald', 1.5, 2.0, 1.0, 2.5),
                                     ('levy', 0.0, 0.3, -0.5, 12.0)]:
        dev = make_dev(64, 64, seed=123, cap=8192, move_distr=nat.DIST[distr], move_p1=p1,
                       move_p2=p2, dir_mu=mu, dir_kappa=kappa)
        rng = np.random.RandomState(1)
        ids = np.sort(rng.choice(10**6, n, replace=False))
        upload_simple(dev, rng.rand(n) * 64, rng.rand(n) * 64, ids=ids)
        dev.step_index = 17
        th, ds = dev.op_move_draws()
        th_o, ds_o = D.move_draws(123, ids, 17, distr, p1, p2, mu, kappa)
        # same uniform bits; logf/cosf/expf/acosf rounding differs by ulps
        assert np.abs(th - th_o).max() < 5e-4, distr
        if distr == 'levy':
            ok = ds_o < 1e4                 # heavy tail: 1/z^2 amplifies rounding
            assert np.abs(ds[ok] / ds_o[ok] - 1).max() < 2e-2
        else:
            assert np.abs(ds / ds_o - 1).max() < 1e-4, distr
        dev.close()


# ------------------------------------------------------------------ A10
@pytest.mark.parametrize('tag', ['sparse', 'free'])
def test_crossover_matches_reference_bit_exact(tag):
    g = load_golden('g1_crossover')
    pg = g[tag + '_parents_g']
    N, L, _ = pg.shape
    ids = g[tag + '_parent_ids']
    paths = g[tag + '_subsetters'][:, 1::2]
    nb = g[tag + '_n_births']
    keys = O.reference_key_layout(nb, g[tag + '_recomb_keys'])
    row_of = {int(i): k for k, i in enumerate(ids)}
    prow = np.array([[row_of[int(a)], row_of[int(b)]] for a, b in g[tag + '_pairs']])
    parent_slots = np.repeat(prow, nb, axis=0)
    B = parent_slots.shape[0]
    dev = make_dev(24, 24, L=L, cap=1024)
    rng = np.random.RandomState(0)
    upload_simple(dev, rng.rand(N) * 24, rng.rand(N) * 24, ids=ids)
    dev.upload_genomes(O.pack_genomes(pg))
    dev.set_recomb_paths(O.pack_bits(paths))
    dev.op_crossover(parent_slots, keys, g[tag + '_start_homs'])
    assert dev.N == N + B
    child = dev.download_genomes(np.arange(N, N + B))
    np.testing.assert_array_equal(O.unpack_genomes(child, L), g[tag + '_child_g'])
    # parents untouched; ids continue from max id
    par = dev.download_genomes(np.arange(N))
    np.testing.assert_array_equal(O.unpack_genomes(par, L), pg)
    idd = dev.download(native().F_ID)
    np.testing.assert_array_equal(idd[N:], ids.max() + 1 + np.arange(B))
    dev.close()


@pytest.mark.parametrize('L,n_paths,rate', [(1000, 64, 0.002), (1000, 64, 0.5),
                                            (4097, 33, 0.0007), (130, 5, 0.3),
                                            (64, 3, 0.0), (100000, 16, 1e-5)])
def test_crossover_matches_oracle_random(L, n_paths, rate):
    rng = np.random.RandomState(L + n_paths)
    N, B = 200, 700
    if L >= 100000:
        N, B = 60, 150
    g = (rng.rand(N, L, 2) < 0.5).astype(np.uint8)
    cross = (rng.rand(n_paths, L) < rate).astype(np.uint8)
    cross[:, 0] = 0
    paths = O.recomb_paths(cross)
    geno = O.pack_genomes(g)
    pk = O.pack_bits(paths)
    parent_slots = rng.randint(0, N, (B, 2))
    keys = rng.randint(0, n_paths, (B, 2))
    starts = rng.randint(0, 2, (B, 2))
    dev = make_dev(32, 32, L=L, cap=N + B + 8)
    upload_simple(dev, rng.rand(N) * 32, rng.rand(N) * 32)
    dev.upload_genomes(geno)
    dev.set_recomb_paths(pk)
    dev.op_crossover(parent_slots, keys, starts)
    child = dev.download_genomes(np.arange(N, N + B))
    ref = O.crossover(geno, pk, parent_slots, keys, starts)
    np.testing.assert_array_equal(child, ref)
    dev.close()


def test_crossover_edge_cases():
    # zero births is a no-op; bad inputs raise instead of faulting
    nat = native()
    L = 200
    rng = np.random.RandomState(3)
    g = (rng.rand(10, L, 2) < 0.5).astype(np.uint8)
    dev = make_dev(16, 16, L=L, cap=64)
    upload_simple(dev, rng.rand(10) * 16, rng.rand(10) * 16)
    dev.upload_genomes(O.pack_genomes(g))
    dev.set_recomb_paths(O.pack_bits(np.zeros((2, L), np.uint8)))
    dev.op_crossover(np.zeros((0, 2)), np.zeros((0, 2)), np.zeros((0, 2)))
    assert dev.N == 10
    with pytest.raises(nat.GnxError):
        dev.op_crossover([[0, 99]], [[0, 0]], [[0, 0]])
    with pytest.raises(nat.GnxError):
        dev.op_crossover([[0, 1]], [[0, 2]], [[0, 0]])
    with pytest.raises(nat.GnxError):       # capacity
        dev.op_crossover(np.zeros((60, 2)), np.zeros((60, 2)), np.zeros((60, 2)))
    # all-zero paths + start homologue s copy hom s of each parent
    dev.op_crossover([[2, 5], [7, 7]], [[0, 1], [1, 0]], [[0, 1], [1, 0]])
    ch = O.unpack_genomes(dev.download_genomes([10, 11]), L)
    np.testing.assert_array_equal(ch[0, :, 0], g[2, :, 0])
    np.testing.assert_array_equal(ch[0, :, 1], g[5, :, 1])
    np.testing.assert_array_equal(ch[1, :, 0], g[7, :, 1])
    np.testing.assert_array_equal(ch[1, :, 1], g[7, :, 0])
    dev.close()


# ------------------------------------------------------------------ A12/A15/A14
@pytest.mark.parametrize('tag', ['codom', 'dom'])
def test_phenotype_fitness_death_probs_vs_reference(tag):
    nat = native()
    g = load_golden('g3_phenotype_fitness')
    G = g[tag + '_g']
    N, L, _ = G.shape
    rasts = g[tag + '_rasts'].astype(np.float32)
    _, H, W = rasts.shape
    z_ref = g[tag + '_z']
    n_trt = z_ref.shape[1]
    dev = make_dev(W, H, rasts=rasts, L=L, n_traits=n_trt, cap=1024,
                   K_layer=0, K_factor=0.6)
    for t in range(n_trt):
        par = g['%s_t%i_par' % (tag, t)]
        dev.set_trait(t, g['%s_t%i_loci' % (tag, t)], g['%s_t%i_alpha' % (tag, t)],
                      int(par[0]), par[1], par[2], bool(par[3]))
    dev.set_dominance(g[tag + '_dom'])
    upload_simple(dev, g[tag + '_x'], g[tag + '_y'])
    dev.upload_genomes(O.pack_genomes(G))
    z = dev.download(nat.F_Z).T
    # tolerance: z is stored as f32 (reference f64)
    np.testing.assert_allclose(z, z_ref, rtol=0, atol=1.5e-7)
    e = dev.download(nat.F_E).T
    np.testing.assert_allclose(e, g[tag + '_e'], rtol=0, atol=6e-8)
    # death probabilities with an arbitrary density field: give the kernel a
    # node field, compute d per cell with the oracle from the same spline
    jx, jy = dev.lattice_dims()
    lat = O.DensityLattice((W, H))
    assert (jx, jy) == tuple(lat.J)
    rng = np.random.RandomState(5)
    VN = rng.rand(jy, jx) * 2.0
    VP = rng.rand(jy, jx) * 0.3
    p_dev, d_dev = dev.op_death_probs(True, VN, VP)
    x32 = g[tag + '_x'].astype(np.float32)
    y32 = g[tag + '_y'].astype(np.float32)
    cx, cy = x32.astype(int), y32.astype(int)
    Nr = O.spline_raster(lat, VN)
    Pr = O.spline_raster(lat, VP)
    K = rasts[0].astype(np.float64) * 0.6
    _, _, _, d_r = O.calc_d(Nr, K, Pr, 0.5, 0.2, 1, 0, 1)
    np.testing.assert_allclose(d_dev, d_r[cy, cx], rtol=1e-9, atol=1e-12)
    lyr, phi, gamma, ua = [], [], [], []
    for t in range(n_trt):
        par = g['%s_t%i_par' % (tag, t)]
        lyr.append(int(par[0])); phi.append(par[1]); gamma.append(par[2]); ua.append(bool(par[3]))
    w = O.fitness_traits(e.astype(np.float64), z.astype(np.float64), lyr, phi, gamma, ua)
    np.testing.assert_allclose(p_dev, O.prob_death(d_dev, w), rtol=1e-10, atol=1e-12)
    fit = dev.download(nat.F_FIT)
    # vs the reference's own fitness (f32 e and z => 1e-6)
    np.testing.assert_allclose(fit, g[tag + '_w'], rtol=2e-6)
    # and the reference's p_death for its own d values, via the same formula
    np.testing.assert_allclose(O.prob_death(g[tag + '_d_at'], fit.astype(np.float64)),
                               g[tag + '_p_death'], rtol=0, atol=2e-6)
    dev.close()


def test_deleterious_fitness_and_max_age():
    nat = native()
    g = load_golden('g3_phenotype_fitness')
    G = g['delet_g']
    N, L, _ = G.shape
    dev = make_dev(24, 24, L=L, cap=512, max_age=3)
    rng = np.random.RandomState(1)
    age = rng.randint(0, 7, N).astype(np.int32)
    upload_simple(dev, rng.rand(N) * 24, rng.rand(N) * 24, age=age)
    dev.upload_genomes(O.pack_genomes(G))
    dev.set_deleterious(g['delet_loci'], g['delet_s'])
    jx, jy = dev.lattice_dims()
    p, d = dev.op_death_probs(True, np.full((jy, jx), 0.3), None)
    fit = dev.download(nat.F_FIT)
    np.testing.assert_allclose(fit, g['delet_w'], rtol=2e-7)
    exp = np.where(age > 3, 1.0, 1 - (1 - d) * g['delet_w'])
    np.testing.assert_allclose(p, exp, rtol=1e-12)
    dev.close()


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_demography_algebra_vs_reference(tag):
    """d raster from the reference's own N, K, n_pairs rasters: feed node
    fields that reproduce constant-per-node values is not possible for an
    arbitrary raster, so check the per-cell algebra through a degenerate
    lattice: one individual per cell reads d at its cell."""
    nat = native()
    g = load_golden('g5_demography')
    R, b, lam, dmin, dmax = g[tag + '_par']
    # the algebra is per cell; evaluate it on a flat density field for every
    # distinct (N, K, n_pairs) combination the kernel can see: constant nodes
    H, W = 12, 12
    rng = np.random.RandomState(2)
    K = (rng.rand(H, W) * 2).astype(np.float32)
    K[0, :3] = 0
    dev = make_dev(W, H, rasts=K[None] / 2.0, K_factor=2.0, R=R, b=b,
                   n_births_lambda=lam, d_min=dmin, d_max=dmax, cap=512)
    xs, ys = np.meshgrid(np.arange(W) + 0.25, np.arange(H) + 0.75)
    upload_simple(dev, xs.ravel(), ys.ravel())
    jx, jy = dev.lattice_dims()
    for Nval, Pval in [(0.0, 0.0), (1.7, 0.2), (0.0, 0.3), (3.0, 0.0)]:
        p, d = dev.op_death_probs(False, np.full((jy, jx), Nval),
                                  np.full((jy, jx), Pval))
        Nr = np.full((H, W), Nval)
        Kr = (K / 2.0).astype(np.float32).astype(np.float64) * 2.0
        _, _, _, d_ref = O.calc_d(Nr, Kr, np.full((H, W), Pval), R, b, lam, dmin, dmax)
        np.testing.assert_allclose(d.reshape(H, W), d_ref, rtol=1e-12, atol=1e-15)
        np.testing.assert_array_equal(p, d)
    dev.close()


# ------------------------------------------------------------------ A2
@pytest.mark.parametrize('tag', ['lognormal', 'wald', 'levy'])
def test_movement_transform_vs_reference(tag):
    nat = native()
    g = load_golden('g7_movement')
    dim = g[tag + '_dim']
    W, H = int(dim[0]), int(dim[1])
    rasts = g['e_rasts'].astype(np.float32)
    dev = make_dev(W, H, rasts=rasts, cap=1024)
    x0, y0 = g[tag + '_x0'], g[tag + '_y0']
    upload_simple(dev, x0, y0)
    th = g[tag + '_theta'].astype(np.float32)
    ds = g[tag + '_dist'].astype(np.float32)
    dev.op_move(th, ds)
    x1, y1 = dev.download(nat.F_X), dev.download(nat.F_Y)
    # exact f32 restatement (oracle in f32 with the same inputs): cosf/sinf ulps
    ox, oy = O.move_transform(x0.astype(np.float32), y0.astype(np.float32), th, ds,
                              (W, H), dtype=np.float32)
    step = np.maximum(ds, 1.0)
    assert (np.abs(x1 - ox) <= 4e-7 * step + 4e-6).all()
    assert (np.abs(y1 - oy) <= 4e-7 * step + 4e-6).all()
    # vs the reference (f64): f32 positions => 2^-19 relative on [0, dim)
    fin = np.isfinite(g[tag + '_dist']) & (g[tag + '_dist'] < 1e3)
    assert np.abs(x1 - g[tag + '_x1'])[fin].max() < 2e-5 * max(W, 1) / 24 * 4
    assert np.abs(y1 - g[tag + '_y1'])[fin].max() < 2e-5 * max(H, 1) / 24 * 4
    assert (x1 >= 0).all() and (x1 <= np.float32(W - 0.001)).all()
    # environment re-sampled at the new cells (Species._set_e)
    e = dev.download(nat.F_E).T
    np.testing.assert_array_equal(e, O.gather_e(list(rasts), x1, y1))
    dev.close()


def test_dispersal_retry_vs_reference():
    g = load_golden('g7_movement')
    dim = g['disp_dim']
    dev = make_dev(int(dim[0]), int(dim[1]), cap=1024)
    A = 8
    th = g['disp_theta'][:A].astype(np.float32)
    ds = g['disp_dist'][:A].astype(np.float32)
    assert g['disp_used'].max() < A
    ox, oy, used = dev.op_dispersal(g['disp_mx'], g['disp_my'], th, ds)
    np.testing.assert_array_equal(used, g['disp_used'])
    assert np.abs(ox - g['disp_x']).max() < 2e-5
    assert np.abs(oy - g['disp_y']).max() < 2e-5
    assert (ox > 0).all() and (oy > 0).all()
    dev.close()


# ------------------------------------------------------------------ A13
@pytest.mark.parametrize('W,H,ww,kind', [(64, 48, -1, 'random'), (300, 200, 10.0, 'random'),
                                         (301, 187, 7.0, 'peaks'), (1024, 1024, -1, 'peaks'),
                                         (2048, 2048, -1, 'random'), (257, 2048, 40.0, 'flat'),
                                         (200, 200, 3.0, 'random')])
def test_density_maximum_equals_a_scan_of_every_cell(W, H, ww, kind, monkeypatch):
    """N.max() (the clip of _calc_dNdt, ops/demography.py:116): the kernels evaluate the density
    spline only at the cells that can hold a row's maximum between two lattice nodes (the first,
    the last, those next to a root of the cubic's derivative).  That is the SAME double as the scan
    of every cell (GNX_NMAX_SCAN_ALL=1) - and the oracle's raster maximum to rounding - for random
    fields, fields with sharp peaks and lattices as fine as the scan threshold (a constant field:
    to the last bits, the cells differ by their rounding only);
    through k_nmax here, through the step's fused kernel in the test below."""
    dev = make_dev(W, H, cap=256, window_width=ww, K_factor=1.0)
    upload_simple(dev, np.array([1.5, 3.0]), np.array([2.5, 1.0]))
    jx, jy = dev.lattice_dims()
    lat = O.DensityLattice((W, H), None if ww < 0 else ww)
    assert (jx, jy) == tuple(lat.J)
    rng = np.random.RandomState(W + H)
    for rep in range(4):
        if kind == 'random':
            V = rng.rand(jy, jx) * 3.0
        elif kind == 'flat':
            V = np.full((jy, jx), 0.75 * (rep + 1))
        else:
            V = np.full((jy, jx), 0.05)
            for _ in range(3):
                V[rng.randint(jy), rng.randint(jx)] = 5.0 + rng.rand()
        monkeypatch.setenv('GNX_NMAX_SCAN_ALL', '0')
        dev.op_death_probs(False, V, V * 0.1)
        fast = dev.density_nmax()
        monkeypatch.setenv('GNX_NMAX_SCAN_ALL', '1')
        dev.op_death_probs(False, V, V * 0.1)
        full = dev.density_nmax()
        if kind == 'flat':
            # a constant field: every cell's value is the constant up to rounding, and which cell's
            # rounding is the largest is all the two scans can differ in
            assert abs(fast - full) <= 4 * np.spacing(full), (rep, fast, full)
        else:
            assert fast == full, (rep, fast, full)
        exp = O.spline_raster(lat, V).max()
        assert abs(fast - exp) <= 1e-12 * exp, (rep, fast, exp)
    dev.close()


def test_step_density_maximum_equals_a_scan_of_every_cell(monkeypatch):
    """the same through gnx_step's fused lattice + N.max() kernel, on a population that clumps"""
    got = {}
    for scan_all in ('0', '1'):
        monkeypatch.setenv('GNX_NMAX_SCAN_ALL', scan_all)
        dev = make_dev(400, 300, cap=60000, seed=5, mating_radius=4.0, K_factor=0.2, window_width=12.0)
        dev.init_population(12000)
        vals = []
        for t in range(12):
            dev.step(True, False)
            vals.append(dev.density_nmax())
        got[scan_all] = (vals, dev.N)
        dev.close()
    assert got['0'] == got['1']
    assert min(got['0'][0]) > 0.0 and got['0'][1] > 5000


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_density_vs_oracle_and_reference(tag):
    g = load_golden('g4_density')
    dim = tuple(int(v) for v in g[tag + '_dim'])
    ww = float(g[tag + '_ww'][0])
    x = g[tag + '_x'].astype(np.float32)
    y = g[tag + '_y'].astype(np.float32)
    dev = make_dev(dim[0], dim[1], cap=16384, window_width=ww)
    lat = O.DensityLattice(dim, ww)
    assert dev.lattice_dims() == tuple(lat.J)
    nodes, rast = dev.op_density(x, y)
    V = lat.node_density(x, y)
    np.testing.assert_allclose(nodes, V, rtol=1e-13)            # integer counts / areas
    mine = O.spline_raster(lat, V)
    np.testing.assert_allclose(rast, mine, rtol=1e-9, atol=1e-10)
    # vs the reference's griddata raster: stated tolerance
    ref = np.clip(g[tag + '_dens'], 0, None)
    diff = np.abs(rast - ref)
    from test_oracle_golden import DENSITY_BOUNDS       # per fixture, measured x 1.15
    assert diff.mean() <= DENSITY_BOUNDS[tag][0] * ref.mean()
    assert diff.max() <= DENSITY_BOUNDS[tag][1] * ref.max()
    # empty input -> zero raster
    nodes0, rast0 = dev.op_density(np.zeros(0), np.zeros(0))
    assert (nodes0 == 0).all() and (rast0 == 0).all()
    dev.close()


# ------------------------------------------------------------------ A6/A7
def _slot_maps(dev, ids_before):
    """slot order after the device's cell sort -> original index"""
    ids_now = dev.download(native().F_ID)
    pos = {int(i): k for k, i in enumerate(ids_before)}
    return np.array([pos[int(i)] for i in ids_now])


# (mode, hash cells per mating radius: GNX_CELL_DIV=2 is the opt-in grid of half-radius cells
# with the 5 x 5 candidate block, csrc/gnx_api.hip: setup_hash_grid)
PAIR_CASES = [('uniform', 1), ('nearest', 1), ('inverse', 1), ('uniform', 2), ('inverse', 2)]


@pytest.mark.parametrize('mode,div', PAIR_CASES)
def test_find_pairs_vs_oracle(mode, div, monkeypatch):
    monkeypatch.setenv('GNX_CELL_DIV', str(div))
    nat = native()
    rng = np.random.RandomState(9)
    n, W, H, r = 3000, 80, 60, 2.5
    x = (rng.rand(n) * W).astype(np.float32)
    y = (rng.rand(n) * H).astype(np.float32)
    x[:40] = x[40:80]           # coincident individuals
    y[:40] = y[40:80]
    ids = np.sort(rng.choice(10**5, n, replace=False))
    mm = {'uniform': nat.MATE_UNIFORM, 'nearest': nat.MATE_NEAREST,
          'inverse': nat.MATE_INVERSE}[mode]
    dev = make_dev(W, H, cap=4096, seed=77, mating_radius=r, mate_mode=mm, b=0.4)
    upload_simple(dev, x, y, ids=ids)
    dev.step_index = 5
    keep = rng.rand(n) < 0.4
    mate, pairs = dev.op_find_pairs(keep)
    o = _slot_maps(dev, ids)                      # slot -> original index
    mate_o = np.where(mate >= 0, o[np.maximum(mate, 0)], -1)
    got = np.full(n, -2)
    got[o] = mate_o
    exp = O.choose_mates(x, y, ids, r, 77, 5, mode=mode, dim=(W, H))
    # the device searches only for individuals whose own Bernoulli(b) draw keeps
    # their pair (nobody else's mate is ever used); the others report -1
    assert (got[~keep] == -1).all()
    got, exp_k = got[keep], exp[keep]
    np.testing.assert_array_equal(got, exp_k)
    if mode != 'inverse':
        # the reference de-duplicates unordered pairs (set of frozensets,
        # ops/mating.py:63); which orientation survives is unspecified
        pr = O.pairs_from_mates(exp, keep)
        mine = [frozenset((int(o[a]), int(o[b]))) for a, b in pairs]
        assert len(mine) == len(set(mine))
        assert set(mine) == {frozenset((int(a), int(b))) for a, b in pr}
    dev.close()


@pytest.mark.parametrize('mode,div', PAIR_CASES)
def test_find_pairs_at_the_edge_of_a_clump(mode, div, monkeypatch):
    """The hard case of the index sampling: a focal individual with two in-radius
    neighbours among ~400 candidates (a dense blob in the next hash cell, out of reach).
    Most of its tries are rejected, so the block's waves take it over (a Philox block per
    lane) and many end in the exact scan; every one must still equal the oracle's walk of
    the same stream (utils/spatial.py:209-241)."""
    monkeypatch.setenv('GNX_CELL_DIV', str(div))
    nat = native()
    rng = np.random.RandomState(31)
    W, H, r = 80, 60, 2.5
    xs, ys = [], []
    centres = [(11.25, 11.25), (41.25, 31.25), (61.25, 13.75), (23.75, 46.25), (68.75, 48.75)]
    for cx, cy in centres:
        xs.append(cx + 0.15 * rng.randn(400))
        ys.append(cy + 0.15 * rng.randn(400))
        th = np.arange(12) * (2 * np.pi / 12) + rng.rand() * 0.3
        xs.append(cx + 3.3 * np.cos(th))
        ys.append(cy + 3.3 * np.sin(th))
    xs.append(rng.rand(300) * W)
    ys.append(rng.rand(300) * H)
    x = np.concatenate(xs).astype(np.float32)
    y = np.concatenate(ys).astype(np.float32)
    n = x.size
    ids = np.sort(rng.choice(10**5, n, replace=False))
    mm = {'uniform': nat.MATE_UNIFORM, 'nearest': nat.MATE_NEAREST,
          'inverse': nat.MATE_INVERSE}[mode]
    dev = make_dev(W, H, cap=4096, seed=123, mating_radius=r, mate_mode=mm, b=1.0)
    upload_simple(dev, x, y, ids=ids)
    dev.step_index = 2
    keep = np.ones(n, bool)
    mate, _ = dev.op_find_pairs(keep)
    o = _slot_maps(dev, ids)
    got = np.full(n, -2)
    got[o] = np.where(mate >= 0, o[np.maximum(mate, 0)], -1)
    exp = O.choose_mates(x, y, ids, r, 123, 2, mode=mode, dim=(W, H))
    np.testing.assert_array_equal(got, exp)
    ring = np.concatenate([np.arange(400, 412) + 412 * k for k in range(len(centres))])
    assert (exp[ring] >= 0).all()
    if mode != 'nearest':
        # the case is what it claims to be: a good share of the ring's individuals run out
        # of tries (about M/2 of them) and take the exact scan
        n_fb = O.mate_fallbacks(x, y, ids, r, 123, 2, mode=mode, dim=(W, H))[ring].sum()
        assert n_fb >= (8 if div == 1 else 4), n_fb
    dev.close()


def test_nearest_pairs_vs_reference_kdtree():
    nat = native()
    g = load_golden('g8_pairing')
    c = g['kd_coords']
    # dyadic coordinates make the f32 distance test exact: round to 2^-10
    x = (np.round(c[:, 0] * 1024) / 1024).astype(np.float32)
    y = (np.round(c[:, 1] * 1024) / 1024).astype(np.float32)
    r = float(g['kd_radius'][0])
    dev = make_dev(24, 24, cap=1024, mating_radius=r, mate_mode=nat.MATE_NEAREST)
    upload_simple(dev, x, y)
    mate, _ = dev.op_find_pairs(np.ones(len(x), np.uint8))
    o = _slot_maps(dev, np.arange(len(x)))
    got = np.full(len(x), -1)
    got[o] = np.where(mate >= 0, o[np.maximum(mate, 0)], -1)
    exp = O.choose_mates(x, y, np.arange(len(x)), r, 0, 0, mode='nearest',
                         dtype=np.float64)
    np.testing.assert_array_equal(got, exp)
    # the reference's KD-tree pairs on the unrounded coordinates agree except
    # where rounding to 2^-10 changed the nearest neighbour
    ref = g['kd_nearest_pairs']
    ref_m = np.full(len(x), -1)
    ref_m[ref[:, 0]] = ref[:, 1]
    assert (ref_m == got).mean() > 0.99
    dev.close()


@pytest.mark.parametrize('tag', ['sex', 'sex_age', 'asex_age'])
def test_pair_filters_sex_and_age(tag):
    nat = native()
    rng = np.random.RandomState(4)
    n = 2000
    x = (rng.rand(n) * 50).astype(np.float32)
    y = (rng.rand(n) * 50).astype(np.float32)
    sex = rng.randint(0, 2, n).astype(np.uint8)
    age = rng.randint(0, 5, n).astype(np.int32)
    sexed = tag.startswith('sex')
    ra = (1, 3) if tag == 'sex_age' else ((2, 2) if tag == 'asex_age' else (0, 0))
    dev = make_dev(50, 50, cap=4096, seed=3, mating_radius=2.0, sexed=int(sexed),
                   repro_age=ra)
    upload_simple(dev, x, y, age=age, sex=sex)
    keep = rng.rand(n) < 0.5
    mate, pairs = dev.op_find_pairs(keep)
    o = _slot_maps(dev, np.arange(n))
    exp_mate = O.choose_mates(x, y, np.arange(n), 2.0, 3, 0, dim=(50, 50))
    if sexed:
        pr = O.sexed_pairs(exp_mate, keep, sex)
        pr = O.repro_age_filter(pr, age, ra, True)
    else:
        has = (exp_mate >= 0) & keep
        has &= (age >= ra[0]) & (age[np.maximum(exp_mate, 0)] >= ra[0])
        pr = O.pairs_from_mates(np.where(has, exp_mate, -1), has)
    if sexed:       # orientation matters: female focal first
        mine = {(int(o[a]), int(o[b])) for a, b in pairs}
        assert mine == {(int(a), int(b)) for a, b in pr}
    else:
        mine = [frozenset((int(o[a]), int(o[b]))) for a, b in pairs]
        assert len(mine) == len(set(mine))
        assert set(mine) == {frozenset((int(a), int(b))) for a, b in pr}
    dev.close()


def test_panmixia_pairs():
    rng = np.random.RandomState(6)
    n = 5000
    ids = np.arange(n) + 100
    dev = make_dev(40, 40, cap=8192, seed=21, mating_radius=-1.0, b=0.3)
    upload_simple(dev, rng.rand(n) * 40, rng.rand(n) * 40, ids=ids)
    dev.step_index = 2
    mate, pairs = dev.op_find_pairs(None)
    ids_now = dev.download(native().F_ID)
    f, m = D.panmixia_draws(21, ids_now, 2, n)
    keep = D.keep_draws(21, ids_now, 2, 0.3)
    sel = keep & (f != m)
    exp = np.stack([f[sel], m[sel]], 1)
    # the pair list is in trial order = the canonical (hash cell, id) slot order of the
    # sorted population (the order offspring ids are handed out in)
    np.testing.assert_array_equal(pairs, exp)
    assert abs(len(pairs) - 0.3 * n) < 5 * np.sqrt(n * 0.3 * 0.7)
    dev.close()


# ------------------------------------------------------------------ A16 / G
def test_mortality_compaction_and_row_recycling():
    nat = native()
    rng = np.random.RandomState(8)
    N, L = 500, 300
    g = (rng.rand(N, L, 2) < 0.5).astype(np.uint8)
    dev = make_dev(32, 32, L=L, cap=640)
    x = (rng.rand(N) * 32).astype(np.float32)
    upload_simple(dev, x, rng.rand(N) * 32, ids=np.arange(N) * 3)
    dev.upload_genomes(O.pack_genomes(g))
    dev.set_recomb_paths(O.pack_bits(np.zeros((1, L), np.uint8)))
    dead = rng.rand(N) < 0.37
    dev.op_mortality(dead)
    n2 = dev.N
    assert n2 == (~dead).sum()
    np.testing.assert_array_equal(dev.download(nat.F_ID), (np.arange(N) * 3)[~dead])
    np.testing.assert_array_equal(dev.download(nat.F_X), x[~dead])
    G2 = O.unpack_genomes(dev.download(nat.F_GENO), L)
    np.testing.assert_array_equal(G2, g[~dead])
    # offspring reuse freed rows without clobbering survivors
    B = 640 - n2
    ps = rng.randint(0, n2, (B, 2))
    dev.op_crossover(ps, np.zeros((B, 2)), np.zeros((B, 2)))
    G3 = O.unpack_genomes(dev.download(nat.F_GENO), L)
    np.testing.assert_array_equal(G3[:n2], g[~dead])
    np.testing.assert_array_equal(G3[n2:, :, 0], g[~dead][ps[:, 0], :, 0])
    rows = dev.download(nat.F_GROW)
    assert len(set(rows.tolist())) == len(rows)
    # everyone dies -> extinct, not an error
    dev.op_mortality(np.ones(dev.N, np.uint8))
    assert dev.N == 0
    dev.close()


def test_starting_genomes_exact_counts_and_oracle():
    g = load_golden('g9_starting_genomes')
    N = int(g['N'][0])
    n = O.starting_mutation_counts(N, g['p'])
    L = len(n)
    dev = make_dev(24, 24, L=L, cap=256, seed=9)
    rng = np.random.RandomState(0)
    upload_simple(dev, rng.rand(N) * 24, rng.rand(N) * 24)
    dev.assign_genomes(n)
    G = dev.download(native().F_GENO)
    np.testing.assert_array_equal(G, O.starting_genomes(N, L, n, seed=9))
    np.testing.assert_array_equal(O.unpack_genomes(G, L).sum(axis=(0, 2)), n)
    dev.close()


def test_mutation_sets_bits():
    rng = np.random.RandomState(2)
    N, L = 50, 500
    dev = make_dev(16, 16, L=L, cap=64)
    upload_simple(dev, rng.rand(N) * 16, rng.rand(N) * 16)
    dev.upload_genomes(np.zeros((N, 2, dev.W64), np.uint64))
    slots = rng.randint(0, N, 20)
    loci = rng.choice(L, 20, replace=False)
    homs = rng.randint(0, 2, 20)
    dev.mutate(slots, loci, homs)
    G = O.unpack_genomes(dev.download(native().F_GENO), L)
    exp = np.zeros((N, L, 2), np.int8)
    exp[slots, loci, homs] = 1
    np.testing.assert_array_equal(G, exp)
    dev.close()


# ------------------------------------------------------------------ whole step
def test_step_invariants_and_reproducibility():
    nat = native()

    def run(seed):
        W = H = 64
        rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))])
        L = 400
        dev = make_dev(W, H, rasts=rasts, L=L, n_traits=1, cap=8192, seed=seed,
                       mating_radius=3.0, K_factor=0.5)
        dev.set_trait(0, [5, 77, 200, 333], [0.1, -0.1, 0.1, -0.1], 1, 0.05, 1.0, False)
        dev.init_population(2000)
        for _ in range(12):
            dev.step(True, False)
        n0 = dev.N
        dev.assign_genomes(np.full(L, n0))        # p = 0.5
        rng = np.random.RandomState(1)
        paths = O.recomb_paths((rng.rand(50, L) < 0.01).astype(np.uint8) *
                               (np.arange(L) > 0))
        dev.set_recomb_paths(O.pack_bits(paths))
        hist = []
        for _ in range(15):
            dev.step(False, True)
            hist.append(dev.counts())
        ids = dev.download(nat.F_ID)
        x = dev.download(nat.F_X)
        G = dev.download(nat.F_GENO)
        dev.close()
        return hist, ids, x, G

    h1, ids1, x1, G1 = run(5)
    h2, ids2, x2, G2 = run(5)
    h3, ids3, _, _ = run(6)
    assert h1 == h2
    np.testing.assert_array_equal(ids1, ids2)
    np.testing.assert_array_equal(x1, x2)
    np.testing.assert_array_equal(G1, G2)
    assert h1 != h3
    assert len(set(ids1.tolist())) == len(ids1)
    Ns = np.array([h[0] for h in h1])
    # logistic regulation towards sum(K) = 0.5 * 64 * 64 = 2048
    assert 1200 < Ns.mean() < 3000
    assert all(h[1] > 0 and h[2] > 0 for h in h1)


def test_whole_model_envelopes_vs_reference_on_device():
    """Same check as tests/test_oracle_golden.py::test_whole_model_envelopes_vs_reference
    with the HIP path doing the stepping."""
    nat = native()
    g = load_golden('g10_envelopes')
    W = H = 30
    L = 60
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))])
    rng = np.random.RandomState(0)
    paths = O.pack_bits(O.recomb_paths((rng.rand(60, L) < 0.5).astype(np.uint8)
                                       * (np.arange(L) > 0)))
    ref = dict(burn=[], first=[], main=[])
    mine = dict(burn=[], first=[], main=[])
    drift_ref, drift_mine = [], []
    n_seeds = int(g['n_seeds'][0])
    # two device runs per reference run (the device's side of every ratio then carries
    # half the sampling variance of the reference's 24 runs)
    for s, seed0 in [(s, seed0) for seed0 in (200, 700) for s in range(1, n_seeds + 1)]:
        nb = int(g['s%i_nburn' % s][0])
        R = g['s%i_Nt' % s]
        if seed0 == 200:
            ref['burn'].append(R[10:nb].mean())
            ref['first'].append(R[nb:nb + 20].mean())
            ref['main'].append(R[-50:].mean())
        dev = make_dev(W, H, rasts=rasts, L=L, n_traits=3, cap=4096, seed=seed0 + s,
                       mating_radius=4.0, K_factor=0.5)
        for t in range(3):
            par = g['s%i_t%i_par' % (s, t)]
            dev.set_trait(t, g['s%i_t%i_loci' % (s, t)], g['s%i_t%i_alpha' % (s, t)],
                          int(par[0]), par[1], par[2], bool(par[3]))
        dev.set_recomb_paths(paths)
        dev.init_population(300)
        Nt = []
        for _ in range(60):
            dev.step(True, False)
            Nt.append(dev.N)
        dev.assign_genomes(O.starting_mutation_counts(dev.N, np.full(L, 0.5)))
        for _ in range(100):
            dev.step(False, True)
            Nt.append(dev.N)
        Nt = np.array(Nt)
        mine['burn'].append(Nt[10:60].mean())
        mine['first'].append(Nt[60:80].mean())
        mine['main'].append(Nt[-50:].mean())
        # genetic drift: allele frequencies of the neutral loci after 100 generations
        # (start 0.5; the spread measures 1/(2 Ne), i.e. mate choice, birth and death
        # variances and recombination all at once)
        sel = np.concatenate([g['s%i_t%i_loci' % (s, t)] for t in range(3)])
        neutral = np.setdiff1d(np.arange(L), sel)
        c1, _ = dev.stats_locus_counts()
        drift_mine.append((c1 / (2.0 * dev.N))[neutral] - 0.5)
        if seed0 == 200:
            drift_ref.append(g['s%i_freq' % s][neutral] - 0.5)
        dev.close()
    v_ref = np.mean(np.concatenate(drift_ref) ** 2)
    v_mine = np.mean(np.concatenate(drift_mine) ** 2)
    # 24 reference runs, twice as many device runs: the runs are the samples (a run's loci
    # drift together) and the ratio carries ~8 % sampling error (bootstrap over runs,
    # tests/test_oracle_golden.py); measured 0.97, the numpy oracle 0.90 over its 48 runs
    m = {k: (np.mean(ref[k]), np.mean(mine[k])) for k in ref}
    print('drift variance ratio %.3f' % (v_mine / v_ref),
          {k: round(v[1] / v[0] - 1, 4) for k, v in m.items()})
    assert 0.85 < v_mine / v_ref < 1.18, (v_mine, v_ref)
    assert abs(m['burn'][1] / m['burn'][0] - 1) < 0.03, m
    assert abs(m['first'][1] / m['first'][0] - 1) < 0.06, m
    assert abs(m['main'][1] / m['main'][0] - 1) < 0.06, m


@pytest.mark.parametrize('div', [1, 2])
def test_device_step_matches_oracle_step_counts(div, monkeypatch):
    """The oracle's whole step uses the device's random streams, so the two populations go
    through the same integer decisions: births, deaths and the set of living ids must agree
    EXACTLY, step after step, until a decision sits on a rounding tie - the device's
    logf / cosf / f32 expressions differ from numpy's in the last bits.  The first step
    that differs must show such a tie, and the test names it: a death draw u within 1e-5
    of its probability p, an individual whose two positions fall into different hash or
    raster cells, or a pair of individuals within 1e-4 of the mating radius.  After the
    first flip the two runs are different populations and only their statistics are
    compared."""
    import gnx_step as S
    monkeypatch.setenv('GNX_CELL_DIV', str(div))      # (2: hash cells of half a mating radius)
    nat = native()
    W = H = 40
    L = 128
    radius = 3.0
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)
    rng = np.random.RandomState(4)
    paths = O.pack_bits(O.recomb_paths((rng.rand(32, L) < 0.02).astype(np.uint8)
                                       * (np.arange(L) > 0)))
    loci = np.array([3, 40, 77, 101])
    alpha = np.array([0.1, -0.1, 0.1, -0.1])
    seed = 31
    dev = make_dev(W, H, rasts=rasts, L=L, n_traits=1, cap=8192, seed=seed,
                   mating_radius=radius, K_factor=0.6)
    dev.set_trait(0, loci, alpha, 1, 0.05, 1.0, False)
    dev.set_recomb_paths(paths)
    dev.init_population(900)
    st = S.State(rasts, S.Params(mating_radius=radius, K_factor=0.6), seed, L=L,
                 traits=[dict(loci=loci, alpha=alpha, layer=1, phi=0.05, gamma=1.0,
                              univ_adv=False)], paths_packed=paths)
    st.init_population(900)
    np.testing.assert_array_equal(dev.download(nat.F_X), st.x)

    def by_id(ids, *arrs):
        o = np.argsort(ids)
        return (ids[o],) + tuple(a[o] for a in arrs)

    def witness(xd, yd, xo, yo, ids, death):
        """a rounding tie that explains a differing decision of this step, or None"""
        cs = radius * (1.0 + 1e-9) / div
        for name, ad, ao in (('hash cell x', xd / cs, xo / cs), ('hash cell y', yd / cs, yo / cs),
                             ('raster cell x', xd, xo), ('raster cell y', yd, yo)):
            k = np.nonzero(np.floor(ad) != np.floor(ao))[0]
            if k.size:
                return 'id %d: %s, device %r oracle %r' % (ids[k[0]], name, ad[k[0]], ao[k[0]])
        d = np.hypot(xo[:, None] - xo[None, :], yo[:, None] - yo[None, :])
        i, j = np.nonzero(np.abs(d - radius) < 1e-4)
        if i.size:
            return 'ids %d, %d: distance %r against radius %r' % (ids[i[0]], ids[j[0]],
                                                                  d[i[0], j[0]], radius)
        m = np.abs(death['u'] - death['p'])
        k = int(np.argmin(m))
        if m[k] < 1e-5:
            return 'id %d: death draw u = %r against p = %r' % (death['ids'][k], death['u'][k],
                                                              death['p'][k])
        return None

    synced, exact_steps, first_flip = True, 0, None
    for t in range(20):
        burn = t < 10
        if t == 10:
            dev.assign_genomes(O.starting_mutation_counts(dev.N, np.full(L, 0.5)))
            st.assign_genomes(O.starting_mutation_counts(st.N, np.full(L, 0.5)))
        dev.age()
        dev.move()
        st.Nt.append(st.N)
        S.move(st, inc_age=True)
        if synced:
            ids_d, xd, yd = by_id(dev.download(nat.F_ID), dev.download(nat.F_X),
                                  dev.download(nat.F_Y))
            ids_o, xo, yo = by_id(st.id.copy(), st.x.copy(), st.y.copy())
            np.testing.assert_array_equal(ids_d, ids_o)
            assert max(np.abs(xd - xo).max(), np.abs(yd - yo).max()) < 2e-4     # f32 cosf/logf
        dev.pop_dynamics(burn, not burn)
        dev.step_index = dev.step_index + 1
        _, B, Dth = S.pop_dynamics(st, burn, not burn)
        st.step += 1
        n_dev, b_dev, d_dev = dev.counts()
        if synced:
            same = (b_dev == B and d_dev == Dth and
                    np.array_equal(np.sort(dev.download(nat.F_ID)), np.sort(st.id)))
            if same:
                exact_steps += 1
            else:
                w = witness(xd.astype(np.float64), yd.astype(np.float64), xo.astype(np.float64),
                            yo.astype(np.float64), ids_o, st.last_death)
                assert w is not None, (
                    'step %d: births %d / %d, deaths %d / %d differ and no decision of the '
                    'step sits on a rounding tie' % (t, b_dev, B, d_dev, Dth))
                first_flip = (t, w)
                synced = False
        else:
            # two different (equally distributed) populations from here on
            assert abs(b_dev - B) <= max(6, 0.2 * B), (t, b_dev, B, first_flip)
            assert abs(d_dev - Dth) <= max(6, 0.2 * Dth), (t, d_dev, Dth, first_flip)
            assert abs(n_dev - st.N) <= max(8, 0.08 * st.N), (t, n_dev, st.N, first_flip)
    assert exact_steps >= 3, first_flip   # the first steps agree exactly before a tie turns up
    print('exact steps:', exact_steps, 'first flip:', first_flip)
    dev.close()


# ------------------------------------------------------------------ A3 on device
@pytest.mark.parametrize('mixture', [True, False])
def test_conductance_directions_vs_reference_and_oracle(mixture):
    """Directions drawn from a conductance surface: (i) the device draws equal
    the oracle's restatement of the same stream (LDS-tiled and fallback gather
    paths), (ii) their circular moments per cell match the reference's LUT."""
    nat = native()
    g = load_golden('g11_conductance')
    rast = g['rast'].astype(np.float32)
    H, W = rast.shape
    per = 1500
    cy, cx = np.mgrid[0:H, 0:W]
    x = (np.repeat(cx.ravel(), per) + 0.5).astype(np.float32)
    y = (np.repeat(cy.ravel(), per) + 0.5).astype(np.float32)
    n = x.size
    ids = np.arange(n) + 7
    dev = make_dev(W, H, rasts=rast[None], cap=n + 64, seed=5,
                   move_surf=nat.SURF_MIXTURE if mixture else nat.SURF_UNIMODAL,
                   move_surf_layer=0, move_surf_kappa=12.0)
    upload_simple(dev, x, y, ids=ids)
    dev.step_index = 3
    th, _ = dev.op_move_draws()
    tho = D.surf_directions(5, ids, 3, P.OP_MOVE_SURF, rast, x, y, mixture, 12.0)
    d = np.abs(th - tho)
    d = np.minimum(d, 2 * np.pi - d)
    assert (d < 2e-3).mean() > 0.9995       # same picks; acosf/logf ulps only
    tag = 'mix' if mixture else 'uni'
    tol = 4.5 / np.sqrt(g['approx_len'][0]) + 4.5 / np.sqrt(per)
    mc = np.cos(th).reshape(H * W, per).mean(1)
    ms = np.sin(th).reshape(H * W, per).mean(1)
    assert np.abs(mc - g[tag + '_mean_cos'].ravel()).max() < tol
    assert np.abs(ms - g[tag + '_mean_sin'].ravel()).max() < tol
    dev.close()


def test_conductance_gather_lds_and_fallback_agree():
    """A block whose individuals are spread over a box larger than the LDS tile
    takes the global-gather fallback; both must give the oracle's draws."""
    nat = native()
    rng = np.random.RandomState(3)
    W = H = 160
    rast = rng.rand(H, W).astype(np.float32)
    n = 4096
    for spread in (False, True):
        if spread:
            x = (rng.rand(n) * W).astype(np.float32)      # box = whole landscape > tile
            y = (rng.rand(n) * H).astype(np.float32)
        else:
            x = (20 + rng.rand(n) * 30).astype(np.float32)  # box ~ 32 x 32 cells
            y = (40 + rng.rand(n) * 30).astype(np.float32)
        dev = make_dev(W, H, rasts=rast[None], cap=n + 8, seed=9, move_surf=nat.SURF_MIXTURE,
                       move_surf_layer=0, move_surf_kappa=6.0)
        upload_simple(dev, x, y)
        th, _ = dev.op_move_draws()
        tho = D.surf_directions(9, np.arange(n), 0, P.OP_MOVE_SURF, rast, x, y, True, 6.0)
        d = np.abs(th - tho)
        d = np.minimum(d, 2 * np.pi - d)
        assert (d < 2e-3).mean() > 0.9995, spread
        dev.close()


# ------------------------------------------------------------------ statistics
def test_stats_vs_reference_golden():
    """het / maf / ld of the reference (sim/stats.py:359-421) on its genotypes"""
    d = load_golden('g12_stats')
    g = d['g']
    N, L, _ = g.shape
    dev = make_dev(16, 16, L=L, cap=128)
    rng = np.random.RandomState(0)
    upload_simple(dev, rng.rand(N) * 16, rng.rand(N) * 16)
    dev.upload_genomes(O.pack_genomes(g, dev.W64))
    c1, ch = dev.stats_locus_counts()
    np.testing.assert_array_equal(ch / N, d['het'])
    f1 = c1 / (2 * N)
    np.testing.assert_array_equal(np.where(f1 > 0.5, 1 - f1, f1), d['maf'])
    np.testing.assert_allclose(dev.stats_ld(np.arange(L)), d['ld'], rtol=1e-10, atol=1e-14,
                               equal_nan=True)
    dev.close()


def test_stats_vs_oracle_after_births_and_deaths():
    """counts follow the genome-row indirection: run real steps (rows recycled,
    slots permuted), then compare with the oracle on the downloaded genotypes;
    L not a multiple of 64 and a monomorphic locus (r2 not finite, as numpy's)."""
    nat = native()
    rng = np.random.RandomState(3)
    W = H = 48
    L = 777
    dev = make_dev(W, H, L=L, cap=8192, seed=5, mating_radius=3.0, K_factor=0.5)
    dev.init_population(1500)
    for _ in range(6):
        dev.step(True, False)
    n0 = dev.N
    n1 = rng.randint(0, 2 * n0 + 1, L)
    n1[5] = 0
    dev.assign_genomes(n1)
    paths = O.recomb_paths((rng.rand(50, L) < 0.01).astype(np.uint8) * (np.arange(L) > 0))
    dev.set_recomb_paths(O.pack_bits(paths))
    for _ in range(5):
        dev.step(False, False)
    Nn = dev.N
    assert Nn > 0
    g = O.unpack_genomes(dev.download(nat.F_GENO), L)
    assert g.shape[0] == Nn
    c1, ch = dev.stats_locus_counts()
    np.testing.assert_array_equal(ch / Nn, O.stats_het(g))
    f1 = c1 / (2 * Nn)
    np.testing.assert_array_equal(np.where(f1 > 0.5, 1 - f1, f1), O.stats_maf(g))
    loci = np.sort(rng.choice(L, 200, replace=False))
    loci[0] = 5
    r2 = dev.stats_ld(loci)
    exp = O.stats_ld(g[:, loci, :])
    fin = np.isfinite(exp)
    assert not fin[0, 1]
    np.testing.assert_array_equal(np.isfinite(r2), fin)
    np.testing.assert_allclose(r2[fin], exp[fin], rtol=1e-9, atol=1e-13)
    dev.close()
