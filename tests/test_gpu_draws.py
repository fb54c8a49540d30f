"""One parity test per random-draw kernel of the HIP path (csrc/gnx_rng.h op codes):
what the device decides from its Philox streams equals what the oracle's restatement
of the same streams (oracle/gnx_draws.py) decides - exact for integer decisions,
ulp-bounded for floats.  OP_MOVE_* / OP_MOVE_SURF / OP_MATE_PICK are covered in
test_gpu_parity.py; here: OP_BIRTHS (Poisson, ops/mating.py:120-126), OP_OFFSPRING
(start homologues, path keys, sex), OP_DISPERSAL, OP_DEATH, OP_PAIR_KEEP, and the
burn-in spatial tester against the reference's own series (sim/burnin.py:44-59).
Everything goes through the C-ABI.  Needs an MI355X."""
import numpy as np
import pytest

import gnx_oracle as O
import gnx_draws as D
import philox as P
from conftest import load_golden
from test_gpu_parity import make_dev, upload_simple, native

pytestmark = pytest.mark.gpu


def _population(rng, n, W, H, id0=1000):
    ids = np.sort(rng.choice(10 ** 6, n, replace=False)) + id0
    x = (rng.rand(n) * W).astype(np.float32)
    y = (rng.rand(n) * H).astype(np.float32)
    return x, y, ids


@pytest.mark.parametrize('lam', [0.4, 1.7, 4.0])
def test_poisson_births_match_oracle(lam):
    """k_births: max(Poisson(lambda), 1) per pair from the focal parent's stream equals
    O.poisson_knuth on the same uniforms, pair by pair; offspring ids ascend in
    (pair, birth) order (structs/species.py:614-648), pairs in the canonical
    (hash cell, id) order of their focal individual"""
    rng = np.random.RandomState(int(lam * 10))
    W = H = 48
    seed, step = 77, 9
    n = 4000
    x, y, ids = _population(rng, n, W, H)
    dev = make_dev(W, H, cap=4 * n, seed=seed, mating_radius=3.0, n_births_fixed=0,
                   n_births_lambda=lam, K_factor=2.0)
    upload_simple(dev, x, y, ids=ids)
    dev.step_index = step
    dev.pop_dynamics_mate(True)
    child, par, _, _, _ = dev.last_births(with_gametes=False)
    B = child.size
    assert B > 300
    focal = par[:, 0]
    pos = {int(i): k for k, i in enumerate(ids)}
    who = np.array([pos[int(f)] for f in focal])
    key = O.pair_order_keys(x[who], y[who], focal, (W, H), 3.0)
    assert (np.diff(key) >= 0).all()                         # canonical pair order
    np.testing.assert_array_equal(child, ids.max() + 1 + np.arange(B))
    uf, counts = np.unique(focal, return_counts=True)
    exp = D.births_draws(seed, uf, step, lam)
    np.testing.assert_array_equal(counts, exp)
    # every child of a pair has the same mate
    for f in uf[:50]:
        assert np.unique(par[focal == f, 1]).size == 1
    dev.close()


@pytest.mark.parametrize('sexed', [0, 1])
def test_offspring_and_dispersal_draws_match_oracle(sexed):
    """k_offspring: start homologues, recombination keys (OP_OFFSPRING block 0), sex
    (block 1) exact; position = dispersal (OP_DISPERSAL, retry loop of
    ops/movement.py:98-141) from the parents' midpoint, f32"""
    nat = native()
    rng = np.random.RandomState(5 + sexed)
    W = H = 40
    L, n_paths = 256, 37
    seed, step = 2024, 4
    n = 3000
    x, y, ids = _population(rng, n, W, H)
    # a band of parents hugs the low edges so that the dispersal retry loop is exercised
    x[:300] = rng.rand(300).astype(np.float32) * 0.05
    sex = rng.randint(0, 2, n).astype(np.uint8)
    dev = make_dev(W, H, L=L, cap=4 * n, seed=seed, mating_radius=3.0, sexed=sexed,
                   p_male=0.3, disp_p1=-1.0, disp_p2=0.6, K_factor=2.0)
    upload_simple(dev, x, y, ids=ids, sex=sex)
    g = (rng.rand(n, L, 2) < 0.5).astype(np.uint8)
    dev.upload_genomes(O.pack_genomes(g))
    dev.set_recomb_paths(O.pack_bits(O.recomb_paths(
        (rng.rand(n_paths, L) < 0.02).astype(np.uint8) * (np.arange(L) > 0))))
    dev.step_index = step
    dev.pop_dynamics_mate(False)
    child, par, keys, starts, xy = dev.last_births()
    B = child.size
    assert B > 150
    st_o, keys_o, sex_o = D.offspring_draws(seed, child, step, n_paths, sexed, 0.3)
    np.testing.assert_array_equal(starts, st_o)
    np.testing.assert_array_equal(keys, keys_o)
    id_all = dev.download(nat.F_ID)
    sex_all = dev.download(nat.F_SEX)
    x_all = dev.download(nat.F_X)
    y_all = dev.download(nat.F_Y)
    order = np.argsort(id_all)
    pos = order[np.searchsorted(id_all[order], child)]
    np.testing.assert_array_equal(sex_all[pos], sex_o)
    if sexed:                      # female focal, male mate (ops/mating.py:41-55)
        pf = order[np.searchsorted(id_all[order], par[:, 0])]
        pm = order[np.searchsorted(id_all[order], par[:, 1])]
        assert (sex_all[pf] == 0).all() and (sex_all[pm] == 1).all()
    # positions: midpoint of the parents, then the dispersal draws
    ppos = {int(i): (a, b) for i, a, b in zip(ids, x, y)}
    mx = np.array([(ppos[int(a)][0] + ppos[int(b)][0]) / np.float32(2) for a, b in par],
                  np.float32)
    my = np.array([(ppos[int(a)][1] + ppos[int(b)][1]) / np.float32(2) for a, b in par],
                  np.float32)
    th, ds = D.dispersal_draws(seed, child, step, 'lognormal', -1.0, 0.6)
    ox, oy, used = O.dispersal(mx, my, th, ds, (W, H), dtype=np.float32)
    assert (used > 0).sum() > 3                              # retries happened
    np.testing.assert_allclose(xy[:, 0], ox, atol=2e-4)
    np.testing.assert_allclose(xy[:, 1], oy, atol=2e-4)
    np.testing.assert_array_equal(x_all[pos], xy[:, 0])
    dev.close()


@pytest.mark.parametrize('distr,p1,p2', [('lognormal', -1.0, 0.6), ('wald', 1.5, 2.0),
                                         ('levy', 0.0, 0.3)])
@pytest.mark.parametrize('surf', ['none', 'mixture', 'unimodal'])
def test_dispersal_surface_and_distance_distributions(surf, distr, p1, p2):
    """A11, the branches of _do_dispersal nobody had run (ops/movement.py:98-141): the
    offspring's direction drawn from a dispersal surface at the parents' midpoint
    (spp._disp_surf, :104-108; utils/spatial.py:182-184 - stream OP_DISP_SURF, eight blocks
    per attempt) instead of the uniform angle, and wald / levy dispersal distances
    (:110-118) beside lognormal.  Device positions == the oracle's restatement of the same
    streams through O.dispersal (retry loop included)."""
    nat = native()
    rng = np.random.RandomState(31 + len(distr))
    W = H = 40
    seed, step = 606, 7
    n = 3000
    x, y, ids = _population(rng, n, W, H)
    x[:400] = rng.rand(400).astype(np.float32) * 0.05      # parents on the low edge: retries
    rast = rng.rand(H, W).astype(np.float32)
    rast[rng.rand(H, W) < 0.15] = 0.0                       # zero-conductance cells too
    rasts = np.stack([np.ones((H, W), np.float32), rast])
    kw = {}
    if surf != 'none':
        kw = dict(disp_surf=nat.SURF_MIXTURE if surf == 'mixture' else nat.SURF_UNIMODAL,
                  disp_surf_layer=1, disp_surf_kappa=9.0)
    dev = make_dev(W, H, rasts=rasts, cap=4 * n, seed=seed, mating_radius=3.0,
                   disp_distr=nat.DIST[distr], disp_p1=p1, disp_p2=p2, K_factor=2.0, **kw)
    upload_simple(dev, x, y, ids=ids)
    dev.step_index = step
    dev.pop_dynamics_mate(True)
    child, par, _, _, xy = dev.last_births(with_gametes=False)
    B = child.size
    assert B > 150
    ppos = {int(i): (a, b) for i, a, b in zip(ids, x, y)}
    mx = np.array([(ppos[int(a)][0] + ppos[int(b)][0]) / np.float32(2) for a, b in par],
                  np.float32)
    my = np.array([(ppos[int(a)][1] + ppos[int(b)][1]) / np.float32(2) for a, b in par],
                  np.float32)
    th, ds = D.dispersal_draws(seed, child, step, distr, p1, p2)
    if surf != 'none':
        th = np.stack([D.surf_directions(seed, child, step, P.OP_DISP_SURF, rast, mx, my,
                                         surf == 'mixture', 9.0, first_blk=8 * a)
                       for a in range(th.shape[0])])
    ox, oy, used = O.dispersal(mx, my, th, ds, (W, H), dtype=np.float32)
    assert (used > 0).sum() > 3                              # the retry loop ran
    # Every offspring, no allowance: 2^-17 x the landscape's width (a few f32 ulps of a
    # coordinate, libm against numpy) up to distances of 5 cells; beyond, the draw's own
    # conditioning - a distance is a function of the normal deviate z, whose Box-Muller
    # evaluation differs by dz ~ 2^-21 between the two libraries: lognormal d(dist) = p2 dist dz,
    # wald d(dist) <= ~2 dist / |z| dz, levy dist = p1 + p2 / z^2, d(dist) = 2 (dist - p1) / |z| dz
    # with |z| = sqrt(p2 / (dist - p1)).
    du = ds[used, np.arange(B)].astype(np.float64)
    ok = du < 5.0
    assert ok.mean() > 0.5
    err = np.maximum(np.abs(xy[:, 0] - ox), np.abs(xy[:, 1] - oy))
    dz = 2.0 ** -21
    if distr == 'levy':
        cond = 2.0 * (du - p1) / np.sqrt(p2 / np.maximum(du - p1, 1e-30)) * dz
    elif distr == 'wald':
        cond = 2.0 * du * dz / 0.05           # (|z| >= 0.05 for all but a per-mille of the draws)
    else:
        cond = p2 * du * dz
    tol = 2.0 ** -17 * W + np.where(ok, 0.0, cond)
    srt = np.argsort(-(err - tol))
    assert (err <= tol).all(), (surf, distr, int((err > tol).sum()), B,
                                [(float(err[i]), float(du[i]), float(tol[i])) for i in srt[:8]])
    assert (xy >= 0).all() and (xy[:, 0] <= W - 0.001 + 1e-4).all()
    if surf != 'none':
        # and the surface matters: not the uniform angles of the plain branch
        th_u, _ = D.dispersal_draws(seed, child, step, distr, p1, p2)
        ux, uy, _ = O.dispersal(mx, my, th_u, ds, (W, H), dtype=np.float32)
        assert (np.abs(xy[:, 0] - ux)[ok] > 1e-2).mean() > 0.5
    dev.close()


def test_death_draws_match_oracle():
    """k_alive: dead = u(id, step, OP_DEATH) < p_death.  With d_min = d_max the death
    probability is the same constant for everybody (ops/demography.py:158-164, no
    selection), so the survivors are exactly the ids whose draw is >= d"""
    nat = native()
    rng = np.random.RandomState(8)
    W = H = 32
    seed, step, d = 99, 12, 0.3
    n = 6000
    x, y, ids = _population(rng, n, W, H)
    dev = make_dev(W, H, cap=2 * n, seed=seed, mating_radius=2.0, d_min=d, d_max=d)
    upload_simple(dev, x, y, ids=ids)
    dev.step_index = step
    dev.pop_dynamics_mate(True)
    id_before = dev.download(nat.F_ID)
    dev.pop_dynamics_die(True, False)
    id_after = dev.download(nat.F_ID)
    u = D.death_draws(seed, id_before, step)
    exp = np.sort(id_before[~(u.astype(np.float64) < d)])
    np.testing.assert_array_equal(np.sort(id_after), exp)
    assert dev.counts()[2] == id_before.size - exp.size
    dev.close()


def test_pair_keep_draws_match_oracle():
    """Bernoulli(b) thinning of the pairs (structs/species.py:2210-2214) is the focal
    individual's own OP_PAIR_KEEP draw: the focal ids of the device's pair list are
    exactly the kept individuals that have a neighbour, minus the reciprocal duplicates
    (ops/mating.py:62-65: of (i,m) and (m,i) the one with the smaller focal id stays)"""
    nat = native()
    rng = np.random.RandomState(21)
    W = H = 30
    seed, step, b = 5150, 3, 0.35
    n = 2500
    x, y, ids = _population(rng, n, W, H)
    dev = make_dev(W, H, cap=2 * n, seed=seed, mating_radius=2.5, b=b)
    upload_simple(dev, x, y, ids=ids)
    dev.step_index = step
    mate, pairs = dev.op_find_pairs()
    id_s = dev.download(nat.F_ID)                            # slot order after the cell sort
    keep = D.keep_draws(seed, id_s, step, b)
    assert ((mate >= 0) <= keep).all()                       # only kept individuals searched
    has = mate >= 0
    drop = np.zeros(n, bool)
    idx = np.nonzero(has)[0]
    m = mate[idx]
    drop[idx] = has[m] & (mate[m] == idx) & (id_s[m] < id_s[idx])
    exp = np.sort(id_s[has & ~drop])
    np.testing.assert_array_equal(np.sort(id_s[pairs[:, 0]]), exp)
    # and every kept individual without a mate really has no neighbour within the radius
    lone = np.nonzero(keep & ~has)[0]
    xs, ys = dev.download(nat.F_X), dev.download(nat.F_Y)
    for i in lone[:200]:
        d2 = (xs - xs[i]) ** 2 + (ys - ys[i]) ** 2
        d2[i] = 1e9
        assert d2.min() > np.float32(2.5) ** 2
    dev.close()


def test_spatial_tester_vs_reference():
    """G16: gnx_spatial_diff_stats (k_cell_counts + k_diff_stats) reproduces the series the
    reference's SpatialTester records over a model's first burn-in steps"""
    g = load_golden('g16_spatial_tester')
    for s in (1, 2):
        dim = tuple(int(v) for v in g['s%i_dim' % s])
        off = np.concatenate([[0], np.cumsum(g['s%i_n' % s])])
        dev = make_dev(dim[0], dim[1], cap=4096, seed=1)
        counts = np.zeros((dim[1], dim[0]))
        for t in range(len(off) - 1):
            x = g['s%i_x' % s][off[t]:off[t + 1]]
            y = g['s%i_y' % s][off[t]:off[t + 1]]
            upload_simple(dev, x, y)
            m, sd = dev.spatial_diff_stats()
            assert abs(m - g['s%i_mean' % s][t]) < 1e-12, (s, t)
            assert abs(sd - g['s%i_std' % s][t]) < 1e-9, (s, t)
            counts, mo, so = O.spatial_diff_stats(counts, x.astype(np.float32),
                                                  y.astype(np.float32), dim)
            assert abs(m - mo) < 1e-12 and abs(sd - so) < 1e-9
        np.testing.assert_array_equal(dev.download_raster(native().R_COUNTS),
                                      g['s%i_counts' % s])
        dev.close()
