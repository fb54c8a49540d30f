"""Edge cases of the hot path on the device: empty and tiny populations, coincident
individuals, one crowded cell, landscape borders, minimal genomes, capacity errors and
extinction (the reference's behaviour for each is cited at the check)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_parity import make_dev, native, upload_simple      # noqa: E402
import gnx_oracle as O                                            # noqa: E402

pytestmark = pytest.mark.gpu


def test_empty_population_steps_and_downloads():
    nat = native()
    dev = make_dev(16, 16, L=70, n_traits=0, cap=64, mating_radius=2.0)
    dev.set_recomb_paths(O.pack_bits(np.zeros((2, 70), np.uint8)))
    upload_simple(dev, np.zeros(0, np.float32), np.zeros(0, np.float32))
    assert dev.counts() == (0, 0, 0)
    for burn in (True, False):
        dev.step(burn, not burn)
        assert dev.counts() == (0, 0, 0)
    assert dev.download(nat.F_X).size == 0 and dev.download(nat.F_ID).size == 0
    dev.close()


def test_single_individual_has_no_mate_and_keeps_its_genome():
    nat = native()
    dev = make_dev(16, 16, L=70, cap=64, mating_radius=3.0, b=1.0, d_min=0.0, d_max=0.0)
    dev.set_recomb_paths(O.pack_bits(np.zeros((2, 70), np.uint8)))
    upload_simple(dev, [8.5], [8.5])
    g = np.zeros((1, 2, dev.W64), np.uint64)
    g[0, 0, 0] = 0x5
    g[0, 1, 1] = 0x3f
    dev.upload_genomes(g)
    mate, pairs = dev.op_find_pairs(np.ones(1, np.uint8))
    assert mate.tolist() == [-1] and len(pairs) == 0
    for _ in range(3):
        dev.step(False, False)
    assert dev.counts()[0] == 1 and dev.counts()[1] == 0        # d_max = 0: nobody dies
    np.testing.assert_array_equal(dev.download(nat.F_GENO), g)
    assert dev.download(nat.F_AGE).tolist() == [3]
    dev.close()


def test_coincident_pair_uniform_vs_inverse_distance():
    nat = native()
    for mode, expect_pair in ((nat.MATE_UNIFORM, True), (nat.MATE_NEAREST, True),
                              (nat.MATE_INVERSE, False)):
        dev = make_dev(16, 16, cap=64, mating_radius=1.0, mate_mode=mode, b=1.0)
        upload_simple(dev, [4.25, 4.25], [7.5, 7.5])
        mate, pairs = dev.op_find_pairs(np.ones(2, np.uint8))
        # distance 0 is within any radius; 1/d weighting cannot use it
        # (utils/spatial.py:225-241 would divide by zero there)
        assert (len(pairs) == 1) == expect_pair, mode
        assert sorted(mate.tolist()) == ([0, 1] if expect_pair else [-1, -1])
        dev.close()


def test_everyone_in_one_cell():
    """5000 individuals inside one hash cell: every focal scans all of them"""
    rng = np.random.RandomState(6)
    n, r = 5000, 4.0
    x = (20 + rng.rand(n) * 3.9).astype(np.float32)
    y = (20 + rng.rand(n) * 3.9).astype(np.float32)
    ids = np.sort(rng.choice(10**6, n, replace=False))
    dev = make_dev(64, 64, cap=8192, seed=4, mating_radius=r, b=0.3)
    upload_simple(dev, x, y, ids=ids)
    keep = rng.rand(n) < 0.3
    mate, pairs = dev.op_find_pairs(keep)
    from test_gpu_parity import _slot_maps
    o = _slot_maps(dev, ids)
    exp = O.choose_mates(x, y, ids, r, 4, 0, dim=(64, 64))
    got = np.full(n, -2)
    got[o] = np.where(mate >= 0, o[np.maximum(mate, 0)], -1)
    np.testing.assert_array_equal(got[keep], exp[keep])
    pr = O.pairs_from_mates(exp, keep)
    assert {frozenset((int(o[a]), int(o[b]))) for a, b in pairs} == \
        {frozenset((int(a), int(b))) for a, b in pr}
    dev.close()


def test_individuals_on_the_landscape_border():
    """positions 0 and dim - 0.001 (the reference's clip bounds, ops/movement.py:88-92)
    sit in the first / last cell and take part in every kernel"""
    nat = native()
    W, H = 33, 17                      # not multiples of the cell or window sizes
    rast = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))])
    dev = make_dev(W, H, rasts=rast, cap=4096, mating_radius=2.0, K_factor=2.0)
    rng = np.random.RandomState(1)
    n = 800
    x = np.concatenate([np.zeros(200), np.full(200, W - 0.001), rng.rand(400) * W])
    y = np.concatenate([rng.rand(200) * H, rng.rand(200) * H, np.zeros(200),
                        np.full(200, H - 0.001)])
    upload_simple(dev, x.astype(np.float32), y.astype(np.float32))
    for _ in range(6):
        dev.step(True, False)
    xx, yy = dev.download(nat.F_X), dev.download(nat.F_Y)
    assert dev.N > 0 and (xx >= 0).all() and (xx <= np.float32(W - 0.001)).all()
    assert (yy >= 0).all() and (yy <= np.float32(H - 0.001)).all()
    e = dev.download(nat.F_E)
    np.testing.assert_array_equal(e[1], rast[1].astype(np.float32)[yy.astype(int), xx.astype(int)])
    dev.close()


@pytest.mark.parametrize('L', [1, 63, 64, 65, 1025])
def test_small_and_ragged_genome_lengths(L):
    """crossover of L not a multiple of the 64-bit word / 128-bit chunk / 1024-bit row pad"""
    nat = native()
    rng = np.random.RandomState(L)
    n_par, n_off = 40, 200
    g = rng.randint(0, 2, (n_par, L, 2)).astype(np.uint8)
    cross = (rng.rand(16, L) < 0.2).astype(np.uint8)
    cross[:, 0] = 0
    pk = O.pack_bits(O.recomb_paths(cross))
    geno = O.pack_genomes(g)
    dev = make_dev(8, 8, L=L, cap=512)
    assert geno.shape[2] == dev.W64 and dev.W64 % 16 == 0
    upload_simple(dev, rng.rand(n_par) * 8, rng.rand(n_par) * 8)
    dev.upload_genomes(geno)
    dev.set_recomb_paths(pk)
    parents = rng.randint(0, n_par, (n_off, 2))
    keys = rng.randint(0, 16, (n_off, 2))
    starts = rng.randint(0, 2, (n_off, 2))
    dev.op_crossover(parents, keys, starts)
    child = dev.download_genomes(np.arange(n_par, n_par + n_off))
    np.testing.assert_array_equal(child, O.crossover(geno, pk, parents, keys, starts))
    # padding bits beyond L stay zero (nothing counts them, but they must not leak)
    bits = O.unpack_genomes(child, dev.W64 * 64)
    assert bits[:, L:, :].sum() == 0
    dev.close()


def test_capacity_overflow_is_an_error_not_a_crash():
    nat = native()
    dev = make_dev(16, 16, cap=256, mating_radius=4.0, b=1.0, K_factor=50.0)
    dev.init_population(250)
    with pytest.raises(nat.GnxError, match='capacity'):
        for _ in range(5):
            dev.step(True, False)
    dev.close()
    with pytest.raises(nat.GnxError):
        dev2 = make_dev(16, 16, cap=64)
        dev2.init_population(65)


def test_model_extinction_stops_the_iteration():
    """reference sim/model.py:776-787: extinction ends the iteration without an exception"""
    import geonomics_amd as gnx
    from test_gpu_model_api import small_params
    p = small_params(T=40, traits=False, L=16)
    p['comm']['species']['spp_0']['mortality'].update({'d_min': 0.9, 'd_max': 1.0})
    p['comm']['species']['spp_0']['mating'].update({'b': 0.01})
    mod = gnx.make_model(p)
    mod.run()
    spp = mod.comm[0]
    # the queue stops at the extinction (sim/model.py:741-745), before _set_Nt
    assert spp.extinct and len(spp) == 0 and spp.Nt[-1] > 0
    assert len(spp.Nt) < 40 + 30
