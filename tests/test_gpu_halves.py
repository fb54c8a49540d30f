"""Shared genome blocks (csrc/gnx_half.h): where a gamete's recombination path has no
switch point the gamete is the parent's homologue bit for bit (ops/mating.py:165-167: the
subsetter is constant there), so the child refers to the parent's block instead of copying
it.  Nothing visible may depend on that: same genotypes as with every gamete copied and
with any number of blocks per homologue, a mutation reaches the mutated individual only,
and after a collection every block is either in use or free.  Needs an MI355X."""
import numpy as np
import pytest

import gnx_oracle as O
from test_gpu_parity import make_dev, native

pytestmark = pytest.mark.gpu

L = 1200          # 32 words per homologue = 2 lines of 128 bytes: 2 blocks by default


def _model(seed=23, defer=True, L=L):
    nat = native()
    W = H = 40
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)
    dev = make_dev(W, H, rasts=rasts, L=L, n_traits=1, cap=16384, seed=seed,
                   mating_radius=3.0, K_factor=1.2, max_age=5)
    dev.set_defer_crossover(defer)
    rng = np.random.RandomState(5)
    loci = np.sort(rng.choice(min(L, 1200), 30, replace=False))
    dev.set_trait(0, loci, 0.05 * np.where(np.arange(30) % 2, -1.0, 1.0), 1, 0.3, 1.0, False)
    # one expected crossover per gamete: e^-1 of the 256 paths have no switch point
    paths = O.recomb_paths((rng.rand(256, L) < 1.0 / L).astype(np.uint8) * (np.arange(L) > 0))
    n_pure = int((paths.max(axis=1) == paths.min(axis=1)).sum())
    assert 60 < n_pure < 130
    dev.set_recomb_paths(O.pack_bits(paths))
    dev.init_population(1800)
    for _ in range(4):
        dev.step(True, False)
    dev.assign_genomes(O.starting_mutation_counts(dev.N, np.full(L, 0.5)))
    return dev, nat


def _check(dev):
    # (after a collection: every physical block is either referred to by somebody alive or free)
    rows, broken, _gc_runs, used, free, total = (int(v) for v in dev.debug_halves())
    assert broken == 0
    assert used + free == total
    assert used <= 2 * rows
    return rows, used


def _genotypes(dev, nat):
    ids = dev.download(nat.F_ID)
    o = np.argsort(ids)
    return ids[o], dev.download(nat.F_GENO)[o]


@pytest.mark.parametrize('defer', [True, False])
def test_shared_half_rows_do_not_change_genotypes(defer, monkeypatch):
    a, nat = _model(defer=defer)
    monkeypatch.setenv('GNX_XO_ALIAS', '0')
    b, _ = _model(defer=defer)
    monkeypatch.delenv('GNX_XO_ALIAS')
    shared = 0
    for t in range(14):
        a.step(False, True)
        b.step(False, True)
        assert a.counts() == b.counts(), t
        rows, used = _check(a)
        shared = max(shared, 2 * rows - used)
        rb, ub = _check(b)
        assert ub == 2 * rb                   # nothing shared when aliasing is off
    assert shared > 200                       # half-rows with more than one referrer
    ia, ga = _genotypes(a, nat)
    ib, gb = _genotypes(b, nat)
    np.testing.assert_array_equal(ia, ib)
    np.testing.assert_array_equal(ga, gb)
    a.close()
    b.close()


def test_mutation_copies_a_shared_half_row_first():
    dev, nat = _model(seed=41)
    for t in range(8):
        dev.step(False, True)
    ids0, g0 = _genotypes(dev, nat)
    rows, used = _check(dev)
    assert used < 2 * rows
    n = ids0.size
    rng = np.random.RandomState(1)
    # mutate every third individual (slots, not ids: map through the id order)
    slot_ids = dev.download(nat.F_ID)
    slots = np.arange(0, n, 3).astype(np.int64)
    loci = rng.randint(0, L, slots.size).astype(np.int32)
    homs = rng.randint(0, 2, slots.size).astype(np.uint8)
    dev.mutate(slots, loci, homs)
    ids1, g1 = _genotypes(dev, nat)
    np.testing.assert_array_equal(ids0, ids1)
    exp = g0.copy()
    pos = np.searchsorted(ids0, slot_ids[slots])
    for p, l, hh in zip(pos, loci, homs):
        exp[p, hh, l >> 6] |= np.uint64(1) << np.uint64(l & 63)
    np.testing.assert_array_equal(g1, exp)    # nobody else's genome moved
    r2, u2 = _check(dev)
    assert r2 == rows and u2 >= used          # copies were made, counts still add up
    # and the population goes on as before
    for t in range(3):
        dev.step(False, True)
        _check(dev)
    dev.close()


def test_blocks_per_homologue_do_not_change_genotypes(monkeypatch):
    """L = 5000: 80 words = 5 lines per homologue: one block or five"""
    monkeypatch.setenv('GNX_HALF_BLOCKS', '5')
    a, nat = _model(L=5000)
    monkeypatch.setenv('GNX_HALF_BLOCKS', '1')
    b, _ = _model(L=5000)
    monkeypatch.delenv('GNX_HALF_BLOCKS')
    for t in range(10):
        a.step(False, True)
        b.step(False, True)
        assert a.counts() == b.counts(), t
        ra, ua = _check(a)
        rb, ub = _check(b)
        assert ra == 5 * rb                       # counted in blocks
    assert 2 * ra - ua > 5 * (2 * rb - ub)        # finer blocks: more of them shared
    ia, ga = _genotypes(a, nat)
    ib, gb = _genotypes(b, nat)
    np.testing.assert_array_equal(ia, ib)
    np.testing.assert_array_equal(ga, gb)
    # a mutation in each of the five blocks of somebody's homologue
    slots = np.full(5, 7, np.int64)
    loci = (np.arange(5) * 1024 + 3).astype(np.int32)
    for dev in (a, b):
        dev.mutate(slots, loci, np.zeros(5, np.uint8))
        _check(dev)
    ia, ga = _genotypes(a, nat)
    ib, gb = _genotypes(b, nat)
    np.testing.assert_array_equal(ga, gb)
    a.close()
    b.close()
