"""The crossover of a step's offspring is deferred behind the death draws (survivors only,
on a second stream) - gnx_set_defer_crossover.  Draws are keyed by id, so the deferred and
the immediate order of operations must give the same population bit for bit; any access to
genomes between the two halves of the step falls back to the immediate path.  Needs an
MI355X."""
import numpy as np
import pytest

import gnx_oracle as O
from test_gpu_parity import make_dev, native

pytestmark = pytest.mark.gpu


def _model(defer, L=900, seed=17, dense=False):
    nat = native()
    W = H = 40
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)
    dev = make_dev(W, H, rasts=rasts, L=L, n_traits=2, cap=16384, seed=seed,
                   mating_radius=3.0, K_factor=1.2, max_age=6)
    dev.set_defer_crossover(defer)
    rng = np.random.RandomState(3)
    loci = np.sort(rng.choice(L, 80, replace=False))
    dev.set_trait(0, loci[:70], 0.05 * np.where(np.arange(70) % 2, -1.0, 1.0), 1, 0.3, 1.0, False)
    dev.set_trait(1, loci[70:71], np.array([0.25]), 1, 0.1, 2.0, True)
    dev.set_deleterious(loci[71:], np.full(9, 0.02))
    rate = 0.3 if dense else 0.003
    dev.set_recomb_paths(O.pack_bits(O.recomb_paths(
        (rng.rand(64, L) < rate).astype(np.uint8) * (np.arange(L) > 0))))
    dev.init_population(1800)
    for _ in range(4):
        dev.step(True, False)
    dev.assign_genomes(O.starting_mutation_counts(dev.N, np.full(L, 0.5)))
    return dev, nat


def _state(dev, nat):
    ids = dev.download(nat.F_ID)
    o = np.argsort(ids)
    return dict(ids=ids[o], x=dev.download(nat.F_X)[o], y=dev.download(nat.F_Y)[o],
                age=dev.download(nat.F_AGE)[o], z=dev.download(nat.F_Z)[:, o],
                fit=dev.download(nat.F_FIT)[o], g=dev.download(nat.F_GENO)[o])


@pytest.mark.parametrize('dense', [False, True])
def test_deferred_crossover_equals_immediate(dense):
    a, nat = _model(True, dense=dense)
    b, _ = _model(False, dense=dense)
    skipped = 0
    for t in range(12):
        a.step(False, True)
        b.step(False, True)
        assert a.counts() == b.counts(), t
        births = a.counts()[1]
        assert b.last_crossover_births == births
        assert a.last_crossover_births <= births
        skipped += births - a.last_crossover_births
    assert skipped > 50                  # offspring that died at age 0 never got a genome
    sa, sb = _state(a, nat), _state(b, nat)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    # rows in use are distinct and the free stack holds the rest (no leak, no double use)
    for dev in (a, b):
        rows = dev.download(nat.F_GROW)
        assert rows.min() >= 0 and np.unique(rows).size == rows.size
    a.close()
    b.close()


def test_genome_access_between_mate_and_die_materialises():
    """reading the offspring's genomes between gnx_pop_dynamics_mate and _die (the
    pedigree recorder, mutation, statistics do) cuts them at once, for every birth"""
    a, nat = _model(True, seed=5)
    b, _ = _model(False, seed=5)
    for t in range(5):
        for dev in (a, b):
            n0 = dev.N
            dev.pop_dynamics_mate(False)
        B = a.counts()[1]
        assert B == b.counts()[1] and B > 0
        if t % 2 == 0:
            slots = np.arange(n0, n0 + B)
            np.testing.assert_array_equal(a.download_genomes(slots), b.download_genomes(slots))
        else:
            c1a, cha = a.stats_locus_counts()
            c1b, chb = b.stats_locus_counts()
            np.testing.assert_array_equal(c1a, c1b)
        for dev in (a, b):
            dev.pop_dynamics_die(False, True)
            dev.step_index = dev.step_index + 1
        assert a.counts() == b.counts()
        assert a.last_crossover_births == B          # fell back to every birth
    sa, sb = _state(a, nat), _state(b, nat)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    a.close()
    b.close()


def test_overlap_mode_and_table_spread_do_not_change_results(monkeypatch):
    """how the crossover shares the GPU with the next step (gnx_set_crossover_overlap,
    gnx_set_crossover_split) and where the genome rows sit in HBM (GNX_ROW_SPREAD) are
    scheduling / placement choices: the population after 10 steps is the same bit for bit"""
    ref, nat = _model(True, seed=23)
    alt, _ = _model(True, seed=23)
    alt.set_crossover_overlap(True)
    split, _ = _model(True, seed=23)
    split.set_crossover_split(400)
    monkeypatch.setenv('GNX_ROW_SPREAD', '1')
    compact, _ = _model(True, seed=23)
    monkeypatch.delenv('GNX_ROW_SPREAD')
    for t in range(10):
        for dev in (ref, alt, split, compact):
            dev.step(False, True)
        assert ref.counts() == alt.counts() == split.counts() == compact.counts(), t
    s0 = _state(ref, nat)
    for dev in (alt, split, compact):
        s1 = _state(dev, nat)
        for k in s0:
            np.testing.assert_array_equal(s0[k], s1[k], err_msg=k)
    # the spread table really is spread: rows are multiples of the stride
    r_ref, r_cmp = ref.download(nat.F_GROW), compact.download(nat.F_GROW)
    stride = np.gcd.reduce(r_ref[r_ref > 0])
    assert stride >= 2 and np.gcd.reduce(r_cmp[r_cmp > 0]) == 1
    for dev in (ref, alt, split, compact):
        dev.close()


def test_id_ordered_index_sort_equals_full_sort(monkeypatch):
    """The cell sort runs over an index of the slots in id order with the cell as its only
    key (gnx_internal.h: ord) instead of sorting (cell, id) keys; the index is rebuilt after
    an upload in arbitrary id order and survives repeated sorts, injected deaths and births.
    Same population, bit for bit, as with GNX_ORD_SORT=0."""
    nat = native()
    W = H = 40
    rng = np.random.RandomState(12)
    n = 3000
    ids = rng.permutation(n * 3)[:n].astype(np.int64)           # NOT ascending
    x = (rng.rand(n) * W).astype(np.float32)
    y = (rng.rand(n) * H).astype(np.float32)
    g = rng.randint(0, 2 ** 62, (n, 2, 16), dtype=np.int64).astype(np.uint64)
    g[:, :, 15] = 0                                             # L = 900: bits beyond stay clear
    g[:, :, 14] &= np.uint64((1 << (900 - 14 * 64)) - 1)
    paths = O.pack_bits(O.recomb_paths((rng.rand(64, 900) < 0.002).astype(np.uint8) *
                                       (np.arange(900) > 0)))
    devs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('GNX_ORD_SORT', flag)
        dev = make_dev(W, H, L=900, cap=16384, seed=5, mating_radius=3.0, K_factor=1.9, max_age=7)
        from test_gpu_parity import upload_simple
        upload_simple(dev, x, y, ids=ids)
        dev.upload_genomes(g)
        dev.set_recomb_paths(paths)
        devs.append(dev)
    monkeypatch.delenv('GNX_ORD_SORT')
    a, b = devs
    for t in range(14):
        if t == 3:                     # a sort with no mortality behind it, then another
            for dev in devs:
                dev.op_find_pairs(None)
        if t == 6:                     # deaths injected in slot order (same slots: same order)
            ia, ib = a.download(nat.F_ID), b.download(nat.F_ID)
            np.testing.assert_array_equal(ia, ib)
            dead = (ia % 5 == 0).astype(np.uint8)
            for dev in devs:
                dev.op_mortality(dead)
        for dev in devs:
            dev.step(False, True)
        assert a.counts() == b.counts(), t
    for f in (nat.F_ID, nat.F_X, nat.F_Y, nat.F_AGE, nat.F_GENO):     # same slots, same order
        np.testing.assert_array_equal(a.download(f), b.download(f))
    assert a.counts()[0] > 1500
    a.close()
    b.close()


@pytest.mark.parametrize('W,radius,n', [(30, 1.0, 20000), (200, 1.0, 20000), (1500, 1.0, 20000),
                                        (500, 1.0, 20000), (200, 1.0, 300000)])
def test_cell_sort_digit_places(W, radius, n):
    """The cell sort (csrc/gnx_prim.hip: k_keys_hist + the own Onesweep passes gnx_os::k_pass)
    with two places of 8-bit digits (900 and 4*10^4 hash cells), two of 9 (2.5*10^5 cells: 18
    bits) and three of 8 (2.25*10^6), and with 74 tiles of 4 096 keys (the look-back walks more
    than one batch of predecessors): the sorted population is in (cell, id) order - the oracle's
    stable sort by cell of the id-ordered population - sort after sort (the scratch is wiped by
    k_permute, not by a fill), with ids uploaded in arbitrary order."""
    nat = native()
    from test_gpu_parity import upload_simple
    rng = np.random.RandomState(W)
    ids = rng.permutation(n * 4)[:n].astype(np.int64)
    x = (rng.rand(n) * W).astype(np.float32)
    y = (rng.rand(n) * W).astype(np.float32)
    x[:500] = x[500:1000]               # crowded cells
    y[:500] = y[500:1000]
    dev = make_dev(W, W, cap=max(32768, 2 * n), seed=3, mating_radius=radius)
    upload_simple(dev, x, y, ids=ids)
    inv_cs, ncx, _ = O.hash_grid((W, W), radius)      # (the device multiplies by the reciprocal)
    for rep in range(3):
        dev.op_find_pairs(None)
        got_id = dev.download(nat.F_ID)
        gx, gy = dev.download(nat.F_X), dev.download(nat.F_Y)
        cx = np.minimum(ncx - 1, (gx.astype(np.float64) * inv_cs).astype(np.int64))
        cy = np.minimum(ncx - 1, (gy.astype(np.float64) * inv_cs).astype(np.int64))
        cell = cy * ncx + cx
        key = cell * (4 * n) + got_id
        assert (np.diff(key) > 0).all(), (W, rep)
        assert np.array_equal(np.sort(got_id), np.sort(ids))
        o = np.argsort(ids)
        np.testing.assert_array_equal(gx[np.argsort(got_id)], x[o])
        if rep == 0:                    # some die in between: the index is compacted, then sorted again
            dead = (got_id % 7 == 0).astype(np.uint8)
            dev.op_mortality(dead)
            keep = ~np.isin(ids, got_id[dead != 0])
            ids, x, y = ids[keep], x[keep], y[keep]
    dev.close()


def test_step_with_counted_digits_equals_the_queue_path():
    """gnx_step lets the movement kernel count the cell sort's digits and the first radix pass
    gather its keys (no kernel in front of the passes); the function queue's calls (gnx_age,
    gnx_move, gnx_pop_dynamics: keys and counts by k_cells / k_keys_hist) do not.  2.5*10^5 hash
    cells = two places of NINE-bit digits (the metric workload has two of eight): the same
    population either way, id by id, burn-in and main steps."""
    nat = native()
    W = 500
    rasts = np.stack([np.ones((W, W)), np.tile(np.linspace(0, 1, W), (W, 1))]).astype(np.float32)

    def mk():
        dev = make_dev(W, W, rasts=rasts, L=0, n_traits=0, cap=131072, seed=9, mating_radius=1.0,
                       K_factor=0.12)
        dev.init_population(30000)
        return dev

    a, b = mk(), mk()
    for t in range(6):
        a.step(True, False)
        b.age()
        b.move()
        b.pop_dynamics(True, False)
        b.step_index = b.step_index + 1
        assert a.counts() == b.counts(), t
    oa, ob = np.argsort(a.download(nat.F_ID)), np.argsort(b.download(nat.F_ID))
    for f in (nat.F_ID, nat.F_X, nat.F_Y, nat.F_AGE):
        np.testing.assert_array_equal(a.download(f)[oa], b.download(f)[ob])
    assert a.counts()[0] > 10000 and a.counts()[1] > 500
    a.close()
    b.close()


def test_in_place_compaction_equals_stable_copy(monkeypatch):
    """Mortality's compaction in place (k_fill_lists + k_fill, csrc/gnx_kernels_demog.hip: the
    survivors of the tail fill the holes of the dead) against the stable copy
    (GNX_COMPACT_FILL=0): the same individuals with the same positions, ages, phenotypes and
    genomes after 15 steps with deferred crossover, id by id - and NOT the same slots."""
    nat = native()
    from test_gpu_parity import upload_simple
    W = H = 40
    L = 900
    rng = np.random.RandomState(21)
    n = 3000
    ids = rng.permutation(n * 3)[:n].astype(np.int64)
    x = (rng.rand(n) * W).astype(np.float32)
    y = (rng.rand(n) * H).astype(np.float32)
    g = rng.randint(0, 2 ** 62, (n, 2, 16), dtype=np.int64).astype(np.uint64)
    g[:, :, 15] = 0
    g[:, :, 14] &= np.uint64((1 << (L - 14 * 64)) - 1)
    paths = O.pack_bits(O.recomb_paths((rng.rand(64, L) < 0.002).astype(np.uint8) *
                                       (np.arange(L) > 0)))
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)
    devs = []
    for flag in ('1', '0'):
        # (the stable copy's twin also permutes every column in one kernel: the split
        # permutation - k_permute + k_permute_rest on the side stream - is compared as well)
        monkeypatch.setenv('GNX_COMPACT_FILL', flag)
        monkeypatch.setenv('GNX_PERMUTE_SPLIT', flag)
        dev = make_dev(W, H, rasts=rasts, L=L, n_traits=1, cap=16384, seed=8, mating_radius=3.0,
                       K_factor=1.9, max_age=9)
        dev.set_trait(0, np.array([5, 300, 611, 842]), np.array([0.1, -0.1, 0.1, -0.1]), 1, 0.05,
                      1.0, False)
        upload_simple(dev, x, y, ids=ids)
        dev.upload_genomes(g)
        dev.set_recomb_paths(paths)
        dev.set_z()
        devs.append(dev)
    monkeypatch.delenv('GNX_COMPACT_FILL')
    monkeypatch.delenv('GNX_PERMUTE_SPLIT')
    a, b = devs
    moved = False
    for t in range(15):
        for dev in devs:
            dev.step(False, True)
        assert a.counts() == b.counts(), t
        ia, ib = a.download(nat.F_ID), b.download(nat.F_ID)
        moved = moved or not np.array_equal(ia, ib)
        oa, ob = np.argsort(ia), np.argsort(ib)
        np.testing.assert_array_equal(ia[oa], ib[ob])
        for f in (nat.F_X, nat.F_Y, nat.F_AGE, nat.F_FIT, nat.F_GENO):
            np.testing.assert_array_equal(a.download(f)[oa], b.download(f)[ob], err_msg=str((t, f)))
        for f in (nat.F_Z, nat.F_E):                  # [traits or layers][N]
            np.testing.assert_array_equal(a.download(f)[:, oa], b.download(f)[:, ob],
                                          err_msg=str((t, f)))
    assert moved and a.counts()[0] > 1500
    a.close()
    b.close()


@pytest.mark.parametrize('variant', ['sparse', 'dense', 'no_graph', 'tile_major_ids',
                                     'host_driven_walk'])
def test_small_path_equals_default_path(variant, monkeypatch):
    """gnx_walk - every count of the step on the device, grids sized by the capacity, one
    captured HIP graph per step, no read-back (the path BASELINE configs[1] and [2] take) -
    against T calls of gnx_step, the many-kernel host-driven path: same seeds -> the same
    population, id by id (positions, ages, phenotypes, fitness, genotypes), the same counts
    in every step, burn-in steps and main steps, walks cut into pieces with host-driven
    steps, a mutation and a forced block collection in between."""
    if variant == 'no_graph':
        monkeypatch.setenv('GNX_DD_GRAPH', '0')
    if variant == 'host_driven_walk':
        # gnx_walk as it takes the metric workload (too large for the device-driven step): gnx_step
        # per step, every step but the last moving the population for the next one right after
        # its own death draws, on the uncompacted slots (gnx_l_move_ahead) - same population
        monkeypatch.setenv('GNX_DD', '0')
    dense = variant == 'dense'
    a, nat = _model(True, dense=dense, seed=31)
    b, _ = _model(True, dense=dense, seed=31)
    if variant == 'tile_major_ids':
        # offspring ids virtual tile by virtual tile (gnx_set_id_order 1, the Model API's default):
        # the newborns' ids do not ascend with their slots - the device-driven step files them
        # behind the id-ordered index itself
        a.set_id_order(1)
        b.set_id_order(1)
    hist = []

    def both(T, burn=False):
        for _ in range(T):
            n0 = a.N
            a.step(burn, not burn)
            hist.append((n0, a.counts()[1], a.counts()[2]))
        b.walk(T, burn, not burn)

    both(7)
    assert a.counts()[0] == b.counts()[0]
    n, births, deaths = b.walk_history()
    assert [tuple(int(v) for v in r) for r in zip(n, births, deaths)] == hist[-7:]
    sa, sb = _state(a, nat), _state(b, nat)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    # a mutation (host-side access to the genomes), a host-driven step on both, a collection
    victim = sa['ids'][5]
    for dev in (a, b):
        slot = np.nonzero(dev.download(nat.F_ID) == victim)[0].astype(np.int64)
        dev.mutate(slot, np.array([11], np.int32), np.array([1], np.uint8))
        dev.step(False, True)
    assert b.debug_halves()[1] == 0          # (runs a collection of the shared genome blocks)
    both(9)
    both(1)
    both(2)
    assert a.counts() == b.counts()
    ta, tb = a.totals(), b.totals()
    if variant == 'host_driven_walk':
        assert tb.pop('dd_steps') == 0 and ta.pop('dd_steps') == 0
    else:
        assert tb.pop('dd_steps') >= 16 and ta.pop('dd_steps') == 0     # b really took the other path
    assert ta == tb, (ta, tb)
    sa, sb = _state(a, nat), _state(b, nat)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    for dev in (a, b):
        rows = dev.download(nat.F_GROW)
        assert rows.min() >= 0 and np.unique(rows).size == rows.size
        assert dev.debug_halves()[1] == 0    # no broken block references
    a.close()
    b.close()


@pytest.mark.parametrize('variant', ['device_driven', 'host_driven'])
def test_walk_that_outgrows_the_capacity_fails_loudly(variant, monkeypatch):
    """a gnx_walk whose births do not fit the preallocated slots returns
    'capacity exceeded', and gnx_walk_history holds ONLY the steps that completed cleanly -
    not the step that dropped its births, nor anything enqueued behind it - and those equal
    the steps of a handle with room, count for count.  (Species._walk_on_device re-raises:
    the population of such a handle is not a population of the model any more.)"""
    if variant == 'host_driven':
        monkeypatch.setenv('GNX_DD', '0')
    nat = native()
    W = H = 40
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)

    def mk(cap):
        dev = make_dev(W, H, rasts=rasts, L=0, n_traits=0, cap=cap, seed=4, mating_radius=3.0,
                       K_factor=2.0)
        dev.init_population(400)
        return dev

    a, b = mk(16384), mk(1024)
    seq = []
    for _ in range(40):
        n0 = a.N
        a.step(True, False)
        seq.append((n0, a.counts()[1], a.counts()[2]))
    assert max(n + bb for n, bb, _ in seq) > 1024          # (the small handle must overflow)
    with pytest.raises(nat.GnxError, match='capacity exceeded'):
        b.walk(40, True, False)
    n, births, deaths = b.walk_history()
    got = [tuple(int(v) for v in r) for r in zip(n, births, deaths)]
    first_bad = next(k for k, (n0, bb, _) in enumerate(seq) if n0 + bb > 1024)
    assert 0 < len(got) <= first_bad
    assert got == seq[:len(got)]
    a.close()
    b.close()


def test_small_path_burn_in_and_growth():
    """device-driven burn-in steps (no genomes) from a small founder population that grows
    several-fold to its carrying capacity: counts per step equal the host-driven path's"""
    nat = native()
    W = H = 40
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)

    def mk():
        dev = make_dev(W, H, rasts=rasts, L=0, n_traits=0, cap=16384, seed=4, mating_radius=3.0,
                       K_factor=2.0)
        dev.init_population(400)
        return dev

    a, b = mk(), mk()
    seq = []
    for _ in range(40):
        n0 = a.N
        a.step(True, False)
        seq.append((n0, a.counts()[1], a.counts()[2]))
    b.walk(40, True, False)
    n, births, deaths = b.walk_history()
    assert [tuple(int(v) for v in r) for r in zip(n, births, deaths)] == seq
    assert a.N == b.N > 1500
    assert b.totals()['dd_steps'] >= 38
    for f in (nat.F_ID, nat.F_X, nat.F_Y, nat.F_AGE):
        ia, ib = np.argsort(a.download(nat.F_ID)), np.argsort(b.download(nat.F_ID))
        np.testing.assert_array_equal(a.download(f)[ia], b.download(f)[ib])
    a.close()
    b.close()
