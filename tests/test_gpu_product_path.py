"""The crossover path the PRODUCT runs, pinned to the oracle bit for bit.

`gnx_op_crossover` (the immediate job builder) is matched against the reference's outputs in
test_gpu_parity.py.  A time step of the product takes another route: the crossover is
deferred behind the death draws (survivors only, second stream), a homologue is 14 blocks at
L = 10^5 and a block without a switch point refers to the parent's block, blocks nobody
alive refers to come back through a mark-and-sweep collection that fires when the free
stack runs low, and mutations copy a shared block first.  Here that route itself is
replayed by the oracle: every step's births (parents, path keys, start homologues:
`gnx_last_births`, which does not touch the pending crossover) go through
`O.crossover` (ops/mating.py:130-214) on the host's own copy of the parents' genomes,
injected mutations (ops/mutation.py:90-125) are applied to that copy, and the genomes the
device holds are compared with it for every living individual - newborns bit for bit, and
everybody born earlier unchanged.  Free-block headroom is small, so the collector runs on
its own several times while crossovers are in flight.  Needs an MI355X."""
import threading

import numpy as np
import pytest

import gnx_oracle as O
from test_gpu_parity import native

pytestmark = pytest.mark.gpu

L = 100000          # 1568 words per homologue = 98 lines of 128 bytes: 20 blocks of 5 lines (the
                    # last one reaches past the homologue: the table is laid out for 100 lines)
NB = 20
W = H = 40
N0 = 1500          # settles at ~1530 after mortality (K_factor 1.0), ~310 births per step
STEPS = 130
# Rows for N + a step's births and little more: 2 * NB * 2050 physical blocks, of which the living
# refer to about half at the steady state, so the free stack drops below what a step's job
# builder may take (2 * NB * births) every ~45 steps and the collector runs: 3 times in 130 steps
# with sparse paths, in almost every step with dense masks (one block per homologue, none shared).
CAP_ROWS = 2050


def _paths(dense, n_paths=192, seed=7):
    rng = np.random.RandomState(seed)
    if dense == 'clustered':
        # sparse paths (at most 24 switch points) whose switch points crowd: 4 .. 9 of them
        # inside one stretch of 3000 loci - more than three in one 7168-locus block, which the
        # crossover's jobs cannot carry inline (GNX_BP_MORE: the kernel walks the path's list)
        cross = np.zeros((n_paths, L), np.uint8)
        for k in range(n_paths):
            lo = rng.randint(1, L - 3000)
            cross[k, lo + rng.choice(3000, rng.randint(4, 10), replace=False)] = 1
            cross[k, rng.choice(np.arange(1, L), rng.randint(0, 6), replace=False)] = 1
        assert cross.sum(axis=1).max() <= 24
        return O.pack_bits(O.recomb_paths(cross))
    rate = 0.5 if dense else 1.0 / L
    cross = (rng.rand(n_paths, L) < rate).astype(np.uint8)
    cross[:, 0] = 0
    return O.pack_bits(O.recomb_paths(cross))


def _make(paths, seed=29, cap_inds=4096, cap_rows=CAP_ROWS, overlap=0, trait=True, N=N0,
          W=W, H=H, K_factor=1.0, upload=True, phi=0.05, max_age=-1):
    nat = native()
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)
    dev = nat.Device(W, H, 2, L=L, n_traits=1 if trait else 0, cap_inds=cap_inds,
                     cap_rows=cap_rows, seed=seed)
    dev.upload_rasters(rasts)
    dev.set_species_params(nat.default_species_params(mating_radius=3.0, K_factor=K_factor,
                                                      max_age=max_age))
    rng = np.random.RandomState(3)
    if trait:
        loci = np.sort(rng.choice(L, 12, replace=False))
        dev.set_trait(0, loci, 0.08 * np.where(np.arange(12) % 2, -1.0, 1.0), 1, phi, 1.0, False)
    dev.set_recomb_paths(paths)
    g = None
    if upload:
        x = rng.rand(N) * W
        y = rng.rand(N) * H
        dev.upload_population(x, y, rng.randint(0, 4, N), np.zeros(N), np.arange(N))
        g = rng.randint(0, 2 ** 63, (N, 2, dev.W64), dtype=np.int64).astype(np.uint64)
        g ^= rng.randint(0, 2, g.shape).astype(np.uint64) << np.uint64(63)
        g[:, :, L // 64] &= np.uint64((1 << (L % 64)) - 1)
        g[:, :, L // 64 + 1:] = 0
        dev.upload_genomes(g)
        dev.set_z()
    dev.set_defer_crossover(True)
    if overlap:
        dev.set_crossover_overlap(overlap)
    return dev, g


class HostGenomes:
    """the oracle's copy of every genome: id -> uint64 [2][W64]"""

    def __init__(self, ids, g, paths):
        self.g = {int(i): g[k] for k, i in enumerate(ids)}
        self.paths = paths
        self.born = 0

    def births(self, child, par, keys, starts):
        """O.crossover (ops/mating.py:130-214) of this step's births"""
        if child.size == 0:
            return
        uniq, inv = np.unique(par.ravel(), return_inverse=True)
        pg = np.stack([self.g[int(i)] for i in uniq])
        kids = O.crossover(pg, self.paths, inv.reshape(par.shape), keys, starts)
        for k, c in enumerate(child):
            assert int(c) not in self.g
            self.g[int(c)] = kids[k]
        self.born += child.size

    def mutate(self, ids, loci, homs):
        """a mutation sets allele 1 at one locus of one homologue of ONE individual
        (ops/mutation.py:90-125; DESIGN 2, reference semantics undefined off-tskit)"""
        for i, l, hh in zip(ids, loci, homs):
            a = self.g[int(i)].copy()             # (parents keep theirs)
            a[hh, l >> 6] |= np.uint64(1) << np.uint64(l & 63)
            self.g[int(i)] = a

    def check(self, dev, nat, tag=''):
        ids = dev.download(nat.F_ID)
        got = dev.download(nat.F_GENO)
        assert np.unique(ids).size == ids.size
        exp = np.stack([self.g[int(i)] for i in ids])
        bad = np.nonzero((got != exp).any(axis=(1, 2)))[0]
        assert bad.size == 0, '%s: %d of %d genomes differ from the oracle replay (first id %d)' % (
            tag, bad.size, ids.size, int(ids[bad[0]]))
        alive = set(int(i) for i in ids)
        for i in [i for i in self.g if i not in alive]:      # the dead are never read again
            del self.g[i]
        return ids.size


def _split_step(dev, host, t, mutate_rng=None, trait_loci=()):
    """one step the way Species._do_pop_dynamics drives it (structs/species.py here;
    reference structs/species.py:822-833): age + move, mate, [mutation], die"""
    dev.age()
    dev.move()
    n0 = dev.N
    dev.pop_dynamics_mate(False)
    B = dev.counts()[1]
    child, par, keys, starts, _ = dev.last_births()
    assert dev.genome_info()['deferred'] == (1 if B else 0)     # reading births cut nothing
    host.births(child, par, keys, starts)
    n_mut = 0
    if mutate_rng is not None and B > 0:
        n_mut = min(B, 9)
        k = mutate_rng.choice(B, n_mut, replace=False)
        loci = mutate_rng.randint(1, L, n_mut).astype(np.int32)
        while np.isin(loci, trait_loci).any():
            loci = mutate_rng.randint(1, L, n_mut).astype(np.int32)
        homs = mutate_rng.randint(0, 2, n_mut).astype(np.uint8)
        dev.mutate((n0 + k).astype(np.int64), loci, homs)       # joins: every birth is cut now
        host.mutate(child[k], loci, homs)
    dev.pop_dynamics_die(False, True)
    dev.step_index = dev.step_index + 1
    return B, n_mut


@pytest.mark.parametrize('overlap', [0, 1])
def test_model_step_path_matches_oracle_crossover(overlap):
    """the split step of the Model API: deferred + 20 shared blocks + natural collections +
    mutations in every third step; genomes == oracle replay at every checkpoint"""
    nat = native()
    paths = _paths(False)
    dev, g = _make(paths, overlap=overlap)
    info = dev.genome_info()
    assert info['NB'] == NB and info['sparse'] == 1 and info['BW'] == 80 and dev.W64 == 1568
    host = HostGenomes(np.arange(N0), g, paths)
    rng = np.random.RandomState(11)
    births = muts = not_cut = 0
    for t in range(STEPS):
        B, m = _split_step(dev, host, t, rng if t % 3 == 2 else None)
        births += B
        muts += m
        not_cut += B - dev.last_crossover_births
        if t % 26 == 25 or t == STEPS - 1:
            host.check(dev, nat, 'step %d' % t)
    gc = dev.genome_info()['gc_runs']
    assert gc >= 2, 'the collector never ran on its own (gc_runs = %d)' % gc
    assert births > 25000 and muts > 300
    assert not_cut > 1500           # offspring that died at age 0 never got a genome
    # bookkeeping after all that: nothing broken, used + free = all, blocks are shared
    rows, broken, _, used, free, total = (int(v) for v in dev.debug_halves())
    assert broken == 0 and used + free == total and used < 2 * rows
    host.check(dev, nat, 'after the final collection')
    dev.close()


@pytest.mark.parametrize('lines,nb', [(4, 25), (6, 17), (7, 14), (8, 13)])
def test_other_block_sizes_match_oracle_crossover(lines, nb, monkeypatch):
    """GNX_BLOCK_LINES: blocks of 4 / 6 / 7 / 8 lines instead of the default 5 (25 / 17 / 14 / 13
    blocks per homologue; all but 7 with a last block that reaches past the homologue; above 16
    the fused job builder's other variant): the same replay, shorter"""
    monkeypatch.setenv('GNX_BLOCK_LINES', str(lines))
    nat = native()
    paths = _paths(False)
    dev, g = _make(paths, overlap=0)
    info = dev.genome_info()
    assert info['NB'] == nb and info['BW'] == 16 * lines and info['sparse'] == 1
    host = HostGenomes(np.arange(N0), g, paths)
    rng = np.random.RandomState(5)
    for t in range(60):
        _split_step(dev, host, t, rng if t % 3 == 2 else None)
        if t % 20 == 19:
            host.check(dev, nat, 'step %d' % t)
    rows, broken, _, used, free, total = (int(v) for v in dev.debug_halves())
    assert broken == 0 and used + free == total and used < 2 * rows
    host.check(dev, nat, 'after the final collection')
    dev.close()


@pytest.mark.parametrize('dense,overlap', [(False, 0), (False, 1), (True, 0), ('clustered', 0)])
def test_fused_step_path_matches_oracle_crossover(dense, overlap):
    """gnx_step (the bench's path).  Its births cannot be read mid-step, so a twin driven
    through the split path supplies them (draws are keyed by id and step: same decisions,
    asserted); the oracle replay is compared with BOTH devices."""
    nat = native()
    paths = _paths(dense)
    a, g = _make(paths, overlap=overlap)
    b, _ = _make(paths, overlap=0)
    info = a.genome_info()
    assert info['NB'] == (1 if dense is True else NB) and info['sparse'] == (0 if dense is True else 1)
    host = HostGenomes(np.arange(N0), g, paths)
    for t in range(STEPS):
        a.step(False, True)
        _split_step(b, host, t)
        assert a.counts() == b.counts(), t
        if t % 32 == 31 or t == STEPS - 1:
            np.testing.assert_array_equal(np.sort(a.download(nat.F_ID)),
                                          np.sort(b.download(nat.F_ID)))
            # (check a first: checking prunes the dead from the host copy, same for both)
            host.check(a, nat, 'gnx_step, step %d' % t)
            host.check(b, nat, 'split step, step %d' % t)
    for dev in (a, b):
        gc = dev.genome_info()['gc_runs']
        assert gc >= 2, 'the collector never ran on its own (gc_runs = %d)' % gc
    assert host.born > 25000
    a.close()
    b.close()


@pytest.mark.parametrize('overlap,library', [(0, False), (1, False), (0, True), (1, True)])
def test_two_tiles_match_oracle_crossover(overlap, library):
    """two tiles (threads of this process, device-resident transport) through gnx_tile_*:
    deferred crossover per tile, gametes of ghost mates cut on the owning tile, migrants
    carrying their genomes.  Each tile's births are read in the stepper's after-births hook.
    overlap = 1: the crossover runs beside the whole next step, i.e. it may still be in flight
    when the tile serves its neighbours' gamete requests (the service waits for it).
    library: every step inside the library's own protocol (gnx_tile_step_begin, the hook,
    gnx_tile_step_end - csrc/gnx_comm.hip), the oracle replaying the LIBRARY path's births."""
    import torch
    from _local_comm import Hub, LocalComm
    from geonomics_amd.parallel import DeviceShard, TiledStepper
    nat = native()
    paths = _paths(False)
    Wt, Ht, N = 80, 40, 3000
    rng = np.random.RandomState(3)
    x = rng.rand(N) * Wt
    y = rng.rand(N) * Ht
    age = rng.randint(0, 4, N)
    W64 = nat.load().gnx_words_per_hom(L)
    g = rng.randint(0, 2 ** 63, (N, 2, W64), dtype=np.int64).astype(np.uint64)
    g[:, :, L // 64] &= np.uint64((1 << (L % 64)) - 1)
    g[:, :, L // 64 + 1:] = 0
    host = HostGenomes(np.arange(N), g, paths)
    lock = threading.Lock()
    hub = Hub(2, library_group=library)
    errs, sizes, gcs, pending = [], [0, 0], [0, 0], [[], []]

    def body(rank):
        try:
            torch.cuda.set_device(0)
            comm = LocalComm(hub, rank)
            dev, _ = _make(paths, cap_inds=4096, cap_rows=2150, N=N, W=Wt, H=Ht, upload=False,
                           overlap=overlap)
            shard = DeviceShard(dev)
            stepper = TiledStepper(shard, comm, Wt, Ht, 3.0, move=True, max_id=N - 1,
                                   fixed_births=1, use_library=library)
            assert stepper.v3 == library
            mine = stepper.rank_of(x, y) == rank
            dev.upload_population(x[mine], y[mine], age[mine], np.zeros(mine.sum()),
                                  np.arange(N)[mine])
            dev.upload_genomes(g[mine])
            dev.set_z()
            dev.set_defer_crossover(True)
            shard.has_genomes = True

            def after(first_id, total):
                pending[rank].append(dev.last_births()[:4])

            for t in range(STEPS):
                del pending[rank][:]
                stepper.step(False, True, after_births=after)
                # both tiles' births of the step go in before anybody's next step reads them
                hub.barrier.wait()
                if rank == 0:
                    for r in range(2):
                        for rec in pending[r]:
                            host.births(*rec)
                hub.barrier.wait()
                if t % 26 == 25 or t == STEPS - 1:
                    with lock:
                        ids = dev.download(nat.F_ID)
                        got = dev.download(nat.F_GENO)
                        exp = np.stack([host.g[int(i)] for i in ids])
                        bad = int((got != exp).any(axis=(1, 2)).sum())
                        assert bad == 0, 'tile %d step %d: %d genomes differ' % (rank, t, bad)
                    hub.barrier.wait()
            sizes[rank] = dev.N
            gcs[rank] = dev.genome_info()['gc_runs']
            dev.close()
        except BaseException as e:       # noqa: BLE001 - re-raised in the main thread
            errs.append(e)
            hub.abort()

    ths = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errs:
        raise errs[0]
    assert min(sizes) > 1000 and host.born > 25000
    assert min(gcs) >= 2, gcs
