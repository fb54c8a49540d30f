"""Full-size checks (BASELINE configs[1]: 2048 x 2048, N = 10^6, L = 10^5) through
properties that need no oracle at that size: exact allele-count checksums of the
crossover's routing, integer conservation laws of a step, uniqueness of ids and genome
rows, determinism."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import gnx_oracle as O                                   # noqa: E402

pytestmark = pytest.mark.gpu

NBITS = 20          # 2^20 > 10^6 ids
TAG0 = 31000        # loci that carry the id tag


def _build(cfg_name='c4_metric'):
    import bench
    from geonomics_amd import _native as nat
    cfg = dict(bench.WORKLOADS[cfg_name])
    dev, _, _ = bench.build_device(cfg, seed=42, device=0)
    return bench, nat, cfg, dev


def test_crossover_routing_checksum_at_metric_size():
    """Every individual carries its id, bit by bit, on BOTH homologues at loci
    [31000, 31020) of an otherwise empty genome, so whatever path and start homologue a
    gamete uses it must carry its parent's tag: child homologue 0 = tag(parent A),
    homologue 1 = tag(parent B), all other bits 0.  Checked exactly for all 2 x 10^5
    offspring through the per-locus allele counts (a checksum over 10^6 + 2 x 10^5
    genomes) and bit for bit on a sample of rows."""
    bench, nat, cfg, dev = _build()
    N, L = dev.N, cfg['L']
    assert N == 1_000_000 and L == 100_000
    dev.set_recomb_paths(bench.sparse_paths(cfg['n_paths'], L, 43, dev.W64))
    dev.assign_genomes(np.zeros(L, np.int32))
    ids = dev.download(nat.F_ID)
    slots = np.arange(N, dtype=np.int64)
    bit = (ids[:, None] >> np.arange(NBITS)[None, :]) & 1            # [N, NBITS]
    who, which = np.nonzero(bit)
    for hom in (0, 1):
        dev.mutate(slots[who], (TAG0 + which).astype(np.int32),
                   np.full(who.size, hom, np.uint8))
    c1, ch = dev.stats_locus_counts()
    np.testing.assert_array_equal(c1[TAG0:TAG0 + NBITS], 2 * bit.sum(axis=0))
    assert c1.sum() == 2 * bit.sum() and ch.sum() == 0
    rng = np.random.RandomState(5)
    B = 200_000
    parents = rng.randint(0, N, (B, 2)).astype(np.int32)
    keys = rng.randint(0, cfg['n_paths'], (B, 2)).astype(np.int32)
    starts = rng.randint(0, 2, (B, 2)).astype(np.uint8)
    dev.op_crossover(parents, keys, starts)
    assert dev.N == N + B
    # checksum over all N + B genomes
    exp = 2 * bit.sum(axis=0) + bit[parents[:, 0]].sum(axis=0) + bit[parents[:, 1]].sum(axis=0)
    c1b, chb = dev.stats_locus_counts()
    np.testing.assert_array_equal(c1b[TAG0:TAG0 + NBITS], exp)
    assert c1b.sum() == exp.sum()                       # nothing anywhere else
    # heterozygous at tag locus k: exactly the offspring whose parents differ at bit k
    np.testing.assert_array_equal(
        chb[TAG0:TAG0 + NBITS], (bit[parents[:, 0]] != bit[parents[:, 1]]).sum(axis=0))
    # bit for bit on 3000 offspring rows (75 MB)
    pick = np.sort(rng.choice(B, 3000, replace=False))
    rows = O.unpack_genomes(dev.download_genomes(N + pick), L)        # [n, L, 2]
    np.testing.assert_array_equal(rows[:, TAG0:TAG0 + NBITS, 0], bit[parents[pick, 0]])
    np.testing.assert_array_equal(rows[:, TAG0:TAG0 + NBITS, 1], bit[parents[pick, 1]])
    assert rows.sum() == bit[parents[pick]].sum()
    dev.close()


@pytest.mark.parametrize('workload', ['c2', 'c3', 'c4_metric'])
def test_step_invariants_at_baseline_sizes(workload):
    """six main steps of BASELINE configs[1], [2] and the metric workload ([3] with 10^5 loci): N' = N + births - deaths, ids unique and
    ascending offspring ids, genome rows unique, positions on the landscape, cell
    order sorted, density bins sum to N, and the run is reproducible - through gnx_step and,
    with the same signature, through gnx_walk (at 10^5 individuals the device-driven step:
    counts on the device, one graph launch per step; tests/runtime/runtime_test.py:155-164
    is the loop it stands for)."""
    def run(via='step'):
        bench, nat, cfg, dev = _build(workload)
        for _ in range(3):
            dev.step(True, False)
        bench.setup_genomes(dev, cfg, 42)
        hist = []
        max_id = int(dev.download(nat.F_ID).max())
        for _ in range(6):
            n0 = dev.N
            if via == 'walk':
                dev.walk(1, False, True)
                assert [int(v[-1]) for v in dev.walk_history()][0] == n0
            else:
                dev.step(False, True)
            n1, b, d = dev.counts()
            assert n1 == n0 + b - d and b > 0 and d > 0
            ids = dev.download(nat.F_ID)
            assert ids.size == n1 and np.unique(ids).size == n1
            new = ids[ids > max_id]
            # this step's offspring ids form one block of b consecutive ids above every
            # id handed out before (the newest individuals of earlier steps may be dead)
            assert 0 < new.size <= b and new.max() - new.min() < b
            max_id = int(new.max())
            hist.append((n1, b, d))
            # integer conservation of the density counts (taken before the mortality)
            assert dev.get_bins(0).sum() == n0 + b
        x, y = dev.download(nat.F_X), dev.download(nat.F_Y)
        assert (x >= 0).all() and (x <= np.float32(cfg['W'] - 0.001)).all()
        assert (y >= 0).all() and (y <= np.float32(cfg['H'] - 0.001)).all()
        rows = dev.download(nat.F_GROW)
        assert np.unique(rows).size == rows.size and rows.min() >= 0
        c1, ch = dev.stats_locus_counts()
        f = c1 / (2.0 * dev.N)
        # start_p = 0.5 at N = 10^6: six generations of drift move no site far
        assert 0.49 < f.mean() < 0.51 and f.min() > 0.45 and f.max() < 0.55
        assert abs(ch.mean() / dev.N - 0.5) < 0.01
        sig = (hist, int(ids.sum()), int(np.bitwise_xor.reduce(ids)), float(x.sum()),
               c1[:64].tolist())
        dev.close()
        return sig
    sig = run()
    assert sig == run()
    assert sig == run('walk')


def _tiles_equal_one_device(cfg, grid, n_paths, n_burn, n_main, nbits, library=False):
    """A landscape of grid[0] x grid[1] tiles of `cfg`, once on one device and once as
    tiles (threads of this process, tests/_local_comm.py in place of RCCL, device-resident
    transport).  Genomes carry each founder's id (both homologues, an otherwise empty
    genome), so the two runs start from the same genomes; after burn-in and main steps
    with selection the two populations must be identical: ids, positions, ages,
    phenotypes, per-locus allele counts."""
    import threading
    import torch
    import bench
    from geonomics_amd import _native as nat
    from geonomics_amd.parallel import Comm, DeviceShard, TiledStepper
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _local_comm import Hub, LocalComm
    L = cfg['L']
    R, C = grid
    n_tiles = R * C
    tag0 = min(TAG0, L - 64)

    def tag_genomes(dev):
        dev.set_recomb_paths(bench.sparse_paths(n_paths, L, 43, dev.W64))
        dev.assign_genomes(np.zeros(L, np.int32))
        ids = dev.download(nat.F_ID)
        bit = (ids[:, None] >> np.arange(nbits)[None, :]) & 1
        who, which = np.nonzero(bit)
        for hom in (0, 1):
            dev.mutate(who.astype(np.int64), (tag0 + which).astype(np.int32),
                       np.full(who.size, hom, np.uint8))
        dev.set_z()

    def summary(dev):
        ids = dev.download(nat.F_ID)
        o = np.argsort(ids)
        c1, ch = dev.stats_locus_counts()
        return dict(ids=ids[o], x=dev.download(nat.F_X)[o], y=dev.download(nat.F_Y)[o],
                    age=dev.download(nat.F_AGE)[o], z=dev.download(nat.F_Z)[:, o],
                    c1=c1.astype(np.int64), ch=ch.astype(np.int64))

    def run(world):
        hub = Hub(world, library_group=library and world > 1)
        res, errs = [None] * world, []

        def body(rank):
            try:
                torch.cuda.set_device(0)
                if world > 1:
                    dev, _, _ = bench.build_device(cfg, seed=42, device=0, grid=grid, rank=rank,
                                                   cap_factor=1.7)
                else:       # the same landscape and the same initial population on one device
                    whole = dict(cfg, W=cfg['W'] * C, H=cfg['H'] * R, N=cfg['N'] * n_tiles)
                    dev, _, _ = bench.build_device(whole, seed=42, device=0, host_init=True)
                comm = LocalComm(hub, rank) if world > 1 else Comm(None)
                shard = DeviceShard(dev)
                # world 1: one tile that spans the whole landscape
                st = TiledStepper(shard, comm, cfg['W'] * C, cfg['H'] * R, 10.0, move=True,
                                  max_id=n_tiles * cfg['N'] - 1,
                                  grid=grid if world > 1 else (1, 1), fixed_births=1,
                                  use_library=library)
                assert st.v3 == library
                hist = [st.step(True, False) for _ in range(n_burn)]
                tag_genomes(dev)
                shard.has_genomes = True
                hist += [st.step(False, True) for _ in range(n_main)]
                res[rank] = (summary(dev), hist)
                dev.close()
            except BaseException as e:       # noqa: BLE001
                errs.append(e)
                hub.abort()
        ths = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        [t.start() for t in ths]
        [t.join(timeout=800) for t in ths]
        if errs:
            raise errs[0]
        return res

    one = run(1)[0]
    many = run(n_tiles)
    for r in many:
        assert r[1] == one[1]                             # global (N, births, deaths) per step
    ids = np.concatenate([r[0]['ids'] for r in many])
    o = np.argsort(ids)
    np.testing.assert_array_equal(ids[o], one[0]['ids'])
    for k in ('x', 'y', 'age'):
        np.testing.assert_array_equal(np.concatenate([r[0][k] for r in many])[o], one[0][k], k)
    np.testing.assert_array_equal(np.concatenate([r[0]['z'] for r in many], axis=1)[:, o],
                                  one[0]['z'])
    for k in ('c1', 'ch'):
        np.testing.assert_array_equal(sum(r[0][k] for r in many), one[0][k], k)
    assert one[0]['c1'][tag0:tag0 + nbits].sum() > cfg['N'] * n_tiles // 2
    assert len(ids) > 0.9 * cfg['N'] * n_tiles
    return one, many


def test_two_tiles_equal_one_device_at_metric_size():
    """a 4096 x 2048 landscape with 2 x 10^6 individuals and 10^5 loci: two 2048^2 tiles of
    the metric workload equal the one-device run"""
    import bench
    _tiles_equal_one_device(dict(bench.WORKLOADS['c4_metric']), (1, 2), n_paths=2000,
                            n_burn=2, n_main=3, nbits=21)


@pytest.mark.timeout(1500)
def test_c5_eight_tiles_equal_one_device_at_size():
    """BASELINE.json configs[4] (C5) AT SIZE: a 4096 x 4096 landscape tiled 2 x 4 (tiles
    1024 wide, 2048 tall), 10^7 individuals, conductance-surface movement, 4 selected
    traits; L = 10^4 so that the one-device run and the eight tiles fit one MI355X
    (SURVEY 8d allows it).  The eight in-process tiles exchange migrants with genomes,
    halos, pair lists, gametes and density bins exactly as eight ranks would, with plain
    device copies in place of RCCL; the result equals the one-device run bit for bit."""
    import bench
    cfg = dict(bench.WORKLOADS['c5_tile'], L=10_000, n_paths=2000)
    # (library=True: every step is ONE gnx_tile_step call, the library issues the exchanges -
    # csrc/gnx_comm.hip; the two-tile test above goes through the Python-driven protocol)
    one, many = _tiles_equal_one_device(cfg, (2, 4), n_paths=2000, n_burn=2, n_main=2,
                                        nbits=24, library=True)
    assert len(one[0]['ids']) > 9_000_000
    sizes = [len(r[0]['ids']) for r in many]
    assert min(sizes) > 500_000                           # every tile carries its share
