"""Tiling on the HIP path: several processes (one per tile) share the one GPU of
the test box and talk over gloo; the code path is the production one except for
the transport (RCCL on a multi-GPU node).  A tiled run must reproduce the
single-tile run bit for bit, and the single-tile stepper must equal gnx_step."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_tiling_cpu import launch          # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world,mode', [(2, 'fixed'), (4, 'poisson'), (2, 'panmixia'),
                                        (4, 'panmixia')])
def test_tiled_device_run_is_bit_identical_to_single_tile(tmp_path, world, mode):
    steps = 8
    one = launch('gloo', 'device', 1, steps, str(tmp_path / 'one.npz'), mode)
    many = launch('gloo', 'device', world, steps, str(tmp_path / 'many.npz'), mode)
    assert one['hist'].tolist() == many['hist'].tolist()
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(one[k], many[k], err_msg=k)
    assert many['bytes_sent'] > 0
    assert len(one['ids']) > 500


def test_tiled_device_matches_oracle_protocol(tmp_path):
    """device tiles vs oracle tiles: same ids alive, same genomes for the
    individuals both runs share (positions differ by cosf/logf ulps, which can
    flip a rare decision - so compare the first steps only)."""
    dev = launch('gloo', 'device', 2, 5, str(tmp_path / 'dev.npz'))
    ora = launch('gloo', 'oracle', 2, 5, str(tmp_path / 'ora.npz'))
    common = np.intersect1d(dev['ids'], ora['ids'])
    assert len(common) > 0.97 * max(len(dev['ids']), len(ora['ids']))
    di = np.searchsorted(dev['ids'], common)
    oi = np.searchsorted(ora['ids'], common)
    same = (dev['geno'][di] == ora['geno'][oi]).all(axis=(1, 2))
    assert same.mean() > 0.97
    assert np.abs(dev['x'][di] - ora['x'][oi]).max() < 1e-3


def test_single_tile_stepper_equals_gnx_step():
    """TiledStepper at world 1 and the fused gnx_step give the same run."""
    from _tiling_worker import config, make_device_shard
    from geonomics_amd import _native as nat
    from geonomics_amd.parallel import Comm, TiledStepper
    import gnx_oracle as O
    cfg = config()
    sh, dev_a = make_device_shard(cfg)
    _, dev_b = make_device_shard(cfg)
    stepper = TiledStepper(sh, Comm(None), cfg['W'], cfg['H'], cfg['radius'],
                           max_id=cfg['N0'] - 1)
    # (the stepper hands its steps to gnx_tile_step when it can - fixed births - which numbers
    # the offspring virtual tile by virtual tile: gnx_step in the same order)
    if stepper.v3:
        dev_b.set_id_order(1)
    for t in range(7):
        burn = t < 3
        if t == 3:
            n = O.starting_mutation_counts(dev_a.N, np.full(cfg['L'], 0.5))
            dev_a.assign_genomes(n)
            dev_b.assign_genomes(n)
            sh.has_genomes = True
        stepper.step(burn, not burn)
        dev_b.step(burn, not burn)
        assert dev_a.counts() == dev_b.counts()
    for f in (nat.F_ID, nat.F_X, nat.F_Y, nat.F_AGE):
        a, b = dev_a.download(f), dev_b.download(f)
        oa, ob = np.argsort(dev_a.download(nat.F_ID)), np.argsort(dev_b.download(nat.F_ID))
        np.testing.assert_array_equal(a[oa], b[ob])
    ga = dev_a.download(nat.F_GENO)[np.argsort(dev_a.download(nat.F_ID))]
    gb = dev_b.download(nat.F_GENO)[np.argsort(dev_b.download(nat.F_ID))]
    np.testing.assert_array_equal(ga, gb)


def test_panmixia_on_tiles_equals_gnx_step(tmp_path):
    """mating_radius None (reference structs/species.py:2178-2194) on a tiled landscape: two tiles
    over gloo (every tile holds everybody's record, the pairs of its own focal individuals) give
    the population gnx_step gives on ONE device - ids, positions, ages, phenotypes, genotypes"""
    from _tiling_worker import config, make_device_shard
    from geonomics_amd import _native as nat
    import gnx_oracle as O
    cfg = config()
    cfg['radius'] = None
    steps = 8
    many = launch('gloo', 'device', 2, steps, str(tmp_path / 'many.npz'), 'panmixia')
    _, dev = make_device_shard(cfg)
    hist = []
    for t in range(steps):
        burn = t < cfg['n_burn']
        if t == cfg['n_burn']:
            ids = np.sort(dev.download(nat.F_ID))
            n_site = O.starting_mutation_counts(dev.N, np.full(cfg['L'], 0.5))
            G = O.starting_genomes(dev.N, cfg['L'], n_site, cfg['seed'])
            dev.upload_genomes(G[np.searchsorted(ids, dev.download(nat.F_ID))])
        dev.step(burn, not burn)
        hist.append(dev.counts())
    assert [list(h) for h in hist] == many['hist'].tolist()
    o = np.argsort(dev.download(nat.F_ID))
    np.testing.assert_array_equal(dev.download(nat.F_ID)[o], many['ids'])
    np.testing.assert_array_equal(dev.download(nat.F_X)[o], many['x'])
    np.testing.assert_array_equal(dev.download(nat.F_Y)[o], many['y'])
    np.testing.assert_array_equal(dev.download(nat.F_AGE)[o], many['age'])
    np.testing.assert_array_equal(dev.download(nat.F_Z)[0][o], many['z'])
    np.testing.assert_array_equal(dev.download(nat.F_GENO)[o], many['geno'])
    assert len(many['ids']) > 500 and many['bytes_sent'] > 0


# ---- device-resident transport (what RCCL carries on a multi-GPU node) -----------------
def _run_threads(world, steps, fixed, library=False, expect_v3=None, walk=False):
    """`world` tiles as threads of this process, LocalComm between them; the
    same schedule as _tiling_worker.run; returns the gathered final population."""
    import threading
    import torch
    from _local_comm import Hub, LocalComm
    from _tiling_worker import config, make_device_shard
    from geonomics_amd import _native as nat
    from geonomics_amd.parallel import Comm, TiledStepper
    import gnx_oracle as O
    cfg = config()
    # library: the tiles meet inside libgnxhip.so as well and the stepper hands every step to
    # gnx_tile_step (one C call per step, exchanges issued by the library)
    hub = Hub(world, library_group=library and world > 1)
    res, errs = [None] * world, []

    def body(rank):
        try:
            torch.cuda.set_device(0)
            comm = LocalComm(hub, rank) if world > 1 else Comm(None)
            shard, dev = make_device_shard(cfg, fixed)
            stepper = TiledStepper(shard, comm, cfg['W'], cfg['H'], cfg['radius'], move=True,
                                   max_id=cfg['N0'] - 1, fixed_births=1 if fixed else 0,
                                   use_library=library)
            assert stepper.dev_transport == (world > 1)
            assert stepper.v3 == (bool(library) if expect_v3 is None else expect_v3)
            shard.export_migrants()
            hist = []
            for t in range(steps):
                burn = t < cfg['n_burn']
                if t == cfg['n_burn']:
                    n_tot = comm.allreduce_sum(np.array([shard.counts()[0]], np.int64))[0]
                    ids = dev.download(nat.F_ID)
                    all_ids = np.sort(np.concatenate(comm.allgather_i64(ids)))
                    n_site = O.starting_mutation_counts(int(n_tot), np.full(cfg['L'], 0.5))
                    G = O.starting_genomes(int(n_tot), cfg['L'], n_site, cfg['seed'])
                    dev.upload_genomes(G[np.searchsorted(all_ids, ids)])
                    shard.has_genomes = True
                    if walk:
                        # the main steps in ONE call: gnx_tile_walk, no compaction between them
                        last, n_sum, b_sum = stepper.walk(steps - t, False, True)
                        hist.append((last, n_sum, b_sum))
                        break
                hist.append(stepper.step(burn, not burn))
            res[rank] = dict(ids=dev.download(nat.F_ID), x=dev.download(nat.F_X),
                             y=dev.download(nat.F_Y), age=dev.download(nat.F_AGE),
                             z=dev.download(nat.F_Z)[0], geno=dev.download(nat.F_GENO),
                             hist=hist, bytes_sent=stepper.bytes_sent)
            dev.close()
        except BaseException as e:       # noqa: BLE001 - re-raised in the main thread
            errs.append(e)
            hub.abort()

    ths = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    if errs:
        raise errs[0]
    ids = np.concatenate([r['ids'] for r in res])
    order = np.argsort(ids)
    out = {k: np.concatenate([r[k] for r in res])[order]
           for k in ('ids', 'x', 'y', 'age', 'z', 'geno')}
    out['hist'] = res[0]['hist'] if walk else np.array(res[0]['hist'])
    out['bytes_sent'] = sum(r['bytes_sent'] for r in res)
    return out


@pytest.mark.parametrize('world,fixed', [(2, True), (4, False), (8, True)])
def test_device_resident_transport_is_bit_identical(world, fixed):
    steps = 8
    one = _run_threads(1, steps, fixed)
    many = _run_threads(world, steps, fixed)
    assert one['hist'].tolist() == many['hist'].tolist()
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(one[k], many[k], err_msg=k)
    assert many['bytes_sent'] > 0 and len(one['ids']) > 500


@pytest.mark.parametrize('world,fixed', [(2, True), (4, True), (8, True), (2, False), (4, False)])
def test_library_tile_step_is_bit_identical(world, fixed):
    """gnx_tile_step: the whole tiled step in ONE call into the library, which issues the
    exchanges itself (csrc/gnx_comm.hip).  On this one-GPU box the tiles are threads of one
    process and the transport is the library's local one (device copies behind a barrier of
    the threads; on a node it is grouped ncclSend / ncclRecv on the same code path): the tiled
    run equals the one-tile run bit for bit, and equals the run the Python-driven protocol
    (TiledStepper._step_v2) gives.  fixed = False: Poisson births (ops/mating.py:120-126) - the
    virtual tiles' BIRTH counts travel and the pairs' ranks are in births (round 5; round 4 sent
    such species through the Python-driven protocol)."""
    steps = 8
    one = _run_threads(1, steps, fixed, library=True)
    many = _run_threads(world, steps, fixed, library=True)
    assert one['hist'].tolist() == many['hist'].tolist()
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(one[k], many[k], err_msg=k)
    assert len(one['ids']) > 500


def test_library_tile_step_with_half_radius_cells(monkeypatch):
    """GNX_CELL_DIV=2 (opt-in): hash cells of half a mating radius, so the halo of two radii is a
    ring of FOUR cells (csrc/gnx_tile.hip: ring = 2 * cell_ref) and the candidate block 5 x 5 -
    the tiled run still equals the one-tile run bit for bit."""
    monkeypatch.setenv('GNX_CELL_DIV', '2')
    steps = 8
    one = _run_threads(1, steps, True, library=True)
    many = _run_threads(4, steps, True, library=True)
    assert one['hist'].tolist() == many['hist'].tolist()
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(one[k], many[k], err_msg=k)
    assert len(one['ids']) > 500


@pytest.mark.parametrize('world,fixed', [(2, True), (4, True), (2, False), (1, True)])
def test_tile_walk_without_compactions_is_bit_identical(world, fixed, monkeypatch):
    """gnx_tile_walk: the main steps of the run in ONE call, the dead left in their slots between
    two steps (the next movement and routing skip them, the imports go behind the uncompacted
    stretch, the cell sort removes them with the emigrants) - the same population as step by
    step with a compaction in every step, id by id, and the same sums of N and births."""
    steps = 9
    ref = _run_threads(world, steps, fixed, library=True)
    monkeypatch.setenv('GNX_TILE_LAZY', '1')       # (off by default: no gain measured on one GPU)
    got = _run_threads(world, steps, fixed, library=True, walk=True)
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(ref[k], got[k], err_msg=k)
    assert len(ref['ids']) > 500
    last, n_sum, b_sum = got['hist'][-1]
    main = ref['hist'][3:]                          # (N after the step, births, deaths), exact
    assert tuple(int(v) for v in last) == tuple(int(v) for v in main[-1])
    assert b_sum == int(main[:, 1].sum())
    # N at the start of main step t = N after step t - 1
    assert n_sum == int(ref['hist'][2:-1, 0].sum())


def test_virtual_tile_boundaries_are_exact():
    """the virtual tile of a pair was classified with a reciprocal,
    (int)(x * (8 / W)), while tile ownership is exact on the integer boundaries - for W = 1000 the
    float just below x = 500 went to virtual-tile column 4 although tile 0 owns it, two tiles
    numbered pairs of one virtual tile independently and two offspring got the same id.  A
    landscape 1000 wide (8 / W is not a power of two), a third of the individuals sitting on
    nextafter(boundary, 0) of the virtual tiles' and the tiles' boundaries, nobody moving: two
    tiles through the library's protocol hand out unique ids and equal the one-device run."""
    import threading
    import torch
    from _local_comm import Hub, LocalComm
    from geonomics_amd import _native as nat
    from geonomics_amd.parallel import Comm, DeviceShard, TiledStepper
    W, H, N = 1000, 40, 6000
    rng = np.random.RandomState(11)
    x = (rng.rand(N) * W).astype(np.float32)
    y = (rng.rand(N) * H).astype(np.float32)
    edge = rng.rand(N) < 0.35
    x[edge] = np.nextafter(np.float32(125) * rng.randint(1, 8, edge.sum()).astype(np.float32),
                           np.float32(0))
    edge_y = rng.rand(N) < 0.2
    y[edge_y] = np.nextafter(np.float32(5) * rng.randint(1, 8, edge_y.sum()).astype(np.float32),
                             np.float32(0))
    assert (x < W).all() and (y < H).all() and (x == np.nextafter(np.float32(500), np.float32(0))).sum() > 100
    rasts = np.ones((2, H, W), np.float32)

    def make():
        dev = nat.Device(W, H, 2, L=0, n_traits=0, cap_inds=32768, cap_rows=16, seed=5, device=0)
        dev.upload_rasters(rasts)
        dev.set_species_params(nat.default_species_params(mating_radius=3.0, K_factor=0.2,
                                                          move=0))
        return dev

    def run(world):
        hub = Hub(world, library_group=world > 1)
        res, errs = [None] * world, []

        def body(rank):
            try:
                torch.cuda.set_device(0)
                comm = LocalComm(hub, rank) if world > 1 else Comm(None)
                dev = make()
                stepper = TiledStepper(DeviceShard(dev), comm, W, H, 3.0, move=False, max_id=N - 1,
                                       fixed_births=1, use_library=True)
                assert stepper.v3
                mine = stepper.rank_of(x, y) == rank
                dev.upload_population(x[mine], y[mine], np.zeros(mine.sum()), np.zeros(mine.sum()),
                                      np.arange(N)[mine])
                hist = [stepper.step(True, False) for _ in range(5)]
                res[rank] = dict(ids=dev.download(nat.F_ID), x=dev.download(nat.F_X),
                                 y=dev.download(nat.F_Y), hist=hist)
                dev.close()
            except BaseException as e:       # noqa: BLE001
                errs.append(e)
                hub.abort()

        ths = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        if errs:
            raise errs[0]
        ids = np.concatenate([r['ids'] for r in res])
        o = np.argsort(ids, kind='stable')
        return dict(ids=ids[o], x=np.concatenate([r['x'] for r in res])[o],
                    y=np.concatenate([r['y'] for r in res])[o], hist=res[0]['hist'])

    one, two = run(1), run(2)
    assert len(np.unique(two['ids'])) == len(two['ids']), 'two offspring share an id'
    assert len(one['ids']) > 3000 and one['hist'][-1][1] > 100        # births happened
    assert one['hist'] == two['hist']
    for k in ('ids', 'x', 'y'):
        np.testing.assert_array_equal(one[k], two[k], err_msg=k)


def test_failed_self_test_drops_the_library_path_on_every_rank(monkeypatch):
    """the transport's self-test fails on ONE of two ranks (fault injection in
    gnx_comm_selftest): both drop the library's communicator, agree on it, and step through
    TiledStepper._step_v2 - the run equals one that never asked for the library"""
    ref = _run_threads(2, 5, True, library=False)
    monkeypatch.setenv('GNX_COMM_SELFTEST_FAIL', '1')
    got = _run_threads(2, 5, True, library=True, expect_v3=False)
    for k in ('ids', 'x', 'y', 'age', 'geno'):
        np.testing.assert_array_equal(got[k], ref[k])


def test_rccl_world1_through_the_library():
    """a one-rank RCCL communicator made by the library itself (ncclGetUniqueId,
    ncclCommInitRank through the dlopen-ed librccl) and steps through gnx_tile_step: what a
    one-GPU box can run of the RCCL path; equals gnx_step."""
    from _tiling_worker import config, make_device_shard
    from geonomics_amd import _native as nat
    cfg = config()
    sh, dev_a = make_device_shard(cfg)
    _, dev_b = make_device_shard(cfg)
    dev_a.tile_set(1, 1, 0, 0)
    # the RCCL calls themselves, not the one-rank shortcuts: ncclAllGather / ncclAllReduce and
    # a group of ncclSend / ncclRecv (to itself) carry the self-test's words and the step's
    import os
    os.environ['GNX_COMM_FORCE_RCCL'] = '1'
    try:
        dev_a.comm_init_rccl(nat.comm_unique_id(), 0, 1)
    finally:
        del os.environ['GNX_COMM_FORCE_RCCL']
    dev_a.comm_selftest()
    dev_a.set_max_id(cfg['N0'] - 1)
    dev_b.set_id_order(1)            # gnx_tile_step numbers offspring virtual tile by virtual tile
    import gnx_oracle as O
    for t in range(11):
        burn = t < 3
        if t == 3:
            # main steps with genomes and selection: the count gathers and the densities' sum of
            # every one of them go through ncclAllGather / ncclAllReduce
            n = O.starting_mutation_counts(dev_a.N, np.full(cfg['L'], 0.5))
            dev_a.assign_genomes(n)
            dev_b.assign_genomes(n)
        if t % 2:
            n, b, d = dev_a.tile_step(burn, not burn, True)
        else:                         # (... and in two calls, as with a host hook in between)
            first, total = dev_a.tile_step_begin(burn)
            assert total == dev_a.counts()[1] and first + total - 1 >= cfg['N0'] - 1
            n, b, d = dev_a.tile_step_end(burn, not burn, True)
        dev_b.step(burn, not burn)
        assert (n, b, d) == dev_b.counts()
    oa, ob = np.argsort(dev_a.download(nat.F_ID)), np.argsort(dev_b.download(nat.F_ID))
    for f in (nat.F_ID, nat.F_X, nat.F_Y, nat.F_AGE):
        np.testing.assert_array_equal(dev_a.download(f)[oa], dev_b.download(f)[ob])
    np.testing.assert_array_equal(dev_a.download(nat.F_Z)[:, oa], dev_b.download(nat.F_Z)[:, ob])
    np.testing.assert_array_equal(dev_a.download(nat.F_GENO)[oa], dev_b.download(nat.F_GENO)[ob])
    assert dev_a.counts()[0] > 500
    dev_a.close()
    dev_b.close()


def test_tile_major_offspring_ids_on_one_device():
    """gnx_set_id_order(1): offspring ids are handed out virtual tile by virtual tile (8 x 8
    over the landscape, row-major), inside a virtual tile in the (hash cell, focal id) order of
    the pairs - so the focal parents of the children, taken in id order, walk through the
    virtual tiles in ascending order, every id is handed out once, and the same births happen
    as in the default order (only their numbering differs)."""
    from _tiling_worker import config, make_device_shard
    from geonomics_amd import _native as nat
    cfg = config()
    _, a = make_device_shard(cfg)
    _, b = make_device_shard(cfg)
    b.set_id_order(1)
    for t in range(3):
        ids = b.download(nat.F_ID)
        xb, yb = b.download(nat.F_X), b.download(nat.F_Y)
        for dev in (a, b):
            dev.age()
            dev.move()
        ids = b.download(nat.F_ID)
        xb, yb = b.download(nat.F_X), b.download(nat.F_Y)
        a.pop_dynamics_mate(True)
        b.pop_dynamics_mate(True)
        ca, pa, _, _, xya = a.last_births(with_gametes=False)
        cb, pb, _, _, xyb = b.last_births(with_gametes=False)
        assert len(cb) > 100
        assert sorted(cb.tolist()) == list(range(int(cb.min()), int(cb.min()) + len(cb)))
        if t == 0:      # (same ids so far: the same pairs in both devices, differently numbered;
            #             from here on the draws - keyed by id - part ways)
            key = lambda par: sorted(map(tuple, par.tolist()))
            assert key(pa) == key(pb) and sorted(cb.tolist()) == sorted(ca.tolist())
        # focal parents in the children's id order: ascending virtual tiles
        o = np.argsort(cb)
        slot = {int(i): k for k, i in enumerate(ids.tolist())}
        fo = np.array([slot[int(p)] for p in pb[o, 0]])
        vt = (np.minimum(7, (yb[fo] * 8 / cfg['H']).astype(int)) * 8 +
              np.minimum(7, (xb[fo] * 8 / cfg['W']).astype(int)))
        assert (np.diff(vt) >= 0).all() and len(set(vt.tolist())) > 8
        for dev in (a, b):
            dev.pop_dynamics_die(True, False)
            dev.step_index = dev.step_index + 1
    a.close()
    b.close()


def test_device_transport_equals_host_transport(tmp_path):
    """the same tiled run through host-staged payloads (gloo processes)"""
    host = launch('gloo', 'device', 2, 8, str(tmp_path / 'h.npz'), 'fixed')
    devt = _run_threads(2, 8, True)
    assert host['hist'].tolist() == devt['hist'].tolist()
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(host[k], devt[k], err_msg=k)


def test_nccl_backend_world1_wraps_library_memory():
    """RCCL refuses two ranks on one GPU, so the multi-rank RCCL path cannot run
    on this box; a 1-rank group still checks what is specific to it: tensors
    over library-owned device memory handed to RCCL collectives."""
    import torch
    import torch.distributed as dist
    from _tiling_worker import config, make_device_shard
    from geonomics_amd.parallel import Comm, dev_bytes
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        comm = Comm(dist)
        assert comm.device == 'cuda'
        sh, dev = make_device_shard(config())
        dev.tile_set(1, 1, 0, 0)
        dev.tile_pairs(True)
        sh.finish_births(True)
        ptr, n = dev.tile_bins_ptr()
        before = np.concatenate([dev.get_bins(0), dev.get_bins(1)])
        t = dev_bytes(ptr, n * 4).view(torch.int32)
        np.testing.assert_array_equal(t.cpu().numpy(), before)
        comm.allreduce_dev_(t)                       # sum over 1 rank: unchanged, in place
        np.testing.assert_array_equal(np.concatenate([dev.get_bins(0), dev.get_bins(1)]), before)
        assert before[:n // 2].sum() == dev.N
        mat = comm.count_matrix(np.array([0]))
        assert mat.tolist() == [[0]]
        P, p_ids, _ = dev.tile_pair_ptrs()
        ids = comm.allgather_var(dev_bytes(p_ids, P * 8).view(torch.int64))[0]
        np.testing.assert_array_equal(ids.cpu().numpy(), dev.tile_pair_info()[0])
        dev.close()
    finally:
        dist.destroy_process_group()


# ---- the Model API over several ranks (structs/tiled.py) -----------------------------------
def _run_model(tmp_path, world, traits, tag, extra=()):
    import subprocess
    from test_tiling_cpu import free_port
    out = str(tmp_path / ('%s.npz' % tag))
    wd = tmp_path / ('wd_' + tag)
    wd.mkdir()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_tiled_model_worker.py')
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        if world > 1:
            env.update({'WORLD_SIZE': str(world), 'RANK': str(r), 'LOCAL_RANK': '0',
                        'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                        'GNX_DIST_BACKEND': 'gloo'})
        else:
            for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
                env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, worker, out, str(int(traits)), str(wd),
                                       *extra], env=env))
    from _procs import wait_all
    assert wait_all(procs) == [0] * world
    return np.load(out), wd


def test_model_over_two_ranks_equals_single_process_when_neutral(tmp_path):
    """the unchanged model script under torch.distributed: a neutral species' demography
    and positions do not depend on genotypes, so the whole trajectory equals the
    single-GPU run; genomes differ only by which homologues got the starting 1-alleles"""
    one, _ = _run_model(tmp_path, 1, False, 'one')
    two, wd = _run_model(tmp_path, 2, False, 'two')
    assert int(two['world']) == 2
    for k in ('Nt', 'births', 'deaths', 'nburn', 'ids'):
        np.testing.assert_array_equal(one[k], two[k], err_msg=k)
    np.testing.assert_array_equal(one['xy'], two['xy'])
    np.testing.assert_allclose(one['N_rast'], two['N_rast'], rtol=1e-12, atol=1e-12)
    # Species._calc_density over the tiles: everybody's positions, the same raster on every rank
    np.testing.assert_allclose(one['dens'], two['dens'], rtol=1e-12, atol=1e-12)
    assert one['dens'].shape == one['N_rast'].shape and one['dens'].max() > 0
    # exact global starting counts: round(2 N p) ones per site (structs/genome.py:1124-1130)
    for r in (one, two):
        assert int(r['n0']) == int(r['n_at_assign'])
        np.testing.assert_array_equal(r['site_counts0'], np.full(48, int(r['n0'])))
    assert two['g'].shape == one['g'].shape and 0.3 < two['g'].mean() < 0.7
    # linkage statistics over the tiles: global chromosome counts, the oracle's r^2
    import gnx_oracle as O
    exp = O.stats_ld(two['g'][:, ::3, :])
    fin = np.isfinite(exp)
    np.testing.assert_allclose(two['ld'][fin], exp[fin], rtol=1e-9, atol=1e-13)
    assert two['het'].shape == (48,) and 0.3 < two['het'].mean() < 0.7
    # rank 0 wrote the files, once (Model.walk leaves mod.it at -1, as the reference does)
    base = wd / 'GNX_mod-api_test' / 'it--1' / 'spp-spp_0'
    names = sorted(os.listdir(base))
    assert 'mod-api_test_it--1_spp-spp_0_HET.csv' in names
    assert 'mod-api_test_it--1_t-14_spp-spp_0.vcf' in names
    vcf = open(base / 'mod-api_test_it--1_t-14_spp-spp_0.vcf').read().splitlines()
    ids = [int(v) for v in vcf[3].split('\t')[9:]]
    assert len(ids) == 20 and set(ids) <= set(two['ids'].tolist())
    col = {i: k for k, i in enumerate(two['ids'].tolist())}
    got = np.array([[[int(c) for c in cell.split('|')] for cell in l.split('\t')[9:]]
                    for l in vcf[4:]])
    np.testing.assert_array_equal(got, np.transpose(two['g'][[col[i] for i in ids]], (1, 0, 2)))


def test_model_over_two_ranks_with_panmixia(tmp_path):
    """mating_radius None through the Model API (reference structs/species.py:2178-2194): the
    unchanged model script over two ranks gives the single-process trajectory - TiledSpecies
    raised NotImplementedError until round 6"""
    one, _ = _run_model(tmp_path, 1, False, 'one', extra=('panmixia',))
    two, _ = _run_model(tmp_path, 2, False, 'two', extra=('panmixia',))
    assert int(two['world']) == 2 and int(two['v3']) == 0
    for k in ('Nt', 'births', 'deaths', 'nburn', 'ids'):
        np.testing.assert_array_equal(one[k], two[k], err_msg=k)
    np.testing.assert_array_equal(one['xy'], two['xy'])
    assert len(two['ids']) > 100 and two['births'].sum() > 100


def test_model_over_two_ranks_with_selection(tmp_path):
    one, _ = _run_model(tmp_path, 1, True, 'one')
    two, _ = _run_model(tmp_path, 2, True, 'two')
    np.testing.assert_array_equal(one['Nt'][:int(one['nburn'])], two['Nt'][:int(two['nburn'])])
    assert abs(one['Nt'][-10:].mean() - two['Nt'][-10:].mean()) < 0.15 * one['Nt'][-10:].mean()
    assert two['z'].shape == (len(two['ids']), 2)
    # phenotypes follow the gathered genotypes (z = 0.5 + sum(gt * alpha), monogenic z = gt)
    assert np.isfinite(two['z']).all() and 0 <= two['z'][:, 1].min() and two['z'][:, 1].max() <= 1


def test_model_over_two_ranks_with_mutation(tmp_path):
    """neutral loci start at 0 and mutate (mu_neut > 0): every 1-allele in the population
    is a mutation drawn during the run, and the two-rank run must place exactly the
    mutations of the single-GPU run - every rank draws the same list from the host
    generator and applies those whose offspring it owns (structs/tiled.py)"""
    one, _ = _run_model(tmp_path, 1, False, 'one', extra=('mutate',))
    two, _ = _run_model(tmp_path, 2, False, 'two', extra=('mutate',))
    for k in ('Nt', 'births', 'deaths', 'ids'):
        np.testing.assert_array_equal(one[k], two[k], err_msg=k)
    assert one['site_counts0'].sum() == 0 and two['site_counts0'].sum() == 0
    assert one['g'].sum() > 0                       # mutations happened and were inherited
    np.testing.assert_array_equal(one['g'], two['g'])
    # 'use_tskit': the pedigree recorded on one GPU and over two ranks reproduces the
    # device genotypes (edges + mutation rows, structs/pedigree.py)
    assert int(one['ped_ok']) == 1 and int(two['ped_ok']) == 1


def _run_model_threads(tmp_path, world, traits, tag, **kw):
    """the ranks of a Model run as THREADS of this process, each with its own device handle on
    the one GPU and the in-process communicator (tests/_local_comm.py) handed to its Model: the
    tiles meet inside libgnxhip.so and every step goes through the library's own protocol
    (gnx_tile_step / _begin + _end) - what runs under RCCL on a node"""
    import threading
    import torch
    from _local_comm import Hub, LocalComm
    from _tiled_model_worker import run_model
    from geonomics_amd.sim import model as M
    wd = tmp_path / ('wd_' + tag)
    wd.mkdir()
    cwd = os.getcwd()
    os.chdir(wd)
    hub = Hub(world, library_group=True)
    res, errs = [None] * world, []

    def body(rank):
        try:
            torch.cuda.set_device(0)
            M._rehearsal.comm = LocalComm(hub, rank)
            res[rank] = run_model(traits, rank=rank, world=world, **kw)
        except BaseException as e:       # noqa: BLE001 - re-raised in the main thread
            errs.append(e)
            hub.abort()
        finally:
            M._rehearsal.comm = None

    ths = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    try:
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=600)
    finally:
        os.chdir(cwd)
    if errs:
        raise errs[0]
    return res, wd


@pytest.mark.parametrize('variant', ['neutral', 'mutate', 'poisson', 'selection'])
def test_model_over_two_tiles_through_the_library(tmp_path, variant):
    """Model.walk over two ranks with every step inside the library's tile protocol
    (stepper.v3): Nt, births, deaths, ids and positions equal the one-process run - which hands
    out the same tile-major offspring ids -, with mutations (the host's hook between
    gnx_tile_step_begin and _end places the one-process run's mutations: genotypes equal),
    with Poisson births (the virtual tiles' BIRTH counts travel), with selection"""
    kw = dict(mutate=variant == 'mutate', poisson=variant == 'poisson')
    traits = variant == 'selection'
    one, _ = _run_model(tmp_path, 1, traits, 'one',
                        extra=tuple(k for k, v in kw.items() if v))
    two, _ = _run_model_threads(tmp_path, 2, traits, 'two', **kw)
    for r in two:
        assert r['v3'] == 1 and r['id_order'] == 1, 'the steps did not go through the library'
    assert int(one['id_order']) == 1
    a, b = two
    for k in ('Nt', 'births', 'deaths', 'ids', 'xy', 'g'):          # accessors are global
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    if variant == 'selection':
        # (which homologues got the starting 1-alleles differs between one and two tiles:
        # selection then acts on different genotypes - the burn-in is the same run)
        nb = int(one['nburn'])
        np.testing.assert_array_equal(one['Nt'][:nb], a['Nt'][:nb])
        assert abs(one['Nt'][-10:].mean() - a['Nt'][-10:].mean()) < 0.15 * one['Nt'][-10:].mean()
        assert np.isfinite(a['z']).all() and a['z'].shape == (len(a['ids']), 2)
        return
    for k in ('Nt', 'births', 'deaths', 'nburn', 'ids'):
        np.testing.assert_array_equal(one[k], a[k], err_msg=k)
    np.testing.assert_array_equal(one['xy'], a['xy'])
    np.testing.assert_allclose(one['N_rast'], a['N_rast'], rtol=1e-12, atol=1e-12)
    if variant == 'mutate':
        assert one['g'].sum() > 0
        np.testing.assert_array_equal(one['g'], a['g'])
        assert int(one['ped_ok']) == 1 and a['ped_ok'] == 1
    if variant == 'poisson':
        assert (one['births'] > 0).any()


def test_device_transport_across_processes(tmp_path, monkeypatch):
    """the device-resident exchanges between separate PROCESSES (separate address spaces,
    torch.distributed point-to-point and collectives on device tensors that wrap the
    library's memory), carried by gloo on the 1-GPU box - what RCCL carries on a node"""
    one = launch('gloo', 'device', 1, 8, str(tmp_path / 'one.npz'), 'poisson')
    monkeypatch.setenv('GNX_COMM_DEVICE', 'cuda')
    many = launch('gloo', 'device', 4, 8, str(tmp_path / 'many.npz'), 'poisson')
    assert int(many['dev_transport']) == 1 and int(one['dev_transport']) == 0
    assert one['hist'].tolist() == many['hist'].tolist()
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(one[k], many[k], err_msg=k)
    assert many['bytes_sent'] > 0
