"""Spatial tiling of one species over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the
MI355X node, "gloo" in CPU rehearsals).  The landscape is cut into a uniform
R x C grid of tiles; rank = r * C + c owns tile (r, c) and the individuals in
it.  `TiledStepper.step` runs one time step of the reference's function queue
(age, movement, pop dynamics) with the exchanges listed in
csrc/gnx_tile.hip: migrants, halo, pair lists, gametes and the density bins.

The only collectives are tiny (pair lists, integer density bins, counters);
the byte movers are neighbour point-to-point messages (migrants with their
genomes, gametes), which is what xGMI's point-to-point links are good at.

The stepper talks to a *shard*: `DeviceShard` (the HIP library) in production;
the CPU rehearsals in tests/ drive the same stepper with the numpy oracle
(oracle/gnx_shard.py) as the shard.  Because every random draw is keyed by
individual id and every choice is order-independent, a tiled run reproduces the
single-tile run bit for bit (tests/test_tiling_cpu.py, tests/test_gpu_tiling.py).
"""
import os
import numpy as np

from . import _native as nat


def tile_grid(world):
    """R x C with R <= C, both powers of two where possible (8 -> 2 x 4)."""
    r = int(np.floor(np.sqrt(world)))
    while world % r:
        r -= 1
    return r, world // r


class DeviceShard:
    """The shard interface on top of a _native.Device."""

    def __init__(self, dev):
        self.dev = dev
        self.n_traits = dev.n_traits
        self.W64 = dev.W64
        self.has_genomes = False

    def tile_set(self, R, C, r, c):
        self.dev.tile_set(R, C, r, c)

    def age_and_move(self, move):
        if move:
            self.dev.age()
            self.dev.move()
        else:
            self.dev.age()

    def export_migrants(self):
        return self.dev.tile_export_migrants(self.has_genomes)

    def import_individuals(self, rec, z, geno):
        if rec.size:
            self.dev.tile_import(rec, z, geno if self.has_genomes else None)

    def export_halo(self):
        return self.dev.tile_export_halo()

    def import_ghosts(self, rec):
        if rec.size:
            self.dev.tile_import_ghosts(rec)

    def pairs(self, burn):
        return self.dev.tile_pairs(burn)

    def pair_info(self):
        return self.dev.tile_pair_info()

    def get_bins(self, which):
        return self.dev.get_bins(which)

    def set_bins(self, which, b):
        self.dev.set_bins(which, b)

    def offspring(self, burn, id_base, goff):
        return self.dev.tile_offspring(burn, id_base, goff)

    def get_requests(self):
        return self.dev.tile_get_requests()

    def serve_gametes(self, pids, keys, starts):
        return self.dev.tile_serve_gametes(pids, keys, starts)

    def put_gametes(self, child_k, data):
        self.dev.tile_put_gametes(child_k, data)

    def finish_births(self, burn):
        self.dev.tile_finish_births(burn)

    def die(self, burn, with_selection, have_pairs):
        self.dev.tile_die(burn, with_selection, have_pairs)

    def counts(self):
        return self.dev.counts()

    def set_max_id(self, v):
        self.dev.set_max_id(v)

    def advance_step(self):
        self.dev.step_index = self.dev.step_index + 1

    def synchronize(self):
        self.dev.synchronize()


class _StdoutToStderr:
    """file descriptor 1 points at stderr while the block runs: gloo announces every group it
    connects on stdout ("[Gloo] Rank 0 is connected to ...", from C++), and a benchmark's stdout
    is one JSON line"""

    def __enter__(self):
        import sys
        try:
            sys.stdout.flush()
            self._saved = os.dup(1)
            os.dup2(2, 1)
        except OSError:
            self._saved = None
        return self

    def __exit__(self, *exc):
        import sys
        if self._saved is not None:
            try:
                sys.stdout.flush()
            except Exception:
                pass
            os.dup2(self._saved, 1)
            os.close(self._saved)
        return False


class Comm:
    """Thin layer over torch.distributed: variable-size all-to-all of byte
    buffers by point-to-point messages, all-gather-v and sum all-reduce."""

    def __init__(self, dist=None):
        self.dist = dist
        # the host side of a step is a few tiny tensor ops; torch's intra-op thread
        # pool (one spinning thread per core) only burns the box's CPU quota and
        # stalls the launching thread
        try:
            import torch
            torch.set_num_threads(1)
        except Exception:
            pass
        if dist is None or not dist.is_initialized():
            self.rank, self.world, self.device = 0, 1, 'cpu'
            self.dist = None
            self.stream_ordered = True
        else:
            self.rank = dist.get_rank()
            self.world = dist.get_world_size()
            import os
            # tensors of the collectives live where the backend can move them; rehearsals
            # can force device tensors over another backend (GNX_COMM_DEVICE=cuda)
            self.device = os.environ.get('GNX_COMM_DEVICE') or (
                'cuda' if dist.get_backend() == 'nccl' else 'cpu')
            # RCCL's operations are ordered on streams; gloo touches the buffers from the host
            self.stream_ordered = dist.get_backend() == 'nccl'
            # the CPU side group of host_allgather is made HERE, by every rank (new_group is a
            # collective), and the ranks agree on whether they all have it: a rank that fell
            # back to the default group while the others use the side group would deadlock
            self._hgrp = None
            if dist.get_backend() != 'gloo':
                import torch
                ok = 1
                try:
                    with _StdoutToStderr():
                        self._hgrp = dist.new_group(backend='gloo')
                except Exception:
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    self._hgrp = None

    def _t(self, a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def allreduce_sum(self, a):
        if self.dist is None:
            return a
        t = self._t(a)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_var(self, t):
        """All ranks' 1-d int64 tensors (variable length) -> list of tensors that
        stay on the communication device."""
        import torch
        if self.dist is None:
            return [t]
        n = torch.tensor([t.numel()], dtype=torch.int64, device=self.device)
        ns = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(ns, n)
        ns = [int(v.item()) for v in ns]
        m = max(max(ns), 1)
        buf = torch.zeros(m, dtype=torch.int64, device=self.device)
        buf[:t.numel()] = t
        out = [torch.zeros_like(buf) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        return [o[:k] for o, k in zip(out, ns)]

    def allgather_i64(self, a):
        """All ranks' 1-d int64 arrays (variable length) -> list per rank."""
        a = np.ascontiguousarray(a, dtype=np.int64)
        if self.dist is None:
            return [a]
        import torch
        n = torch.tensor([a.size], dtype=torch.int64, device=self.device)
        ns = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(ns, n)
        ns = [int(v.item()) for v in ns]
        m = max(max(ns), 1)
        buf = torch.zeros(m, dtype=torch.int64, device=self.device)
        buf[:a.size] = self._t(a)
        out = [torch.zeros_like(buf) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        return [o[:k].cpu().numpy() for o, k in zip(out, ns)]

    # -- device-resident transport (backend nccl = RCCL) -----------------------------
    def count_matrix(self, counts):
        """every rank's per-destination counts -> int64 [src][dst] on the host"""
        import torch
        c = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64)).to(self.device)
        out = [torch.zeros_like(c) for _ in range(self.world)]
        self.dist.all_gather(out, c)
        return torch.stack(out).cpu().numpy()

    def exchange_dev(self, parts, mat):
        """parts: [(uint8 device tensor grouped by destination rank, bytes per
        element)]; mat[src][dst] = element counts.  One batch of isend/irecv for
        all parts; returns the received tensors, grouped by source rank."""
        import torch
        me = self.rank
        soff = np.concatenate([[0], np.cumsum(mat[me])])
        roff = np.concatenate([[0], np.cumsum(mat[:, me])])
        ops, recv = [], []
        for t, unit in parts:
            r = torch.empty(int(roff[-1]) * unit, dtype=torch.uint8, device=t.device)
            recv.append(r)
            for peer in range(self.world):
                if peer == me:
                    if mat[me, me]:
                        r[roff[me] * unit:roff[me + 1] * unit] = \
                            t[soff[me] * unit:soff[me + 1] * unit]
                    continue
                if mat[me, peer]:
                    ops.append(self.dist.P2POp(
                        self.dist.isend, t[soff[peer] * unit:soff[peer + 1] * unit], peer))
                if mat[peer, me]:
                    ops.append(self.dist.P2POp(
                        self.dist.irecv, r[roff[peer] * unit:roff[peer + 1] * unit], peer))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        # the library reads these buffers from its own stream
        torch.cuda.current_stream().synchronize()
        return recv

    def allreduce_dev_(self, t):
        import torch
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        torch.cuda.current_stream().synchronize()
        return t

    # -- tile2: nothing here blocks the host except the count all-gathers -----------------
    def host_allgather(self, a):
        """every rank's int64 vector (same length everywhere) -> int64 [world][len] on the
        host.  The vectors are a few dozen words; they travel through a CPU (gloo) group
        when there is one beside RCCL, so that no GPU stream is waited for."""
        a = np.ascontiguousarray(a, dtype=np.int64)
        if self.dist is None:
            return a[None, :]
        import torch
        grp = self._host_group()
        if grp is not None:
            t = torch.from_numpy(a)
            out = [torch.zeros_like(t) for _ in range(self.world)]
            self.dist.all_gather(out, t, group=grp)
            return torch.stack(out).numpy()
        t = torch.from_numpy(a).to(self.device)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return torch.stack(out).cpu().numpy()

    def _host_group(self):
        return getattr(self, '_hgrp', None)

    def exchange_multi(self, groups):
        """groups: [(parts, mat)] as for exchange_dev, all in ONE batch of isend / irecv and
        without a host synchronisation: the received tensors are valid for work enqueued on
        the current stream (req.wait() orders the stream behind RCCL, it does not block)."""
        import torch
        me = self.rank
        ops, out = [], []
        for parts, mat in groups:
            soff = np.concatenate([[0], np.cumsum(mat[me])])
            roff = np.concatenate([[0], np.cumsum(mat[:, me])])
            recv = []
            for t, unit in parts:
                r = torch.empty(int(roff[-1]) * unit, dtype=torch.uint8, device=t.device)
                recv.append(r)
                for peer in range(self.world):
                    if peer == me:
                        if mat[me, me]:
                            r[roff[me] * unit:roff[me + 1] * unit] = \
                                t[soff[me] * unit:soff[me + 1] * unit]
                        continue
                    if mat[me, peer]:
                        ops.append(self.dist.P2POp(
                            self.dist.isend, t[soff[peer] * unit:soff[peer + 1] * unit], peer))
                    if mat[peer, me]:
                        ops.append(self.dist.P2POp(
                            self.dist.irecv, r[roff[peer] * unit:roff[peer + 1] * unit], peer))
            out.append(recv)
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        return out

    def allgather_known(self, t, ns):
        """all ranks' 1-d int64 device tensors whose lengths ns[r] every rank knows already
        -> list of tensors (no count exchange, no host synchronisation)"""
        import torch
        m = max(int(max(ns)), 1)
        buf = torch.zeros(m, dtype=torch.int64, device=t.device)
        buf[:t.numel()] = t
        out = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        return [o[:int(k)] for o, k in zip(out, ns)]

    def allreduce_async_(self, t):
        """sum all-reduce in place, ordered on the current stream (no host wait)"""
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def alltoallv(self, send):
        """send: list (len world) of uint8 arrays -> list of received arrays."""
        if self.dist is None:
            return [send[0]]
        import torch
        counts = np.array([s.size for s in send], dtype=np.int64)
        mat = np.stack(self.allgather_i64(counts))          # mat[src][dst]
        recv = [None] * self.world
        ops = []
        keep = []
        for peer in range(self.world):
            if peer == self.rank:
                recv[peer] = send[peer]
                continue
            if mat[self.rank, peer] > 0:
                t = self._t(send[peer].view(np.uint8))
                keep.append(t)
                ops.append(self.dist.P2POp(self.dist.isend, t, peer))
            if mat[peer, self.rank] > 0:
                r = torch.empty(int(mat[peer, self.rank]), dtype=torch.uint8, device=self.device)
                recv[peer] = r
                ops.append(self.dist.P2POp(self.dist.irecv, r, peer))
            else:
                recv[peer] = np.zeros(0, np.uint8)
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        return [r.cpu().numpy() if hasattr(r, 'cpu') else r for r in recv]


class _DevMem:
    """Device memory owned by libgnxhip.so, exposed through the CUDA array
    interface so that torch can wrap it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1',
                                         'data': (int(ptr), False), 'version': 2}


def dev_bytes(ptr, nbytes):
    """uint8 tensor over [ptr, ptr + nbytes) of the current device (no copy)."""
    import torch
    if not nbytes:
        return torch.empty(0, dtype=torch.uint8, device='cuda')
    return torch.as_tensor(_DevMem(ptr, nbytes), device='cuda')


def _cat(chunks, dtype, shape_tail=()):
    chunks = [c for c in chunks if c is not None and c.size]
    if not chunks:
        return np.zeros((0,) + tuple(shape_tail), dtype=dtype)
    return np.concatenate(chunks)


def _everybody(comm, flag):
    """every rank says yes - over the CPU side group when there is one (no device, no RCCL call
    involved: this is what decides whether anybody enters one)"""
    if comm.world == 1:
        return bool(flag)
    mine = np.array([1 if flag else 0], np.int64)
    if getattr(comm, '_hgrp', None) is not None:
        return int(comm.host_allgather(mine).sum()) == comm.world
    return int(comm.allreduce_sum(mine)[0]) == comm.world


def _rccl_rendezvous(comm, dev):
    """The ranks of `comm` join the library's RCCL communicator, or none of them does:
    -> (joined 0 | 1, why).  ncclCommInitRank is a collective nobody can be called back from: a
    rank that cannot follow (no librccl, no id) must say so BEFORE anybody enters it.  So first
    every rank probes its librccl (gnx_comm_probe) and the ranks agree on it; then rank 0 makes
    the id, it travels once through the launcher's CPU side group (128 bytes, no device involved;
    a rank 0 that cannot make one says so with None) and the ranks agree that everybody holds
    it; only then do they join - and the join itself has a deadline (GNX_COMM_INIT_TIMEOUT_S):
    past it a rank says why and ends its process with a non-zero code (csrc/gnx_comm.hip).
    GNX_COMM_PROBE_FAIL=<rank>: fault injection for the tests."""
    import os
    ok, why = 1, None
    try:
        nat.comm_probe()
        if os.environ.get('GNX_COMM_PROBE_FAIL', '') == str(comm.rank):
            raise nat.GnxError('injected failure (GNX_COMM_PROBE_FAIL)')
    except Exception as e:
        ok, why = 0, e
    if not _everybody(comm, ok):
        return 0, why or 'librccl is not usable on some rank'
    box = [None]
    if comm.rank == 0:
        try:
            box[0] = nat.comm_unique_id()
        except Exception as e:
            why = e
    hgrp = getattr(comm, '_hgrp', None)
    if hgrp is not None:
        import torch
        comm.dist.broadcast_object_list(box, src=0, group=hgrp, device=torch.device('cpu'))
    else:
        comm.dist.broadcast_object_list(box, src=0)
    if not _everybody(comm, box[0] is not None):
        return 0, why or 'no communicator id reached some rank'
    dev.comm_init_rccl(box[0], comm.rank, comm.world)
    return 1, None


class TiledStepper:
    def __init__(self, shard, comm, W, H, mating_radius, move=True, max_id=-1,
                 grid=None, fixed_births=0, use_library=None):
        self.shard = shard
        self.comm = comm
        self.W, self.H = W, H
        self.R, self.C = grid if grid is not None else tile_grid(comm.world)
        assert self.R * self.C == comm.world
        assert W % self.C == 0 and H % self.R == 0, 'tiles must divide the landscape'
        self.tw, self.th = W // self.C, H // self.R
        self.r, self.c = divmod(comm.rank, self.C)
        # panmixia (mating_radius None, reference structs/species.py:2178-2194): every tile holds
        # everybody - its own individuals and all the others as ghosts - and the pairs whose focal
        # individual it owns; the Python-driven protocol with host-staged payloads (it need not be
        # fast: an all-to-all of the whole population's records every step)
        self.panmixia = mating_radius is None or float(mating_radius) < 0
        self.radius = -1.0 if self.panmixia else float(mating_radius)
        if comm.world > 1 and not self.panmixia:
            # the halo is made of whole hash cells (2 rings); it must come from the
            # adjacent tiles only
            cs = max(self.radius * (1.0 + 1e-9), max(W, H) / 2048.0)
            assert 3 * cs <= min(self.tw, self.th), (
                'tiles must be at least 3 hash cells (3 x mating_radius) wide')
        self.move = move
        self.fixed_births = int(fixed_births)   # > 0: every pair has this many births
        self.max_id = int(max_id)            # global maximum id handed out
        shard.tile_set(self.R, self.C, self.r, self.c)
        self.bytes_sent = 0
        import os
        # payloads stay in GPU memory when the transport can move it (RCCL);
        # gloo rehearsals stage through the host
        want = os.environ.get('GNX_TILE_TRANSPORT')
        self.dev_transport = (comm.world > 1 and hasattr(shard, 'dev') and
                              getattr(comm, 'device', 'cpu') == 'cuda')
        if want == 'host':
            self.dev_transport = False
        elif want == 'device':
            assert hasattr(shard, 'dev') and getattr(comm, 'device', 'cpu') == 'cuda', (
                'GNX_TILE_TRANSPORT=device needs the HIP shard and a device-capable backend')
            self.dev_transport = comm.world > 1
        if self.panmixia:
            self.dev_transport = False
        self.profile = bool(os.environ.get('GNX_TILE_PROFILE'))
        self.phase_s = {}
        self._t0 = 0.0
        # the device-driven protocol (tile2, include/gnx_hip.h): the HIP shard with a
        # transport that moves device memory, or a single tile
        self.v2 = (hasattr(shard, 'dev') and os.environ.get('GNX_TILE_V2', '1') != '0' and
                   (comm.world == 1 or self.dev_transport) and not self.panmixia)
        # One C call per step, the exchanges issued by the library itself on its own stream
        # (gnx_tile_step, csrc/gnx_comm.hip: grouped ncclSend / ncclRecv, KB-sized collectives,
        # no torch.distributed call and no Python between the phases of a step); with a per-step
        # hook (mutations, pedigree rows: after_births) two calls, the hook between them
        # (gnx_tile_step_begin / _end).  Poisson births included (round 5).  GNX_TILE_V3=0: off.
        # gnx_tile_step hands out offspring ids virtual tile by virtual tile (gnx_set_id_order 1);
        # the Python-driven protocols do so too when the device is set to that order (the Model
        # API's default where the landscape allows it), else in the (hash cell, focal id) order
        # of the whole landscape: two runs agree id by id when they use the same one.
        # use_library: None = GNX_TILE_V3 (default on).
        self.v3 = False
        if use_library is None:
            # (GNX_ID_ORDER=0 asks for the (hash cell, focal id) offspring order of the WHOLE
            # landscape everywhere: gnx_tile_step numbers tile-major, so that setting takes the
            # Python-driven protocol, which hands the pairs' order keys around - ADVICE r5)
            use_library = (os.environ.get('GNX_TILE_V3', '1') != '0' and
                           os.environ.get('GNX_ID_ORDER', '1') != '0')
        if self.v2 and use_library:
            self.v3 = self._join_library_comm()
        self._ext = None           # the library's stream as a torch stream
        self._evcache = os.environ.get('GNX_TILE_EVCACHE', '1') != '0'
        self._keep = []            # tensors the library reads until the end of the step
        self._n_start = None       # global population at the start of the coming step
        self._pre = None           # global population before the last step's deaths

    def _join_library_comm(self):
        import os
        dev, comm = self.shard.dev, self.comm
        # (tile-major offspring ids: a fixed 8 x 8 blocking of the landscape the tiles are unions of)
        if self.W % 8 or self.H % 8 or 8 % self.R or 8 % self.C:
            return False
        group = getattr(comm, 'local_group', None)
        rccl = comm.dist is not None and comm.dist.get_backend() == 'nccl'
        if comm.world > 1 and group is None and not rccl:
            return False

        everybody = lambda flag: _everybody(comm, flag)     # noqa: E731

        joined, why = 1, None
        try:
            if comm.world == 1:
                dev.comm_init_single()
            elif group is not None:               # tiles as threads of one process (tests)
                dev.comm_local_join(group, comm.rank)
            else:
                joined, why = _rccl_rendezvous(comm, dev)
        except Exception as e:           # (whatever it was: the ranks agree below)
            joined, why = 0, e

        def give_up(what):
            import sys
            if comm.rank == 0:
                print('geonomics_amd: the tiles step through torch.distributed, not through the '
                      'library\'s own communicator (%s: %s)' % (what, why), file=sys.stderr)
            return False
        # every rank or none: a rank that could not join must not leave the others waiting in
        # the library's collectives
        if not everybody(joined):
            if joined:
                dev.comm_free()
            return give_up('joining failed on some rank')
        # known words through the transport once, before a population depends on it; a transport
        # that does not deliver them is dropped by every rank (the step then goes through
        # TiledStepper._step_v2 and torch.distributed), loudly
        if os.environ.get('GNX_TILE_SELFTEST', '1') != '0':
            ok = True
            try:
                dev.comm_selftest()
            except nat.GnxError as e:
                ok, why = False, e
            if not everybody(ok):
                dev.comm_free()
                return give_up('self-test failed')
        dev.set_max_id(self.max_id)
        return True

    def rank_of(self, x, y):
        c = np.minimum(self.C - 1, (np.asarray(x) // self.tw).astype(np.int64))
        r = np.minimum(self.R - 1, (np.asarray(y) // self.th).astype(np.int64))
        return r * self.C + c

    # -- exchanges -------------------------------------------------------------------
    def _exchange_records(self, dest, rec, z, geno):
        w = self.comm.world
        nt, W64 = self.shard.n_traits, self.shard.W64
        send = []
        for p in range(w):
            sel = dest == p
            parts = [rec[sel].view(np.uint8).ravel()]
            if nt:
                parts.append(np.ascontiguousarray(z[sel]).view(np.uint8).ravel())
            if geno is not None:
                parts.append(np.ascontiguousarray(geno[sel]).view(np.uint8).ravel())
            send.append(np.concatenate(parts) if sel.any() else np.zeros(0, np.uint8))
        self.bytes_sent += sum(s.size for i, s in enumerate(send) if i != self.comm.rank)
        recv = self.comm.alltoallv(send)
        per = nat.IND_REC.itemsize + 4 * nt + (16 * W64 if geno is not None else 0)
        recs, zs, gs = [], [], []
        for p, buf in enumerate(recv):
            if p == self.comm.rank or buf.size == 0:
                continue
            n = buf.size // per
            o = 0
            recs.append(buf[o:o + n * nat.IND_REC.itemsize].view(nat.IND_REC))
            o += n * nat.IND_REC.itemsize
            if nt:
                zs.append(buf[o:o + 4 * nt * n].view(np.float32).reshape(n, nt))
                o += 4 * nt * n
            if geno is not None:
                gs.append(buf[o:o + 16 * W64 * n].view(np.uint64).reshape(n, 2, W64))
        return (_cat(recs, nat.IND_REC),
                _cat(zs, np.float32, (nt,)) if nt else None,
                _cat(gs, np.uint64, (2, W64)) if geno is not None else None)

    # -- device-resident exchanges (see csrc/gnx_tile.hip, "Device-resident transport")
    def _migrate_dev(self):
        dev = self.shard.dev
        geno = self.shard.has_genomes
        nt, W64 = self.shard.n_traits, self.shard.W64
        counts, (p_rec, p_z, p_g) = dev.tile_export_migrants_dev()
        self._tick('  mig export')
        assert counts[self.comm.rank] == 0
        mat = self.comm.count_matrix(counts)
        self._tick('  mig counts')
        n = int(counts.sum())
        parts = [(dev_bytes(p_rec, n * 32), 32)]
        if nt:
            parts.append((dev_bytes(p_z, n * 4 * nt), 4 * nt))
        if geno:
            parts.append((dev_bytes(p_g, n * 16 * W64), 16 * W64))
        self.bytes_sent += sum(t.numel() for t, _ in parts)
        got = self.comm.exchange_dev(parts, mat)
        self._tick('  mig exchange')
        m = int(mat[:, self.comm.rank].sum())
        if m:
            dev.tile_import_dev(m, got[0].data_ptr(), got[1].data_ptr() if nt else 0,
                                got[-1].data_ptr() if geno else 0)
        self._tick('  mig import')

    def _halo_dev(self):
        dev = self.shard.dev
        counts, p_rec = dev.tile_export_halo_dev()
        self._tick('  halo export')
        mat = self.comm.count_matrix(counts)
        n = int(counts.sum())
        self.bytes_sent += n * 32
        got = self.comm.exchange_dev([(dev_bytes(p_rec, n * 32), 32)], mat)
        self._tick('  halo exchange')
        m = int(mat[:, self.comm.rank].sum())
        if m:
            dev.tile_import_ghosts_dev(m, got[0].data_ptr())
        self._tick('  halo import')

    def _offspring_dev(self, burn):
        """pair order on the device: all-gather the pairs' order keys (hash cell << 40 |
        focal id, ascending on every tile), searchsorted, and hand the offsets to the
        library without leaving GPU memory"""
        import torch
        dev = self.shard.dev
        P, p_ids, p_nb = dev.tile_pair_ptrs()
        mine = dev_bytes(p_ids, P * 8).view(torch.int64)
        all_ids = self.comm.allgather_var(mine)
        fixed = self.fixed_births
        all_nb = None
        if not fixed:
            nb = dev_bytes(p_nb, P * 4).view(torch.int32).to(torch.int64) if P else mine
            all_nb = self.comm.allgather_var(nb)
        goff = torch.zeros(P, dtype=torch.int64, device=mine.device)
        total_births = total_pairs = 0
        for r in range(self.comm.world):
            li = all_ids[r]
            if fixed:
                if P:
                    goff += torch.searchsorted(li, mine) * int(fixed)
                total_births += int(fixed) * li.numel()
            else:
                cum = torch.zeros(li.numel() + 1, dtype=torch.int64, device=mine.device)
                if li.numel():
                    cum[1:] = torch.cumsum(all_nb[r], 0)
                if P:
                    goff += cum[torch.searchsorted(li, mine)]
                total_births += int(cum[-1].item())
            total_pairs += li.numel()
        torch.cuda.current_stream().synchronize()
        n_req = dev.tile_offspring_dev(burn, self.max_id + 1, goff.data_ptr() if P else 0)
        return n_req, total_births, total_pairs

    def _gametes_dev(self):
        dev = self.shard.dev
        W64 = self.shard.W64
        counts, p_req = dev.tile_group_requests()
        mat = self.comm.count_matrix(counts)
        n = int(counts.sum())
        got = self.comm.exchange_dev([(dev_bytes(p_req, n * 16), 16)], mat)
        m = int(mat[:, self.comm.rank].sum())
        p_out = dev.tile_serve_gametes_dev(m, got[0].data_ptr() if m else 0)
        self.bytes_sent += m * 8 * W64
        back = self.comm.exchange_dev([(dev_bytes(p_out, m * 8 * W64), 8 * W64)], mat.T.copy())
        if n:
            dev.tile_put_gametes_dev(n, back[0].data_ptr())

    def _bins_dev(self):
        import torch
        ptr, n = self.shard.dev.tile_bins_ptr()
        self.comm.allreduce_dev_(dev_bytes(ptr, n * 4).view(torch.int32))

    def _migrate(self):
        if self.dev_transport:
            return self._migrate_dev()
        rec, z, geno = self.shard.export_migrants()
        dest = self.rank_of(rec['x'], rec['y']) if rec.size else np.zeros(0, np.int64)
        if self.comm.world == 1:
            assert rec.size == 0
            return
        if geno is None and self.shard.has_genomes:
            geno = np.zeros((0, 2, self.shard.W64), np.uint64)
        rec2, z2, g2 = self._exchange_records(dest, rec, z, geno)
        self.shard.import_individuals(rec2, z2, g2)

    def _halo(self):
        if self.comm.world == 1:
            return
        if self.dev_transport:
            return self._halo_dev()
        rec = self.shard.export_halo()
        w = self.comm.world
        send = [np.zeros(0, np.uint8)] * w
        if self.panmixia:
            # everybody to everybody
            blob = rec.view(np.uint8).ravel()
            send = [blob if p != self.comm.rank else np.zeros(0, np.uint8) for p in range(w)]
            self.bytes_sent += blob.size * (w - 1)
            recv = self.comm.alltoallv(send)
            ghosts = _cat([b.view(nat.IND_REC) for p, b in enumerate(recv)
                           if p != self.comm.rank and b.size], nat.IND_REC)
            self.shard.import_ghosts(ghosts)
            return
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == 0 and dy == 0:
                    continue
                rr, cc = self.r + dy, self.c + dx
                if not (0 <= rr < self.R and 0 <= cc < self.C):
                    continue
                bit = 1 << ((dy + 1) * 3 + (dx + 1))
                sel = (rec['nbr_mask'] & bit) != 0
                peer = rr * self.C + cc
                if sel.any():
                    send[peer] = np.concatenate([send[peer], rec[sel].view(np.uint8).ravel()])
        self.bytes_sent += sum(s.size for s in send)
        recv = self.comm.alltoallv(send)
        ghosts = _cat([b.view(nat.IND_REC) for p, b in enumerate(recv)
                       if p != self.comm.rank and b.size], nat.IND_REC)
        # a corner individual reaches a tile through one message only, but make the
        # ghost list unique by id anyway
        if ghosts.size:
            _, first = np.unique(ghosts['id'], return_index=True)
            ghosts = ghosts[np.sort(first)]
        self.shard.import_ghosts(ghosts)

    def _pair_offsets(self):
        """Global offspring offset of every local pair: pairs are ordered by their key
        (hash cell << 40 | focal id: the canonical order of the cell-sorted population)
        over ALL tiles (each tile's list is already sorted), births are numbered in that
        order.  offset(v) = sum over tiles of the births of that tile's pairs with
        key < v (searchsorted on the comm device)."""
        import torch
        ids, nb = self.shard.pair_info()
        if self.comm.world == 1:        # one tile: the local order is the global order
            nb64 = nb.astype(np.int64)
            goff = np.concatenate([[0], np.cumsum(nb64)[:-1]]) if ids.size else nb64
            return goff.astype(np.int64), int(nb64.sum()), ids.size
        dev = self.comm.device
        mine = torch.from_numpy(ids).to(dev)
        all_ids = self.comm.allgather_var(mine)
        fixed = self.fixed_births
        all_nb = None if fixed else self.comm.allgather_var(
            torch.from_numpy(nb.astype(np.int64)).to(dev))
        goff = torch.zeros(ids.size, dtype=torch.int64, device=dev)
        total_births = 0
        total_pairs = 0
        for r in range(self.comm.world):
            li = all_ids[r]
            if fixed:
                if ids.size:
                    goff += torch.searchsorted(li, mine) * int(fixed)
                total_births += int(fixed) * li.numel()
            else:
                cum = torch.zeros(li.numel() + 1, dtype=torch.int64, device=dev)
                if li.numel():
                    cum[1:] = torch.cumsum(all_nb[r], 0)
                if ids.size:
                    goff += cum[torch.searchsorted(li, mine)]
                total_births += int(cum[-1].item())
            total_pairs += li.numel()
        return goff.cpu().numpy(), total_births, total_pairs

    def _gametes(self, n_req):
        w = self.comm.world
        if w == 1:
            return
        pid, ck, key, st, px, py = self.shard.get_requests()
        owner = self.rank_of(px, py) if n_req else np.zeros(0, np.int64)
        req_dt = np.dtype([('pid', np.int64), ('key', np.int32), ('start', np.int32)])
        send, order = [], []
        for p in range(w):
            sel = np.nonzero(owner == p)[0]
            order.append(sel)
            q = np.zeros(sel.size, req_dt)
            q['pid'], q['key'], q['start'] = pid[sel], key[sel], st[sel]
            send.append(q.view(np.uint8).ravel())
        recv = self.comm.alltoallv(send)
        back = []
        for p, buf in enumerate(recv):
            if p == self.comm.rank or buf.size == 0:
                back.append(np.zeros(0, np.uint8))
                continue
            q = buf.view(req_dt)
            data = self.shard.serve_gametes(q['pid'], q['key'], q['start'].astype(np.uint8))
            back.append(np.ascontiguousarray(data).view(np.uint8).ravel())
        self.bytes_sent += sum(b.size for b in back)
        got = self.comm.alltoallv(back)
        W64 = self.shard.W64
        for p, buf in enumerate(got):
            if p == self.comm.rank or buf.size == 0:
                continue
            self.shard.put_gametes(ck[order[p]], buf.view(np.uint64).reshape(-1, W64))

    # -- one time step -------------------------------------------------------------------
    def _tick(self, name):
        """host wall time per phase (GNX_TILE_PROFILE=1); the device is
        synchronised at each mark so the figures include kernel time"""
        if not self.profile:
            return
        import time
        sync = getattr(getattr(self.shard, 'dev', None), 'synchronize', None)
        if sync:
            sync()
        now = time.perf_counter()
        if name is not None:
            self.phase_s[name] = self.phase_s.get(name, 0.0) + now - self._t0
        self._t0 = now

    # -- tile2: one time step, three host waits -----------------------------------------
    def _lib_to_torch(self):
        """collectives enqueued from here on see what the library has enqueued so far: a
        stream dependency under RCCL; a transport that reads device memory from the host
        (the gloo rehearsals) needs the work done"""
        import torch
        if not getattr(self.comm, 'stream_ordered', True):
            self.shard.dev.synchronize()
            return
        # (one event per direction, recorded again at every hand-over: Stream.wait_stream makes
        # a new event each time, 30 us of host time a call and ten calls a step)
        if self._ext is None:
            self._ext = torch.cuda.ExternalStream(self.shard.dev.stream_ptr())
            self._ev_l2t = torch.cuda.Event()
            self._ev_t2l = torch.cuda.Event()
        if self._evcache:
            self._ev_l2t.record(self._ext)
            torch.cuda.current_stream().wait_event(self._ev_l2t)
        else:
            torch.cuda.current_stream().wait_stream(self._ext)

    def _torch_to_lib(self):
        """library work enqueued from here on sees what torch / RCCL have enqueued so far"""
        import torch
        if not getattr(self.comm, 'stream_ordered', True):
            torch.cuda.current_stream().synchronize()
            return
        if self._ext is None:
            self._ext = torch.cuda.ExternalStream(self.shard.dev.stream_ptr())
            self._ev_l2t = torch.cuda.Event()
            self._ev_t2l = torch.cuda.Event()
        if self._evcache:
            self._ev_t2l.record(torch.cuda.current_stream())
            self._ext.wait_event(self._ev_t2l)
        else:
            self._ext.wait_stream(torch.cuda.current_stream())

    def _step_v2(self, burn, with_selection, after_births, exact):
        import torch
        sh, dev, comm = self.shard, self.shard.dev, self.comm
        w, me = comm.world, comm.rank
        nt, W64 = sh.n_traits, sh.W64
        geno = sh.has_genomes
        self._tick(None)
        self._keep = []
        # 1. age + movement, routing (wait 1), ONE exchange: migrants and ghosts
        cnt = dev.tile2_move_route(self.move)
        self._tick('move + route')
        if w > 1:
            mats = comm.host_allgather(cnt.reshape(-1)).reshape(w, 2, w)
            m_mig, m_gh = mats[:, 0, :], mats[:, 1, :]
            p_rec, p_z, p_g, p_gh = dev.tile2_route_ptrs()
            n_mig, n_gh = int(cnt[0].sum()), int(cnt[1].sum())
            parts = [(dev_bytes(p_rec, n_mig * 32), 32)]
            if nt:
                parts.append((dev_bytes(p_z, n_mig * 4 * nt), 4 * nt))
            if geno:
                parts.append((dev_bytes(p_g, n_mig * 16 * W64), 16 * W64))
            ghosts = [(dev_bytes(p_gh, n_gh * 32), 32)]
            self.bytes_sent += sum(t.numel() for t, _ in parts) + n_gh * 32
            self._lib_to_torch()
            got_m, got_g = comm.exchange_multi([(parts, m_mig), (ghosts, m_gh)])
            self._keep += got_m + got_g
            self._torch_to_lib()
            dev.tile2_import(int(m_mig[:, me].sum()), got_m[0].data_ptr(),
                             got_m[1].data_ptr() if nt else 0,
                             got_m[-1].data_ptr() if geno else 0,
                             int(m_gh[:, me].sum()), got_g[0].data_ptr())
        self._tick('exchange + import')
        # 2. pairs (wait 2); pair order and gamete requests from ONE count all-gather
        P, B, req = dev.tile2_pairs(burn)
        self._tick('pairs')
        # births per pair when they are fixed (the species' parameters, the same on every
        # rank), else 0: Poisson counts travel with the pair keys
        lam = self.fixed_births or dev.births_fixed_lambda
        P_, p_ids, p_nb = dev.tile_pair_ptrs_nosync()
        tile_major = dev.id_order == 1
        if tile_major:
            # offspring ids virtual tile by virtual tile (gnx_set_id_order 1): this tile's births
            # per virtual tile travel with the counts, their exclusive sums are the bases - no
            # pair key leaves the tile
            vt_mine = dev.tile2_vt_counts()
            if w > 1:
                mat2 = comm.host_allgather(np.concatenate([[P, B], req, vt_mine]))
                Ps, Bs, m_req = mat2[:, 0], mat2[:, 1], mat2[:, 2:2 + w]
                total_births, total_pairs = int(Bs.sum()), int(Ps.sum())
                vt = mat2[:, 2 + w:].sum(0)
                dev.tile2_vt_bases(np.concatenate([[0], np.cumsum(vt)[:-1]]))
            else:
                total_births, total_pairs = B, P
            goff = None
        elif w > 1:
            mat2 = comm.host_allgather(np.concatenate([[P, B], req]))
            Ps, Bs, m_req = mat2[:, 0], mat2[:, 1], mat2[:, 2:]
            total_births, total_pairs = int(Bs.sum()), int(Ps.sum())
            mine = dev_bytes(p_ids, P * 8).view(torch.int64)
            self._lib_to_torch()
            all_ids = comm.allgather_known(mine, Ps)
            all_nb = None
            if not lam:
                nb = dev_bytes(p_nb, P * 4).view(torch.int32).to(torch.int64) if P else mine
                all_nb = comm.allgather_known(nb, Ps)
            goff = torch.zeros(max(P, 1), dtype=torch.int64, device=mine.device)[:P]
            for r in range(w):
                li = all_ids[r]
                if not P or not li.numel():
                    continue
                if lam:
                    goff += torch.searchsorted(li, mine) * int(lam)
                else:
                    cum = torch.zeros(li.numel() + 1, dtype=torch.int64, device=mine.device)
                    cum[1:] = torch.cumsum(all_nb[r], 0)
                    goff += cum[torch.searchsorted(li, mine)]
        else:
            # one tile: the pairs' global offsets are their local ones (the library's own
            # prefix sums), nothing to compute
            total_births, total_pairs = B, P
            goff = None
        if goff is not None:
            self._keep.append(goff)
            self._torch_to_lib()
        p_req = dev.tile2_offspring(burn, self.max_id + 1,
                                    goff.data_ptr() if (P and goff is not None) else 0)
        self.max_id += total_births
        sh.set_max_id(self.max_id)
        self._tick('offspring')
        # gametes of ghost mates: requests out, gametes back
        if w > 1 and not burn and geno:
            n_req = int(req.sum())
            self._lib_to_torch()
            (got_r,), = comm.exchange_multi([([(dev_bytes(p_req, n_req * 24), 24)], m_req)])
            self._keep.append(got_r)
            m = int(m_req[:, me].sum())
            self._torch_to_lib()
            p_out = dev.tile2_serve(m, got_r.data_ptr() if m else 0)
            self.bytes_sent += m * 8 * W64
            self._lib_to_torch()
            (back,), = comm.exchange_multi([([(dev_bytes(p_out, m * 8 * W64), 8 * W64)],
                                            m_req.T.copy())])
            self._keep.append(back)
            self._torch_to_lib()
            if n_req:
                dev.tile2_put(n_req, back.data_ptr())
        self._tick('gametes')
        ptr, n_words = dev.tile2_finish_births(burn)
        if after_births is not None and total_births > 0:
            after_births(self.max_id - total_births + 1, total_births)
        # ONE all-reduce: both density fields and the counters
        if w > 1:
            self._lib_to_torch()
            comm.allreduce_async_(dev_bytes(ptr, n_words * 4).view(torch.int32))
            self._torch_to_lib()
        self._tick('finish + all-reduce')
        # 3. densities, death probabilities, mortality (wait 3)
        n_pre, b_glob, d_prev = dev.tile2_die(burn, with_selection, total_pairs > 0)
        sh.advance_step()
        self._keep = []
        self._tick('die')
        if exact:
            n, b, d = sh.counts()
            tot = comm.allreduce_sum(np.array([n, b, d], dtype=np.int64)) if w > 1 else (n, b, d)
            return int(tot[0]), int(tot[1]), int(tot[2])
        # (N at the start of this step = N before the previous step's deaths - those deaths)
        n_start = (self._pre - d_prev) if self._pre is not None else n_pre - b_glob
        self._pre = n_pre
        return int(n_start), int(b_glob), int(d_prev)

    def walk(self, T, burn, with_selection, exact=True):
        """T time steps with nothing between them -> ((N, births, deaths) of the last step as
        step() reports them, sum over the steps of the global N at the start, sum of the births).
        Through the library (gnx_tile_walk: no compaction between the steps) when it drives the
        tiles; else T calls of step()."""
        if T <= 0:
            return (0, 0, 0), 0, 0
        if self.v3 and os.environ.get('GNX_TILE_WALK', '1') != '0':
            self.shard.set_max_id(self.max_id)
            last, n_sum, b_sum = self.shard.dev.tile_walk(T, burn, with_selection, exact)
            self.max_id += b_sum
            self.bytes_sent = self.shard.dev.comm_bytes_sent
            return last, n_sum, b_sum
        n_sum = b_sum = 0
        last = (0, 0, 0)
        for _ in range(T):
            last = self.step(burn, with_selection, exact=False if self.v2 else True)
            n_sum += last[0]
            b_sum += last[1]
        return last, n_sum, b_sum

    def step(self, burn, with_selection, after_births=None, exact=True):
        """one time step; `after_births(first_id, total_births)` runs once every
        offspring of the step has its genome and phenotype (mutations go there).
        Returns the global (N after the step, births, deaths).  With exact=False the
        device-driven protocol skips the collective those take and returns the counts
        that rode on the step's own all-reduce: (N at the START of the step, births,
        deaths of the PREVIOUS step)."""
        if self.v3:
            # (the library keeps the global maximum id while it drives the steps; an upload in
            # between - every tile its own share - leaves the tile's own maximum there)
            self.shard.set_max_id(self.max_id)
        if self.v3 and after_births is None:
            n, b, d = self.shard.dev.tile_step(burn, with_selection, exact)
            self.max_id += b
            self.bytes_sent = self.shard.dev.comm_bytes_sent
            return n, b, d
        if self.v3:
            # the step in two library calls, the host's work on the newborns between them
            dev = self.shard.dev
            first, total = dev.tile_step_begin(burn)
            self.max_id = first + total - 1
            # the host's work on the newborns may fail on ONE rank (a mutation or pedigree hook):
            # that rank must not simply leave - the others would wait inside the step's all-reduce
            # for ever (only the communicator's init has a deadline).  The ranks agree over the
            # CPU side group that every hook succeeded before anybody enters tile_step_end;
            # otherwise all of them drop the half-done step (gnx_tile_step_abort) and raise.
            hook_err = None
            if total > 0:
                try:
                    after_births(first, total)
                except Exception as e:          # noqa: BLE001 - re-raised below, on every rank
                    hook_err = e
            if not _everybody(self.comm, hook_err is None):
                dev.tile_step_abort()
                if hook_err is not None:
                    raise hook_err
                raise RuntimeError('a tiled step was abandoned after its births: the host hook '
                                   '(mutation / pedigree) failed on another rank')
            n, b, d = dev.tile_step_end(burn, with_selection, exact)
            self.bytes_sent = dev.comm_bytes_sent
            return n, b, d
        if self.v2:
            return self._step_v2(burn, with_selection, after_births, exact)
        sh = self.shard
        self._tick(None)
        sh.age_and_move(self.move)
        self._tick('age+move')
        self._migrate()
        self._tick('migrants')
        self._halo()
        self._tick('halo')
        P, B = sh.pairs(burn)
        self._tick('pairs')
        if hasattr(sh, 'dev') and sh.dev.id_order == 1:
            # tile-major offspring ids: 64 birth counts per tile instead of every pair's key
            vt = self.comm.allreduce_sum(np.concatenate([sh.dev.tile2_vt_counts(), [P]]))
            total_births, total_pairs = int(vt[:64].sum()), int(vt[64])
            if self.comm.world > 1:
                sh.dev.tile2_vt_bases(np.concatenate([[0], np.cumsum(vt[:64])[:-1]]))
            if self.dev_transport:
                n_req = sh.dev.tile_offspring_dev(burn, self.max_id + 1, 0)
            else:
                n_req = sh.offspring(burn, self.max_id + 1, None)
        elif self.dev_transport:
            n_req, total_births, total_pairs = self._offspring_dev(burn)
        else:
            goff, total_births, total_pairs = self._pair_offsets()
            self._tick('pair order')
            n_req = sh.offspring(burn, self.max_id + 1, goff)
        self.max_id += total_births
        sh.set_max_id(self.max_id)
        self._tick('offspring+crossover')
        if not burn and sh.has_genomes:
            if self.dev_transport:
                self._gametes_dev()
            else:
                self._gametes(n_req)
        self._tick('gametes')
        sh.finish_births(burn)
        if after_births is not None and total_births > 0:
            after_births(self.max_id - total_births + 1, total_births)
        # one all-reduce for both density fields (individuals, pair midpoints)
        if self.dev_transport:
            self._bins_dev()
        else:
            b0, b1 = sh.get_bins(0), sh.get_bins(1)
            both = self.comm.allreduce_sum(np.concatenate([b0, b1]))
            sh.set_bins(0, both[:b0.size])
            sh.set_bins(1, both[b0.size:])
        self._tick('phenotype + bins')
        sh.die(burn, with_selection, total_pairs > 0)
        sh.advance_step()
        n, b, d = sh.counts()
        tot = self.comm.allreduce_sum(np.array([n, b, d], dtype=np.int64))
        self._tick('die + counts')
        return int(tot[0]), int(tot[1]), int(tot[2])
