"""In-process stand-in for torch.distributed: the ranks are THREADS of one
process, each with its own libgnxhip handle on the one GPU, and `LocalComm`
moves device tensors between them with plain device copies.  It implements the
interface TiledStepper uses for the device-resident transport (the one RCCL
serves on a multi-GPU node), so the whole device path - grouping by
destination, device-address imports, on-device pair order, gamete service -
runs on a 1-GPU box; only the RCCL calls themselves are replaced."""
import threading

import numpy as np


class Hub:
    def __init__(self, world, library_group=False):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        # library_group: the tiles also meet inside libgnxhip.so (gnx_comm_local_*), so that
        # TiledStepper can hand the whole step to gnx_tile_step
        self.group = None
        if library_group:
            from geonomics_amd import _native as nat
            self.group = nat.comm_local_create(world)

    def abort(self):
        self.barrier.abort()
        if self.group is not None:
            from geonomics_amd import _native as nat
            nat.comm_local_abort(self.group)


class LocalComm:
    device = 'cuda'

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world = hub, rank, hub.world
        self.dist = None
        self.local_group = hub.group

    def _swap(self, obj):
        self.hub.slots[self.rank] = obj
        self.hub.barrier.wait()
        out = list(self.hub.slots)
        self.hub.barrier.wait()
        return out

    # like RCCL: operations are ordered on streams, nothing waits for the device.  Every
    # thread's torch work goes to the one default stream of the process, so "every rank has
    # ENQUEUED its copies" (a barrier of the threads) is all a producer needs to know before
    # it orders its own stream behind the default stream (TiledStepper._torch_to_lib).
    # GNX_LOCALCOMM_SYNC=1: drain the device at every collective (round 2's behaviour).
    stream_ordered = True

    def _done(self):
        """nobody reuses a posted buffer before every rank has copied from it"""
        import os
        import torch
        if os.environ.get('GNX_LOCALCOMM_SYNC'):
            torch.cuda.synchronize()
        self.hub.barrier.wait()

    # host-side small collectives
    def allreduce_sum(self, a):
        return np.sum(np.stack(self._swap(np.asarray(a).copy())), axis=0)

    def allgather_i64(self, a):
        return [x.copy() for x in self._swap(np.ascontiguousarray(a, dtype=np.int64))]

    def count_matrix(self, counts):
        return np.stack(self._swap(np.asarray(counts, dtype=np.int64).copy()))

    # device-side
    def allgather_var(self, t):
        out = [x.clone() for x in self._swap(t)]
        self._done()
        return out

    def allreduce_dev_(self, t):
        import torch
        posted = self._swap(t)
        total = torch.stack([p.clone() for p in posted]).sum(0).to(t.dtype)
        self._done()
        t.copy_(total)
        import os
        if os.environ.get('GNX_LOCALCOMM_SYNC'):
            torch.cuda.synchronize()
        return t

    def exchange_dev(self, parts, mat):
        import torch
        me = self.rank
        posted = self._swap([t for t, _ in parts])
        recv = []
        for k, (t, unit) in enumerate(parts):
            chunks = []
            for peer in range(self.world):
                off = int(mat[peer, :me].sum())
                n = int(mat[peer, me])
                chunks.append(posted[peer][k][off * unit:(off + n) * unit])
            recv.append(torch.cat(chunks).clone() if chunks else
                        torch.empty(0, dtype=torch.uint8, device='cuda'))
        self._done()
        return recv

    def alltoallv(self, send):
        got = self._swap([np.asarray(s).copy() for s in send])
        return [got[p][self.rank] for p in range(self.world)]

    # -- tile2 (parallel.py: Comm.host_allgather / exchange_multi / allgather_known) ------
    def host_allgather(self, a):
        return np.stack(self._swap(np.ascontiguousarray(a, dtype=np.int64).copy()))

    def exchange_multi(self, groups):
        return [self.exchange_dev(parts, mat) for parts, mat in groups]

    def allgather_known(self, t, ns):
        return self.allgather_var(t)

    def allreduce_async_(self, t):
        return self.allreduce_dev_(t)
