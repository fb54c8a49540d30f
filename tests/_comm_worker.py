"""Worker for tests/test_tiling_cpu.py::test_tile2_comm_helpers: the collective helpers of the
device-driven tile protocol (geonomics_amd/parallel.py: Comm.host_allgather, exchange_multi,
allgather_known, allreduce_async_) over gloo on CPU tensors, world_size ranks.

    python tests/_comm_worker.py <world> <rank> <port>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(world, rank, port):
    import torch
    import torch.distributed as dist
    from geonomics_amd.parallel import Comm
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    comm = Comm(dist)
    assert comm.world == world and comm.rank == rank and not comm.stream_ordered
    # counts: every rank's vector, in rank order, on the host
    mine = np.arange(5, dtype=np.int64) + 100 * rank
    got = comm.host_allgather(mine)
    assert got.shape == (world, 5)
    for r in range(world):
        np.testing.assert_array_equal(got[r], np.arange(5) + 100 * r)
    # two groups of variable-size messages in one batch; a group may send to itself
    rng = np.random.RandomState(7)                       # the same matrices on every rank
    mat_a = rng.randint(0, 4, (world, world))
    mat_b = rng.randint(0, 3, (world, world))
    np.fill_diagonal(mat_a, 0)

    def payload(mat, unit, tag):
        """what `src` sends to each `dst`: unit bytes per element, value = f(src, dst, k)"""
        out = {}
        for src in range(world):
            parts = []
            for dst in range(world):
                for k in range(mat[src, dst]):
                    parts.append(np.full(unit, (tag + 7 * src + 3 * dst + k) % 251, np.uint8))
            out[src] = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        return out

    pa, pb = payload(mat_a, 8, 1), payload(mat_b, 24, 5)
    ta, tb = torch.from_numpy(pa[rank].copy()), torch.from_numpy(pb[rank].copy())
    (ra,), (rb,) = comm.exchange_multi([([(ta, 8)], mat_a), ([(tb, 24)], mat_b)])

    def expect(mat, unit, tag):
        parts = []
        for src in range(world):
            for k in range(mat[src, rank]):
                parts.append(np.full(unit, (tag + 7 * src + 3 * rank + k) % 251, np.uint8))
        return np.concatenate(parts) if parts else np.zeros(0, np.uint8)

    np.testing.assert_array_equal(ra.numpy(), expect(mat_a, 8, 1))
    np.testing.assert_array_equal(rb.numpy(), expect(mat_b, 24, 5))
    # keys of known lengths, padded all-gather
    ns = [3 + 2 * r for r in range(world)]
    t = torch.arange(ns[rank], dtype=torch.int64) + 1000 * rank
    allk = comm.allgather_known(t, ns)
    for r in range(world):
        np.testing.assert_array_equal(allk[r].numpy(), np.arange(ns[r]) + 1000 * r)
    # in-place sum
    v = torch.full((6,), rank + 1, dtype=torch.int32)
    comm.allreduce_async_(v)
    assert (v == world * (world + 1) // 2).all()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
