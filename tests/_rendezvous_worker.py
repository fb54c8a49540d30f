"""Worker for tests/test_tiling_cpu.py::test_rccl_rendezvous_is_all_or_none: the agreement that
precedes the library's ncclCommInitRank (geonomics_amd/parallel.py: _rccl_rendezvous) over gloo,
with a stand-in device that must never be asked to join.

    python tests/_rendezvous_worker.py <world> <rank> <port> <scenario: probe | id>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class NoDevice:
    def comm_init_rccl(self, *a):
        raise AssertionError('a rank entered the collective init although another cannot follow')


def main(world, rank, port, scenario):
    import torch.distributed as dist
    from geonomics_amd import _native as nat
    from geonomics_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    comm = parallel.Comm(dist)
    if scenario == 'probe':
        # the last rank's librccl "cannot be loaded"
        os.environ['GNX_COMM_PROBE_FAIL'] = str(world - 1)
    else:
        # every probe succeeds, rank 0 cannot make the id: the others learn it from the None
        nat.comm_probe = lambda: None

        def no_id():
            raise nat.GnxError('no id (test)')
        nat.comm_unique_id = no_id
    joined, why = parallel._rccl_rendezvous(comm, NoDevice())
    assert joined == 0 and why is not None, (joined, why)
    # and the ranks are still in step with each other
    assert parallel._everybody(comm, True) and not parallel._everybody(comm, rank != 0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
