"""Rehearsal of the multi-GPU tiling protocol on CPU: torch.distributed with the
gloo backend, world_size 2 and 4, the numpy oracle standing in for the device.
A tiled run must reproduce the single-tile run bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, '_tiling_worker.py')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(backend, shard, world, steps, out, mode='fixed'):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, WORKER, backend, shard, str(world), str(r),
                               str(port), str(steps), out, mode]) for r in range(world)]
    from _procs import wait_all
    rcs = wait_all(procs)
    assert rcs == [0] * world, rcs
    return np.load(out)


@pytest.mark.parametrize('world,mode', [(2, 'fixed'), (4, 'poisson'), (2, 'panmixia')])
def test_tiled_oracle_run_is_bit_identical_to_single_tile(tmp_path, world, mode):
    steps = 8
    one = launch('gloo', 'oracle', 1, steps, str(tmp_path / 'one.npz'), mode)
    many = launch('gloo', 'oracle', world, steps, str(tmp_path / 'many.npz'), mode)
    assert one['hist'].tolist() == many['hist'].tolist()
    for k in ('ids', 'x', 'y', 'age', 'z', 'geno'):
        np.testing.assert_array_equal(one[k], many[k], err_msg=k)
    assert len(set(one['ids'].tolist())) == len(one['ids']) > 500
    # something actually crossed the tile borders
    assert many['bytes_sent'] > 0 and one['bytes_sent'] == 0
    # genomes carry real variation (the check is not vacuous)
    assert 0.3 < np.unpackbits(one['geno'].view(np.uint8)).mean() * (one['geno'].shape[2] * 64 / 192) < 0.7


@pytest.mark.parametrize('world', [2, 3])
def test_tile2_comm_helpers(world):
    """the collective helpers of the device-driven tile protocol (host count all-gather, the
    one-batch multi-group exchange incl. messages to oneself, the padded all-gather of known
    lengths, the in-place all-reduce) over gloo on CPU tensors"""
    port = free_port()
    worker = os.path.join(HERE, '_comm_worker.py')
    procs = [subprocess.Popen([sys.executable, worker, str(world), str(r), str(port)])
             for r in range(world)]
    from _procs import wait_all
    assert wait_all(procs) == [0] * world


@pytest.mark.parametrize('scenario', ['probe', 'id'])
def test_rccl_rendezvous_is_all_or_none(scenario):
    """Before any rank enters the library's ncclCommInitRank - a collective nobody can be called
    back from - the ranks agree over the CPU group that every one of them can load librccl
    (gnx_comm_probe) and holds rank 0's id.  One rank whose probe fails, or a rank 0 that cannot
    make an id: NO rank asks its device to join, every rank returns (0, why) and they stay in
    step (three gloo ranks)."""
    port = free_port()
    worker = os.path.join(HERE, '_rendezvous_worker.py')
    procs = [subprocess.Popen([sys.executable, worker, '3', str(r), str(port), scenario])
             for r in range(3)]
    from _procs import wait_all
    assert wait_all(procs, timeout=120) == [0, 0, 0]
