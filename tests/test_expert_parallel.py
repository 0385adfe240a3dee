"""Expert-parallel path: world_size-2 gloo rehearsal on the CPU, and the real kernels with two ranks
sharing one GPU (gloo group, payload staged through the host)."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ep_worker  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(mode, kind, world=2):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(ep_worker.run, args=(world, _free_port(), mode, kind, d), nprocs=world, join=True)
        oks = sorted(os.listdir(d))
        assert oks == [f"ok_{r}" for r in range(world)], oks
        return [open(os.path.join(d, f)).read() for f in oks]


@pytest.mark.parametrize("kind", ["stablelm", "phi"])
def test_expert_parallel_exchange_cpu_gloo(kind):
    info = _spawn("cpu", kind)
    assert all("dropped=" in i for i in info)


def test_expert_parallel_two_groups_of_two_cpu_gloo():
    """world 4 = 2 expert-parallel groups x 2 ranks (the layout bench.py uses at N = 8: EP 4 x 2 replicas): every rank
    creates every group, a rank's expert shard is rank % ep, the exchange stays inside its group."""
    info = _spawn("cpu", "stablelm", world=4)
    assert len(info) == 4


def test_default_chunk_count_is_the_same_on_every_rank_cpu_gloo():
    """MOE_EP_CHUNKS unset, skewed per-rank token counts around the rule's thresholds, a bandwidth
    override on one rank only -- the ranks must agree on the number of exchanges."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(ep_worker.run_chunk_agreement, args=(2, _free_port(), d), nprocs=2, join=True)
        assert sorted(os.listdir(d)) == ["ok_0", "ok_1"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["stablelm", "phi"])
def test_expert_parallel_two_ranks_one_gpu(kind):
    info = _spawn("gpu", kind)
    print(info)


@pytest.mark.gpu
def test_exchange_path_on_rccl_single_rank():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(ep_worker.run_single_rank_rccl, args=(_free_port(), d), nprocs=1, join=True)
        assert os.listdir(d) == ["ok_0"]
