"""Expert-parallel process groups, mirroring ``deepspeed.utils.groups._create_expert_and_data_parallel``
(SURVEY 8a row a14): with world size W and expert-parallel size ``ep``

* expert-parallel group i   = ranks [i*ep, (i+1)*ep)          (consecutive ranks; experts sharded inside)
* expert-data-parallel group j = ranks {j, j+ep, j+2ep, ...}  (replicas of the same expert shard)

Groups are keyed by the name ``ep_size_{ep}`` -- the string is load-bearing: the reference trainer looks for
``'ep_size'`` in optimizer param-group names (``moellava/train/llava_trainer.py:452``).
On ROCm the ``nccl`` backend is RCCL; the all-to-all inside an EP group runs over xGMI peer links.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch.distributed as dist

_EXPERT_PARALLEL_GROUP: Dict[str, object] = {}
_EXPERT_DATA_PARALLEL_GROUP: Dict[str, object] = {}


def expert_parallel_ranks(world_size: int, ep_size: int) -> Tuple[List[List[int]], List[List[int]]]:
    """Pure function (unit-testable without a process group): (ep groups, expert-data-parallel groups)."""
    if ep_size < 1 or world_size % ep_size != 0:
        raise ValueError(f"world size {world_size} is not divisible by expert parallel size {ep_size}")
    ep_groups = [list(range(i * ep_size, (i + 1) * ep_size)) for i in range(world_size // ep_size)]
    edp_groups = [list(range(j, world_size, ep_size)) for j in range(ep_size)]
    return ep_groups, edp_groups


def create_expert_and_data_parallel(ep_size: int) -> None:
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised: expert parallelism (ep_size > 1) needs "
                           "init_process_group(backend='nccl') first (RCCL on ROCm)")
    name = f"ep_size_{ep_size}"
    if name in _EXPERT_PARALLEL_GROUP:
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    ep_groups, edp_groups = expert_parallel_ranks(world, ep_size)
    # every rank must create every group, in the same order
    for ranks in edp_groups:
        g = dist.new_group(ranks)
        if rank in ranks:
            _EXPERT_DATA_PARALLEL_GROUP[name] = g
    for ranks in ep_groups:
        g = dist.new_group(ranks)
        if rank in ranks:
            _EXPERT_PARALLEL_GROUP[name] = g


def get_expert_parallel_group(name: str):
    if name not in _EXPERT_PARALLEL_GROUP:
        raise KeyError(f"expert parallel group {name} was not created")
    return _EXPERT_PARALLEL_GROUP[name]


def get_expert_data_parallel_group(name: str):
    if name not in _EXPERT_DATA_PARALLEL_GROUP:
        raise KeyError(f"expert data parallel group {name} was not created")
    return _EXPERT_DATA_PARALLEL_GROUP[name]


def reset() -> None:
    _EXPERT_PARALLEL_GROUP.clear()
    _EXPERT_DATA_PARALLEL_GROUP.clear()
