"""``MoE``: drop-in for ``deepspeed.moe.layer.MoE`` as the reference uses it.

Call sites it must satisfy unchanged (files under /root/reference/moellava/model/language_model):
``llava_stablelm_moe.py:550-560,590-600``, ``llava_phi_moe.py:483-493,522-532``,
``llava_qwen_moe.py:630-640,670-680`` (constructor), ``llava_stablelm_moe.py:163-170`` (3-tuple forward),
``llava_stablelm_moe.py:667-677,709`` (RePaMoE subclasses it and calls ``super().forward``).
Semantics: SURVEY.md 8a rows a1-a2; attribute tree / checkpoint keys: SURVEY.md 8b.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from .experts import Experts
from .sharded_moe import MOELayer, TopKGate


class MoE(nn.Module):
    def __init__(self, hidden_size: int, expert: nn.Module, num_experts: int = 1, ep_size: int = 1, k: int = 1,
                 capacity_factor: float = 1.0, eval_capacity_factor: float = 1.0, min_capacity: int = 4,
                 use_residual: bool = False, noisy_gate_policy: Optional[str] = None, drop_tokens: bool = True,
                 use_rts: bool = True, use_tutel: bool = False, enable_expert_tensor_parallelism: bool = False,
                 top2_2nd_expert_sampling: bool = True) -> None:
        super().__init__()
        assert num_experts % ep_size == 0, \
            f"Number of experts ({num_experts}) should be divisible by expert parallel size ({ep_size})"
        assert noisy_gate_policy is None or noisy_gate_policy in ['None', 'Jitter', 'RSample'], \
            'Unsupported noisy_gate_policy: ' + str(noisy_gate_policy)
        if enable_expert_tensor_parallelism:
            raise NotImplementedError("enable_expert_tensor_parallelism: the reference has no tensor parallelism")
        self.use_residual = use_residual
        self.enable_expert_tensor_parallelism = enable_expert_tensor_parallelism
        self.ep_size = ep_size
        self.expert_group_name = f"ep_size_{self.ep_size}"
        self.num_experts = num_experts
        self.num_local_experts = num_experts // self.ep_size

        experts = Experts(expert, self.num_local_experts, self.expert_group_name)
        gate = TopKGate(hidden_size, num_experts, k, capacity_factor, eval_capacity_factor, min_capacity,
                        None if noisy_gate_policy in (None, 'None') else noisy_gate_policy, drop_tokens, use_rts,
                        None, top2_2nd_expert_sampling)
        self.deepspeed_moe = MOELayer(gate, experts, self.expert_group_name, self.ep_size,
                                      self.num_local_experts, use_tutel=use_tutel)
        if self.use_residual:
            self.mlp = expert
            # coefficient is used for weighted sum of the output of expert and mlp
            self.coefficient = nn.Linear(hidden_size, 2)

    def set_deepspeed_parallelism(self, use_data_before_expert_parallel_: bool = False) -> None:
        """Create / look up the expert-parallel group (upstream: called by the DeepSpeed engine).  EP groups are
        consecutive ranks, expert-data-parallel groups strided."""
        from . import groups
        if self.ep_size == 1:
            return
        groups.create_expert_and_data_parallel(self.ep_size)
        self.deepspeed_moe._set_ep_group(groups.get_expert_parallel_group(self.expert_group_name))

    def set_ep_group(self, group) -> None:
        self.deepspeed_moe._set_ep_group(group)

    def forward(self, hidden_states: Tensor, used_token: Optional[Tensor] = None, *args, **kwargs):
        """Returns ``(output, l_aux, exp_counts)``: output has the input's shape and dtype, ``l_aux`` is a 0-dim
        fp32 tensor, ``exp_counts`` an int64 ``[num_experts]`` tensor (pre-drop counts)."""
        if self.ep_size > 1 and self.deepspeed_moe.ep_group is None:
            self.set_deepspeed_parallelism()
        output = self.deepspeed_moe(hidden_states, used_token)
        if self.use_residual:
            # residual MoE: small dense blend, left to torch (not on the hot path the reference
            # ships: use_residual defaults to False, moellava/train/train.py:97)
            output_mlp = self.mlp(hidden_states)
            if isinstance(output_mlp, tuple):
                output_mlp = output_mlp[0]
            coef = F.softmax(self.coefficient(hidden_states), dim=-1)
            output = output * coef[..., 0:1] + output_mlp * coef[..., 1:]
        return output, self.deepspeed_moe.l_aux, self.deepspeed_moe.exp_counts
