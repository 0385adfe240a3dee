"""``TopKGate`` / ``MOELayer``: drop-ins for ``deepspeed.moe.sharded_moe``.

The dense ``[S,E,C]`` mask pipeline of upstream is replaced by the index-form routing kernel and the grouped
expert GEMM; the Python here only orchestrates launches (and, for ``ep_size > 1``, the RCCL all-to-all).
"""
from __future__ import annotations

import math
import warnings
from typing import Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import ops
from .experts import Experts


def capacity_of(num_tokens: int, num_experts: int, capacity_factor: float, min_capacity: int) -> int:
    """Upstream ``_capacity``: ``ceil((S/E) * cf)`` where the product is taken against a 0-dim
    fp32 tensor, i.e. fp32(S/E) * fp32(cf) rounded to fp32, then ``max(., min_capacity)``.  Host-side scalar
    arithmetic (numpy float32 reproduces the fp32 rounding bit for bit)."""
    prod = np.float32(num_tokens / num_experts) * np.float32(capacity_factor)
    cap = int(np.ceil(prod))
    return max(cap, int(min_capacity))


def gumbel_rsample(shape, device) -> Tensor:
    """Gumbel(0,1) sample; upstream draws ``Gumbel(0,1).rsample`` on the logits' device."""
    u = torch.rand(shape, dtype=torch.float32, device=device).clamp_(min=1e-10, max=1.0 - 1e-7)
    return -torch.log(-torch.log(u))


class TopKGate(nn.Module):
    """Gate: ``wg`` is an ``nn.Linear(model_dim, num_experts, bias=False)`` named ``wg`` (hooked by name in
    ``moellava/utils.py:38-46``); logits are always computed in fp32 from ``input.float()`` and
    ``wg.weight.float()``."""

    def __init__(self, model_dim: int, num_experts: int, k: int = 1, capacity_factor: float = 1.0,
                 eval_capacity_factor: float = 1.0, min_capacity: int = 8, noisy_gate_policy: Optional[str] = None,
                 drop_tokens: bool = True, use_rts: bool = True, ep_group=None,
                 top2_2nd_expert_sampling: bool = True) -> None:
        super().__init__()
        if k != 1 and k != 2:
            raise ValueError("Only top-1 and top-2 gatings are supported.")
        self.wg = nn.Linear(model_dim, num_experts, bias=False)
        self.ep_group = ep_group
        self.k = k
        self.capacity_factor = capacity_factor
        self.eval_capacity_factor = eval_capacity_factor
        self.min_capacity = min_capacity
        self.noisy_gate_policy = noisy_gate_policy
        self.drop_tokens = drop_tokens
        self.use_rts = use_rts
        self.top2_2nd_expert_sampling = top2_2nd_expert_sampling
        #: optional [S,E] fp32 tensor used instead of a fresh Gumbel sample (parity tests inject the oracle's noise)
        self.noise_override: Optional[Tensor] = None

    def _set_ep_group(self, ep_group):
        self.ep_group = ep_group

    def logits(self, x: Tensor) -> Tensor:
        lg = ops.gate_logits(x, self.wg.weight.detach())
        # upstream 0.9.5 calls self.wg(input) so forward hooks fire; 0.17.5 uses F.linear and they do not.
        # The reference's gating-logit capture (moellava/utils.py:38-46) relies on them: fire them by hand.
        if self.wg._forward_hooks:
            for hook in list(self.wg._forward_hooks.values()):
                res = hook(self.wg, (x,), lg)
                if res is not None:
                    lg = res
        return lg

    def forward(self, x: Tensor, used_token: Optional[Tensor] = None) -> ops.Routing:
        """Returns the index-form routing (upstream returns dense combine/dispatch masks; nothing in the
        reference tree calls the gate directly)."""
        if self.k != 2:
            raise NotImplementedError("k=1 (top1gating with RTS) is not implemented in the HIP path yet")
        if used_token is not None:
            raise NotImplementedError("used_token is a top-1 feature upstream; not implemented")
        if self.noisy_gate_policy is not None:
            raise NotImplementedError("noisy_gate_policy is only used by top-1 gating upstream; not implemented")
        lg = self.logits(x)
        S, E = lg.shape
        cf = self.capacity_factor if self.training else self.eval_capacity_factor
        cap = capacity_of(S, E, cf * 2, self.min_capacity)
        noise, seed = None, 0
        if self.top2_2nd_expert_sampling:
            if self.noise_override is not None:
                noise = self.noise_override
            else:
                # Gumbel noise is generated inside the routing kernel (Philox keyed by this seed); the seed comes
                # from torch's CPU generator, so torch.manual_seed() makes runs reproducible and no launch is spent
                seed = int(torch.randint(1, 2 ** 62, (1,), dtype=torch.int64).item())
        return ops.routing_top2(lg, cap, noise, self.drop_tokens, noise_seed=seed)


def _group_is_gloo(group) -> bool:
    import torch.distributed as dist
    try:
        return dist.get_backend(group) == "gloo"
    except Exception:
        return False


def _all_to_all_rows(out: Tensor, inp: Tensor, out_splits, in_splits, group) -> None:
    """Variable-split all_to_all_single on dim 0.  RCCL (backend "nccl") moves device buffers directly over
    xGMI.  A gloo group (CPU rehearsals, or several ranks sharing one GPU in the tests) cannot take device
    tensors, so for gloo only the payload is staged through host memory; the compute never is."""
    import torch.distributed as dist
    if inp.is_cuda and _group_is_gloo(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
        return
    dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def _all_gather_counts(local: Tensor, ep: int, group) -> Tensor:
    """[ep, E] int32 matrix of every EP rank's kept counts, on the host."""
    import torch.distributed as dist
    if local.is_cuda and _group_is_gloo(group):
        loc = local.cpu()
        allk = torch.empty((ep * loc.numel(),), dtype=loc.dtype)
        dist.all_gather_into_tensor(allk, loc, group=group)
        return allk.reshape(ep, -1)
    allk = torch.empty((ep * local.numel(),), dtype=local.dtype, device=local.device)   # flat: gloo wants 1-D
    dist.all_gather_into_tensor(allk, local, group=group)
    return allk.cpu().reshape(ep, -1)


class MOELayer(nn.Module):
    """gate -> dispatch -> (all-to-all) -> experts -> (all-to-all) -> combine."""

    def __init__(self, gate: nn.Module, experts: Experts, ep_group_name: str, ep_size: int,
                 num_local_experts: int, use_tutel: bool = False) -> None:
        super().__init__()
        if use_tutel:
            raise NotImplementedError("use_tutel=True: Tutel kernels are CUDA-only and unused by the reference")
        self.gate = gate
        self.experts = experts
        self.ep_group = None
        self.ep_size = ep_size
        self.ep_group_name = ep_group_name
        self.num_local_experts = num_local_experts
        self.l_aux: Optional[Tensor] = None
        self.exp_counts: Optional[Tensor] = None
        self.last_routing: Optional[ops.Routing] = None
        #: rows the local grouped GEMM processed in the last expert-parallel forward (None when ep_size == 1)
        self.last_ffn_rows: Optional[int] = None
        #: set to an ops.PhaseTimer to record HIP events around gate / fc1 / fc2 / combine (wall_clock_breakdown)
        self.timer: Optional[ops.PhaseTimer] = None
        self._warned_grad = False

    def _set_ep_group(self, ep_group):
        self.ep_group = ep_group
        self.gate._set_ep_group(ep_group)

    # ------------------------------------------------------------------------------------------
    def forward(self, *inp) -> Tensor:
        x_in = inp[0]
        used_token = inp[1] if len(inp) > 1 else None
        if torch.is_grad_enabled() and not self._warned_grad and (
                x_in.requires_grad or any(p.requires_grad for p in self.parameters())):
            warnings.warn("moe_llava_amd: the HIP MoE block is forward-only in this release; outputs are detached "
                          "(backward is SURVEY 8f row f1)")
            self._warned_grad = True
        if x_in.dtype != torch.bfloat16:
            raise TypeError(f"MoE block runs in bf16 (the reference trains and serves with --bf16); got {x_in.dtype}")
        M = x_in.shape[-1]
        x = x_in.detach().reshape(-1, M)
        if not x.is_contiguous():
            x = x.contiguous()
        S = x.shape[0]
        t = self.timer
        if t is not None:
            t.mark("gate_begin")
        r = self.gate(x, used_token)
        if t is not None:
            t.mark("gate_end")
        self.l_aux, self.exp_counts, self.last_routing = r.l_aux, r.exp_counts, r
        act, w1, b1, w2, b2 = self.experts.packed(x.device)
        if self.ep_size == 1:
            y = ops.expert_ffn(act, x, r.perm, 1, w1, b1, w2, b2, r.expert_offsets, r.max_rows, timer=t)
        else:
            y = self._expert_parallel(x, r, act, w1, b1, w2, b2)
        out = ops.combine(y, r, S, M)
        if t is not None:
            t.mark("combine_end")
        return out.reshape(x_in.shape)

    # ------------------------------------------------------------------------------------------
    def _expert_parallel(self, x: Tensor, r: ops.Routing, act, w1, b1, w2, b2) -> Tensor:
        """Expert-parallel exchange: rank q of the EP group owns experts [q*El, (q+1)*El).
        Only kept rows travel (variable-split all_to_all_single over RCCL), not upstream's capacity-padded
        [E,C,M] block.  One host sync per call fetches the [ep, E] kept-count matrix that sizes the splits."""
        import torch.distributed as dist
        if self.ep_group is None:
            raise RuntimeError("ep_size > 1 but no expert-parallel group set: call MoE.set_deepspeed_parallelism() "
                               "after torch.distributed.init_process_group")
        ep, El = self.ep_size, self.num_local_experts
        if dist.get_world_size(self.ep_group) != ep:
            raise RuntimeError("expert-parallel group size does not match ep_size")
        me = dist.get_rank(self.ep_group)
        E = ep * El
        M = x.shape[1]
        # every rank learns every rank's kept counts: [ep(src), E]; the one host sync of the call
        kept_h = _all_gather_counts(r.kept_counts, ep, self.ep_group).to(torch.int64)
        send_splits = kept_h[me].reshape(ep, El).sum(1).tolist()     # rows I send to each rank
        recv_mat = kept_h[:, me * El:(me + 1) * El]                  # [src, le] rows I receive
        recv_splits = recv_mat.sum(1).tolist()
        n_send, n_recv = int(sum(send_splits)), int(sum(recv_splits))
        self.last_ffn_rows = n_recv
        send = ops.gather_rows(x, r.perm, 1, r.expert_offsets[E:E + 1], n_send)
        recv = torch.empty((max(n_recv, 1), M), dtype=torch.bfloat16, device=x.device)
        _all_to_all_rows(recv[:n_recv], send[:n_send], recv_splits, send_splits, self.ep_group)
        # received rows are ordered (src, local expert): ep*El groups, group g uses expert g % El
        offs_h = torch.zeros(ep * El + 1, dtype=torch.int32)
        offs_h[1:] = torch.cumsum(recv_mat.reshape(-1), 0).to(torch.int32)
        offs = offs_h.to(x.device, non_blocking=True)
        y_recv = ops.expert_ffn(act, recv, None, 0, w1, b1, w2, b2, offs, n_recv, timer=self.timer)
        y = torch.empty((max(n_send, r.max_rows, 1), M), dtype=torch.bfloat16, device=x.device)
        _all_to_all_rows(y[:n_send], y_recv[:n_recv], send_splits, recv_splits, self.ep_group)
        return y
