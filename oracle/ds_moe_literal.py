"""Literal (dense-mask) CPU restatement of the DeepSpeed MoE layer -- TEST ORACLE ONLY.

PARITY UNPINNED (see ``oracle/__init__.py``): upstream ``deepspeed==0.17.5`` is
not available offline; this file restates its published algorithm op for op,
following the transcription in SURVEY.md section 8(a) rows a1-a8.  Reference-side
anchors in /root/reference:

* constructor call the module must accept:
  ``moellava/model/language_model/llava_stablelm_moe.py:550-560``
* ``(hidden, l_aux, exp_counts)`` 3-tuple contract:
  ``moellava/model/language_model/llava_stablelm_moe.py:163-170``
* attribute tree ``.deepspeed_moe.experts.deepspeed_experts`` / ``.gate.wg``:
  ``llava_stablelm_moe.py:561,693-699``, ``moellava/utils.py:38-46``
* expert MLP forms: ``stablelm/modeling_stablelm_epoch.py:156-168``,
  ``phi/modeling_phi.py:203-215``, ``qwen/modeling_qwen.py:589-606``

Upstream files restated (names only; source not in the container):
``deepspeed/moe/sharded_moe.py`` (_capacity, top1gating, top2gating, TopKGate,
MOELayer, _AllToAll), ``deepspeed/moe/experts.py`` (Experts),
``deepspeed/moe/layer.py`` (MoE).

Everything here is dense: masks are ``[S, E]`` integer tensors, combine weights
and the dispatch mask are ``[S, E, C]``.  Cost is O(S*E*C*M); use it at small S
and use ``oracle.ds_moe_index`` (proved equal to this form in the tests) at
large S.
"""
from __future__ import annotations

import copy
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def capacity_of(num_tokens: int, num_experts: int, capacity_factor: float,
                min_capacity: int) -> int:
    """C = ceil((S/E) * cf) with the multiply done against a
    0-dim fp32 tensor (so in fp32), then max(C, min_capacity)."""
    cf = torch.tensor(capacity_factor)                       # fp32 0-dim, as upstream
    cap = torch.ceil((num_tokens / num_experts) * cf).to(torch.int64)
    mc = torch.tensor(min_capacity)
    if cap < mc:
        cap = mc.to(torch.int64)
    return int(cap)


def gumbel_noise(shape, seed: int) -> Tensor:
    """Gumbel(0,1) samples, -log(-log(U)); deterministic via ``seed`` so tests can
    inject identical noise into oracle and kernel."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(shape, generator=g, dtype=torch.float32).clamp_(1e-10, 1.0 - 1e-7)
    return -torch.log(-torch.log(u))


def _one_hot_float(idx: Tensor, n: int) -> Tensor:
    return F.one_hot(idx, num_classes=n).float()


# --------------------------------------------------------------------------- #
# gating (dense form)
# --------------------------------------------------------------------------- #
def top2gating_dense(logits: Tensor, capacity_factor: float, min_capacity: int,
                     drop_tokens: bool = True,
                     noise: Optional[Tensor] = None,
                     ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """SURVEY 8(a) row a5.  ``noise`` (same shape as logits) plays the role of the
    Gumbel sample upstream adds when ``top2_2nd_expert_sampling`` is on; ``None``
    is the deterministic branch.  Returns (l_aux, combine[S,E,C], dispatch[S,E,C],
    exp_counts[E])."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    first = torch.argmax(gates, dim=1)
    m1 = F.one_hot(first, num_classes=E)
    l2 = logits if noise is None else logits + noise
    second = torch.argmax(l2.masked_fill(m1.bool(), float("-inf")), dim=1)
    m2 = F.one_hot(second, num_classes=E)

    pos1 = torch.cumsum(m1, dim=0) - 1
    pos2 = torch.cumsum(m2, dim=0) - 1
    pos2 = pos2 + torch.sum(m1, dim=0, keepdim=True)          # behind ALL first choices (pre-drop)

    me = torch.mean(gates, dim=0)
    ce = torch.mean(m1.float(), dim=0)
    l_aux = torch.mean(me * ce) * E * E
    exp_counts = torch.sum(m1 + m2, dim=0).detach()          # pre-drop

    if drop_tokens:
        C = capacity_of(S, E, capacity_factor * 2, min_capacity)
        m1 = m1 * torch.lt(pos1, C)
        m2 = m2 * torch.lt(pos2, C)
    else:
        C = int(torch.max(exp_counts))                       # single-process form of the all-reduce(MAX)

    pos1_s = torch.sum(pos1 * m1, dim=1)
    pos2_s = torch.sum(pos2 * m2, dim=1)
    m1f, m2f = m1.float(), m2.float()
    g1 = torch.einsum("se,se->s", gates, m1f)
    g2 = torch.einsum("se,se->s", gates, m2f)
    den = torch.clamp(g1 + g2, min=torch.finfo(g1.dtype).eps)
    g1 = g1 / den
    g2 = g2 / den
    w1 = torch.einsum("s,se->se", g1, m1f)
    w2 = torch.einsum("s,se->se", g2, m2f)
    c1 = torch.einsum("se,sc->sec", w1, _one_hot_float(pos1_s, C))
    c2 = torch.einsum("se,sc->sec", w2, _one_hot_float(pos2_s, C))
    combine = c1 + c2
    return l_aux, combine, combine.bool(), exp_counts


def top1gating_dense(logits: Tensor, capacity_factor: float, min_capacity: int,
                     used_token: Optional[Tensor] = None,
                     drop_tokens: bool = True, use_rts: bool = True,
                     rts_uniform: Optional[Tensor] = None,
                     noise: Optional[Tensor] = None,
                     ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Upstream ``top1gating`` (SURVEY 8(a) footnote / 8(f) row f4).  ``noise`` is the
    Gumbel sample of noisy_gate_policy 'RSample' (the expert is then argmax(logits + noise)
    instead of argmax(gates)); 'Jitter' acts on the gate INPUT and is restated in TopKGateRef.forward.
    Without ``use_rts`` upstream's topk runs over a 0/1 column, so which C member tokens
    survive is whatever torch.topk does with ties (unspecified); this restatement then
    keeps the FIRST C member tokens in token order, the rule the kernels implement.  With ``use_rts`` upstream draws U(0,1) noise per (token, expert) and
    keeps, per expert, the ``C`` tokens with the largest ``mask*noise``;
    ``rts_uniform`` injects that noise (required for determinism when use_rts)."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    C = capacity_of(S, E, capacity_factor, min_capacity)
    first = torch.argmax(logits + noise if noise is not None else gates, dim=1)
    m1 = F.one_hot(first, num_classes=E)
    if used_token is not None:
        m1 = torch.einsum("s,se->se", used_token, m1)
    exp_counts = torch.sum(m1, dim=0).detach()
    if not drop_tokens:
        C = max(int(torch.max(exp_counts)), min_capacity)
    me = torch.mean(gates, dim=0)
    ce = torch.mean(m1.float(), dim=0)
    l_aux = torch.sum(me * ce) * E
    if use_rts:
        if rts_uniform is None:
            raise ValueError("oracle needs rts_uniform injected when use_rts=True")
        score = m1 * rts_uniform
    else:
        # descending in token order inside (0, 1]: first member tokens win
        score = m1 * ((S - torch.arange(S, dtype=torch.float64)) / S)[:, None]
    if C > S:
        raise AssertionError("No. of tokens (batch-size) should be greater than min_capacity")
    top_idx = torch.topk(score, k=C, dim=0)[1]
    m1 = m1 * torch.zeros_like(m1).scatter_(0, top_idx, 1)
    pos1 = torch.cumsum(m1, dim=0) - 1
    pos1_s = torch.sum(pos1 * m1, dim=1)
    gates = gates * m1.float()
    combine = torch.einsum("se,sc->sec", gates, _one_hot_float(pos1_s, C))
    return l_aux, combine, combine.bool(), exp_counts


# --------------------------------------------------------------------------- #
# modules (same attribute tree as upstream)
# --------------------------------------------------------------------------- #
class TopKGateRef(nn.Module):
    """SURVEY 8(a) row a3."""

    def __init__(self, model_dim: int, num_experts: int, k: int = 1,
                 capacity_factor: float = 1.0, eval_capacity_factor: float = 1.0,
                 min_capacity: int = 8, drop_tokens: bool = True, use_rts: bool = True,
                 top2_2nd_expert_sampling: bool = True, noisy_gate_policy: Optional[str] = None):
        super().__init__()
        self.noisy_gate_policy = noisy_gate_policy
        if k not in (1, 2):
            raise ValueError("Only top-1 and top-2 gatings are supported.")
        self.wg = nn.Linear(model_dim, num_experts, bias=False)
        self.k = k
        self.capacity_factor = capacity_factor
        self.eval_capacity_factor = eval_capacity_factor
        self.min_capacity = min_capacity
        self.drop_tokens = drop_tokens
        self.use_rts = use_rts
        self.top2_2nd_expert_sampling = top2_2nd_expert_sampling
        # injected randomness (oracle only): [S,E] tensors or None
        self.noise: Optional[Tensor] = None
        self.rts_uniform: Optional[Tensor] = None
        #: [S,M] fp32 factors of noisy_gate_policy 'Jitter' (upstream draws Uniform(1 - 1e-2, 1 + 1e-2).rsample(x.shape)
        #: inside multiplicative_jitter; the oracle takes the sample from outside, like its other randomness)
        self.jitter: Optional[Tensor] = None
        self.last_logits: Optional[Tensor] = None
        #: oracle-only: use these fp32 logits instead of the CPU GEMV ("given fp32 gate logits" parity contract)
        self.logits_override: Optional[Tensor] = None

    def forward(self, x: Tensor, used_token: Optional[Tensor] = None):
        input_fp32 = x.float()
        # upstream TopKGate.forward: `if self.noisy_gate_policy == 'Jitter' and self.training:
        #     input_fp32 = multiplicative_jitter(input_fp32, device=input.device)`  (= x * uniform sample)
        if self.noisy_gate_policy == "Jitter" and self.training:
            if self.jitter is None:
                raise ValueError("oracle needs gate.jitter injected for noisy_gate_policy='Jitter' in training")
            input_fp32 = input_fp32 * self.jitter.reshape(input_fp32.shape)
        logits = F.linear(input_fp32, self.wg.weight.float())
        if self.logits_override is not None:
            assert self.logits_override.shape == logits.shape
            # value of the override, gradient of the linear (backward parity runs keep the graph)
            logits = logits + (self.logits_override - logits).detach()
        self.last_logits = logits.detach()
        cf = self.capacity_factor if self.training else self.eval_capacity_factor
        if self.k == 1:
            return top1gating_dense(logits, cf, self.min_capacity, used_token,
                                    self.drop_tokens, self.use_rts, self.rts_uniform,
                                    # upstream hands top1gating `noisy_gate_policy if self.training else None`
                                    self.noise if (self.noisy_gate_policy == "RSample" and self.training) else None)
        noise = None
        if self.top2_2nd_expert_sampling:
            if self.noise is None:
                raise ValueError("oracle needs gate.noise injected when top2_2nd_expert_sampling=True")
            noise = self.noise
        return top2gating_dense(logits, cf, self.min_capacity, self.drop_tokens, noise)


class ExpertsRef(nn.Module):
    """deep copies of the expert, chunk over dim 1, loop, cat."""

    def __init__(self, expert: nn.Module, num_local_experts: int = 1,
                 expert_group_name: Optional[str] = None):
        super().__init__()
        self.deepspeed_experts = nn.ModuleList(
            [copy.deepcopy(expert) for _ in range(num_local_experts)])
        self.num_local_experts = num_local_experts
        for e in self.deepspeed_experts:
            for p in e.parameters():
                p.allreduce = False
                p.group_name = expert_group_name

    def forward(self, inputs: Tensor) -> Tensor:
        outs = []
        for chunk, expert in zip(inputs.chunk(self.num_local_experts, dim=1),
                                 self.deepspeed_experts):
            y = expert(chunk)
            if isinstance(y, tuple):
                y = y[0]
            outs.append(y)
        return torch.cat(outs, dim=1)


def _all_to_all(group, t: Tensor) -> Tensor:
    """equal-split all_to_all_single on dim 0 (identity when
    there is no group / a single rank)."""
    if group is None:
        return t
    import torch.distributed as dist
    if dist.get_world_size(group) == 1:
        return t
    t = t.contiguous()
    out = torch.empty_like(t)
    dist.all_to_all_single(out, t, group=group)
    return out


class MOELayerRef(nn.Module):
    """SURVEY 8(a) row a6."""

    def __init__(self, gate: nn.Module, experts: nn.Module, ep_group_name: str,
                 ep_size: int, num_local_experts: int):
        super().__init__()
        self.gate = gate
        self.experts = experts
        self.ep_group = None
        self.ep_size = ep_size
        self.ep_group_name = ep_group_name
        self.num_local_experts = num_local_experts
        self.l_aux = None
        self.exp_counts = None

    def _set_ep_group(self, group):
        self.ep_group = group

    def forward(self, *inp):
        x = inp[0]
        d_model = x.shape[-1]
        flat = x.reshape(-1, d_model)
        used = inp[1] if len(inp) > 1 else None
        self.l_aux, combine, dispatch, self.exp_counts = self.gate(flat, used)
        buf = torch.einsum("sec,sm->ecm", dispatch.type_as(x), flat)
        buf = _all_to_all(self.ep_group, buf)
        buf = buf.reshape(self.ep_size, self.num_local_experts, -1, d_model)
        y = self.experts(buf)
        y = _all_to_all(self.ep_group, y)
        y = y.reshape(self.ep_size * self.num_local_experts, -1, d_model)
        out = torch.einsum("sec,ecm->sm", combine.type_as(x), y)
        return out.reshape(x.shape)


class MoERef(nn.Module):
    """SURVEY 8(a) rows a1-a2 (``deepspeed.moe.layer.MoE``)."""

    def __init__(self, hidden_size: int, expert: nn.Module, num_experts: int = 1,
                 ep_size: int = 1, k: int = 1, capacity_factor: float = 1.0,
                 eval_capacity_factor: float = 1.0, min_capacity: int = 4,
                 use_residual: bool = False, noisy_gate_policy=None,
                 drop_tokens: bool = True, use_rts: bool = True, use_tutel: bool = False,
                 enable_expert_tensor_parallelism: bool = False,
                 top2_2nd_expert_sampling: bool = True):
        super().__init__()
        assert num_experts % ep_size == 0, \
            f"Number of experts ({num_experts}) should be divisible by expert parallel size ({ep_size})"
        assert noisy_gate_policy in (None, "RSample"), "oracle covers noisy_gate_policy None / 'RSample' only"
        self.use_residual = use_residual
        self.ep_size = ep_size
        self.expert_group_name = f"ep_size_{ep_size}"
        self.num_experts = num_experts
        self.num_local_experts = num_experts // ep_size
        experts = ExpertsRef(expert, self.num_local_experts, self.expert_group_name)
        gate = TopKGateRef(hidden_size, num_experts, k, capacity_factor, eval_capacity_factor,
                           min_capacity, drop_tokens, use_rts, top2_2nd_expert_sampling, noisy_gate_policy)
        self.deepspeed_moe = MOELayerRef(gate, experts, self.expert_group_name,
                                         ep_size, self.num_local_experts)
        if use_residual:
            self.mlp = expert
            self.coefficient = nn.Linear(hidden_size, 2)

    def set_ep_group(self, group):
        self.deepspeed_moe._set_ep_group(group)

    def forward(self, hidden_states: Tensor, used_token: Optional[Tensor] = None):
        out = self.deepspeed_moe(hidden_states, used_token)
        if self.use_residual:
            res = self.mlp(hidden_states)
            if isinstance(res, tuple):
                res = res[0]
            coef = F.softmax(self.coefficient(hidden_states), dim=-1)
            out = out * coef[..., 0:1] + res * coef[..., 1:]
        return out, self.deepspeed_moe.l_aux, self.deepspeed_moe.exp_counts
