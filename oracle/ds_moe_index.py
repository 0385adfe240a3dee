"""Index-form CPU restatement of the DeepSpeed MoE layer -- TEST ORACLE ONLY.

PARITY UNPINNED (see ``oracle/__init__.py``).  Same algorithm as
``oracle.ds_moe_literal`` but without the dense ``[S,E,C]`` masks, following
SURVEY.md section 8(a') line by line, so that it finishes in seconds at
S = 22 528 ... 1 M tokens where the dense form is quadratic.  The tests prove
routing (expert ids, slots, drops, exp_counts) bit-identical and weights /
l_aux bit-identical to the dense form at small S.

Outputs per token s and choice j in {0,1}:
  expert_idx[s,j]  int32   chosen expert
  slot[s,j]        int32   position inside the expert's capacity buffer, -1 = dropped
  weight[s,j]      fp32    normalised combine weight (0 when dropped)
plus exp_counts[E] int64 (pre-drop), kept_counts[E] int32, l_aux fp32 scalar.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor


def _exclusive_rank(idx: Tensor, num_experts: int) -> Tensor:
    """p[s] = #{s' < s : idx[s'] == idx[s]} (token order)."""
    oh = F.one_hot(idx, num_classes=num_experts)
    return (torch.cumsum(oh, dim=0) - 1).gather(1, idx[:, None])[:, 0]


def route_top2(logits: Tensor, capacity: int, noise: Optional[Tensor] = None,
               drop_tokens: bool = True) -> Dict[str, Tensor]:
    """SURVEY 8(a') routing.  ``logits`` fp32 [S,E]; ``capacity`` already computed
    by ``ds_moe_literal.capacity_of(S, E, cf*2, min_capacity)``."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    i1 = torch.argmax(gates, dim=1)
    l2 = (logits if noise is None else logits + noise).clone()
    l2.scatter_(1, i1[:, None], float("-inf"))
    i2 = torch.argmax(l2, dim=1)
    n1 = torch.bincount(i1, minlength=E)
    n2 = torch.bincount(i2, minlength=E)
    exp_counts = (n1 + n2).to(torch.int64)
    me = torch.mean(gates, dim=0)
    ce = n1.float() / S
    l_aux = torch.mean(me * ce) * E * E
    p1 = _exclusive_rank(i1, E)
    p2 = _exclusive_rank(i2, E) + n1[i2]
    if not drop_tokens:
        capacity = int(exp_counts.max())
    k1 = p1 < capacity
    k2 = p2 < capacity
    zero = torch.zeros((), dtype=gates.dtype)
    g1 = torch.where(k1, gates.gather(1, i1[:, None])[:, 0], zero)
    g2 = torch.where(k2, gates.gather(1, i2[:, None])[:, 0], zero)
    den = torch.clamp(g1 + g2, min=torch.finfo(torch.float32).eps)
    w1 = g1 / den
    w2 = g2 / den
    slot1 = torch.where(k1, p1, torch.full_like(p1, -1))
    slot2 = torch.where(k2, p2, torch.full_like(p2, -1))
    kept = torch.bincount(i1[k1], minlength=E) + torch.bincount(i2[k2], minlength=E)
    return {
        "gates": gates,
        "expert_idx": torch.stack([i1, i2], 1).to(torch.int32),
        "slot": torch.stack([slot1, slot2], 1).to(torch.int32),
        "weight": torch.stack([w1, w2], 1),
        "exp_counts": exp_counts,
        "kept_counts": kept.to(torch.int32),
        "l_aux": l_aux,
        "capacity": capacity,
    }


def route_top1_rts(logits: Tensor, capacity: int, rts_uniform: Optional[Tensor],
                   used_token: Optional[Tensor] = None, noise: Optional[Tensor] = None,
                   drop_tokens: bool = True) -> Dict[str, Tensor]:
    """Index form of upstream ``top1gating`` with random token selection: per
    expert keep the ``capacity`` member tokens with the largest uniform score;
    slots are ranks among the KEPT tokens in token order; weight = gate (not
    renormalised).  ``rts_uniform=None`` = no RTS: the first ``capacity`` member tokens
    stay.  Equal uniform scores resolve to the lower token index.  ``noise``: 'RSample'
    policy, expert = argmax(logits + noise).  ``drop_tokens=False``: nothing is dropped."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    i1 = torch.argmax(logits + noise if noise is not None else gates, dim=1)
    member = torch.ones(S, dtype=torch.bool) if used_token is None else used_token.reshape(-1) != 0
    n1 = torch.bincount(i1[member], minlength=E)
    me = torch.mean(gates, dim=0)
    ce = n1.float() / S
    l_aux = torch.sum(me * ce) * E
    if not drop_tokens:
        capacity = int(n1.max())
    if rts_uniform is not None:
        score = rts_uniform.gather(1, i1[:, None])[:, 0].double()
    else:
        score = torch.zeros(S, dtype=torch.float64)
    keep = torch.zeros(S, dtype=torch.bool)
    for e in range(E):
        rows = torch.nonzero(member & (i1 == e))[:, 0]
        if rows.numel() > capacity:
            # stable descending sort: equal scores keep token order
            order = torch.sort(-score[rows], stable=True)[1]
            keep[rows[order[:capacity]]] = True
        else:
            keep[rows] = True
    slot = torch.full((S,), -1, dtype=torch.int64)
    for e in range(E):
        rows = torch.nonzero(keep & (i1 == e))[:, 0]
        slot[rows] = torch.arange(rows.numel())
    w = torch.where(keep, gates.gather(1, i1[:, None])[:, 0], torch.zeros(()))
    return {
        "gates": gates,
        "expert_idx": i1[:, None].to(torch.int32),
        "slot": slot[:, None].to(torch.int32),
        "weight": w[:, None],
        "exp_counts": n1.to(torch.int64),
        "kept_counts": torch.bincount(i1[keep], minlength=E).to(torch.int32),
        "l_aux": l_aux,
        "capacity": capacity,
    }


def dispatch_order(route: Dict[str, Tensor], num_experts: int):
    """(expert_offsets[E+1], perm[sum kept]) with perm[offsets[e]+slot] =
    token*K + choice -- the compact, expert-sorted row list the kernels use."""
    idx, slot = route["expert_idx"].long(), route["slot"].long()
    K = idx.shape[1]
    kept = route["kept_counts"].long()
    offs = torch.zeros(num_experts + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(kept, 0)
    perm = torch.full((int(offs[-1]),), -1, dtype=torch.int64)
    tok = torch.arange(idx.shape[0])
    for j in range(K):
        m = slot[:, j] >= 0
        perm[offs[idx[m, j]] + slot[m, j]] = tok[m] * K + j
    return offs.to(torch.int32), perm.to(torch.int32)


def moe_forward_index(x: Tensor, route: Dict[str, Tensor], experts: List[torch.nn.Module]) -> Tensor:
    """SURVEY 8(a') dispatch / expert / combine: exact row copy in, per-expert MLP
    on the kept rows only (in slot order), combine weights rounded to x.dtype then
    accumulated in fp32 and rounded once (what the ``sec,ecm->sm`` einsum does
    when at most two terms per row are non-zero)."""
    S, M = x.shape
    idx, slot, w = route["expert_idx"].long(), route["slot"].long(), route["weight"]
    K = idx.shape[1]
    acc = torch.zeros(S, M, dtype=torch.float32)
    wq = w.to(x.dtype).float()
    for e, expert in enumerate(experts):
        for j in range(K):
            rows = torch.nonzero((idx[:, j] == e) & (slot[:, j] >= 0) & (w[:, j] != 0))[:, 0]
            if rows.numel() == 0:
                continue
            y = expert(x[rows])
            if isinstance(y, tuple):
                y = y[0]
            acc[rows] += wq[rows, j, None] * y.float()
    return acc.to(x.dtype)
