"""``TopKGate`` / ``MOELayer``: drop-ins for ``deepspeed.moe.sharded_moe``.

The dense ``[S,E,C]`` mask pipeline of upstream is replaced by the index-form routing kernel and the grouped
expert GEMM; the Python here only orchestrates launches (and, for ``ep_size > 1``, the RCCL all-to-all).
"""
from __future__ import annotations

import ctypes
import os

import math
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
        #: optional [S,E] fp32 uniforms for top-1 random token selection (parity tests inject the oracle's)
        self.rts_uniform_override: Optional[Tensor] = None
        self.last_logits: Optional[Tensor] = None
        #: seed of the multiplicative gate-input jitter of the last forward (0 = none); fixed value for tests
        self.last_jitter_seed: int = 0
        self.jitter_seed_override: Optional[int] = None

    def _set_ep_group(self, ep_group):
        self.ep_group = ep_group

    def logits(self, x: Tensor, jitter_seed: int = 0) -> Tensor:
        wg = self.wg
        if not module_has_hooks(wg):
            return ops.gate_logits(x, wg.weight.detach(), jitter_seed)
        # upstream 0.9.5 calls self.wg(input) so forward hooks fire; 0.17.5 uses F.linear and they do not.
        # The reference's gating-logit capture (moellava/utils.py:38-46) relies on them.  With any hook registered the
        # kernel runs as the module's forward INSIDE nn.Module.__call__, so every kind of hook keeps torch's own
        # semantics (forward hooks with or without kwargs, pre-hooks that may replace the input, always-called and
        # global hooks); the hooks see the fp32 [S, E] logits as output.
        def kernel_forward(inp, *args, **kwargs):
            if args or kwargs:
                raise TypeError("wg takes one positional input (a hook handed it more)")
            return ops.gate_logits(inp, wg.weight.detach(), jitter_seed)
        wg.__dict__["forward"] = kernel_forward          # instance attribute: shadows nn.Linear.forward for this call
        try:
            lg = wg(x)
        finally:
            wg.__dict__.pop("forward", None)
        if not isinstance(lg, Tensor) or lg.dtype != torch.float32 or lg.shape != (x.shape[0], wg.weight.shape[0]) \
                or lg.device != x.device:
            raise TypeError("a forward hook on wg replaced the logits by something that is not an fp32 [S, E] tensor on "
                            "the activations' device")
        return lg.contiguous()

    def forward(self, x: Tensor, used_token: Optional[Tensor] = None) -> ops.Routing:
        """Returns the index-form routing (upstream returns dense combine/dispatch masks; nothing in the
        reference tree calls the gate directly)."""
        # upstream applies a noisy-gate policy only while training: 'Jitter' multiplies the fp32 gate input by
        # U(1 - 1e-2, 1 + 1e-2) when `self.training` (multiplicative_jitter), and top1gating is handed
        # `noisy_gate_policy if self.training else None`.  The factors are generated inside the gate kernel from one
        # seed (torch's CPU generator); the backward kernels regenerate them from the same seed.
        jseed = 0
        if self.noisy_gate_policy == "Jitter" and self.training:
            jseed = self.jitter_seed_override if self.jitter_seed_override is not None else _fresh_seed()
            if jseed == 0:
                raise ValueError("jitter_seed_override = 0 would switch the jitter off (0 means 'no jitter' in the C ABI)")
        self.__dict__["last_jitter_seed"] = jseed
        if self.k == 2:
            used_token = None   # upstream top2gating takes no used_token: TopKGate.forward drops it for k = 2
                                # (RePaMoE.forward passes *args through blindly, llava_stablelm_moe.py:709)
        lg = self.logits(x, jseed)
        self.__dict__["last_logits"] = lg   # kept for the backward pass (plain attribute, fast path)
        S, E = lg.shape
        if self.k == 1:
            cf = self.capacity_factor if self.training else self.eval_capacity_factor
            return self._top1(lg, cf, used_token)
        cap, noise, seed = self._top2_args(S, E)
        return ops.routing_top2(lg, cap, noise, self.drop_tokens, noise_seed=seed)

    def _top2_args(self, S: int, E: int):
        """(capacity, noise tensor or None, noise seed) of one top-2 routing call."""
        cf = self.capacity_factor if self.training else self.eval_capacity_factor
        cap = capacity_of(S, E, cf * 2, self.min_capacity)
        noise, seed = None, 0
        if self.top2_2nd_expert_sampling:
            if self.noise_override is not None:
                noise = self.noise_override
            else:
                # Gumbel noise is generated inside the routing kernel (Philox keyed by this seed); the seed comes
                # from torch's CPU generator, so torch.manual_seed() makes runs reproducible and no launch is spent
                seed = _fresh_seed()
        return cap, noise, seed

    def _top1(self, lg: Tensor, cf: float, used_token: Optional[Tensor]) -> ops.Routing:
        """Upstream top1gating: capacity from cf (not doubled), RSample noise on the choice, random
        token selection among an overflowing expert's tokens, weight = gate of the kept token."""
        S, E = lg.shape
        cap = capacity_of(S, E, cf, self.min_capacity)
        if S < self.min_capacity:
            raise AssertionError("No. of tokens (batch-size) should be greater than min_capacity. Either set "
                                 "min_capacity to 0 or increase your batch size.")
        noise, nseed = None, 0
        if self.noisy_gate_policy == "RSample" and self.training:     # eval routing is noise-free upstream
            noise = self.noise_override
            nseed = 0 if noise is not None else _fresh_seed()
        uni, rseed = None, 0
        if self.use_rts:
            uni = self.rts_uniform_override
            rseed = 0 if uni is not None else _fresh_seed()
        return ops.routing_top1(lg, cap, used_token, self.use_rts, uni, rseed, noise, nseed, self.drop_tokens)


DECODE_MAX_TOKENS = 64      # calls of at most this many tokens take the cached decode plan (the GEMV-style kernels' range)


class _DecodePlan:
    """Everything one decode-size forward (S <= 64 tokens, one GPU, top-2, no grad) needs, built once per
    (S, dtype, device, stream, training flag) and reused: the routing arena, logits and the h / y workspace, and the
    argument block of ``moe_block_top2_fwd_args`` with every constant field filled in.  A call then rewrites three
    fields (x, the Gumbel seed, out) and makes ONE ctypes call with ONE argument: at S = 1 the kernels take ~45 us and
    the host side of the call was what bounded the step (profiles/r03/decode_host_*).

    The buffers are REUSED: ``l_aux``, ``exp_counts``, ``gate.last_logits`` and ``last_routing`` of such a call are views
    that the module's next decode-size call overwrites (clone what must survive a later call); the output is a
    fresh tensor every time."""

    __slots__ = ("args", "ref", "routing", "logits", "hy", "packed", "S", "M", "cap", "sampling", "fn", "l_aux", "exp_counts",
                 "dtype", "dev")

    def __init__(self, layer: "MOELayer", x: Tensor, packed, st: int):
        from . import _lib
        gate, experts = layer.gate, layer.experts
        act, w1, b1, w2, b2 = packed
        wgw = gate.wg.weight
        S, M = x.shape
        E, I = wgw.shape[0], w2.shape[2]
        # the same validation the general path does per call, once
        ops._check_ffn_weights(act, x, w1, b1, w2, b2)
        ops._need(wgw, "wg.weight", wgw.dtype, 2)
        if wgw.shape[1] != M or wgw.device != x.device or w1.shape[0] != E:
            raise ValueError("wg.weight / expert weights do not match x")
        cf = gate.capacity_factor if gate.training else gate.eval_capacity_factor
        cap = capacity_of(S, E, cf * 2, gate.min_capacity)
        max_rows = min(2 * S, E * cap) if gate.drop_tokens else 2 * S
        lib = _lib.load()
        ws_bytes = lib.moe_routing_workspace_bytes(S, E)
        r = ops.Routing.allocate(S, E, 2, max_rows, cap, ws_bytes, x.device)
        rows = max(max_rows, 1)
        self.logits = torch.empty((S, E), dtype=torch.float32, device=x.device)
        self.hy = torch.empty((rows * (I + M),), dtype=x.dtype, device=x.device)
        a = _lib.BlockTop2Args()
        a.wg, a.wg_dtype, a.logits = wgw.data_ptr(), ops._wg_dt(wgw, x), self.logits.data_ptr()
        a.noise, a.noise_seed, a.noise_seed_dev = None, 0, None
        a.S, a.M, a.E, a.I, a.act, a.capacity = S, M, E, I, act, cap
        a.drop_tokens, a.n_gated = int(bool(gate.drop_tokens)), experts.n_gated
        for name in ("expert_idx", "slot", "weight", "exp_counts", "kept_counts", "expert_offsets", "perm", "l_aux"):
            setattr(a, name, r.ptr(name))
        a.routing_ws, a.routing_ws_bytes = r.ptr("workspace"), ws_bytes
        a.w1, a.b1 = w1.data_ptr(), (b1.data_ptr() if b1 is not None else None)
        a.w2, a.b2 = w2.data_ptr(), (b2.data_ptr() if b2 is not None else None)
        hp = self.hy.data_ptr()
        a.h_ws, a.y = hp, hp + rows * I * 2
        a.splitk_ws, a.splitk_ws_bytes = None, 0         # decode-size launches take the GEMV-style kernel: nothing to split
        a.dtype, a.stream = ops._dt(x), st
        self.args, self.ref = a, ctypes.byref(a)
        self.routing, self.packed, self.S, self.M, self.cap = r, packed, S, M, cap
        self.sampling = bool(gate.top2_2nd_expert_sampling)
        self.fn = lib.moe_block_top2_fwd_args
        self.l_aux, self.exp_counts = r.l_aux, r.exp_counts      # views, created once
        self.dtype, self.dev = x.dtype, x.device
        r._keep = (self.hy, wgw)

    def run(self, gate: "TopKGate", x: Tensor) -> Tensor:
        a = self.args
        seed_dev = gate.__dict__.get("noise_seed_dev")            # graph.GraphedMoE: the seed lives on the device
        noise = gate.noise_override
        if not self.sampling:
            a.noise, a.noise_seed, a.noise_seed_dev = None, 0, None
        elif noise is not None:
            if noise.dtype != torch.float32 or tuple(noise.shape) != (self.S, a.E) or noise.device != self.dev \
                    or not noise.is_contiguous():
                raise ValueError("noise_override must be a contiguous fp32 [S, E] tensor on x's device")
            a.noise, a.noise_seed, a.noise_seed_dev = noise.data_ptr(), 0, None
        elif seed_dev is not None:
            a.noise, a.noise_seed, a.noise_seed_dev = None, 0, seed_dev.data_ptr()
        else:
            a.noise, a.noise_seed, a.noise_seed_dev = None, _fresh_seed(), None
        out = torch.empty((self.S, self.M), dtype=self.dtype, device=self.dev)
        a.x, a.out = x.data_ptr(), out.data_ptr()
        rc = self.fn(self.ref)
        if rc:
            from . import _lib
            _lib.check(rc, "moe_block_top2_fwd_args")
        return out


def module_has_hooks(m: nn.Module) -> bool:
    """Any forward / forward-pre hook that nn.Module.__call__ would run for ``m`` (its own or torch's global ones)."""
    import torch.nn.modules.module as tm
    return bool(m._forward_hooks or m._forward_pre_hooks or tm._global_forward_hooks or tm._global_forward_pre_hooks)


def _fresh_seed() -> int:
    return int(torch.randint(1, 2 ** 62, (1,), dtype=torch.int64).item())


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


def _all_to_all_rows_async(out: Tensor, inp: Tensor, out_splits, in_splits, group):
    """Like ``_all_to_all_rows`` but returns a work handle (None when it already completed): with RCCL the
    exchange runs on the communicator's stream, so kernels launched meanwhile on the compute stream overlap it."""
    import torch.distributed as dist
    if inp.is_cuda and _group_is_gloo(group):
        _all_to_all_rows(out, inp, out_splits, in_splits, group)
        return None
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=True)


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
        #: grouped-FFN launches the last expert-parallel forward was split into (chunks of the exchange)
        self.last_ffn_launches: int = 1
        #: run the exchange path even when ep_size == 1 (single-rank group; exercises the RCCL code on one GPU)
        self.force_exchange: bool = False

    def _set_ep_group(self, ep_group):
        self.ep_group = ep_group
        self.gate._set_ep_group(ep_group)

    # ------------------------------------------------------------------------------------------
    def forward(self, *inp) -> Tensor:
        x_in = inp[0]
        used_token = inp[1] if len(inp) > 1 else None
        if x_in.dtype not in ops.HALF_DTYPES:
            # the reference trains and serves with --bf16 (scripts/v1/*/finetune_moe.sh) and builds the model in fp16 in
            # its inference loader (moellava/model/builder.py:70,475; serve/cli.py:67-69): both run, nothing else does
            raise TypeError(f"MoE block runs in bf16 or fp16; got {x_in.dtype}")
        wg = self.gate.wg.weight
        if x_in.numel() == 0:
            return self._empty_forward(x_in, wg)
        if torch.is_grad_enabled():
            params = self.experts.param_list()
            if x_in.requires_grad or wg.requires_grad or any(p.requires_grad for p in params):
                if x_in.dtype != torch.bfloat16:
                    raise NotImplementedError("training runs in bf16 (as the reference's scripts do, --bf16 True); the fp16 "
                                              "path is forward only: wrap inference in torch.no_grad()")
                # training: one autograd node for the whole block; l_aux stays differentiable
                # because the reference adds it to the loss (llava_stablelm_moe.py:461-472)
                out, self.l_aux = _MoEBlockFn.apply(self, used_token, x_in, wg, *params)
                return out
        with ops.stream_scope(x_in):
            out, l_aux, _ = self._run(x_in, used_token, save=False)
        self.__dict__["l_aux"] = l_aux
        return out

    def _empty_forward(self, x_in: Tensor, wg: Tensor) -> Tensor:
        """A call without tokens ([0, T, M] or [B, 0, M]).  Upstream's gating runs on the empty logits: every count is
        0, `l_aux` is a mean over zero tokens (NaN), the experts see only empty capacity slots, the output is empty.
        Nothing is launched.  With expert parallelism the other ranks' rows would still have to be served: refused
        loudly instead of leaving the group waiting in its all-to-all."""
        if not (self.ep_size == 1 and not (self.force_exchange and self.ep_group is not None)):
            raise RuntimeError("MoE block: a rank without tokens cannot take part in the expert-parallel exchange")
        d = self.__dict__
        d["exp_counts"] = torch.zeros(wg.shape[0], dtype=torch.int64, device=x_in.device)
        d["last_routing"] = None
        d["l_aux"] = torch.full((), float("nan"), dtype=torch.float32, device=x_in.device)
        return x_in * 0 if (torch.is_grad_enabled() and x_in.requires_grad) else torch.empty_like(x_in)

    def _run(self, x_in: Tensor, used_token, save: bool):
        """gate -> routing -> grouped expert FFN (through the all-to-all when ep_size > 1) -> combine.
        ``save``: keep what the backward needs and return it as the third value."""
        M = x_in.shape[-1]
        x = x_in.detach().reshape(-1, M)
        if not x.is_contiguous():
            x = x.contiguous()
        S = x.shape[0]
        t = self.timer
        gate = self.gate
        if gate.k == 2:
            used_token = None              # ignored by upstream top2gating
        local = self.ep_size == 1 and not (self.force_exchange and self.ep_group is not None)
        if (local and not save and t is None and used_token is None and gate.k == 2 and not module_has_hooks(gate.wg)
                and not (gate.noisy_gate_policy == "Jitter" and gate.training)):
            # inference on one GPU: the whole block in ONE library call (a decode step is host-bound otherwise)
            if S <= DECODE_MAX_TOKENS and x.is_cuda:
                res = self._decode_step(x_in, x, S)
                if res is not None:
                    return res
            experts = self.experts
            act, w1, b1, w2, b2 = experts.packed(x.device)
            wgw = gate.wg.weight
            seed_dev = gate.__dict__.get("noise_seed_dev")    # set by graph.GraphedMoE: seed lives on the device
            if seed_dev is not None and gate.top2_2nd_expert_sampling and gate.noise_override is None:
                cf = gate.capacity_factor if gate.training else gate.eval_capacity_factor
                cap = capacity_of(S, wgw.shape[0], cf * 2, gate.min_capacity)
                out, lg, r = ops.block_top2(x, wgw, cap, None, 0, gate.drop_tokens, act, w1, b1, w2, b2,
                                            experts.n_gated, noise_seed_dev=seed_dev)
            else:
                cap, noise, seed = gate._top2_args(S, wgw.shape[0])
                out, lg, r = ops.block_top2(x, wgw, cap, noise, seed, gate.drop_tokens, act, w1, b1, w2, b2,
                                            experts.n_gated)
            gate.__dict__["last_logits"] = lg
            d = self.__dict__
            d["exp_counts"], d["last_routing"] = r.exp_counts, r
            return out.reshape(x_in.shape), r.l_aux, None
        if t is not None:
            t.mark("gate_begin")
        r = gate(x, used_token)
        if t is not None:
            t.mark("gate_end")
        d = self.__dict__                  # plain attributes: nn.Module.__setattr__ is slow on this path
        d["exp_counts"], d["last_routing"] = r.exp_counts, r
        act, w1, b1, w2, b2 = self.experts.packed(x.device)
        ng = self.experts.n_gated
        shift = r.top_k - 1                      # perm holds token * top_k + choice
        saved = None
        if local:
            if save:
                y, ab, h = ops.expert_ffn_train(act, x, r.perm, shift, w1, b1, w2, b2, r.expert_offsets, r.max_rows, ng)
                saved = dict(a=x, gather=r.perm, shift=shift, ab=ab, h=h, offs=r.expert_offsets, rows=r.max_rows)
            else:
                y = ops.expert_ffn_routed(act, x, r, w1, b1, w2, b2, timer=t, n_gated=ng)
        else:
            y, saved = self._expert_parallel(x, r, act, w1, b1, w2, b2, save)
        out = ops.combine(y, r, S, M)
        if t is not None:
            t.mark("combine_end")
        if save:
            saved.update(x=x, y=y, r=r, logits=self.gate.last_logits, jseed=self.gate.last_jitter_seed, act=act, w1=w1,
                         w2=w2, ng=ng, has_b1=b1 is not None, has_b2=b2 is not None)
        return out.reshape(x_in.shape), r.l_aux, saved

    def _decode_step(self, x_in: Tensor, x: Tensor, S: int):
        """Decode-size forward through a cached ``_DecodePlan`` (see there for what is reused between calls)."""
        gate, experts = self.gate, self.experts
        d = self.__dict__
        st = ops._stream_ptr(x)
        packed = experts.packed_unchanged()
        if packed is None:
            packed = experts.packed(x.device)
            experts._remember_fast(experts._specs(), packed)
            d["_decode_plans"] = {}
            if experts.__dict__.get("_fast") is None:
                return None                  # RePaMLP masks / externally owned parameters: the general path
        plans = d.get("_decode_plans")
        if plans is None:
            plans = d["_decode_plans"] = {}
        wgw = gate.wg.weight
        key = (S, x.dtype, x.device.index, st, gate.training, wgw.data_ptr(), wgw.dtype, gate.capacity_factor,
               gate.eval_capacity_factor, gate.min_capacity, gate.drop_tokens, gate.top2_2nd_expert_sampling)
        plan = plans.get(key)
        if plan is None or plan.packed is not packed:
            if len(plans) >= 32:
                plans.clear()
            plan = plans[key] = _DecodePlan(self, x, packed, st)
        out = plan.run(gate, x)
        gate.__dict__["last_logits"] = plan.logits
        d["exp_counts"], d["last_routing"] = plan.exp_counts, plan.routing
        return out.reshape(x_in.shape), plan.l_aux, None

    def _backward(self, saved, dout: Tensor, dl_aux: Optional[Tensor]):
        """Gradients of the block: (dx, dwg fp32, dw1, db1, dw2, db2)."""
        x, r = saved["x"], saved["r"]
        S, M = x.shape
        dy, dweight = ops.combine_bwd(dout, saved["y"], r, S, M)
        if "gather" in saved:                    # local path (no exchange in the forward pass)
            da_rows, dw1, db1, dw2, db2 = ops.expert_ffn_bwd(
                saved["act"], saved["a"], saved["gather"], saved["shift"], saved["w1"], saved["w2"], saved["ab"],
                saved["h"], dy, saved["offs"], saved["rows"], saved["has_b1"], saved["has_b2"], saved["ng"])
        else:
            plan = saved["plan"]
            dy_recv = self._exchange_rows_chunked(dy, r, plan, to_experts=True)
            da_recv, dw1, db1, dw2, db2 = ops.expert_ffn_bwd(
                saved["act"], saved["a"], None, 0, saved["w1"], saved["w2"], saved["ab"], saved["h"], dy_recv,
                saved["offs"], plan["n_recv"], saved["has_b1"], saved["has_b2"], saved["ng"])
            da_rows = self._exchange_rows_chunked(da_recv, r, plan, to_experts=False)
        dlogits, dwg = ops.gate_bwd(x, saved["logits"], r, dweight, dl_aux, saved["jseed"])
        dx = ops.dispatch_bwd(da_rows, r, dlogits, self.gate.wg.weight.detach(), S, M, saved["jseed"])
        return dx, dwg, dw1, db1, dw2, db2

    # ------------------------------------------------------------------------------------------
    def _expert_parallel(self, x: Tensor, r: ops.Routing, act, w1, b1, w2, b2, save: bool = False):
        """Expert-parallel exchange: rank q of the EP group owns experts [q*El, (q+1)*El).
        Only kept rows travel (variable-split all_to_all_single over RCCL), not upstream's capacity-padded
        [E,C,M] block.  One host sync per call fetches the [ep, E] kept-count matrix that sizes the splits; everything
        per ROW (send order, return order) is computed on the device (``moe_ep_chunk_rows_fwd``).

        The exchange is pipelined in C chunks (``_ep_chunks``): chunk c takes rows [c*k/C, (c+1)*k/C) of every (source
        rank, expert) pair's k kept rows, so both sides derive all split sizes from the one count matrix.  The all-to-all
        of chunk c+1 (and the return of chunk c-1) run on RCCL's stream while the grouped FFN of chunk c runs on the
        compute stream.  Received rows sit in ONE buffer ordered (chunk, source, local expert): C*ep*El groups, group g
        uses expert g % El -- which is also what the backward pass (one call over all groups) wants."""
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
        dev = x.device
        # every rank learns every rank's kept counts: [ep(src), E]; the one host sync of the call
        k = _all_gather_counts(r.kept_counts, ep, self.ep_group).numpy().astype(np.int64)
        n_send, n_recv = int(k[me].sum()), int(k[:, me * El:(me + 1) * El].sum())
        # The chunk count must be the SAME on every rank of the group (each chunk is one all_to_all_single on every rank):
        # it is derived only from the count matrix, which every rank holds, and from the group's agreed bandwidth.
        sends = k.sum(1)                                             # [ep] rows every rank sends
        recvs = k.reshape(ep, ep, El).sum((0, 2))                    # [ep] rows every rank receives
        C = _ep_chunks(int(min(sends.min(), recvs.min())), int(sends.max()), M, ep, _ep_bandwidth(self.ep_group, ep, x))
        # Everything below the count matrix is a pure function of (k, C, this rank): cached, because identical routing
        # comes back (evaluation over fixed batches, the steps of a benchmark) and because the arithmetic plus the
        # host -> device copy of the offsets cost more host time than the launches they feed.
        cache = self.__dict__.setdefault("_ep_plan_cache", {})
        ckey = (k.tobytes(), C, me, dev.index if dev.type == "cuda" else -1)
        hit = cache.get(ckey)
        if hit is None:
            edges = (k[:, :, None] * np.arange(C + 1, dtype=np.int64)) // C      # [src, e, C+1]
            cnt = edges[:, :, 1:] - edges[:, :, :-1]                             # [src, e, c]
            send_cnt = cnt[me].reshape(ep, El, C).sum(1)                         # [dest, c]
            recv_cnt = cnt[:, me * El:(me + 1) * El, :]                          # [src, le, c]
            send_edges = np.concatenate([[0], np.cumsum(send_cnt.sum(0))]).tolist()       # chunk boundaries in the send buffer
            recv_edges = np.concatenate([[0], np.cumsum(recv_cnt.sum((0, 1)))]).tolist()  # ... and in the receive buffer
            # group offsets of the receive buffer, (chunk, source, local expert) order, in one host->device copy
            offs_h = np.zeros(C * ep * El + 1, dtype=np.int32)
            offs_h[1:] = np.cumsum(recv_cnt.transpose(2, 0, 1).reshape(-1))
            offs = torch.from_numpy(offs_h).to(dev, non_blocking=True)
            plan = dict(C=C, n_send=n_send, n_recv=n_recv, send_edges=send_edges, recv_edges=recv_edges,
                        send_splits=[send_cnt[:, c].tolist() for c in range(C)],
                        recv_splits=[recv_cnt[:, :, c].sum(1).tolist() for c in range(C)])
            if len(cache) >= 64:
                cache.clear()
            hit = cache[ckey] = (plan, offs)
        plan, offs = hit
        send_edges, recv_edges = plan["send_edges"], plan["recv_edges"]
        G1 = ep * El
        self.last_ffn_rows, self.last_ffn_launches = n_recv, C
        # send buffer: kept rows gathered through the dispatch index, chunk-major (device-side positions)
        send = ops.ep_chunk_rows(x, r.perm, r.top_k - 1, r.expert_offsets, C, False, n_send)
        rows = max(n_recv, 1)
        recv = torch.empty((rows, M), dtype=x.dtype, device=dev)
        works = [_all_to_all_rows_async(recv[recv_edges[c]:recv_edges[c + 1]], send[send_edges[c]:send_edges[c + 1]],
                                        plan["recv_splits"][c], plan["send_splits"][c], self.ep_group) for c in range(C)]
        I = w2.shape[2]
        y_recv = torch.empty((rows, M), dtype=x.dtype, device=dev)
        h = torch.empty((rows, I), dtype=x.dtype, device=dev)
        ab = torch.empty((rows, w1.shape[1]), dtype=x.dtype, device=dev) if save else None
        y_cm = torch.empty((max(n_send, 1), M), dtype=x.dtype, device=dev)   # returned rows, chunk-major
        back = []
        # The chunks' grouped FFNs alternate between the compute stream and a side stream: a chunk launch is only a few
        # rounds of 256-row tiles (fc2: 1.4 rounds at the bench's shapes), and with two streams the next chunk's blocks
        # fill the CUs that the tail of the previous launch leaves idle.
        cur = torch.cuda.current_stream(dev) if x.is_cuda else None
        two = cur is not None and C > 1 and os.environ.get("MOE_EP_STREAMS", "2") != "1"
        if two:
            side = self.__dict__.get("_side_stream")
            if side is None or side.device != dev:
                side = torch.cuda.Stream(device=dev)
                self.__dict__["_side_stream"] = side
            side.wait_stream(cur)                                  # the send buffer and the offsets are ready
        ng = self.experts.n_gated
        for c in range(C):
            n_rc = int(recv_edges[c + 1] - recv_edges[c])
            on_side = two and (c & 1)
            with (torch.cuda.stream(side) if on_side else ops._NULL):
                with (ops.stream_scope(recv) if two else ops._NULL):  # the launches follow the stream switch
                    if works[c] is not None:
                        works[c].wait()
                    go = offs[c * G1:(c + 1) * G1 + 1]               # absolute row offsets of this chunk's groups
                    if save:
                        ops.expert_ffn_train(act, recv, None, 0, w1, b1, w2, b2, go, n_rc, ng, out=(y_recv, ab, h))
                    else:
                        ops.expert_ffn(act, recv, None, 0, w1, b1, w2, b2, go, n_rc, timer=self.timer, n_gated=ng,
                                       out=(y_recv, h))
                    back.append(_all_to_all_rows_async(y_cm[send_edges[c]:send_edges[c + 1]],
                                                       y_recv[recv_edges[c]:recv_edges[c + 1]],
                                                       plan["send_splits"][c], plan["recv_splits"][c], self.ep_group))
        if two:
            cur.wait_stream(side)
        for w in back:
            if w is not None:
                w.wait()
        # back to compact (expert-major) order, in a buffer the combine can index up to max_rows
        y = torch.empty((max(n_send, r.max_rows, 1), M), dtype=x.dtype, device=dev)
        ops.ep_chunk_rows(y_cm, None, 0, r.expert_offsets, C, True, n_send, out=y)
        saved = None
        if save:
            saved = dict(a=recv, ab=ab, h=h, offs=offs, plan=plan)
        return y, saved

    def _exchange_rows_chunked(self, src_compact: Tensor, r: ops.Routing, plan, to_experts: bool) -> Tensor:
        """Backward-pass exchanges along the forward route.  ``to_experts``: compact rows of this rank's tokens
        (expert-major) -> receive-buffer order on the expert ranks; else the reverse.  Chunks go out back to back, so
        chunk c+1 is on the wire while chunk c is unpacked (the compute between the two directions is one call)."""
        C, n_send, n_recv = plan["C"], plan["n_send"], plan["n_recv"]
        se, re_ = plan["send_edges"], plan["recv_edges"]
        M = src_compact.shape[1]
        dev = src_compact.device
        if to_experts:
            cm = ops.ep_chunk_rows(src_compact, None, 0, r.expert_offsets, C, False, n_send)
            out = torch.empty((max(n_recv, 1), M), dtype=src_compact.dtype, device=dev)
            works = [_all_to_all_rows_async(out[re_[c]:re_[c + 1]], cm[se[c]:se[c + 1]], plan["recv_splits"][c],
                                            plan["send_splits"][c], self.ep_group) for c in range(C)]
            for w in works:
                if w is not None:
                    w.wait()
            return out
        cm = torch.empty((max(n_send, 1), M), dtype=src_compact.dtype, device=dev)
        works = [_all_to_all_rows_async(cm[se[c]:se[c + 1]], src_compact[re_[c]:re_[c + 1]], plan["send_splits"][c],
                                        plan["recv_splits"][c], self.ep_group) for c in range(C)]
        for w in works:
            if w is not None:
                w.wait()
        out = torch.empty((max(n_send, r.max_rows, 1), M), dtype=src_compact.dtype, device=dev)
        ops.ep_chunk_rows(cm, None, 0, r.expert_offsets, C, True, n_send, out=out)
        return out


_EP_BW = {}          # process group -> per-peer all-to-all bandwidth in GB/s, agreed by all its ranks


def _ep_bandwidth(group, ep: int, x: Tensor) -> float:
    """Per-peer all-to-all bandwidth of an EP group in GB/s: measured ONCE per process group (not per layer) on the first
    expert-parallel forward, which every rank of the group enters together, then agreed (MIN over the ranks), so that
    every rank sizes the chunking of the exchange (``_ep_chunks``) from the same number.  Every rank of the group runs
    the same collectives here whatever its environment: 8 MiB to every peer twice (second run timed), then one
    all-reduce that also carries ``MOE_EP_BW_GBS`` -- an override set on ANY rank is adopted by all.  A gloo group or a
    single rank cannot be measured: 0 (= unknown) unless overridden."""
    key = group
    bw = _EP_BW.get(key)
    if bw is not None:
        return bw
    import torch.distributed as dist
    env = os.environ.get("MOE_EP_BW_GBS")
    env_bw = float(env) if env else float("inf")
    measured = float("inf")
    gloo = _group_is_gloo(group)
    if x.is_cuda and ep > 1 and not gloo:
        per = 8 << 20
        a = torch.empty(ep * per, dtype=torch.uint8, device=x.device)
        b = torch.empty_like(a)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(2):
            ev[0].record()
            dist.all_to_all_single(b, a, group=group)
            ev[1].record()
        torch.cuda.synchronize(x.device)
        ms = ev[0].elapsed_time(ev[1])
        if ms > 0:
            measured = per / (ms * 1e-3) / 1e9      # bytes to ONE peer / time of the whole exchange (links run in parallel)
    if ep > 1 and group is not None:
        t = torch.tensor([measured, env_bw], dtype=torch.float64, device=x.device if (x.is_cuda and not gloo) else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        measured, env_bw = float(t[0]), float(t[1])
    bw = env_bw if math.isfinite(env_bw) else (measured if math.isfinite(measured) else 0.0)
    _EP_BW[key] = bw
    return bw


EP_MAX_CHUNKS = 64   # what moe_ep_chunk_rows_fwd accepts


def _ep_chunks(rows_min: int, send_max: int, M: int = 0, ep: int = 1, bw_gbs: float = 0.0) -> int:
    """Chunks the expert-parallel exchange is pipelined in: 1-4 by the rule below, ``MOE_EP_CHUNKS`` forces a count of up
    to 64 (1 = one exchange each way).  Every argument must be identical on all ranks of the group: ``rows_min`` = the
    fewest rows any rank sends or receives, ``send_max`` = the most rows any rank sends (both from the all-gathered
    count matrix), ``bw_gbs`` the group's agreed per-peer bandwidth.
    Chunking costs ~0.1 ms per extra chunk in launch tails (measured) and hides wire time: it is worth it when
    the bytes to one peer take longer than that on the measured per-peer bandwidth, and a chunk still holds >= 4096
    rows (a few rounds of 256-row tiles).  With no measurement (bw_gbs == 0) the round-1 rule stands: 4 chunks from
    16 384 rows."""
    env = os.environ.get("MOE_EP_CHUNKS")
    if env:
        c = int(env)
        if not 1 <= c <= EP_MAX_CHUNKS:
            raise ValueError(f"MOE_EP_CHUNKS={env}: must be between 1 and {EP_MAX_CHUNKS}")
        return c
    rows = rows_min
    if bw_gbs <= 0.0:
        return 4 if rows >= 16384 else 1
    wire_ms = (send_max * M * 2.0 / max(ep, 1)) / (bw_gbs * 1e9) * 1e3       # one direction, one peer link
    if rows < 8192 or wire_ms < 0.15:
        return 1
    return int(max(2, min(4, rows // 4096, round(wire_ms / 0.15))))


class _MoEBlockFn(torch.autograd.Function):
    """The whole MoE block as one autograd node: forward = the inference launches plus the saved pre-activations,
    backward = ``MOELayer._backward``.  Inputs after ``used_token``: x, wg.weight, then ``Experts.param_list()``."""

    @staticmethod
    def forward(ctx, layer: "MOELayer", used_token, x_in: Tensor, wg: Tensor, *params):
        out, l_aux, saved = layer._run(x_in, used_token, save=True)
        ctx.layer, ctx.saved, ctx.x_shape = layer, saved, x_in.shape
        ctx.wg_dtype = wg.dtype
        return out, l_aux

    @staticmethod
    def backward(ctx, dout: Tensor, dl_aux: Optional[Tensor]):
        layer, saved = ctx.layer, ctx.saved
        ctx.saved = None                        # free the activations as soon as they are used
        M = ctx.x_shape[-1]
        if dout is None:
            dout = torch.zeros(ctx.x_shape, dtype=torch.bfloat16, device=saved["x"].device)
        dout = dout.reshape(-1, M)
        if dout.dtype != torch.bfloat16:
            dout = dout.to(torch.bfloat16)
        if not dout.is_contiguous():
            dout = dout.contiguous()
        if dl_aux is not None:
            dl_aux = dl_aux.to(torch.float32).contiguous()
        dx, dwg, dw1, db1, dw2, db2 = layer._backward(saved, dout, dl_aux)
        pg = layer.experts.split_grads(dw1, db1, dw2, db2)
        return (None, None, dx.reshape(ctx.x_shape) if ctx.needs_input_grad[2] else None,
                dwg.to(ctx.wg_dtype) if ctx.needs_input_grad[3] else None, *pg)
