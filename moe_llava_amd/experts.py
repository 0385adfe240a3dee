"""``Experts``: drop-in for ``deepspeed.moe.experts.Experts``.

Keeps the reference-visible structure -- ``deepspeed_experts`` is an ``nn.ModuleList`` of deep copies of
the expert passed in, every parameter tagged ``allreduce=False`` / ``group_name`` (read by
``moellava/train/llava_trainer.py:822-824`` and ``llava_stablelm_moe.py:693-699``) -- while the grouped
GEMM wants one contiguous ``[E_local, N, K]`` buffer per projection.  Both hold: the packed buffers are
allocated once and every expert ``nn.Linear`` parameter that owns its memory is re-pointed to a *view* into
them, so ``state_dict()`` / ``load_state_dict()`` / in-place optimizer updates hit the packed memory directly.
Parameters that are views into somebody else's larger storage (an engine's flat buffer: DeepSpeed ZeRO-1/2) are
NOT re-pointed: they stay where their owner put them and are copied into the packed buffers on every forward.
A ``data_ptr`` check per forward re-packs after ``.to()`` / ``.cuda()`` replaced the storage; decode-size calls use
the cheaper ``packed_unchanged()`` (identity and address of every parameter against the last full check).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch
from torch import Tensor, nn

from . import ops


@dataclass
class ExpertSpec:
    """How one expert module maps onto the kernel's two GEMMs."""
    act: int                      # ops.ACT_*
    fc1: List[nn.Linear]          # SWIGLU: [a (silu'd), b]; GELU_NEW: [fc1]
    fc2: nn.Linear
    n_gated: int = -1             # SWIGLU: hidden channels >= n_gated skip the SiLU (RePaMLP); -1 = all gated


_ADAPTERS: List[Callable[[nn.Module], Optional[ExpertSpec]]] = []


def register_expert_adapter(fn: Callable[[nn.Module], Optional[ExpertSpec]]) -> None:
    """Register a function mapping an expert module to an ``ExpertSpec`` (or None if it does not apply)."""
    _ADAPTERS.insert(0, fn)


def _is_linear(m) -> bool:
    return isinstance(m, nn.Linear)


def _act_name(m) -> str:
    return type(m).__name__.lower() if m is not None else ""


def _repa_gated(e: nn.Module) -> int:
    """RePaMLP (llava_stablelm_moe.py:613-655, llava_qwen_moe.py:693-738): a bool ``mask`` buffer selects the gated
    channels until ``reparam()`` has shrunk the projections; the reference only ever builds prefix masks
    (``adjust_gated_ratio``).  Returns the prefix length, or -1 for ordinary experts."""
    mask = getattr(e, "mask", None)
    if not isinstance(mask, Tensor) or mask.dtype != torch.bool or getattr(e, "reparamed", False):
        return -1
    n = int(getattr(e, "num_gated_channels", mask.numel()))
    return -1 if n >= mask.numel() else n


def _adapt_builtin(e: nn.Module) -> Optional[ExpertSpec]:
    # StableLM / LLaMA / Mistral / MiniCPM form: down_proj(act_fn(gate_proj(x)) * up_proj(x))
    # (stablelm/modeling_stablelm_epoch.py:156-168)
    if all(_is_linear(getattr(e, n, None)) for n in ("gate_proj", "up_proj", "down_proj")):
        act = getattr(e, "act_fn", None)
        if act is not None and "silu" not in _act_name(act) and "swish" not in _act_name(act):
            raise NotImplementedError(f"gated expert with activation {type(act).__name__}: only SiLU is implemented")
        return ExpertSpec(ops.ACT_SWIGLU, [e.gate_proj, e.up_proj], e.down_proj, _repa_gated(e))
    # Qwen form: c_proj(w1(x) * silu(w2(x)))  (qwen/modeling_qwen.py:589-606): the silu'd projection is w2
    if all(_is_linear(getattr(e, n, None)) for n in ("w1", "w2", "c_proj")):
        return ExpertSpec(ops.ACT_SWIGLU, [e.w2, e.w1], e.c_proj, _repa_gated(e))
    # Phi form: fc2(gelu_new(fc1(x)))  (phi/modeling_phi.py:203-215)
    if all(_is_linear(getattr(e, n, None)) for n in ("fc1", "fc2")):
        act = getattr(e, "activation_fn", None)
        name = _act_name(act)
        if "newgelu" not in name and "gelunew" not in name and "gelu_new" not in name:
            raise NotImplementedError(
                f"fc1/fc2 expert with activation {type(act).__name__}: only gelu_new (tanh form) is implemented")
        return ExpertSpec(ops.ACT_GELU_NEW, [e.fc1], e.fc2)
    return None


def _is_foreign_view(p: Tensor, mine) -> bool:
    """True when ``p`` is a view into a storage larger than itself that is not one of the packed buffers in ``mine``
    (storage base addresses): its memory has another owner."""
    st = p.untyped_storage()
    if st.data_ptr() in mine:
        return False
    return p.storage_offset() != 0 or st.nbytes() > p.numel() * p.element_size()


def _contiguous_state_dict(module, state_dict, prefix, local_metadata):
    """state_dict() hands out the parameters as they are: views into the packed buffers.  Those are contiguous except
    fc2 weights of an expert width that is not a multiple of 64 (RePaMLP after reparam(): ``w2[i, :, :I]`` with a padded
    row stride), which safetensors and ``Tensor.view`` consumers reject: hand out a contiguous copy of those."""
    for k, v in list(state_dict.items()):
        if k.startswith(prefix) and isinstance(v, Tensor) and not v.is_contiguous():
            state_dict[k] = v.contiguous()


def describe_expert(e: nn.Module) -> ExpertSpec:
    for fn in _ADAPTERS + [_adapt_builtin]:
        spec = fn(e)
        if spec is not None:
            return spec
    raise NotImplementedError(
        f"moe_llava_amd has no kernel mapping for expert module {type(e).__name__}; "
        "register one with moe_llava_amd.experts.register_expert_adapter (there is no eager fallback)")


class Experts(nn.Module):
    def __init__(self, expert: nn.Module, num_local_experts: int = 1, expert_group_name: Optional[str] = None):
        super().__init__()
        self.deepspeed_experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(num_local_experts)])
        self.num_local_experts = num_local_experts
        self.expert_group_name = expert_group_name
        for e in self.deepspeed_experts:
            for p in e.parameters():
                p.allreduce = False
                p.group_name = expert_group_name
        self._packed: Optional[Dict[str, Optional[Tensor]]] = None
        self._ptrs: Optional[List[int]] = None
        #: the parameters live in somebody else's flat buffer: never re-pointed, copied into the packed layout per forward
        self._external: bool = False
        #: SwiGLU channels >= n_gated skip the SiLU (RePaMLP experts); -1 = ordinary experts.  Refreshed per forward.
        self.n_gated: int = -1
        self._register_state_dict_hook(_contiguous_state_dict)

    # ---- packed weights ----------------------------------------------------------------------
    def _specs(self) -> List[ExpertSpec]:
        # The mapping module -> (activation, Linears) is fixed for a module's lifetime (reparam() replaces the
        # Parameters inside the same nn.Linear objects); only the RePaMLP gated-channel count can change between
        # calls.  Caching it keeps the per-call host cost low (decode steps are launch-bound).
        cached = getattr(self, "_spec_cache", None)
        if cached is not None and len(cached) == len(self.deepspeed_experts):
            ng = _repa_gated(self.deepspeed_experts[0]) if cached[0].act == ops.ACT_SWIGLU else -1
            if ng == cached[0].n_gated:
                self.__dict__["n_gated"] = ng   # plain attribute: skip nn.Module.__setattr__ on the hot path
                return cached
        specs = [describe_expert(e) for e in self.deepspeed_experts]
        a0, n0 = specs[0].act, len(specs[0].fc1)
        for s in specs:
            if s.act != a0 or len(s.fc1) != n0 or s.n_gated != specs[0].n_gated:
                raise ValueError("all local experts must share one MLP form")
        self.n_gated = specs[0].n_gated
        self._spec_cache = specs
        return specs

    def _params(self, specs) -> List[nn.Parameter]:
        ps = []
        for s in specs:
            for lin in s.fc1 + [s.fc2]:
                pd = lin._parameters                  # direct dict access: nn.Module.__getattr__ is slow on this path
                ps.append(pd["weight"])
                b = pd.get("bias")
                if b is not None:
                    ps.append(b)
        return ps

    def packed(self, device: torch.device):
        """Return (act, w1 [E,N1,M], b1|None, w2 [E,M,I], b2|None), (re)packing when a parameter's storage moved."""
        specs = self._specs()
        params = self._params(specs)
        ptrs = [(p.data_ptr(), tuple(p.shape)) for p in params]   # reparam() keeps the address and shrinks the shape
        if self._packed is not None and ptrs == self._ptrs and self._packed["w1"].device == device:
            pk = self._packed
            if self._external:
                self._copy_in(specs, pk)
            return specs[0].act, pk["w1"], pk["b1"], pk["w2"], pk["b2"]
        dtype = params[0].dtype
        for p in params:
            if p.dtype not in ops.HALF_DTYPES or p.dtype != dtype:
                raise TypeError(f"expert parameters must all be bf16 or all fp16 (got {p.dtype}); cast the model with "
                                ".to(torch.bfloat16) / .to(torch.float16)")
            if p.device != device:
                raise ValueError(f"expert parameter on {p.device}, activations on {device}")
        E = len(specs)
        I, M = specs[0].fc1[0].weight.shape
        nproj = len(specs[0].fc1)
        for s in specs:
            for lin in s.fc1:
                if tuple(lin.weight.shape) != (I, M):
                    raise ValueError("fc1 projections of all local experts must have equal shapes")
            if tuple(s.fc2.weight.shape) != (M, I):
                raise ValueError(f"fc2 weight {tuple(s.fc2.weight.shape)} does not match {(M, I)}")
        if M % 64:
            raise ValueError(f"kernel needs the hidden size to be a multiple of 64 (M={M})")
        # an expert width that is not a multiple of 64 (RePaMLP after reparam(): int(I * ratio)) is padded with zero
        # rows / columns: silu(0) * 0 = 0 and zero columns of fc2 add nothing, gradients of the padding are zero
        Ip = (I + 63) // 64 * 64
        if specs[0].n_gated >= 0 and any(not bool(e.mask[:specs[0].n_gated].all()) or bool(e.mask[specs[0].n_gated:].any())
                                         for e in self.deepspeed_experts):
            raise NotImplementedError("RePaMLP mask is not a prefix mask; only prefix masks are implemented")
        has_b1 = specs[0].fc1[0].bias is not None
        has_b2 = specs[0].fc2.bias is not None
        kw = dict(dtype=dtype, device=device)
        for s in specs:
            if any((lin.bias is not None) != has_b1 for lin in s.fc1):
                raise ValueError("bias must be present on all fc1 projections or none")
            if (s.fc2.bias is not None) != has_b2:
                raise ValueError("bias must be present on all fc2 projections or none")
        # Who owns the parameters' memory?  A parameter that is a view into a larger storage which is not one of OUR
        # packed buffers belongs to somebody else -- an engine that flattened the parameters into one buffer and
        # updates that buffer in place (DeepSpeed ZeRO-1/2 bf16, which is how the reference trains: scripts/zero2.json).
        # Taking such parameters away from their owner would silently stop the weight updates from reaching the
        # kernels, so they are never re-pointed: the packed buffers are then a COPY refreshed on every forward.
        mine = set()
        if self._packed is not None:
            mine = {t.untyped_storage().data_ptr() for t in self._packed.values() if t is not None}
        external = any(_is_foreign_view(p, mine) for p in params)
        if external and not self._external:
            import warnings
            warnings.warn("moe_llava_amd.Experts: expert parameters are views into a buffer owned by someone else (a "
                          "flattening optimizer / engine); they are left where they are and copied into the packed "
                          "kernel layout on every forward (one pass over the expert weights per call)", stacklevel=3)
        w1 = torch.zeros((E, nproj * Ip, M), **kw)
        w2 = torch.zeros((E, M, Ip), **kw)
        b1 = torch.zeros((E, nproj * Ip), **kw) if has_b1 else None
        b2 = torch.zeros((E, M), **kw) if has_b2 else None
        pk = {"w1": w1, "b1": b1, "w2": w2, "b2": b2}
        self._copy_in(specs, pk)
        if not external:
            for lin, wv, bv in self._views(specs, pk):
                lin.weight.data = wv
                if bv is not None:
                    lin.bias.data = bv
        self._packed = pk
        self._external = external
        self._ptrs = [(p.data_ptr(), tuple(p.shape)) for p in self._params(specs)]
        return specs[0].act, w1, b1, w2, b2

    def packed_unchanged(self):
        """The tuple ``packed()`` returned last time if nothing it depends on has moved since (same Parameter objects at
        the same addresses with the same leading size), else None -- the cheap check of the decode fast path, which
        is host-bound.  Experts whose parameters live in somebody else's buffer, or with a RePaMLP mask, always answer
        None (they need the full ``packed()`` on every call)."""
        fast = self.__dict__.get("_fast")
        if fast is None:
            return None
        for d, name, prm, ptr, n0 in fast[0]:
            if d.get(name) is not prm or prm.data_ptr() != ptr or prm.shape[0] != n0:
                self.__dict__["_fast"] = None
                return None
        return fast[1]

    def _remember_fast(self, specs, result) -> None:
        """Called by the decode fast path after a full ``packed()``: record what ``packed_unchanged`` compares."""
        if self._external or specs[0].n_gated >= 0 or any(hasattr(e, "mask") for e in self.deepspeed_experts):
            self.__dict__["_fast"] = None
            return
        chk = []
        for sp in specs:
            for lin in sp.fc1 + [sp.fc2]:
                d = lin._parameters
                for name in ("weight", "bias"):
                    prm = d.get(name)
                    if prm is not None:
                        chk.append((d, name, prm, prm.data_ptr(), prm.shape[0]))
        self.__dict__["_fast"] = (chk, result)

    def _views(self, specs, pk):
        """(nn.Linear, weight view, bias view | None) of every projection in the packed buffers ``pk``."""
        I = specs[0].fc1[0].weight.shape[0]
        Ip = pk["w2"].shape[2]
        for i, s in enumerate(specs):
            for j, lin in enumerate(s.fc1):
                yield lin, pk["w1"][i, j * Ip:j * Ip + I], (pk["b1"][i, j * Ip:j * Ip + I] if pk["b1"] is not None else None)
            yield s.fc2, pk["w2"][i, :, :I], (pk["b2"][i] if pk["b2"] is not None else None)

    def _copy_in(self, specs, pk) -> None:
        """parameters -> packed buffers (a no-op copy onto itself never happens: callers use it only for parameters
        that live elsewhere)."""
        with torch.no_grad():
            for lin, wv, bv in self._views(specs, pk):
                wv.copy_(lin.weight.data)
                if bv is not None:
                    bv.copy_(lin.bias.data)

    # ---- autograd plumbing -------------------------------------------------------------------
    def param_list(self) -> List[nn.Parameter]:
        """Expert parameters in the fixed order ``split_grads`` answers in."""
        return self._params(self._specs())

    def split_grads(self, dw1: Tensor, db1: Optional[Tensor], dw2: Tensor, db2: Optional[Tensor]) -> List[Optional[Tensor]]:
        """Packed gradients -> one tensor per parameter of ``param_list()`` (views for the weights, bf16 casts of the
        fp32 column sums for the biases); None for parameters that do not require grad."""
        specs = self._specs()
        I = specs[0].fc1[0].weight.shape[0]
        Ip = dw2.shape[2]                                  # padded expert width of the packed layout
        out: List[Optional[Tensor]] = []
        for i, s in enumerate(specs):
            for j, lin in enumerate(s.fc1):
                out.append(dw1[i, j * Ip:j * Ip + I] if lin.weight.requires_grad else None)
                if lin.bias is not None:
                    out.append(db1[i, j * Ip:j * Ip + I].to(lin.bias.dtype) if lin.bias.requires_grad else None)
            out.append(dw2[i, :, :I] if s.fc2.weight.requires_grad else None)
            if s.fc2.bias is not None:
                out.append(db2[i].to(s.fc2.bias.dtype) if s.fc2.bias.requires_grad else None)
        return out

    # ---- reference-shaped forward -------------------------------------------------------------
    def forward(self, inputs: Tensor) -> Tensor:
        """``inputs`` [ep, E_local, C, M] (upstream layout): expert ``le`` processes ``inputs[:, le]``.
        Runs the grouped HIP FFN over the ep*E_local equal groups of C rows."""
        if inputs.dim() != 4 or inputs.shape[1] != self.num_local_experts:
            raise ValueError(f"Experts.forward expects [ep, {self.num_local_experts}, C, M], got {tuple(inputs.shape)}")
        ep, el, C, M = inputs.shape
        flat = inputs.contiguous().reshape(ep * el * C, M)
        act, w1, b1, w2, b2 = self.packed(flat.device)
        offs = torch.arange(0, (ep * el + 1) * C, C, dtype=torch.int32, device=flat.device)
        y = ops.expert_ffn(act, flat, None, 0, w1, b1, w2, b2, offs, ep * el * C, n_gated=self.n_gated)
        return y[: ep * el * C].reshape(ep, el, C, M)
