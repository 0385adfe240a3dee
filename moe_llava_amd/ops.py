"""Host-side wrappers of the C ABI: shape / dtype / device validation, buffer allocation through the
torch caching allocator (so every buffer is stream-ordered on the current stream), and the launch.

Nothing here computes on the CPU or in eager torch; a missing library raises (``_lib.load``).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _lib

ACT_SWIGLU = 0
ACT_GELU_NEW = 1
EPI_PLAIN, EPI_SWIGLU, EPI_GELU_NEW = 0, 1, 2
DTYPE_BF16, DTYPE_F16 = 0, 1                 # MOE_DTYPE_* of the C ABI
WG_F32, WG_BF16, WG_F16 = 0, 1, 2            # MOE_WG_*
#: element types of the forward path: bf16 (training, --bf16 serving) and fp16 (the reference's inference loader,
#: moellava/model/builder.py:70,475).  The backward entry points are bf16 only.
HALF_DTYPES = (torch.bfloat16, torch.float16)


def _dt(t: Tensor) -> int:
    """MOE_DTYPE_* of a 16-bit activation / weight tensor."""
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    if t.dtype == torch.float16:
        return DTYPE_F16
    raise TypeError(f"expected bf16 or fp16, got {t.dtype}")


def _wg_dt(wg: Tensor, x: Tensor) -> int:
    """MOE_WG_* of the gate weight: fp32, or the activations' own 16-bit type (what ``model.to(dtype)`` leaves)."""
    if wg.dtype == torch.float32:
        return WG_F32
    if wg.dtype == x.dtype and wg.dtype in HALF_DTYPES:
        return WG_BF16 if wg.dtype == torch.bfloat16 else WG_F16
    raise TypeError(f"wg.weight: expected fp32 or the activations' dtype ({x.dtype}), got {wg.dtype}")


class PhaseTimer:
    """HIP-event marks on the stream the kernels are launched on (torch's current stream), in the spirit of
    upstream MOELayer's ``wall_clock_breakdown`` timers.  ``durations()`` synchronises and returns, per pair of
    consecutive marks of one forward, the list of elapsed milliseconds."""

    def __init__(self):
        self.events = []   # list of (name, event) in record order
        self.enabled = True

    def mark(self, name: str) -> None:
        if not self.enabled:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.events.append((name, ev))

    def reset(self) -> None:
        self.events = []

    def durations(self, begin: str, end: str):
        torch.cuda.synchronize()
        out, start = [], None
        for name, ev in self.events:
            if name == begin:
                start = ev
            elif name == end and start is not None:
                out.append(start.elapsed_time(ev))
                start = None
        return out


_STREAM_SCOPE = None   # (device index, stream pointer) while MOELayer runs one forward; saves repeated lookups


def _stream_ptr(t: Tensor) -> int:
    sc = _STREAM_SCOPE
    if sc is not None and sc[0] == t.device.index:
        return sc[1]
    return torch.cuda.current_stream(t.device).cuda_stream


class stream_scope:
    """``with stream_scope(x):`` -- look the current stream of x's device up once for all ops inside (the stream must
    not be switched inside the block).  At decode sizes the host side of a call costs as much as its GPU side."""

    def __init__(self, t: Tensor):
        self.t = t

    def __enter__(self):
        global _STREAM_SCOPE
        self.prev = _STREAM_SCOPE
        if self.t.is_cuda:
            _STREAM_SCOPE = (self.t.device.index, torch.cuda.current_stream(self.t.device).cuda_stream)
        return self

    def __exit__(self, *a):
        global _STREAM_SCOPE
        _STREAM_SCOPE = self.prev
        return False


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def _on(dev: torch.device):
    """Context that makes ``dev`` the current HIP device; free when it already is (the common case: at decode sizes
    the host-side cost of a call is comparable to its GPU time)."""
    if dev.type == "cuda" and torch.cuda.current_device() == (dev.index if dev.index is not None else 0):
        return _NULL
    return torch.cuda.device(dev)


_SPLITK_WS = {}      # (device index, stream pointer) -> 64 MiB workspace of the split-K tail
_SPLITK_BYTES = None


def _splitk_ws(t: Tensor, st: int):
    """(pointer, bytes) of the split-K workspace of the grouped GEMM launches on stream ``st`` of ``t``'s device
    (include/moe_gfx950.h: moe_splitk_workspace_bytes).  One buffer per (device, stream), allocated on first use and kept:
    launches on one stream are ordered, launches on different streams (the chunks of the expert-parallel exchange
    alternate between two) must not share it.  ``MOE_SPLITK=0`` in the environment: no workspace, tiles stay whole."""
    global _SPLITK_BYTES
    if _SPLITK_BYTES is None:
        import os
        env = os.environ.get("MOE_SPLITK")
        _SPLITK_BYTES = 0 if (env is not None and int(env) < 2) else int(_lib.load().moe_splitk_workspace_bytes())
    if _SPLITK_BYTES == 0 or not t.is_cuda:
        return None, 0
    key = (t.device.index, st)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.empty((_SPLITK_BYTES,), dtype=torch.uint8, device=t.device)
        _SPLITK_WS[key] = ws
    return ws.data_ptr(), _SPLITK_BYTES


def reload_knobs() -> None:
    """Have the library (and this module) read the MOE_* environment knobs again (tests, tuning tools)."""
    global _SPLITK_BYTES
    _SPLITK_BYTES = None
    _lib.load().moe_reload_knobs()


def _need(t: Tensor, name: str, dtype, ndim: Optional[int] = None) -> None:
    if not isinstance(t, Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise _lib.MoEKernelError(f"{name}: must live on the GPU (got {t.device}); there is no CPU path")
    if (t.dtype not in dtype) if isinstance(dtype, tuple) else (t.dtype != dtype):
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")


_ROUTING_FIELDS = ("expert_idx", "slot", "weight", "exp_counts", "kept_counts", "expert_offsets", "perm", "l_aux")


def _pad16(n: int) -> int:
    return (n + 15) & ~15


class Routing:
    """Index-form routing result (SURVEY 8a').  All arrays live on the device:

        expert_idx [S,k] int32 · slot [S,k] int32 (-1 = dropped) · weight [S,k] fp32 · exp_counts [E] int64 (pre-drop)
        kept_counts [E] int32 · expert_offsets [E+1] int32 · perm [max_rows] int32 (token*k + choice) · l_aux 0-dim fp32

    The kernels' outputs sit in ONE allocation; the tensor attributes above are views created on first access (the
    launches only need the addresses, ``ptr(name)``), which keeps the per-call host cost low.  A Routing can also be
    built from ready-made tensors (tests, CPU rehearsals): ``Routing(expert_idx, slot, weight, exp_counts, kept_counts,
    expert_offsets, perm, l_aux, max_rows, capacity)``."""

    def __init__(self, expert_idx=None, slot=None, weight=None, exp_counts=None, kept_counts=None, expert_offsets=None,
                 perm=None, l_aux=None, max_rows: int = 0, capacity: int = 0, used_token: Optional[Tensor] = None):
        self._views = {}
        for name, t in zip(_ROUTING_FIELDS, (expert_idx, slot, weight, exp_counts, kept_counts, expert_offsets, perm, l_aux)):
            if t is not None:
                self._views[name] = t
        self._arena = None
        self._layout = None
        self.max_rows = int(max_rows)      # host bound on expert_offsets[E]
        self.capacity = int(capacity)
        self.used_token = used_token       # [S] int32 mask given to top-1 routing (kept for the backward pass)
        self.S = expert_idx.shape[0] if expert_idx is not None else 0
        self.E = exp_counts.numel() if exp_counts is not None else 0
        self.top_k = expert_idx.shape[1] if expert_idx is not None else 0

    @classmethod
    def allocate(cls, S: int, E: int, k: int, max_rows: int, capacity: int, ws_bytes: int, dev, used_token=None):
        """One device allocation for every routing output + the kernel's workspace."""
        r = cls.__new__(cls)
        lay, off = {}, 0
        for name, nbytes, dtype, shape in (
                ("expert_idx", S * k * 4, torch.int32, (S, k)), ("slot", S * k * 4, torch.int32, (S, k)),
                ("weight", S * k * 4, torch.float32, (S, k)), ("exp_counts", E * 8, torch.int64, (E,)),
                ("kept_counts", E * 4, torch.int32, (E,)), ("expert_offsets", (E + 1) * 4, torch.int32, (E + 1,)),
                ("perm", max(max_rows, 1) * 4, torch.int32, (max(max_rows, 1),)), ("l_aux", 4, torch.float32, ()),
                ("workspace", ws_bytes, torch.uint8, (ws_bytes,))):
            lay[name] = (off, nbytes, dtype, shape)
            off = _pad16(off + nbytes)
        r._arena = torch.empty((max(off, 16),), dtype=torch.uint8, device=dev)
        r._base = r._arena.data_ptr()
        r._layout = lay
        r._views = {}
        r.max_rows, r.capacity, r.used_token = int(max_rows), int(capacity), used_token
        r.S, r.E, r.top_k = S, E, k
        return r

    def ptr(self, name: str) -> int:
        if self._layout is not None:
            return self._base + self._layout[name][0]
        return self._views[name].data_ptr()

    def _view(self, name: str) -> Tensor:
        v = self._views.get(name)
        if v is None:
            off, nbytes, dtype, shape = self._layout[name]
            v = self._arena[off:off + nbytes].view(dtype).view(shape)
            self._views[name] = v
        return v

    expert_idx = property(lambda self: self._view("expert_idx"))
    slot = property(lambda self: self._view("slot"))
    weight = property(lambda self: self._view("weight"))
    exp_counts = property(lambda self: self._view("exp_counts"))
    kept_counts = property(lambda self: self._view("kept_counts"))
    expert_offsets = property(lambda self: self._view("expert_offsets"))
    perm = property(lambda self: self._view("perm"))
    l_aux = property(lambda self: self._view("l_aux"))


def gate_logits(x: Tensor, wg_weight: Tensor, jitter_seed: int = 0) -> Tensor:
    """logits = x.float() @ wg.float().T  ([S,E] fp32) -- TopKGate.forward's F.linear.
    ``jitter_seed`` != 0: x.float() is first multiplied by the U(1 - 1e-2, 1 + 1e-2) factors of that seed
    (noisy_gate_policy='Jitter' while training; ``gate_jitter`` returns the factors)."""
    _need(x, "x", HALF_DTYPES, 2)
    wgd = _wg_dt(wg_weight, x)
    _need(wg_weight, "wg.weight", wg_weight.dtype, 2)
    S, M = x.shape
    E = wg_weight.shape[0]
    if wg_weight.shape[1] != M or wg_weight.device != x.device:
        raise ValueError(f"wg.weight {tuple(wg_weight.shape)} on {wg_weight.device} does not match x {tuple(x.shape)} on {x.device}")
    logits = torch.empty((S, E), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    with _on(x.device):
        rc = lib.moe_gate_logits_fwd(x.data_ptr(), wg_weight.data_ptr(), wgd, logits.data_ptr(), S, M, E, _dt(x),
                                     int(jitter_seed), _stream_ptr(x))
    _lib.check(rc, "moe_gate_logits_fwd")
    return logits


def gate_jitter(jitter_seed: int, S: int, M: int, device) -> Tensor:
    """[S,M] fp32: the multiplicative jitter factors the gate kernels derive from ``jitter_seed`` (tests, inspection)."""
    out = torch.empty((S, M), dtype=torch.float32, device=device)
    lib = _lib.load()
    with _on(out.device):
        rc = lib.moe_gate_jitter_fill(int(jitter_seed), S, M, out.data_ptr(), _stream_ptr(out))
    _lib.check(rc, "moe_gate_jitter_fill")
    return out


def routing_top2(logits: Tensor, capacity: int, noise: Optional[Tensor] = None,
                 drop_tokens: bool = True, noise_seed: int = 0) -> Routing:
    """top2gating in index form from fp32 logits (SURVEY 8a row a5 / 8a').  Second-choice sampling noise is either
    a given [S,E] tensor or generated in the kernel from ``noise_seed`` (non-zero)."""
    _need(logits, "logits", torch.float32, 2)
    S, E = logits.shape
    if S < 1:
        raise ValueError("routing needs at least one token")
    if noise is not None:
        _need(noise, "noise", torch.float32, 2)
        if noise.shape != logits.shape or noise.device != logits.device:
            raise ValueError("noise must match logits in shape and device")
    dev = logits.device
    max_rows = min(2 * S, E * int(capacity)) if drop_tokens else 2 * S
    lib = _lib.load()
    ws_bytes = lib.moe_routing_workspace_bytes(S, E)
    r = Routing.allocate(S, E, 2, max_rows, int(capacity), ws_bytes, dev)
    with _on(dev):
        rc = lib.moe_routing_top2_fwd(logits.data_ptr(), noise.data_ptr() if noise is not None else None,
                                      int(noise_seed) & 0xFFFFFFFFFFFFFFFF, S, E, int(capacity), int(bool(drop_tokens)),
                                      r.ptr("expert_idx"), r.ptr("slot"), r.ptr("weight"), r.ptr("exp_counts"),
                                      r.ptr("kept_counts"), r.ptr("expert_offsets"), r.ptr("perm"), r.ptr("l_aux"),
                                      r.ptr("workspace"), ws_bytes, _stream_ptr(logits))
    _lib.check(rc, "moe_routing_top2_fwd")
    return r


def routing_top1(logits: Tensor, capacity: int, used_token: Optional[Tensor] = None, use_rts: bool = True,
                 rts_uniform: Optional[Tensor] = None, rts_seed: int = 0, noise: Optional[Tensor] = None,
                 noise_seed: int = 0, drop_tokens: bool = True) -> Routing:
    """top1gating in index form.  ``rts_uniform`` / ``noise`` inject the random numbers (parity
    tests); otherwise they are generated in the kernel from the seeds."""
    _need(logits, "logits", torch.float32, 2)
    S, E = logits.shape
    if S < 1:
        raise ValueError("routing needs at least one token")
    dev = logits.device
    for t, name in ((rts_uniform, "rts_uniform"), (noise, "noise")):
        if t is not None:
            _need(t, name, torch.float32, 2)
            if t.shape != logits.shape or t.device != dev:
                raise ValueError(f"{name} must match logits in shape and device")
    ut = None
    if used_token is not None:
        if used_token.numel() != S or used_token.device != dev:
            raise ValueError("used_token must have one entry per token, on the logits' device")
        ut = (used_token.reshape(-1) != 0).to(torch.int32).contiguous()
    max_rows = min(S, E * int(capacity)) if drop_tokens else S
    lib = _lib.load()
    ws_bytes = lib.moe_routing_top1_workspace_bytes(S, E)
    r = Routing.allocate(S, E, 1, max_rows, int(capacity), ws_bytes, dev, used_token=ut)
    with _on(dev):
        rc = lib.moe_routing_top1_fwd(
            logits.data_ptr(), noise.data_ptr() if noise is not None else None, int(noise_seed) & 0xFFFFFFFFFFFFFFFF,
            ut.data_ptr() if ut is not None else None, rts_uniform.data_ptr() if rts_uniform is not None else None,
            int(rts_seed) & 0xFFFFFFFFFFFFFFFF, int(bool(use_rts)), S, E, int(capacity), int(bool(drop_tokens)),
            r.ptr("expert_idx"), r.ptr("slot"), r.ptr("weight"), r.ptr("exp_counts"), r.ptr("kept_counts"),
            r.ptr("expert_offsets"), r.ptr("perm"), r.ptr("l_aux"), r.ptr("workspace"), ws_bytes, _stream_ptr(logits))
    _lib.check(rc, "moe_routing_top1_fwd")
    return r


def _check_ffn_weights(act, a, w1, b1, w2, b2):
    _need(a, "a", HALF_DTYPES, 2)
    _need(w1, "w1", a.dtype, 3)            # activations and expert weights share one 16-bit type
    _need(w2, "w2", a.dtype, 3)
    M = a.shape[1]
    n_local, M2, I = w2.shape
    n1 = 2 * I if act == ACT_SWIGLU else I
    if M2 != M or tuple(w1.shape) != (n_local, n1, M):
        raise ValueError(f"weight shapes w1 {tuple(w1.shape)} / w2 {tuple(w2.shape)} do not match M={M}, act={act}")
    for b, name, n in ((b1, "b1", n1), (b2, "b2", M)):
        if b is not None:
            _need(b, name, a.dtype, 2)
            if tuple(b.shape) != (n_local, n):
                raise ValueError(f"{name}: expected {(n_local, n)}, got {tuple(b.shape)}")
            if b.device != a.device:
                raise ValueError("all operands must be on the same device")
    if w1.device != a.device or w2.device != a.device:
        raise ValueError("all operands must be on the same device")
    return M, I, n_local


def _check_out(t: Tensor, name: str, like: Tensor, cols: int) -> None:
    if (not isinstance(t, Tensor) or t.dtype != like.dtype or t.device != like.device or t.dim() != 2
            or t.shape[1] != cols or not t.is_contiguous()):
        raise ValueError(f"out[{name}]: need a contiguous [rows, {cols}] {like.dtype} tensor on {like.device}")


def _expert_ffn_launch(act, a, gather_ptr, gather_shift, w1, b1, w2, b2, offs_ptr, G, max_rows, M, I, n_local, timer,
                       n_gated, out=None):
    dev = a.device
    if out is not None:
        # caller-provided (y, h) indexed by ABSOLUTE compact row (the chunks of the expert-parallel exchange write their
        # rows of one shared pair of buffers); the caller guarantees they cover every row the group offsets name
        y, h = out
        _check_out(y, "y", a, M)
        _check_out(h, "h", a, I)
        if max_rows == 0:
            return y
    else:
        y = torch.empty((max(max_rows, 1), M), dtype=a.dtype, device=dev)
        if max_rows == 0:
            return y
        h = torch.empty((max_rows, I), dtype=a.dtype, device=dev)
    dt = _dt(a)
    lib = _lib.load()
    b1p = b1.data_ptr() if b1 is not None else None
    b2p = b2.data_ptr() if b2 is not None else None
    st = _stream_ptr(a)
    skp, skb = _splitk_ws(a, st)
    with _on(dev):
        if timer is None:
            rc = lib.moe_expert_ffn_fwd(act, a.data_ptr(), a.stride(0), gather_ptr, gather_shift, w1.data_ptr(), b1p,
                                        w2.data_ptr(), b2p, offs_ptr, G, n_local, max_rows, M, I, n_gated, h.data_ptr(),
                                        y.data_ptr(), skp, skb, dt, st)
            _lib.check(rc, "moe_expert_ffn_fwd")
        else:
            # same two launches through the single-GEMM primitive, with events between them
            epi = EPI_SWIGLU if act == ACT_SWIGLU else EPI_GELU_NEW
            timer.mark("fc1_begin")
            rc = lib.moe_grouped_gemm_fwd(epi, a.data_ptr(), a.stride(0), gather_ptr, gather_shift, w1.data_ptr(), b1p,
                                          offs_ptr, G, n_local, max_rows, M, I, n_gated, h.data_ptr(), I, skp, skb, dt, st)
            _lib.check(rc, "moe_grouped_gemm_fwd(fc1)")
            timer.mark("fc1_end")
            rc = lib.moe_grouped_gemm_fwd(EPI_PLAIN, h.data_ptr(), I, None, 0, w2.data_ptr(), b2p, offs_ptr, G, n_local,
                                          max_rows, I, M, -1, y.data_ptr(), M, skp, skb, dt, st)
            _lib.check(rc, "moe_grouped_gemm_fwd(fc2)")
            timer.mark("fc2_end")
    return y


def expert_ffn(act: int, a: Tensor, gather: Optional[Tensor], gather_shift: int,
               w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor],
               group_offsets: Tensor, max_rows: int, timer: Optional["PhaseTimer"] = None,
               n_gated: int = -1, out=None) -> Tensor:
    """Grouped expert MLP over compact rows; returns y [max_rows, M] (rows past the device-side
    group_offsets[G] are left unwritten).  ``out=(y, h)``: write into these buffers instead (indexed by the absolute
    rows the group offsets name; ``max_rows`` then only bounds the number of rows of THIS call)."""
    M, I, n_local = _check_ffn_weights(act, a, w1, b1, w2, b2)
    _need(group_offsets, "group_offsets", torch.int32, 1)
    G = group_offsets.numel() - 1
    if G < 1:
        raise ValueError("group_offsets needs at least 2 entries")
    if gather is not None:
        _need(gather, "gather", torch.int32, 1)
        if gather.numel() < max_rows:
            raise ValueError("gather shorter than max_rows")
    elif a.shape[0] < max_rows:
        raise ValueError("a has fewer rows than max_rows")
    for t in (group_offsets, gather):
        if t is not None and t.device != a.device:
            raise ValueError("all operands must be on the same device")
    return _expert_ffn_launch(act, a, gather.data_ptr() if gather is not None else None, gather_shift, w1, b1, w2, b2,
                              group_offsets.data_ptr(), G, max_rows, M, I, n_local, timer, n_gated, out)


def expert_ffn_routed(act: int, x: Tensor, r: Routing, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor],
                      timer: Optional["PhaseTimer"] = None, n_gated: int = -1) -> Tensor:
    """``expert_ffn`` over the rows a Routing dispatched (gather through its perm, one group per expert), taking the
    addresses straight from the routing arena."""
    M, I, n_local = _check_ffn_weights(act, x, w1, b1, w2, b2)
    if n_local != r.E:
        raise ValueError(f"{n_local} local experts but the routing has {r.E}")
    return _expert_ffn_launch(act, x, r.ptr("perm"), r.top_k - 1, w1, b1, w2, b2, r.ptr("expert_offsets"), r.E, r.max_rows,
                              M, I, n_local, timer, n_gated)


def block_top2(x: Tensor, wg_weight: Tensor, capacity: int, noise: Optional[Tensor], noise_seed: int, drop_tokens: bool,
               act: int, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], n_gated: int = -1,
               noise_seed_dev: Optional[Tensor] = None):
    """Whole block on one GPU in one library call (gate -> top-2 routing -> expert FFN -> combine).
    Returns (out [S,M] bf16, logits [S,E] fp32, Routing).  ``noise_seed_dev`` (1-element int64 device tensor): the
    Gumbel seed lives on the device and the routing kernel advances it -- the form a HIP graph can replay."""
    M, I, n_local = _check_ffn_weights(act, x, w1, b1, w2, b2)
    wgd = _wg_dt(wg_weight, x)
    dt = _dt(x)
    _need(wg_weight, "wg.weight", wg_weight.dtype, 2)
    S = x.shape[0]
    E = wg_weight.shape[0]
    if wg_weight.shape[1] != M or wg_weight.device != x.device or n_local != E:
        raise ValueError("wg.weight / expert weights do not match x")
    if noise is not None:
        _need(noise, "noise", torch.float32, 2)
        if tuple(noise.shape) != (S, E) or noise.device != x.device:
            raise ValueError("noise must be [S, E] on x's device")
    dev = x.device
    max_rows = min(2 * S, E * int(capacity)) if drop_tokens else 2 * S
    lib = _lib.load()
    ws_bytes = lib.moe_routing_workspace_bytes(S, E)
    r = Routing.allocate(S, E, 2, max_rows, int(capacity), ws_bytes, dev)
    logits = torch.empty((S, E), dtype=torch.float32, device=dev)
    rows = max(max_rows, 1)
    hy = torch.empty((rows * (I + M),), dtype=x.dtype, device=dev)   # h workspace, then y
    out = torch.empty((S, M), dtype=x.dtype, device=dev)
    hp = hy.data_ptr()
    st = _stream_ptr(x)
    skp, skb = _splitk_ws(x, st) if max_rows > 64 else (None, 0)     # decode-size launches take the GEMV-style kernel
    if noise_seed_dev is not None:
        _need(noise_seed_dev, "noise_seed_dev", torch.int64)
        if noise is not None or noise_seed_dev.numel() != 1 or noise_seed_dev.device != dev:
            raise ValueError("noise_seed_dev: one int64 on x's device, and no noise tensor with it")
        with _on(dev):
            rc = lib.moe_block_top2_fwd_dseed(
                x.data_ptr(), wg_weight.data_ptr(), wgd, logits.data_ptr(),
                noise_seed_dev.data_ptr(), S, M, E, I, act,
                int(capacity), int(bool(drop_tokens)), n_gated, r.ptr("expert_idx"), r.ptr("slot"), r.ptr("weight"),
                r.ptr("exp_counts"), r.ptr("kept_counts"), r.ptr("expert_offsets"), r.ptr("perm"), r.ptr("l_aux"),
                r.ptr("workspace"), ws_bytes, w1.data_ptr(), b1.data_ptr() if b1 is not None else None, w2.data_ptr(),
                b2.data_ptr() if b2 is not None else None, hp, hp + rows * I * 2, out.data_ptr(), skp, skb, dt, st)
        _lib.check(rc, "moe_block_top2_fwd_dseed")
        r._keep = hy
        return out, logits, r
    with _on(dev):
        rc = lib.moe_block_top2_fwd(
            x.data_ptr(), wg_weight.data_ptr(), wgd, logits.data_ptr(),
            noise.data_ptr() if noise is not None else None, int(noise_seed) & 0xFFFFFFFFFFFFFFFF, S, M, E, I, act,
            int(capacity), int(bool(drop_tokens)), n_gated, r.ptr("expert_idx"), r.ptr("slot"), r.ptr("weight"),
            r.ptr("exp_counts"), r.ptr("kept_counts"), r.ptr("expert_offsets"), r.ptr("perm"), r.ptr("l_aux"),
            r.ptr("workspace"), ws_bytes, w1.data_ptr(), b1.data_ptr() if b1 is not None else None, w2.data_ptr(),
            b2.data_ptr() if b2 is not None else None, hp, hp + rows * I * 2, out.data_ptr(), skp, skb, dt, st)
    _lib.check(rc, "moe_block_top2_fwd")
    r._keep = hy                                   # y lives in hy; keep it alive with the routing (inspection only)
    return out, logits, r


def gather_rows(src: Tensor, gather: Tensor, gather_shift: int, n_rows_dev: Tensor, max_rows: int) -> Tensor:
    """dst[p] = src[gather[p] >> shift] for p < n_rows_dev[0] (expert-sorted send buffer of the EP all-to-all)."""
    _need(src, "src", HALF_DTYPES, 2)
    _need(gather, "gather", torch.int32, 1)
    _need(n_rows_dev, "n_rows_dev", torch.int32)
    if gather.numel() < max_rows:
        raise ValueError("gather shorter than max_rows")
    dst = torch.empty((max(max_rows, 1), src.shape[1]), dtype=src.dtype, device=src.device)
    lib = _lib.load()
    with _on(src.device):
        rc = lib.moe_gather_rows_fwd(src.data_ptr(), gather.data_ptr(), gather_shift, n_rows_dev.data_ptr(),
                                     max_rows, src.shape[1], dst.data_ptr(), _stream_ptr(src))
    _lib.check(rc, "moe_gather_rows_fwd")
    return dst


def ep_chunk_rows(src: Tensor, gather: Optional[Tensor], gather_shift: int, expert_offsets: Tensor, n_chunks: int,
                  unpack: bool, max_rows: int, out: Optional[Tensor] = None) -> Tensor:
    """Send buffer (``unpack=False``: rows gathered through ``gather`` and laid out chunk-major) or returned rows put back
    in compact expert-major order (``unpack=True``) of the chunked expert-parallel exchange; positions are computed on the
    device from ``expert_offsets`` ([E+1] int32)."""
    _need(src, "src", HALF_DTYPES, 2)
    _need(expert_offsets, "expert_offsets", torch.int32, 1)
    if gather is not None:
        _need(gather, "gather", torch.int32, 1)
        if gather.numel() < max_rows:
            raise ValueError("gather shorter than max_rows")
    E = expert_offsets.numel() - 1
    M = src.shape[1]
    if out is None:
        out = torch.empty((max(max_rows, 1), M), dtype=src.dtype, device=src.device)
    elif out.shape[0] < max_rows or out.shape[1] != M or out.dtype != src.dtype or not out.is_contiguous():
        raise ValueError("out does not match")
    lib = _lib.load()
    with _on(src.device):
        rc = lib.moe_ep_chunk_rows_fwd(src.data_ptr(), gather.data_ptr() if gather is not None else None, gather_shift,
                                       expert_offsets.data_ptr(), E, int(n_chunks), int(bool(unpack)), max_rows, M,
                                       out.data_ptr(), _stream_ptr(src))
    _lib.check(rc, "moe_ep_chunk_rows_fwd")
    return out


def combine(y: Tensor, r: Routing, S: int, M: int) -> Tensor:
    """out[s] = sum_j bf16(w[s,j]) * y[pos(s,j)]  (the ``sec,ecm->sm`` einsum)."""
    _need(y, "y", HALF_DTYPES, 2)
    if y.shape[1] != M or y.shape[0] < r.max_rows:
        raise ValueError(f"y {tuple(y.shape)} too small for max_rows={r.max_rows}, M={M}")
    if r.S != S:
        raise ValueError("routing does not match S")
    out = torch.empty((S, M), dtype=y.dtype, device=y.device)
    lib = _lib.load()
    with _on(y.device):
        rc = lib.moe_combine_fwd(y.data_ptr(), r.ptr("expert_idx"), r.ptr("slot"), r.ptr("weight"),
                                 r.ptr("expert_offsets"), S, M, r.top_k, out.data_ptr(), _dt(y), _stream_ptr(y))
    _lib.check(rc, "moe_combine_fwd")
    return out


# ---------------------------------------------------------------------------------------------------
# backward: thin wrappers of the *_bwd entry points, same validation style
# ---------------------------------------------------------------------------------------------------
def _ffn_shapes(act, a, w1, w2):
    M = a.shape[1]
    n_local, M2, I = w2.shape
    n1 = 2 * I if act == ACT_SWIGLU else I
    if M2 != M or tuple(w1.shape) != (n_local, n1, M):
        raise ValueError(f"weight shapes w1 {tuple(w1.shape)} / w2 {tuple(w2.shape)} do not match M={M}, act={act}")
    return M, I, n1, n_local


def expert_ffn_train(act: int, a: Tensor, gather: Optional[Tensor], gather_shift: int,
                     w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor],
                     group_offsets: Tensor, max_rows: int, n_gated: int = -1, out=None):
    """``expert_ffn`` that keeps what the backward needs: returns (y, ab, h).  ``out=(y, ab, h)``: write into these
    buffers (absolute compact rows, as in ``expert_ffn``)."""
    _need(a, "a", torch.bfloat16, 2)
    _need(w1, "w1", torch.bfloat16, 3)
    _need(w2, "w2", torch.bfloat16, 3)
    _need(group_offsets, "group_offsets", torch.int32, 1)
    M, I, n1, n_local = _ffn_shapes(act, a, w1, w2)
    G = group_offsets.numel() - 1
    if gather is not None:
        _need(gather, "gather", torch.int32, 1)
        if gather.numel() < max_rows:
            raise ValueError("gather shorter than max_rows")
    elif a.shape[0] < max_rows:
        raise ValueError("a has fewer rows than max_rows")
    dev = a.device
    rows = max(max_rows, 1)
    if out is not None:
        y, ab, h = out
        _check_out(y, "y", a, M)
        _check_out(ab, "ab", a, n1)
        _check_out(h, "h", a, I)
    else:
        y = torch.empty((rows, M), dtype=torch.bfloat16, device=dev)
        ab = torch.empty((rows, n1), dtype=torch.bfloat16, device=dev)
        h = torch.empty((rows, I), dtype=torch.bfloat16, device=dev)
    if max_rows == 0:
        return y, ab, h
    lib = _lib.load()
    st = _stream_ptr(a)
    skp, skb = _splitk_ws(a, st)
    with _on(dev):
        rc = lib.moe_expert_ffn_fwd_train(
            act, a.data_ptr(), a.stride(0), gather.data_ptr() if gather is not None else None, gather_shift,
            w1.data_ptr(), b1.data_ptr() if b1 is not None else None, w2.data_ptr(),
            b2.data_ptr() if b2 is not None else None, group_offsets.data_ptr(), G, n_local, max_rows, M, I,
            n_gated, ab.data_ptr(), h.data_ptr(), y.data_ptr(), skp, skb, st)
    _lib.check(rc, "moe_expert_ffn_fwd_train")
    return y, ab, h


def expert_ffn_bwd(act: int, a: Tensor, gather: Optional[Tensor], gather_shift: int, w1: Tensor, w2: Tensor,
                   ab: Tensor, h: Tensor, dy: Tensor, group_offsets: Tensor, max_rows: int,
                   has_b1: bool, has_b2: bool, n_gated: int = -1):
    """Gradients of the grouped expert MLP: (da_rows [max_rows,M] bf16, dw1, db1|None, dw2, db2|None);
    weight gradients bf16 in the packed layouts, bias gradients fp32."""
    for t, name in ((a, "a"), (ab, "ab"), (h, "h"), (dy, "dy")):
        _need(t, name, torch.bfloat16, 2)
    _need(w1, "w1", torch.bfloat16, 3)
    _need(w2, "w2", torch.bfloat16, 3)
    _need(group_offsets, "group_offsets", torch.int32, 1)
    M, I, n1, n_local = _ffn_shapes(act, a, w1, w2)
    G = group_offsets.numel() - 1
    if G % n_local:
        raise ValueError("number of groups must be a multiple of the local expert count")
    rows = max(max_rows, 1)
    if tuple(ab.shape) != (rows, n1) or tuple(h.shape) != (rows, I) or dy.shape[1] != M or dy.shape[0] < max_rows:
        raise ValueError("saved activations / dy do not match the forward call")
    dev = a.device
    da = torch.empty((rows, M), dtype=torch.bfloat16, device=dev)
    dw1 = torch.empty_like(w1)
    dw2 = torch.empty_like(w2)
    db1 = torch.empty((n_local, n1), dtype=torch.float32, device=dev) if has_b1 else None
    db2 = torch.empty((n_local, M), dtype=torch.float32, device=dev) if has_b2 else None
    lib = _lib.load()
    nws = lib.moe_expert_ffn_bwd_workspace_bytes(act, G, n_local, max_rows, M, I)
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    with _on(dev):
        rc = lib.moe_expert_ffn_bwd(
            act, a.data_ptr(), a.stride(0), gather.data_ptr() if gather is not None else None, gather_shift,
            w1.data_ptr(), w2.data_ptr(), ab.data_ptr(), h.data_ptr(), dy.data_ptr(), group_offsets.data_ptr(), G,
            n_local, max_rows, M, I, n_gated, da.data_ptr(), dw1.data_ptr(), db1.data_ptr() if has_b1 else None,
            dw2.data_ptr(), db2.data_ptr() if has_b2 else None, ws.data_ptr(), nws, _stream_ptr(a))
    _lib.check(rc, "moe_expert_ffn_bwd")
    return da, dw1, db1, dw2, db2


def combine_bwd(dout: Tensor, y: Tensor, r: Routing, S: int, M: int):
    """(dy [rows,M] bf16, dweight [S,k] fp32) -- backward of ``combine``."""
    _need(dout, "dout", torch.bfloat16, 2)
    _need(y, "y", torch.bfloat16, 2)
    if tuple(dout.shape) != (S, M) or y.shape[1] != M:
        raise ValueError("dout / y shapes do not match")
    dy = torch.empty_like(y)
    k = r.top_k
    dweight = torch.empty((S, k), dtype=torch.float32, device=y.device)
    lib = _lib.load()
    with _on(y.device):
        rc = lib.moe_combine_bwd(dout.data_ptr(), y.data_ptr(), r.ptr("expert_idx"), r.ptr("slot"),
                                 r.ptr("weight"), r.ptr("expert_offsets"), S, M, k, dy.data_ptr(),
                                 dweight.data_ptr(), _stream_ptr(y))
    _lib.check(rc, "moe_combine_bwd")
    return dy, dweight


def gate_bwd(x: Tensor, logits: Tensor, r: Routing, dweight: Tensor, dl_aux: Optional[Tensor], jitter_seed: int = 0):
    """(dlogits [S,E] fp32, dwg [E,M] fp32) -- backward of top2gating and of the gate's F.linear w.r.t. wg
    (``jitter_seed``: the seed the forward gate call was given)."""
    _need(x, "x", torch.bfloat16, 2)
    _need(logits, "logits", torch.float32, 2)
    _need(dweight, "dweight", torch.float32, 2)
    S, M = x.shape
    E = logits.shape[1]
    if dl_aux is not None:
        _need(dl_aux, "dl_aux", torch.float32)
    dlogits = torch.empty_like(logits)
    dwg = torch.empty((E, M), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nws = lib.moe_gate_bwd_workspace_bytes(S, M, E)
    ws = torch.empty((nws,), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        rc = lib.moe_gate_bwd(x.data_ptr(), logits.data_ptr(), r.ptr("expert_idx"), r.ptr("slot"),
                              dweight.data_ptr(), dl_aux.data_ptr() if dl_aux is not None else None,
                              r.used_token.data_ptr() if r.used_token is not None else None, S, M, E, r.top_k,
                              dlogits.data_ptr(), dwg.data_ptr(), ws.data_ptr(), nws, int(jitter_seed), _stream_ptr(x))
    _lib.check(rc, "moe_gate_bwd")
    return dlogits, dwg


def dispatch_bwd(da_rows: Tensor, r: Routing, dlogits: Optional[Tensor], wg_weight: Optional[Tensor],
                 S: int, M: int, jitter_seed: int = 0) -> Tensor:
    """dx [S,M] bf16: sum of the token's compact-row gradients + the gate term dlogits @ wg (times the jitter factors of
    ``jitter_seed`` when the forward gate call had one)."""
    _need(da_rows, "da_rows", torch.bfloat16, 2)
    E = r.E
    dx = torch.empty((S, M), dtype=torch.bfloat16, device=da_rows.device)
    lib = _lib.load()
    with _on(da_rows.device):
        rc = lib.moe_dispatch_bwd(da_rows.data_ptr(), r.ptr("expert_idx"), r.ptr("slot"),
                                  r.ptr("expert_offsets"),
                                  dlogits.data_ptr() if dlogits is not None else None,
                                  wg_weight.data_ptr() if wg_weight is not None else None,
                                  int(wg_weight is not None and wg_weight.dtype == torch.bfloat16), S, M, E,
                                  r.top_k, dx.data_ptr(), int(jitter_seed), _stream_ptr(da_rows))
    _lib.check(rc, "moe_dispatch_bwd")
    return dx
