"""ctypes binding of libmoe_gfx950.so (the C ABI of ``include/moe_gfx950.h``).

There is deliberately no fallback: if the shared library is missing or a symbol
cannot be resolved, importing the ops raises, so a GPU box can never pass on a
silent eager path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_int64, c_size_t, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
# MOE_GFX950_LIB points the loader at another build of the same ABI (kernel A/B experiments, tools/variants)
LIB_PATH = os.environ.get("MOE_GFX950_LIB") or os.path.join(_PKG, "libmoe_gfx950.so")

ABI_VERSION = 8

# (name, restype, argtypes) -- must mirror include/moe_gfx950.h exactly
_SIGNATURES = [
    ("moe_abi_version", c_int, []),
    ("moe_last_error", c_char_p, []),
    ("moe_gate_logits_fwd", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_uint64, c_void_p]),
    ("moe_gate_jitter_fill", c_int, [c_uint64, c_int, c_int, c_void_p, c_void_p]),
    ("moe_routing_workspace_bytes", c_size_t, [c_int, c_int]),
    ("moe_routing_top2_fwd", c_int,
     [c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("moe_routing_top2_fwd_dseed", c_int,
     [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("moe_routing_top2_fwd_noalloc", c_int,
     [c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_void_p, c_void_p]),
    ("moe_routing_top1_workspace_bytes", c_size_t, [c_int, c_int]),
    ("moe_routing_top1_fwd", c_int,
     [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("moe_splitk_workspace_bytes", c_size_t, []),
    ("moe_reload_knobs", None, []),
    ("moe_expert_ffn_fwd", c_int,
     [c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
      c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    ("moe_grouped_gemm_fwd", c_int,
     [c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
      c_int, c_void_p, c_int64, c_void_p, c_size_t, c_int, c_void_p]),
    ("moe_gather_rows_fwd", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("moe_ep_chunk_rows_fwd", c_int,
     [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("moe_combine_fwd", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    ("moe_block_top2_fwd", c_int,
     [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    ("moe_block_top2_fwd_dseed", c_int,
     [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    ("moe_block_top2_fwd_args", c_int, [c_void_p]),
    # ---- backward ----
    ("moe_expert_ffn_fwd_train", c_int,
     [c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
      c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("moe_expert_ffn_bwd_workspace_bytes", c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    ("moe_expert_ffn_bwd", c_int,
     [c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
      c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
      c_void_p]),
    ("moe_combine_bwd", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("moe_gate_bwd_workspace_bytes", c_size_t, [c_int, c_int, c_int]),
    ("moe_gate_bwd", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
      c_void_p, c_void_p, c_size_t, c_uint64, c_void_p]),
    ("moe_dispatch_bwd", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
      c_uint64, c_void_p]),
]

class BlockTop2Args(ctypes.Structure):
    """``moe_block_top2_args`` of include/moe_gfx950.h (field for field)."""
    _fields_ = [
        ("x", c_void_p), ("wg", c_void_p), ("wg_dtype", c_int), ("logits", c_void_p), ("noise", c_void_p),
        ("noise_seed", c_uint64), ("noise_seed_dev", c_void_p),
        ("S", c_int), ("M", c_int), ("E", c_int), ("I", c_int), ("act", c_int), ("capacity", c_int),
        ("drop_tokens", c_int), ("n_gated", c_int),
        ("expert_idx", c_void_p), ("slot", c_void_p), ("weight", c_void_p), ("exp_counts", c_void_p),
        ("kept_counts", c_void_p), ("expert_offsets", c_void_p), ("perm", c_void_p), ("l_aux", c_void_p),
        ("routing_ws", c_void_p), ("routing_ws_bytes", c_size_t),
        ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
        ("h_ws", c_void_p), ("y", c_void_p), ("out", c_void_p),
        ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_size_t),
        ("dtype", c_int), ("stream", c_void_p),
    ]


EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


class MoEKernelError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the library once.  ``import torch`` must have happened first on a GPU box so that the HIP
    runtime torch ships (same SONAME) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MoEKernelError(
            f"{LIB_PATH} not found: build it with `python -m moe_llava_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU or eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in _SIGNATURES:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise MoEKernelError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.moe_abi_version()
    if v != ABI_VERSION:
        raise MoEKernelError(f"ABI version mismatch: library {v}, python {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().moe_last_error()
        raise MoEKernelError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
