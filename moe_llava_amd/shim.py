"""Install this package under the import path the reference uses.

Every MoE model file of the reference does ``from deepspeed.moe.layer import MoE``
(``moellava/model/language_model/llava_stablelm_moe.py:29`` and the six siblings).  ``install()`` registers
``deepspeed.moe.layer`` / ``.sharded_moe`` / ``.experts`` modules that resolve to the gfx950 implementation, so
those files run unchanged.  If a real DeepSpeed is importable only its ``moe.layer`` / ``moe.sharded_moe`` /
``moe.experts`` entries are overridden; otherwise minimal parent packages are created.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types


def install(force: bool = True) -> None:
    from . import experts, layer, sharded_moe

    def _ensure(name: str) -> types.ModuleType:
        if name in sys.modules:
            return sys.modules[name]
        try:
            if importlib.util.find_spec(name) is not None:
                return importlib.import_module(name)
        except (ImportError, ValueError):
            pass
        m = types.ModuleType(name)
        m.__path__ = []  # mark as package
        sys.modules[name] = m
        return m

    ds = _ensure("deepspeed")
    moe = _ensure("deepspeed.moe")
    if not hasattr(ds, "moe"):
        ds.moe = moe
    for short, mod in (("layer", layer), ("sharded_moe", sharded_moe), ("experts", experts)):
        full = f"deepspeed.moe.{short}"
        if force or full not in sys.modules:
            sys.modules[full] = mod
            setattr(moe, short, mod)
