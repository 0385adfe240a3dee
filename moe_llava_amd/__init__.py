"""MI355X-native sparse-MoE FFN block: drop-in for ``deepspeed.moe.layer.MoE`` as used by MoE-LLaVA.

``from moe_llava_amd import MoE`` (or ``moe_llava_amd.shim.install()`` and keep the reference's
``from deepspeed.moe.layer import MoE``).  Kernels live in ``libmoe_gfx950.so`` (``include/moe_gfx950.h``).
"""
from .layer import MoE  # noqa: F401
from .sharded_moe import MOELayer, TopKGate, capacity_of  # noqa: F401
from .experts import Experts, ExpertSpec, register_expert_adapter  # noqa: F401

__all__ = ["MoE", "MOELayer", "TopKGate", "Experts", "ExpertSpec", "register_expert_adapter", "capacity_of"]
