"""Expert MLP forms of the reference, restated for the oracle -- TEST ORACLE ONLY.

Each class keeps the reference's parameter names (they are the checkpoint ABI,
SURVEY.md 8(b)) and the reference's op order.  Citations into /root/reference:

* StableLM: ``moellava/model/language_model/stablelm/modeling_stablelm_epoch.py:156-168``
* Phi:      ``moellava/model/language_model/phi/modeling_phi.py:203-215``
            (``gelu_new`` = tanh approximation, ``phi/configuration_phi.py:128``)
* Qwen:     ``moellava/model/language_model/qwen/modeling_qwen.py:589-606``
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    """HF ``NewGELUActivation``: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class GeluNew(nn.Module):
    def forward(self, x):
        return gelu_new(x)


class StableLMMLPRef(nn.Module):
    """down_proj(silu(gate_proj(x)) * up_proj(x)), no biases."""

    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.gate_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.up_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.down_proj = nn.Linear(intermediate_size, hidden_size, bias=False)
        self.act_fn = nn.SiLU()

    def forward(self, x):
        return self.down_proj(self.act_fn(self.gate_proj(x)) * self.up_proj(x))


class PhiMLPRef(nn.Module):
    """fc2(gelu_new(fc1(x))), both Linear layers with bias."""

    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.activation_fn = GeluNew()
        self.fc1 = nn.Linear(hidden_size, intermediate_size)
        self.fc2 = nn.Linear(intermediate_size, hidden_size)

    def forward(self, x):
        return self.fc2(self.activation_fn(self.fc1(x)))


class QWenMLPRef(nn.Module):
    """c_proj(w1(x) * silu(w2(x))); ``ff_dim`` is the per-projection width
    (``config.intermediate_size // 2`` in the reference)."""

    def __init__(self, hidden_size: int, ff_dim: int, bias: bool = False):
        super().__init__()
        self.w1 = nn.Linear(hidden_size, ff_dim, bias=bias)
        self.w2 = nn.Linear(hidden_size, ff_dim, bias=bias)
        self.c_proj = nn.Linear(ff_dim, hidden_size, bias=bias)

    def forward(self, x):
        a1 = self.w1(x)
        a2 = self.w2(x)
        return self.c_proj(a1 * F.silu(a2))


class RePaMLPRef(nn.Module):
    """The fork's masked-SwiGLU expert, restated from
    ``moellava/model/language_model/llava_stablelm_moe.py:613-655`` (StableLM parameter names) and
    ``llava_qwen_moe.py:693-738`` (Qwen names: the SiLU goes on ``w2``, ``w1`` is the plain projection).

    forward (``:632-638``): not reparamed -> ``down(where(mask, silu(gate(x)), 1) * up(x))``; reparamed -> plain SwiGLU
    over the shrunk projections.  ``adjust_gated_ratio`` (``:649-655``) sets a prefix mask of
    ``int(intermediate * ratio)`` channels.  ``reparam`` (``:640-647``) returns
    ``(up.weight[n:].T @ down.weight[:, n:].T).T`` and keeps the first ``n`` channels.  TEST INFRASTRUCTURE ONLY."""

    def __init__(self, hidden_size: int, intermediate_size: int, names: str = "stablelm"):
        super().__init__()
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.names = names
        g, u, d = ("gate_proj", "up_proj", "down_proj") if names == "stablelm" else ("w2", "w1", "c_proj")
        self._g, self._u, self._d = g, u, d
        setattr(self, g, nn.Linear(hidden_size, intermediate_size, bias=False))
        setattr(self, u, nn.Linear(hidden_size, intermediate_size, bias=False))
        setattr(self, d, nn.Linear(intermediate_size, hidden_size, bias=False))
        self.gated_ratio = 1.0
        self.num_gated_channels = int(intermediate_size * self.gated_ratio)
        self.register_buffer("mask", torch.ones(intermediate_size, dtype=torch.bool))
        self.adjust_gated_ratio(self.gated_ratio)
        self.reparamed = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        gate, up, down = getattr(self, self._g), getattr(self, self._u), getattr(self, self._d)
        if self.reparamed:
            return down(F.silu(gate(x)) * up(x))
        x_gate = F.silu(gate(x))
        x_mask = torch.where(self.mask[None, None, :] if x_gate.dim() == 3 else self.mask, x_gate, torch.ones_like(x_gate))
        return down(x_mask * up(x))

    def reparam(self):
        if not self.reparamed:
            gate, up, down = getattr(self, self._g), getattr(self, self._u), getattr(self, self._d)
            n = self.num_gated_channels
            w = up.weight[n:].T @ down.weight[:, n:].T
            gate.weight = nn.Parameter(gate.weight[:n])
            up.weight = nn.Parameter(up.weight[:n])
            down.weight = nn.Parameter(down.weight[:, :n])
            self.reparamed = True
            return w.T

    def adjust_gated_ratio(self, gated_ratio):
        self.gated_ratio = gated_ratio
        self.num_gated_channels = int(self.intermediate_size * self.gated_ratio)
        self.mask.fill_(False)
        self.mask[:self.num_gated_channels] = True
