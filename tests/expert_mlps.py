"""The three expert MLP forms the reference feeds to ``MoE(expert=...)``, as plain
``nn.Module`` containers with the reference's parameter names.  ``bench.py`` / ``smoke()`` use them to build
random-init blocks of the right architecture; with the real reference the user passes its own MLP modules.

The ``forward`` of these classes is never called by the HIP path (``Experts`` reads their ``nn.Linear`` parameters
and runs the grouped GEMM); it exists so the modules remain ordinary torch modules.

* StableLM ``MLP``:  ``moellava/model/language_model/stablelm/modeling_stablelm_epoch.py:156-168``
* Phi ``PhiMLP``:    ``moellava/model/language_model/phi/modeling_phi.py:203-215``
* Qwen ``QWenMLP``:  ``moellava/model/language_model/qwen/modeling_qwen.py:589-606``
* the fork's ``RePaMLP`` (masked SwiGLU): ``moellava/model/language_model/llava_stablelm_moe.py:613-655``
  (StableLM names) and ``llava_qwen_moe.py:693-738`` (Qwen names)
"""
from __future__ import annotations

import math

import torch
from torch import nn
from torch.nn import functional as F


class NewGELUActivation(nn.Module):
    """tanh-form GELU (HF ``gelu_new``)."""

    def forward(self, x):
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class StableLMMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.gate_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.up_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.down_proj = nn.Linear(intermediate_size, hidden_size, bias=False)
        self.act_fn = nn.SiLU()

    def forward(self, x):
        return self.down_proj(self.act_fn(self.gate_proj(x)) * self.up_proj(x))


class PhiMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.activation_fn = NewGELUActivation()
        self.fc1 = nn.Linear(hidden_size, intermediate_size)
        self.fc2 = nn.Linear(intermediate_size, hidden_size)

    def forward(self, x):
        return self.fc2(self.activation_fn(self.fc1(x)))


class QWenMLP(nn.Module):
    def __init__(self, hidden_size: int, ff_dim: int, bias: bool = False):
        super().__init__()
        self.w1 = nn.Linear(hidden_size, ff_dim, bias=bias)
        self.w2 = nn.Linear(hidden_size, ff_dim, bias=bias)
        self.c_proj = nn.Linear(ff_dim, hidden_size, bias=bias)

    def forward(self, x):
        return self.c_proj(self.w1(x) * F.silu(self.w2(x)))


class RePaMLP(nn.Module):
    """Masked-SwiGLU expert of the fork: hidden channels ``>= num_gated_channels`` skip the SiLU (their gate is the
    constant 1), so after training they can be folded into one dense ``hidden x hidden`` matrix (``reparam``), which
    leaves a narrower ordinary SwiGLU expert.  ``names`` picks the StableLM or the Qwen parameter names."""

    _NAMES = {"stablelm": ("gate_proj", "up_proj", "down_proj"), "qwen": ("w2", "w1", "c_proj")}

    def __init__(self, hidden_size: int, intermediate_size: int, names: str = "stablelm", bias: bool = False):
        super().__init__()
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self._gate, self._up, self._down = self._NAMES[names]
        setattr(self, self._gate, nn.Linear(hidden_size, intermediate_size, bias=bias))
        setattr(self, self._up, nn.Linear(hidden_size, intermediate_size, bias=bias))
        setattr(self, self._down, nn.Linear(intermediate_size, hidden_size, bias=bias))
        if names == "stablelm":
            self.act_fn = nn.SiLU()
        self.gated_ratio = 1.0
        self.num_gated_channels = intermediate_size
        self.register_buffer("mask", torch.ones(intermediate_size, dtype=torch.bool))
        self.reparamed = False

    def forward(self, x):
        gate, up, down = getattr(self, self._gate), getattr(self, self._up), getattr(self, self._down)
        g = F.silu(gate(x))
        if not self.reparamed:
            g = torch.where(self.mask, g, torch.ones_like(g))
        return down(g * up(x))

    def adjust_gated_ratio(self, gated_ratio: float) -> None:
        self.gated_ratio = gated_ratio
        self.num_gated_channels = int(self.intermediate_size * gated_ratio)
        self.mask.fill_(False)
        self.mask[: self.num_gated_channels] = True

    def reparam(self):
        """Fold the ungated channels: returns the ``[hidden, hidden]`` weight of the equivalent dense Linear and
        shrinks the three projections to the gated channels."""
        if self.reparamed:
            return None
        gate, up, down = getattr(self, self._gate), getattr(self, self._up), getattr(self, self._down)
        n = self.num_gated_channels
        folded = up.weight[n:].T @ down.weight[:, n:].T
        gate.weight = nn.Parameter(gate.weight[:n])
        up.weight = nn.Parameter(up.weight[:n])
        down.weight = nn.Parameter(down.weight[:, :n])
        self.reparamed = True
        return folded.T
