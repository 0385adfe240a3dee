"""The three expert MLP forms the reference feeds to ``MoE(expert=...)``, as plain
``nn.Module`` containers with the reference's parameter names.  ``bench.py`` / ``smoke()`` use them to build
random-init blocks of the right architecture; with the real reference the user passes its own MLP modules.

The ``forward`` of these classes is never called by the HIP path (``Experts`` reads their ``nn.Linear`` parameters
and runs the grouped GEMM); it exists so the modules remain ordinary torch modules.

* StableLM ``MLP``:  ``moellava/model/language_model/stablelm/modeling_stablelm_epoch.py:156-168``
* Phi ``PhiMLP``:    ``moellava/model/language_model/phi/modeling_phi.py:203-215``
* Qwen ``QWenMLP``:  ``moellava/model/language_model/qwen/modeling_qwen.py:589-606``
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
