"""HIP-graph replay of the MoE block forward for a fixed token count (decode loops).

A generation step runs the block on S = batch tokens: five short kernels whose launch cost on the host exceeds their
GPU time.  ``GraphedMoE`` captures ONE forward of a local (ep_size == 1), top-2, no-grad MoE block into a HIP graph
(``torch.cuda.CUDAGraph`` -- hipGraph on ROCm) and replays it per step: one host call instead of the Python path, and no
gaps between the kernels.  What makes the step capturable:

* the C ABI never allocates or synchronises and its launch count does not depend on the data (include/moe_gfx950.h);
* the Gumbel seed of the second-expert sampling lives in device memory and the routing kernel advances it
  (``moe_block_top2_fwd_dseed``), so every replay draws fresh noise although no launch argument changes.

The reference has no counterpart (it is eager PyTorch; `deepspeed.moe.layer.MoE.forward`); outputs are those of the
eager path for the same seed sequence.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops, sharded_moe


class GraphedMoE:
    """``g = GraphedMoE(moe, example)`` then ``out, l_aux, exp_counts = g(x)`` with ``x.shape == example.shape``.

    ``moe`` is a ``moe_llava_amd.MoE`` (or its ``MOELayer``) in eval or train mode; gradients are not recorded.  The
    returned tensors are the graph's static outputs: they are overwritten by the next call (clone what must survive).
    """

    def __init__(self, moe, example: Tensor, seed: Optional[int] = None, warmup: int = 2):
        layer = getattr(moe, "deepspeed_moe", moe)
        gate = layer.gate
        if not example.is_cuda or example.dtype not in ops.HALF_DTYPES:
            raise TypeError("GraphedMoE: example must be a bf16 or fp16 tensor on the GPU")
        if layer.ep_size != 1 or layer.force_exchange:
            raise NotImplementedError("GraphedMoE: expert-parallel steps are not captured (the exchange sizes its "
                                      "splits on the host)")
        if gate.k != 2 or gate.noisy_gate_policy is not None or sharded_moe.module_has_hooks(gate.wg) or layer.timer is not None:
            raise NotImplementedError("GraphedMoE: needs the one-call path (top-2 gate, no noisy_gate_policy, no "
                                      "forward hooks on wg, no phase timer)")
        S = example.numel() // example.shape[-1]
        if S > 4096 or gate.wg.weight.shape[0] > 16:
            raise NotImplementedError("GraphedMoE: the device-side seed needs S <= 4096 tokens and E <= 16 experts")
        self.moe, self.layer, self.gate = moe, layer, gate
        dev = example.device
        if seed is None:
            seed = int(torch.randint(1, 2 ** 62, (1,), dtype=torch.int64).item())
        self.seed_dev = torch.tensor([seed], dtype=torch.int64, device=dev)
        self.x = example.detach().clone()
        self.graph = torch.cuda.CUDAGraph()
        gate.__dict__["noise_seed_dev"] = self.seed_dev
        try:
            with torch.no_grad():
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):                     # warm-up off the default stream (packs the weights)
                    for _ in range(max(1, warmup)):
                        self._forward()
                torch.cuda.current_stream(dev).wait_stream(side)
                with torch.cuda.graph(self.graph):
                    self.out = self._forward()
        finally:
            gate.__dict__.pop("noise_seed_dev", None)
        self.routing = layer.last_routing      # static routing arrays of the captured step (overwritten per replay)

    def _forward(self):
        res = self.moe(self.x)
        if not isinstance(res, tuple):         # a bare MOELayer returns the output only
            res = (res, self.layer.l_aux, self.layer.exp_counts)
        return res

    def __call__(self, x: Tensor):
        if x.shape != self.x.shape or x.dtype != self.x.dtype or x.device != self.x.device:
            raise ValueError(f"GraphedMoE: captured for {tuple(self.x.shape)} {self.x.dtype} on {self.x.device}, got "
                             f"{tuple(x.shape)} {x.dtype} on {x.device}")
        self.x.copy_(x)
        self.graph.replay()
        return self.out
