"""A/B of grouped-GEMM tile choices inside one process (env knobs are read per call).
usage: python tools/tile_sweep.py [config ...]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moe_llava_amd  # noqa: E402
from moe_llava_amd import ops  # noqa: E402
from tests.expert_mlps import StableLMMLP  # noqa: E402

SHAPES = {"decode1": 1, "decode8": 8, "decode32": 32, "S128": 128, "S256": 256, "cfg1": 704, "S1408": 1408,
          "S2432": 2432, "S5000": 5000, "cfg2": 22528}
DEFAULT = ["decode1", "decode32", "S128", "S256", "cfg1", "S1408", "S2432"]
VARIANTS = [("auto", None)] + [(f"{bm},{bnw}", f"{bm},{bnw}") for bm in (32, 64, 128, 256) for bnw in (256, 128, 64)]
if os.environ.get("SWEEP_ORDER"):   # A/B of the tile order (MOE_GEMM_ORDER: 0 = super-rows, 1 = group-major) on a few tiles
    VARIANTS = [(f"{t or 'auto'} order={o}", t, o) for t in (None, "128,256", "128,128", "256,256", "64,256", "256,128")
                for o in ("0", "1")]
elif os.environ.get("SWEEP_AUTO"):    # only the library's own choice (A/B of library builds through MOE_GFX950_LIB)
    VARIANTS = [("auto", None, None)]
else:
    VARIANTS = [(a, b, None) for a, b in VARIANTS]
N_LAYERS = int(os.environ.get("SWEEP_LAYERS", "5"))


def main():
    names = sys.argv[1:] or DEFAULT
    dev = torch.device("cuda", 0)
    M, I, E = 2048, 5632, 4
    # several layers cycled, as in the model (12 MoE layers): one layer's 277 MB of weights would otherwise sit in
    # the 256 MiB Infinity Cache between calls and hide the HBM stream
    layers = []
    timer = ops.PhaseTimer()
    for _ in range(N_LAYERS):
        moe = moe_llava_amd.MoE(M, StableLMMLP(M, I).to(torch.bfloat16), num_experts=E, k=2, capacity_factor=1.5,
                                eval_capacity_factor=2.0, min_capacity=0, top2_2nd_expert_sampling=False)
        for p in moe.deepspeed_moe.experts.parameters():
            p.data.normal_(0, 0.02)
        moe = moe.to(dev).train()
        moe.deepspeed_moe.timer = timer
        layers.append(moe)
    for name in names:
        S = SHAPES[name]
        x = torch.randn(1, S, M, device=dev)
        x[:, : (S * 4) // 5] += 0.5 * torch.randn(M, device=dev).sign()     # "image" rows skew the router (uneven experts)
        x = x.to(torch.bfloat16)
        res = {}
        for rep in range(2):
            for label, tile, order in VARIANTS:
                for key, val in (("MOE_GEMM_TILE", tile), ("MOE_GEMM_ORDER", order)):
                    if val is None:
                        os.environ.pop(key, None)
                    else:
                        os.environ[key] = val
                ops.reload_knobs()                      # the library reads its knobs once
                timer.reset()
                with torch.no_grad():
                    for i in range(5 + 2 * len(layers)):
                        layers[i % len(layers)](x)
                f1 = timer.durations("fc1_begin", "fc1_end")[5:]
                f2 = timer.durations("fc1_end", "fc2_end")[5:]
                res.setdefault(label, []).append((statistics.median(f1) * 1e3, statistics.median(f2) * 1e3))
        print(f"== {name} (S={S}) exp_counts={layers[0].deepspeed_moe.exp_counts.tolist()}")
        for label, v in res.items():
            print(f"   {label:14s} fc1 {min(a for a, _ in v):8.1f} us   fc2 {min(b for _, b in v):8.1f} us")


if __name__ == "__main__":
    main()
