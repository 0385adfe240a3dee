#!/usr/bin/env python3
"""Per-phase timing of the MoE block at the BASELINE.json configs that fit one GPU (single-GPU shapes).
Not the contract bench (that is bench.py); this is the tuning aid: HIP-event phase breakdown per config."""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moe_llava_amd  # noqa: E402
from moe_llava_amd import ops  # noqa: E402
from tests.expert_mlps import PhiMLP, QWenMLP, StableLMMLP  # noqa: E402

CONFIGS = {
    # name: (expert ctor, M, I, E, cf, S, training)
    "cfg1_stablelm_S704": (StableLMMLP, 2048, 5632, 4, 1.5, 704, True),
    "cfg2_stablelm_S22528": (StableLMMLP, 2048, 5632, 4, 1.5, 22528, True),
    "cfg3_phi_S13712": (PhiMLP, 2560, 10240, 4, 1.5, 13712, True),
    "cfg4_qwen_S2432": (QWenMLP, 2048, 5504, 4, 1.5, 2432, True),
    "cfg5_phi8e_S16384": (PhiMLP, 2560, 10240, 8, 1.25, 16384, True),
    "decode_stablelm_S32": (StableLMMLP, 2048, 5632, 4, 2.0, 32, False),
    "decode_stablelm_S1": (StableLMMLP, 2048, 5632, 4, 2.0, 1, False),
}


def run(name, iters, warmup, n_layers, graph=False):
    ctor, M, I, E, cf, S, training = CONFIGS[name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    moe = moe_llava_amd.MoE(M, ctor(M, I).to(torch.bfloat16), num_experts=E, k=2, capacity_factor=cf,
                            eval_capacity_factor=2.0, min_capacity=0)
    for p in moe.deepspeed_moe.experts.parameters():
        p.data.normal_(0, 0.02)
    moe.deepspeed_moe.gate.wg.weight.data.normal_(0, 0.02)
    moe = moe.to(dev).train(training)
    # the model has 12 MoE layers: cycling several copies keeps one layer's weights from sitting in the 256 MiB
    # Infinity Cache between calls (which would flatter the weight-streaming configs)
    import copy
    layers = [moe] + [copy.deepcopy(moe) for _ in range(n_layers - 1)]
    x = torch.randn(1, S, M, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        for i in range(max(warmup, n_layers)):
            layers[i % n_layers](x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            layers[i % n_layers](x)
        e1.record()
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1) / iters
        if graph:
            from moe_llava_amd.graph import GraphedMoE
            import time
            gs = [GraphedMoE(m, x) for m in layers]
            for i in range(n_layers):
                gs[i](x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(iters * 4):
                gs[i % n_layers](x)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - t0) / (iters * 4)
            t0 = time.perf_counter()
            for i in range(iters * 4):
                layers[i % n_layers](x)
            torch.cuda.synchronize()
            te = (time.perf_counter() - t0) / (iters * 4)
            print(f"{name:24s} wall clock per step: eager {te*1e6:7.1f} us   HIP-graph replay {tg*1e6:7.1f} us", flush=True)
        t = ops.PhaseTimer()
        for m in layers:
            m.deepspeed_moe.timer = t
        for i in range(iters):
            layers[i % n_layers](x)
        for m in layers:
            m.deepspeed_moe.timer = None
        ph = {k: statistics.mean(t.durations(a, b)) for k, (a, b) in
              {"gate+route": ("gate_begin", "gate_end"), "fc1": ("fc1_begin", "fc1_end"),
               "fc2": ("fc1_end", "fc2_end"), "combine": ("fc2_end", "combine_end")}.items()}
    rows = int(moe.deepspeed_moe.last_routing.expert_offsets[-1])
    nmat = 3 if ctor is not PhiMLP else 2
    flops = 2.0 * rows * nmat * M * I
    wbytes = E * nmat * M * I * 2
    print(f"{name:24s} {total*1e3:9.1f} us/step  {S/total/1e3:9.2f} Mtok/s  {flops/total/1e9:7.1f} TF  "
          f"weights {wbytes/total/1e6:7.0f} GB/s | " + "  ".join(f"{k} {v*1e3:.1f}us" for k, v in ph.items()), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=list(CONFIGS))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=4, help="distinct weight sets cycled (cold weights)")
    ap.add_argument("--graph", action="store_true", help="also time HIP-graph replays of the step (decode configs)")
    a = ap.parse_args()
    for c in a.configs:
        run(c, a.iters, a.warmup, a.layers, a.graph)
