"""Forward+backward timing of the MoE block at a BASELINE config (default config 2), with a per-kernel breakdown
when run under rocprofv3.  Usage: python tools/train_bench.py [--batch 32] [--steps 5] [--kind stablelm|phi]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moe_llava_amd  # noqa: E402
from tests.expert_mlps import PhiMLP, StableLMMLP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=704)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kind", default="stablelm")
    ap.add_argument("--skew", type=int, default=0, help="1: bench.py's input (image rows share a direction: experts get 30/31/19/20 % of the rows)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    if a.kind == "stablelm":
        M, I, proto = 2048, 5632, StableLMMLP(2048, 5632)
    else:
        M, I, proto = 2560, 10240, PhiMLP(2560, 10240)
    moe = moe_llava_amd.MoE(M, proto, num_experts=4, k=2, capacity_factor=1.5, min_capacity=0).to(torch.bfloat16).to(dev)
    for p in moe.deepspeed_moe.experts.parameters():
        p.data.normal_(0, 0.02)
    moe.deepspeed_moe.gate.wg.weight.data = (torch.randn(4, M, device=dev) * 0.02)
    moe.train()
    S = a.batch * a.tokens
    x = torch.randn(a.batch, a.tokens, M, device=dev)
    if a.skew:
        g = torch.Generator().manual_seed(7)
        u = torch.randn(M, generator=g)
        x[:, :(a.tokens * 576) // 704] += 0.5 * (u * (M ** 0.5) / u.norm()).to(dev)
    x = x.to(torch.bfloat16)
    R = torch.randn(a.batch, a.tokens, M, device=dev).to(torch.bfloat16)

    def step():
        moe.zero_grad(set_to_none=True)
        xg = x.detach().requires_grad_(True)
        out, l_aux, _ = moe(xg)
        e_f.record()
        torch.autograd.backward([out, l_aux], [R, torch.tensor(0.01, device=dev)])

    e0, e_f, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    fw, bw = [], []
    for _ in range(a.steps):
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        fw.append(e0.elapsed_time(e_f))
        bw.append(e_f.elapsed_time(e1))
    fw.sort(); bw.sort()
    f, b = fw[len(fw) // 2], bw[len(bw) // 2]
    flops_f = 2 * S * 6 * M * I if a.kind == "stablelm" else 2 * S * 4 * M * I
    print(json.dumps({"S": S, "kind": a.kind, "fwd_ms": round(f, 3), "bwd_ms": round(b, 3),
                      "fwd_tflops": round(flops_f / f / 1e9, 1), "bwd_tflops": round(2 * flops_f / b / 1e9, 1),
                      "train_tokens_per_s": round(S / (f + b) * 1e3),
                      "exp_counts": moe.deepspeed_moe.exp_counts.tolist()}))


if __name__ == "__main__":
    main()
