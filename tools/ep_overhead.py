"""Cost of the expert-parallel code path without any wire: one rank, real RCCL (single-rank group), config 2.
Compares the local path, the exchange path with one exchange each way, and the chunked pipeline -- the difference
is what gathers, host-side split arithmetic and the smaller grouped-FFN launches cost (the xGMI time itself can only
be measured on a multi-GPU node).  usage: python tools/ep_overhead.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moe_llava_amd  # noqa: E402
from tests.expert_mlps import StableLMMLP  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
M, I, E, S = 2048, 5632, 4, 22528
moe = moe_llava_amd.MoE(M, StableLMMLP(M, I).to(torch.bfloat16), num_experts=E, k=2, capacity_factor=1.5, min_capacity=0)
for p in moe.deepspeed_moe.experts.parameters():
    p.data.normal_(0, 0.02)
moe = moe.to(dev).train()
x = torch.randn(32, 704, M, device=dev).to(torch.bfloat16)


def timeit(label):
    with torch.no_grad():
        for _ in range(5):
            moe(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            moe(x)
        torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 20
    # host time of one call: how long the Python call takes to RETURN with an idle GPU queue (launches are asynchronous;
    # the expert-parallel path additionally waits for the kept-count matrix, its one host sync)
    host = []
    with torch.no_grad():
        for _ in range(10):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            moe(x)
            host.append(time.perf_counter() - t1)
            torch.cuda.synchronize()
    print(f"{label:40s} {step * 1e3:7.3f} ms/step   host time per call {sorted(host)[len(host) // 2] * 1e6:7.1f} us", flush=True)


timeit("local path (ep_size 1, no exchange)")
moe.deepspeed_moe.force_exchange = True
moe.deepspeed_moe._set_ep_group(dist.group.WORLD)
for c in ("1", "2", "4", "8"):
    os.environ["MOE_EP_CHUNKS"] = c
    timeit(f"exchange path, {c} chunk(s), 1-rank RCCL")
if os.environ.get("EP_PROFILE"):
    import cProfile, pstats
    os.environ["MOE_EP_CHUNKS"] = "1"
    pr = cProfile.Profile()
    with torch.no_grad():
        for _ in range(5):
            moe(x)
        torch.cuda.synchronize()
        pr.enable()
        for _ in range(200):
            moe(x)
        pr.disable()
        torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
dist.destroy_process_group()
