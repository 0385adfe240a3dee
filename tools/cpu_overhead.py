import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.getcwd())
import moe_llava_amd
from tests.expert_mlps import StableLMMLP
dev = torch.device("cuda", 0)
M, I, E = 2048, 5632, 4
moe = moe_llava_amd.MoE(M, StableLMMLP(M, I).to(torch.bfloat16), num_experts=E, k=2, capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0).to(dev).eval()
x = torch.randn(1, 32, M, device=dev).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(20): moe(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): moe(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"cpu issue {1e6*(t1-t0)/500:.1f} us/call, total {1e6*(t2-t0)/500:.1f} us/call")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): moe(x)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
