"""Host-side cost of one decode-size MoE call (S = 1): cProfile over the module call, GPU work left asynchronous.
usage: python tools/decode_host_profile.py [S]"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moe_llava_amd  # noqa: E402
from tests.expert_mlps import StableLMMLP  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
M, I = 2048, 5632
moe = moe_llava_amd.MoE(M, StableLMMLP(M, I), num_experts=4, k=2, capacity_factor=1.5, eval_capacity_factor=2.0,
                        min_capacity=0).to(torch.bfloat16).to(dev).eval()
x = torch.randn(1, S, M, device=dev).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(50):
        moe(x)
    torch.cuda.synchronize()
    N = 3000
    t0 = time.perf_counter()
    for _ in range(N):
        moe(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"S={S}: host issue time {1e6 * (t1 - t0) / N:.1f} us per call, wall {1e6 * (t2 - t0) / N:.1f} us per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        moe(x)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(18)
