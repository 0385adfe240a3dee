#!/usr/bin/env python3
"""In-kernel cycle stamps of the weight-gradient kernel (a -DWG_STAMPS build under tools/variants; diagnostic only).
usage: python tools/wstamps.py [variant]   -- one training step at config 2, then the stamps of the LAST wgrad launch (dW1)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = sys.argv[1] if len(sys.argv) > 1 else "wstamps"
os.environ["MOE_GFX950_LIB"] = os.path.join(ROOT, "tools", "variants", f"libmoe_{VAR}.so")
import torch  # noqa: E402
import moe_llava_amd  # noqa: E402
from moe_llava_amd import _lib  # noqa: E402
from tests.expert_mlps import StableLMMLP  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(1)
M, I = 2048, 5632
moe = moe_llava_amd.MoE(M, StableLMMLP(M, I), num_experts=4, k=2, capacity_factor=1.5, min_capacity=0).to(torch.bfloat16).to(dev)
for p in moe.deepspeed_moe.experts.parameters():
    p.data.normal_(0, 0.02)
moe.train()
x = torch.randn(32, 704, M, device=dev).to(torch.bfloat16)
R = torch.randn_like(x)
for _ in range(2):
    moe.zero_grad(set_to_none=True)
    xg = x.detach().requires_grad_(True)
    out, l_aux, _ = moe(xg)
    torch.autograd.backward([out, l_aux], [R, torch.tensor(0.01, device=dev)])
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["MOE_GFX950_LIB"])
buf = (ctypes.c_ulonglong * 512)()
lib.moe_debug_read_wstamps.argtypes = [ctypes.c_void_p]
assert lib.moe_debug_read_wstamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(8, 8, 8)
print("block wave   wait    issue  reads+mfma  steps  kernel   per-step(wait issue comp)")
for b in range(8):
    for w in range(8):
        wt, iss, cp, n, tot = [int(v) for v in a[b, w, :5]]
        if n == 0:
            continue
        print(f"{b:3d} {w:3d} {wt:9d} {iss:8d} {cp:9d} {n:6d} {tot:9d}   {wt/n:7.0f} {iss/n:6.0f} {cp/n:7.0f}")
