#!/usr/bin/env python3
"""Phase stamps of the decode step's gate + routing kernel (-DROUTE_STAMPS build; diagnostic, never a timing claim).
usage: MOE_GFX950_LIB=tools/variants/libmoe_rstamps.so python tools/route_stamps.py [S]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moe_llava_amd
from moe_llava_amd import _lib
from tests.expert_mlps import StableLMMLP
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
layers = [moe_llava_amd.MoE(2048, StableLMMLP(2048, 5632), num_experts=4, k=2, capacity_factor=1.5, eval_capacity_factor=2.0,
                            min_capacity=0).to(torch.bfloat16).to(dev).eval() for _ in range(6)]
x = torch.randn(1, S, 2048, device=dev).to(torch.bfloat16)
lib = _lib.load()
lib.moe_debug_read_route_stamps.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 16)()
names = ["gate GEMV", "fence+barrier", "select/ballot", "barrier", "scan (waves)+barrier", "thread 0", "barrier", "finalize"]
acc = [0] * 8
N = 30
with torch.no_grad():
    for i in range(12 + N):
        layers[i % 6](x)
        torch.cuda.synchronize()
        if i >= 12:
            assert lib.moe_debug_read_route_stamps(buf) == 0
            for j in range(8):
                acc[j] += buf[j + 1] - buf[j]
print(f"S={S}: thread 0's cycles per phase (mean of {N} calls, s_memtime ticks = 100 MHz reference x clock ratio)")
for n, a in zip(names, acc):
    print(f"  {n:24s} {a / N:9.0f}")
print(f"  total                    {sum(acc) / N:9.0f}")
