#!/usr/bin/env python3
"""A/B timing of the gate GEMV (moe_gate_logits_fwd) of library variants in one process, inputs cycled so that x and wg
come from HBM.  usage: python tools/gate_bench.py variant1 variant2 ...  ('prod' = the in-tree library)"""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moe_llava_amd import _lib

def load(name):
    path = _lib.LIB_PATH if name == "prod" else os.path.join(ROOT, "tools", "variants", f"libmoe_{name}.so")
    lib = ctypes.CDLL(path)
    for n, res, args in _lib._SIGNATURES:
        f = getattr(lib, n); f.restype = res; f.argtypes = args
    return lib

dev = torch.device("cuda", 0)
libs = {v: load(v) for v in sys.argv[1:]}
st = torch.cuda.current_stream().cuda_stream
for S, M, E in ((704, 2048, 4), (2432, 2048, 4), (22528, 2048, 4), (13712, 2560, 4), (131072, 2560, 8)):
    ncopy = max(2, min(12, (600 << 20) // (S * M * 2)))
    xs = [torch.randn(S, M, device=dev).to(torch.bfloat16) for _ in range(ncopy)]
    wgs = [torch.randn(E, M, device=dev) * 0.02 for _ in range(ncopy)]
    lg = torch.empty(S, E, device=dev)
    ref = None
    for v, lib in libs.items():
        ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(ncopy):
                rc = lib.moe_gate_logits_fwd(xs[i].data_ptr(), wgs[i].data_ptr(), 0, lg.data_ptr(), S, M, E, 0, 0, st)
                assert rc == 0
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / ncopy * 1e3)
        same = ""
        if ref is None: ref = lg.clone()
        else: same = f"  bit-identical to {sys.argv[1]}: {bool(torch.equal(ref, lg))}"
        print(f"S={S:6d} M={M} E={E} {v:>10s}: median {statistics.median(ts):7.1f} us  min {min(ts):7.1f} us  ({S*M*2/min(ts)/1e6:5.2f} TB/s){same}", flush=True)
