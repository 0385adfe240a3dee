#!/usr/bin/env python3
"""A/B timing of grouped-GEMM library variants in ONE process, interleaved rounds (cdna guide rule 24).
usage: python tools/gemm_bench.py [--shape fc1|fc2|both] [--rows 45056] variant1 variant2 ...
(variant = name of tools/variants/libmoe_<name>.so; 'prod' = the in-tree library)"""
import argparse, ctypes, os, statistics, sys
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

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--rows", type=int, default=45056)
ap.add_argument("--M", type=int, default=2048)
ap.add_argument("--I", type=int, default=5632)
ap.add_argument("--E", type=int, default=4)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--shape", default="both")
ap.add_argument("--skew", type=int, default=1)
ap.add_argument("--sorted", type=int, default=1, help="gather rows in token order (as the routing kernel produces them)")
ap.add_argument("--cold", type=int, default=1, help="weight copies cycled per call (cold weights for decode-size runs)")
ap.add_argument("--gelu", type=int, default=0, help="fc1 = Phi form: W1 [E, I, M] + bias, gelu_new epilogue (default: gated SwiGLU, W1 [E, 2I, M])")
ap.add_argument("--splitk", type=int, default=1, help="1: pass a split-K workspace (product behaviour), 0: none")
ap.add_argument("--pad", type=int, default=0, help="pad leading dims of x / h (and the w buffers, for -DGEMM_EXP_LDW_PAD builds)")
ap.add_argument("--active", type=int, default=0, help="decode-size runs: spread the rows evenly over the first N experts only")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
M, I, E, R = a.M, a.I, a.E, a.rows
S = R // 2
P = a.pad
x = torch.randn(S, M + P, device=dev).to(torch.bfloat16)
N1 = I if a.gelu else 2 * I
w1s = [(torch.randn(E, N1 + 8, M + P, device=dev) * 0.02).to(torch.bfloat16) for _ in range(a.cold)]
b1 = (torch.randn(E, N1 + 8, device=dev) * 0.02).to(torch.bfloat16)
w2s = [(torch.randn(E, M + 8, I + P, device=dev) * 0.02).to(torch.bfloat16) for _ in range(a.cold)]
calls = [0, 0]
if a.skew:
    frac = torch.tensor([0.3026, 0.3103, 0.189, 0.198][:E]) if E == 4 else torch.ones(E) / E
else:
    frac = torch.ones(E) / E
cnt = (frac / frac.sum() * R).long(); cnt[-1] = R - cnt[:-1].sum()
if a.active:
    cnt = torch.zeros(E, dtype=torch.long); cnt[:a.active] = R // a.active; cnt[a.active - 1] += R - cnt.sum()
offs = torch.zeros(E + 1, dtype=torch.int32); offs[1:] = torch.cumsum(cnt, 0).int(); offs = offs.to(dev)
perm = (torch.arange(R, device=dev) if a.sorted else torch.randperm(R, device=dev)).int()
h = torch.empty(R, I + P, dtype=torch.bfloat16, device=dev)
y = torch.empty(R, M, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
SKB = 256 * 256 * 256 * 4 if a.splitk else 0
skws = torch.empty(max(SKB, 16), dtype=torch.uint8, device=dev)
SKP = skws.data_ptr() if a.splitk else None
libs = {v: load(v) for v in a.variants}

def fc1(lib):
    w1 = w1s[calls[0] % a.cold]; calls[0] += 1
    if a.gelu:
        rc = lib.moe_grouped_gemm_fwd(2, x.data_ptr(), M + P, perm.data_ptr(), 1, w1.data_ptr(), b1.data_ptr(), offs.data_ptr(), E, E, R, M, I, -1, h.data_ptr(), I + P, SKP, SKB, 0, st)
    else:
        rc = lib.moe_grouped_gemm_fwd(1, x.data_ptr(), M + P, perm.data_ptr(), 1, w1.data_ptr(), None, offs.data_ptr(), E, E, R, M, I, -1, h.data_ptr(), I + P, SKP, SKB, 0, st)
    assert rc == 0, lib.moe_last_error()
def fc2(lib):
    w2 = w2s[calls[1] % a.cold]; calls[1] += 1
    rc = lib.moe_grouped_gemm_fwd(0, h.data_ptr(), I + P, None, 0, w2.data_ptr(), None, offs.data_ptr(), E, E, R, I, M, -1, y.data_ptr(), M, SKP, SKB, 0, st)
    assert rc == 0, lib.moe_last_error()

shapes = {"fc1": (fc1, 2.0 * R * N1 * M), "fc2": (fc2, 2.0 * R * M * I)}
todo = ["fc1", "fc2"] if a.shape == "both" else [a.shape]
ref = None
for v, lib in libs.items():
    calls[0] = calls[1] = 0
    h.zero_(); y.zero_()
    fc1(lib); fc2(lib)
    torch.cuda.synchronize()
    cur = (h.clone(), y.clone())
    if ref is None:
        ref = cur
    else:
        print(f"{v}: outputs equal to {a.variants[0]}: h {bool(torch.equal(cur[0], ref[0]))} y {bool(torch.equal(cur[1], ref[1]))}", flush=True)
del ref, cur
res = {(v, s): [] for v in libs for s in todo}
for r in range(a.rounds):
    for v, lib in libs.items():
        for s in todo:
            fn, _ = shapes[s]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): fn(lib)
            e1.record(); torch.cuda.synchronize()
            res[(v, s)].append(e0.elapsed_time(e1) / a.iters)
for (v, s), ts in res.items():
    fl = shapes[s][1]
    med, mn = statistics.median(ts), min(ts)
    print(f"{v:>16s} {s}: median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us   {fl/med/1e9:7.1f} TF (median) {fl/mn/1e9:7.1f} TF (best)", flush=True)
