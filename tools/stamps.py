#!/usr/bin/env python3
"""Read the in-kernel cycle stamps of a -DGEMM_STAMPS build (diagnostic; never a timing claim)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.environ.get("STAMPS_VARIANT", "stamps")
os.environ["MOE_NO_PERSIST"] = "1"   # the stamped blocks are virtual blocks GEMM_STAMP_BASE..+7 (1024) of a one-tile-per-block launch
# usage: stamps.py [fc1|fc2] [extra gemm_bench.py arguments, e.g. --rows 1408 --cold 12]
sys.argv = [sys.argv[0], "--rounds", "1", "--iters", "1", "--shape", sys.argv[1] if len(sys.argv) > 1 else "fc2", *sys.argv[2:], VAR]
exec(open(os.path.join(ROOT, "tools", "gemm_bench.py")).read())
lib = libs[VAR]
buf = (ctypes.c_ulonglong * 512)()
lib.moe_debug_read_stamps.argtypes = [ctypes.c_void_p]
assert lib.moe_debug_read_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(8, 8, 8)
print("block wave  wait   issue  reads+mfma  loop   epilogue prologue  nk   per-step(wait issue comp)")
for b in range(8):
    for w in range(8):
        wt, iss, cp, loop, epi, nk, pro = [int(v) for v in a[b, w, :7]]
        if nk == 0: continue
        print(f"{b:3d} {w:3d} {wt:8d} {iss:7d} {cp:9d} {loop:8d} {epi:7d} {pro:7d} {nk:4d}   {wt/nk:7.0f} {iss/nk:6.0f} {cp/nk:7.0f}")
