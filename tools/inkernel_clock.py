#!/usr/bin/env python3
"""In-kernel clock of the grouped GEMM K loop (MI355X_MICROARCH.md, DVFS give-back item 6), -DGEMM_STAMPS build:
>= 2 s of back-to-back launches on random data, then d(s_memtime) / d(s_memrealtime) x 100 MHz across the K loop of the
stamped blocks (median).  Diagnostic: the stamps cost ~10 % of the loop, the number is a clock, not a timing claim.
usage: python tools/inkernel_clock.py [fc1|fc2] [gemm_bench.py arguments]"""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.environ.get("STAMPS_VARIANT", "stamps")
os.environ["MOE_NO_PERSIST"] = "1"
shape = sys.argv[1] if len(sys.argv) > 1 else "fc1"
sys.argv = [sys.argv[0], "--rounds", "1", "--iters", "1", "--shape", shape, *sys.argv[2:], VAR]
exec(open(os.path.join(ROOT, "tools", "gemm_bench.py")).read())
lib = libs[VAR]
fn = shapes[shape][0]
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 2.5:
    for _ in range(50):
        fn(lib)
    torch.cuda.synchronize()
    n += 50
buf = (ctypes.c_ulonglong * 16)()
lib.moe_debug_read_clock.argtypes = [ctypes.c_void_p]
assert lib.moe_debug_read_clock(buf) == 0
clk = [buf[2 * b] / buf[2 * b + 1] * 0.1 for b in range(8) if buf[2 * b + 1]]
print(f"{shape}: {n} launches in {time.perf_counter() - t0:.1f} s; in-kernel clock of the K loop over {len(clk)} stamped blocks: "
      f"median {statistics.median(clk):.3f} GHz (min {min(clk):.3f}, max {max(clk):.3f})")
