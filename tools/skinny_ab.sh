#!/bin/bash
# Decode weight-streaming kernel A/B (round 3): variants given as arguments vs the in-tree library, S = 1 (two active
# experts, one row each) and S = 32 (64 rows over four experts), 12 weight copies cycled (cold weights).
set -e
V="$@"
echo "== S = 1 (2 rows, 2 active experts)"
timeout -k 10 200 python3 tools/gemm_bench.py prod $V --rows 2 --active 2 --cold 12 --iters 24 --rounds 9
echo "== S = 32 (64 rows, 4 experts)"
timeout -k 10 200 python3 tools/gemm_bench.py prod $V --rows 64 --active 4 --cold 12 --iters 24 --rounds 9
echo "== S = 32, Phi shapes (gelu_new + bias)"
timeout -k 10 200 python3 tools/gemm_bench.py prod $V --rows 64 --active 4 --cold 6 --iters 24 --rounds 9 --M 2560 --I 10240 --gelu 1
