#!/bin/bash
# Does the decode GEMV's time follow the busiest CU's share of blocks?  S = 1 (2 active experts), gated fc1 with
# I chosen so that the active blocks (2 x I/32) are 256 / 352 / 384 / 512 / 768.
set -e
for I in 4096 5632 6144 8192 12288; do
  echo "== I = $I: active fc1 blocks = $((2*I/32)), fc1 weight bytes = $((2*2*I*2048*2))"
  timeout -k 10 200 python3 tools/gemm_bench.py prod --rows 2 --active 2 --cold 12 --iters 24 --rounds 7 --I $I 2>&1 | grep -v amdgpu.ids
done
echo "== Phi fc2 at S = 32 with 1 and 2 column groups per block (640 / 320 blocks)"
MOE_SKINNY_NW=1 timeout -k 10 200 python3 tools/gemm_bench.py prod --rows 64 --active 4 --cold 6 --iters 24 --rounds 7 --M 2560 --I 10240 --gelu 1 2>&1 | grep -v amdgpu.ids
MOE_SKINNY_NW=2 timeout -k 10 200 python3 tools/gemm_bench.py prod --rows 64 --active 4 --cold 6 --iters 24 --rounds 7 --M 2560 --I 10240 --gelu 1 2>&1 | grep -v amdgpu.ids
