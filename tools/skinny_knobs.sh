#!/bin/bash
# (column groups per block, K split) sweep of the decode GEMV on Phi shapes at S = 32 and S = 1
set -e
for R in "64 4" "2 2"; do set -- $R
for NW in 1 2; do for KS in 2 4 8; do
  echo "== rows $1 active $2 NW=$NW KS=$KS"
  MOE_SKINNY_NW=$NW MOE_SKINNY_KS=$KS timeout -k 10 200 python3 tools/gemm_bench.py prod --rows $1 --active $2 --cold 6 --iters 24 --rounds 5 --M 2560 --I 10240 --gelu 1 2>&1 | grep " fc"
done; done; done
