#!/bin/bash
# Row-stride experiment (round 3): do 4-KiB-multiple row strides of the GEMM operands (K = 2048: 4096 B; K = 10240: 20480 B)
# cost tile-fill rate through cache-channel conflicts?  Activation strides padded by --pad, weight strides by the ldwN builds.
set -e
R=$PWD
echo "== config 2 shapes (M 2048, I 5632, 45056 rows), pad 0"
timeout -k 10 200 python3 tools/gemm_bench.py prod --pad 0
for P in 32 64 128; do
  echo "== config 2 shapes, activations + weights padded by $P elements"
  timeout -k 10 200 python3 tools/gemm_bench.py prod ldw$P --pad $P
done
echo "== Phi shapes (M 2560, I 10240, 27424 rows, gelu), pad 0"
timeout -k 10 200 python3 tools/gemm_bench.py prod --M 2560 --I 10240 --rows 27424 --gelu 1 --pad 0
echo "== Phi shapes, pad 64"
timeout -k 10 200 python3 tools/gemm_bench.py prod ldw64 --M 2560 --I 10240 --rows 27424 --gelu 1 --pad 64
