#!/bin/bash
# wgrad_tn_kernel ablations (loads only / MFMA only) through the training bench under rocprofv3 --stats
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in prod "$@"; do
  if [ "$v" = prod ]; then unset MOE_GFX950_LIB; else export MOE_GFX950_LIB=$R/tools/variants/libmoe_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abl_$v -o a --output-format csv -- python3 $R/tools/train_bench.py --steps 5 > $R/gpurun_out/abl_$v.log 2>&1
  f=$(ls $R/gpurun_out/abl_$v/*kernel_stats.csv $R/gpurun_out/abl_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $v"; grep -E "wgrad_tn|grouped_gemm" $f | cut -d, -f1-4
done
