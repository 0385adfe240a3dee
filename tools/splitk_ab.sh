#!/bin/bash
# Split-K tail A/B (round 3): the same library with (--splitk 1) and without (--splitk 0) a workspace, one process each
# (the rule that picks the pieces lives in the kernel; the workspace is what switches it on).
run() { echo "== $*"; for sk in 0 1; do echo "-- splitk=$sk"; timeout -k 10 200 python3 tools/gemm_bench.py prod --splitk $sk "$@" 2>&1 | grep -v amdgpu.ids || exit 1; done; }
run --rows 1408 --cold 12                                   # config 1 (S = 704)
run --rows 4864 --I 5504 --cold 5                           # config 4 (Qwen, S = 2432)
run --M 2560 --I 10240 --rows 27424 --gelu 1                # config 3 (Phi)
run                                                         # config 2
