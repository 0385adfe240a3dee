#!/bin/bash
# rocprofv3 kernel stats of the training step with variant builds of the weight-gradient kernel (tools/variants/libmoe_<v>.so).
# usage: tools/run_wg_ablation.sh v1 v2 ...   ('prod' = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = prod ]; then unset MOE_GFX950_LIB; else export MOE_GFX950_LIB=$GRAFT_REPO_ROOT/tools/variants/libmoe_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wg_$v -o t -- python tools/train_bench.py --steps 3 > gpurun_out/wg_$v.log 2>&1 || exit 1
done
