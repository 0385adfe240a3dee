#!/bin/bash
# rocprofv3 kernel stats of the training step with ablated builds of the weight-gradient kernel (loads only / MFMA only).
# Both ablations keep every bounds check; outputs are garbage by design, nothing is compared.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in NO_LOAD NO_MFMA; do
  export MOE_GFX950_LIB=$GRAFT_REPO_ROOT/tools/variants/libmoe_wg_$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wg_$v -o t -- python tools/train_bench.py --steps 3 > gpurun_out/wg_$v.log 2>&1 || exit 1
done
