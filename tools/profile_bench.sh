#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats, then separate --pmc passes (never combined with other trace domains).
# usage: tools/profile_bench.sh TAG [bench.py arguments, e.g. --config 1]
set -e
TAG=$1; shift; R=$PWD; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
X="--no-cpu-baseline --no-train --no-other-configs $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 $X > $OUT.trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o $TAG --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 $X > $OUT.pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_write -o $TAG --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 $X > $OUT.pmc2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o $TAG --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 $X > $OUT.pmc3.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_summary_$TAG.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > /dev/null
cp $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
echo done
