set -e
R=$PWD; OUT=$R/gpurun_out/prof_train_pmc
cd /tmp; export TMPDIR=/tmp
B="python3 $R/tools/train_bench.py --steps 3 --warmup 1"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o t --output-format csv -- $B > $OUT.1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/w -o t --output-format csv -- $B > $OUT.2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/s -o t --output-format csv -- $B > $OUT.3.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_summary_r02b_train.json $OUT/f $OUT/w $OUT/s > /dev/null
