#!/bin/bash
# rocprofv3 passes over one program on the GPU box (run through gpurun from the repo root): kernel trace + stats, then
# separate --pmc passes (never combined with other trace domains), then the summaries.
# usage: tools/profile_any.sh TAG script.py [arguments]      (script path relative to the repo root)
set -e
TAG=$1; shift; PROG=$1; shift; R=$PWD; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- python3 $R/$PROG "$@" > $OUT.trace.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o $TAG --output-format csv -- python3 $R/$PROG "$@" > $OUT.pmc1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_write -o $TAG --output-format csv -- python3 $R/$PROG "$@" > $OUT.pmc2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o $TAG --output-format csv -- python3 $R/$PROG "$@" > $OUT.pmc3.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_summary_$TAG.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > /dev/null
cp $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
# keep the committed CSV readable: our kernels only (torch's set-up kernels have kilobyte-long names)
python3 - $R/gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "moe::" in r["Name"] or "rccl" in r["Name"].lower() or "nccl" in r["Name"].lower()]
w = csv.DictWriter(open(sys.argv[1], "w", newline=""), fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
w.writeheader()
for r in keep:
    w.writerow(r)
PY
echo done $TAG
