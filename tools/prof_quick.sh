# usage: tools/prof_quick.sh TAG bench-args...   -> kernel stats (top kernels) of one bench.py run
TAG=$1; shift; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pq_$TAG -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-train --no-other-configs "$@" > /dev/null 2>&1 || exit 1
f=$(ls $R/gpurun_out/pq_$TAG/*/*kernel_stats.csv $R/gpurun_out/pq_$TAG/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$f" "$TAG" <<'PY'
import csv, sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if "moe::" in r["Name"]:
        print(f'{r["Name"][:84]:84s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
