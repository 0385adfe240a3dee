# split-K tail on / off under rocprofv3 for Phi fc1, config 1 fc1, config 4 (-> profiles/r03/splitk_*.txt)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
for cfg in "phi --M 2560 --I 10240 --rows 27424 --gelu 1 --shape fc1" "c1 --rows 1408 --cold 12 --shape fc1" "c4 --rows 4864 --I 5504 --cold 5 --shape both"; do
  set -- $cfg; tag=$1; shift
  for sk in 0 1; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sk_${tag}_$sk -o t --output-format csv -- python3 $R/tools/gemm_bench.py prod --splitk $sk --rounds 3 "$@" > /dev/null 2>&1 || exit 1
    echo "== $tag splitk=$sk"
    f=$(ls $R/gpurun_out/prof_sk_${tag}_$sk/*/*kernel_stats.csv $R/gpurun_out/prof_sk_${tag}_$sk/*kernel_stats.csv 2>/dev/null | head -1)
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "gemm_kernel" in n or "fixup" in n:
        print(f'{n[:80]:80s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY
  done
done
