set -e
timeout -k 10 600 python3 bench.py > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err
bash tools/profile_bench.sh r03b_decode1 --decode 1
bash tools/profile_bench.sh r03b_decode32 --decode 32
python3 - <<'PY'
import csv, glob
for t in ("decode1", "decode32"):
    print("==", t)
    for r in csv.DictReader(open(f"gpurun_out/r03b_{t}_kernel_stats.csv")):
        if float(r["Percentage"]) > 1.0:
            print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
