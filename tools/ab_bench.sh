#!/bin/bash
# Block-level A/B of library variants through bench.py (separate processes, alternating): tools/ab_bench.sh v1 v2 ...
# (variant = name of tools/variants/libmoe_<name>.so; 'prod' = the in-tree library)
R=$PWD
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = prod ]; then unset MOE_GFX950_LIB; else export MOE_GFX950_LIB=$R/tools/variants/libmoe_$v.so; fi
    timeout -k 10 200 python bench.py --steps 40 --no-cpu-baseline --no-train 2>/dev/null > /tmp/ab_$v.json || exit 1
    python - "$v" /tmp/ab_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d["roofline"]
print(f"{sys.argv[1]:>10s}  {d['ms_per_step']:.4f} ms/step  fc1 {r['avg_launch_ms']:.4f}  fc2 {r['fc2_avg_launch_ms']:.4f}  {d['value']/1e6:.3f} Mtok/s", flush=True)
PY
  done
done
