#!/bin/bash
# Every workload of the bench line under rocprofv3 (kernel stats + PMC summaries), tags <PFX>_<workload>:
# usage: tools/profile_all.sh r03h [workloads...]     (default: all)
set -e
PFX=$1; shift
W=${@:-"cfg2 cfg1 cfg3 cfg4 cfg5 decode1 decode32 train"}
X="--no-cpu-baseline --no-train --no-other-configs"
for w in $W; do
  case $w in
    cfg2) bash tools/profile_any.sh ${PFX}_cfg2 bench.py --steps 12 --warmup 3 $X ;;
    cfg1) bash tools/profile_any.sh ${PFX}_cfg1 bench.py --config 1 --steps 36 --warmup 12 $X ;;
    cfg3) bash tools/profile_any.sh ${PFX}_cfg3 bench.py --config 3 --steps 8 --warmup 3 $X ;;
    cfg4) bash tools/profile_any.sh ${PFX}_cfg4 bench.py --config 4 --steps 24 --warmup 6 $X ;;
    cfg5) bash tools/profile_any.sh ${PFX}_cfg5 bench.py --config 5 --steps 4 --warmup 2 $X ;;
    decode1) bash tools/profile_any.sh ${PFX}_decode1 bench.py --decode 1 --steps 120 ;;
    decode32) bash tools/profile_any.sh ${PFX}_decode32 bench.py --decode 32 --steps 120 ;;
    train) bash tools/profile_any.sh ${PFX}_train tools/train_bench.py --steps 3 --warmup 1 --skew 1 ;;
  esac
done
