# other tiles, with and without the split-K tail, for config 1 shapes (-> profiles/r03/tile_try2_r03i.txt)
for t in "" "256,256" "256,128" "64,256"; do
  for sk in 0 4; do
  echo "== MOE_GEMM_TILE=$t MOE_SPLITK_MIN=$sk (config 1 shapes: rows 1408, cold 12) fc1"
  MOE_SPLITK_MIN=$sk MOE_GEMM_TILE=$t timeout -k 10 200 python3 tools/gemm_bench.py prod --rows 1408 --cold 12 --shape fc1 2>&1 | grep -v amdgpu.ids
  done
done
