# other tiles for config 4 shapes (-> profiles/r03/tile_try_r03*.txt)
for t in "" "256,128" "128,256" "128,128"; do
  echo "== MOE_GEMM_TILE=$t (config 4 shapes: rows 4864, I 5504)"
  MOE_GEMM_TILE=$t timeout -k 10 200 python3 tools/gemm_bench.py prod --rows 4864 --I 5504 --cold 5 --shape fc2 2>&1 | grep -v amdgpu.ids
done
for t in "" "256,128" "128,256"; do
  echo "== MOE_GEMM_TILE=$t (rows 9728, I 5632: S = 4864)"
  MOE_GEMM_TILE=$t timeout -k 10 200 python3 tools/gemm_bench.py prod --rows 9728 --cold 5 --shape fc2 2>&1 | grep -v amdgpu.ids
done
