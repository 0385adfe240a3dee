# super-row height sweep (MOE_GEMM_RS) at Phi width and config 5 size (-> profiles/r03/rs_sweep_r03b.txt)
for rs in 0 2 3 4 5 6 8; do
  echo "== MOE_GEMM_RS=$rs  (phi shapes: fc1 CT 40, fc2 CT 10; rows 27424)"
  MOE_GEMM_RS=$rs timeout -k 10 200 python3 tools/gemm_bench.py prod --M 2560 --I 10240 --rows 27424 --gelu 1 --rounds 5 2>&1 | grep -v amdgpu.ids
done
for rs in 0 3 4 5; do
  echo "== MOE_GEMM_RS=$rs  (config 5 per-GPU shapes: 8 experts, 240000 rows)"
  MOE_GEMM_RS=$rs timeout -k 10 300 python3 tools/gemm_bench.py prod --M 2560 --I 10240 --E 8 --rows 240000 --gelu 1 --rounds 3 --iters 2 --skew 0 2>&1 | grep -v amdgpu.ids
done
