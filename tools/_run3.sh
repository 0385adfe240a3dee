set -e
for sh in fc1 fc2; do timeout -k 10 200 python3 tools/inkernel_clock.py $sh 2>&1 | grep -v amdgpu.ids | tail -1; done | tee gpurun_out/inkernel_clock_r03.txt
timeout -k 10 200 python3 tools/stamps.py fc1 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/stamps_fc1_r03.txt
timeout -k 10 300 python3 tools/ep_overhead.py 2>&1 | grep -E "path" | tee gpurun_out/ep_overhead_r03a.txt
timeout -k 10 600 python -u -m pytest tests/test_expert_parallel.py tests/test_zz_no_alloc.py -m gpu -q --timeout 300 --timeout-method thread 2>&1 | tail -2
