# full GPU test suite in one process -> gpurun_out/r03_gputest_a.log (run on the GPU box: gpurun -- bash tools/gpu_suite.sh)
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -x 2>&1 | tail -30 > gpurun_out/r03_gputest_a.log
echo "rc=$?" >> gpurun_out/r03_gputest_a.log
