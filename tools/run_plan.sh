set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
LORA_CORE_TIMING=1 timeout -k 10 300 python -u tools/pipeline_bench.py > gpurun_out/pipeline_plan.log 2>&1 && tail -7 gpurun_out/pipeline_plan.log
