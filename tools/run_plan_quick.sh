set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_plan.py tests/test_gpu_l2_parity.py tests/test_gpu_device_input.py -x -q --timeout 300 --timeout-method thread > gpurun_out/plan_test.log 2>&1 || { tail -40 gpurun_out/plan_test.log; exit 1; }
tail -3 gpurun_out/plan_test.log
LORA_CORE_TIMING=1 timeout -k 10 300 python -u tools/pipeline_bench.py > gpurun_out/pipeline_plan.log 2>&1 && tail -7 gpurun_out/pipeline_plan.log
