set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 600 python -u tools/run_configs.py 4 > gpurun_out/cfg4.log 2>&1 || { tail -30 gpurun_out/cfg4.log; exit 1; }
cat gpurun_out/cfg4.log
