set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_shard_replay.py tests/test_time_shard.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/shard_test.log 2>&1 || { tail -40 gpurun_out/shard_test.log; exit 1; }
tail -3 gpurun_out/shard_test.log
timeout -k 10 600 python -u tools/run_configs.py 4 > gpurun_out/cfg4.log 2>&1 || { tail -30 gpurun_out/cfg4.log; exit 1; }
cat gpurun_out/cfg4.log
