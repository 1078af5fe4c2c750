cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_conv.py tests/test_gpu_agent.py tests/test_gpu_dist.py -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5r.txt
bash tools/sweep_wq.sh PA_WREDUCE_LAG=1 PA_WREDUCE_LAG=2 PA_WREDUCE_LAG=3 PA_WREDUCE_LAG=4 PA_WREDUCE_LAG=1 PA_WREDUCE_LAG=2 PA_WREDUCE_LAG=3 > gpurun_out/sweep_lag.txt 2>&1
