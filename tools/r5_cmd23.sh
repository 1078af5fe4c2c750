cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_agent.py tests/test_gpu_dist.py -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5s.txt
bash tools/sweep_wq.sh PA_ADAPTER_PAR=0 PA_ADAPTER_PAR=1 PA_ADAPTER_PAR=0 PA_ADAPTER_PAR=1 PA_ADAPTER_PAR=0 PA_ADAPTER_PAR=1 > gpurun_out/sweep_adpar.txt 2>&1
