cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -n 3 > gpurun_out/pytest_r5d.txt
bash tools/sweep_wq.sh PA_CONV1T_DBG=128 PA_WG_GROUP_PIPE=0 "PA_WG_GROUP_PIPE=0 PA_WGRAD_NOPIPE=1" PA_WGRAD_NOPIPE=1 PA_CONV1T_DBG=128 PA_WG_GROUP_PIPE=0 "PA_WG_GROUP_PIPE=0 PA_WGRAD_NOPIPE=1" PA_WGRAD_NOPIPE=1 > gpurun_out/sweep_wq6.txt 2>&1
cd tune; python tools/conv1t_clocks.py 2>&1 | grep -E "dgrad  64" > ../gpurun_out/conv1t_clocks_128_lds.txt;  PA_EPI_DIRECT=1 python tools/conv1t_clocks.py 2>&1 | grep -E "dgrad  64" > ../gpurun_out/conv1t_clocks_128_direct.txt
