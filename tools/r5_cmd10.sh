cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -n 3 > gpurun_out/pytest_r5e.txt
bash tools/sweep_wq.sh PA_CONV1_K32=0 PA_CONV1_K32=1 PA_CONV1_K32=0 PA_CONV1_K32=1 > gpurun_out/sweep_k32.txt 2>&1
cd tune; python tools/conv1t_clocks.py 2>&1 | grep -E "128->256 64x64|256->128 64x64" > ../gpurun_out/conv1t_clocks_k32.txt; PA_CONV1_K32=0 python tools/conv1t_clocks.py 2>&1 | grep -E "128->256 64x64|256->128 64x64" > ../gpurun_out/conv1t_clocks_k64.txt
