cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -n 3 > gpurun_out/pytest_r5f.txt
bash tools/sweep_wq.sh PA_CONV1_C64_BM64=0 PA_CONV1_C64_BM64=1 PA_CONV1_C64_BM64=0 PA_CONV1_C64_BM64=1 > gpurun_out/sweep_c64.txt 2>&1
cd tune; python tools/conv1t_clocks.py 2>&1 | grep -E "64-> 64|64->128" | cut -c1-80 > ../gpurun_out/conv1t_clocks_c64.txt
