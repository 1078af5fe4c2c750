cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -n 3 > gpurun_out/pytest_r5c.txt
bash tools/sweep_wq.sh PA_WG_GROUP=0 PA_STEM_PIPE=1 "PA_STEM_PIPE=1 PA_STEM_SPLITS=256" PA_WG_GROUP_PIPE=0 PA_WG_GROUP_MINPER9=1 PA_WG_GROUP_MINPER9=8 "PA_WG_GROUP_S9=64 PA_WG_GROUP_S1=128" "PA_WG_GROUP_S9=16 PA_WG_GROUP_S1=48" > gpurun_out/sweep_wq5.txt 2>&1
cd tune; python tools/conv1t_clocks.py 2>&1 | grep -E "64-> 64|64->128" > ../gpurun_out/conv1t_clocks_128.txt
