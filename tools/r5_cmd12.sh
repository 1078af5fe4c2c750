cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_conv.py -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5l.txt
g() { echo "PA_WG_GROUP_S9=$1 PA_WG_GROUP_S1=$2 $3"; }
bash tools/sweep_wq.sh PA_WG_GROUP=0 "$(g 32 96)" "$(g 32 64)" "$(g 48 96)" "$(g 64 128)" "$(g 24 64)" "$(g 32 96 PA_WG_GROUP_MINPER9=2)" "$(g 32 96 PA_WG_GROUP_MINPER9=8)" "$(g 32 96 PA_WG_GROUP_MINPER1=1)" "$(g 32 96 PA_WG_GROUP_MINPER1=4)" > gpurun_out/sweep_wq7.txt 2>&1
