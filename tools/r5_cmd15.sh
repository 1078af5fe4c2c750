cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_conv.py tests/test_gpu_agent.py -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5n.txt
g() { echo "PA_WG_GROUP_WGS9=$1 PA_WG_GROUP_WGS1=$2 $3"; }
bash tools/sweep_wq.sh "$(g 96 128)" "$(g 64 128)" "$(g 128 128)" "$(g 96 96)" "$(g 96 192)" "$(g 128 192)" "$(g 64 96)" "$(g 96 128 PA_WG_GROUP_MINPER9=2)" "$(g 96 128 PA_WG_GROUP_MINPER1=2)" > gpurun_out/sweep_wq10.txt 2>&1
bash tools/r5_cmd7.sh
