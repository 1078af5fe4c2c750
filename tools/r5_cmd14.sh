cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5m.txt
bash tools/sweep_wq.sh X=1 X=2 > gpurun_out/sweep_wq9.txt 2>&1
bash tools/r5_cmd7.sh
