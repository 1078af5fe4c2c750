cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5_final.txt
bash tools/prof_round.sh round5 > gpurun_out/round5_prof_round.log 2>&1
python bench.py --stacks 8 --res 384 --bs 16 --dtype fp16 --no-cpu-baseline --no-traffic > gpurun_out/round5_c5_8stack_384_bs16_fp16.json 2>/dev/null
python tools/bench_joint.py > gpurun_out/round5_joint_loop.txt 2>&1
bash tools/ab_r5.sh 2 > gpurun_out/ab_r5c.txt 2>&1
bash tools/trace_step.sh r5final
