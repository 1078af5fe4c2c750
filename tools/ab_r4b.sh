cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py tests/test_gpu_local.py -x -q -m gpu) > gpurun_out/r4b_tests.txt 2>&1; tail -15 gpurun_out/r4b_tests.txt
for cfg in "0 0" "1 0" "0 128" "1 128" "0 0" "1 128"; do
  set -- $cfg
  echo "== PA_CONV3_TRI=$1 PA_FIN_PROLOGUE=$2" >> gpurun_out/r4b_ab.txt
  PA_CONV3_TRI=$1 PA_FIN_PROLOGUE=$2 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-parity --no-traffic --no-floor 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['ms_per_step'], d['ms_per_step_median'], {k:v['ms_per_step'] for k,v in d['roofline']['classes'].items()})" >> gpurun_out/r4b_ab.txt
done
cat gpurun_out/r4b_ab.txt
for t in 0 1; do echo "== conv3 microbench PA_CONV3_TRI=$t"; PA_CONV3_TRI=$t python tools/bench_conv3.py 2>&1 | tail -12; PA_CONV3_TRI=$t python tools/bench_cold.py 2>&1 | tail -12; done > gpurun_out/r4b_micro.txt 2>&1
cat gpurun_out/r4b_micro.txt
