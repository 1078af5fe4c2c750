cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_gpu_net.py tests/test_gpu_local.py tests/test_gpu_agent.py tests/test_gpu_dropout.py -x -q -m gpu) > gpurun_out/r4c_tests.txt 2>&1; tail -6 gpurun_out/r4c_tests.txt
for cfg in "0 0 0" "0 0 1" "0 0 2" "0 0 4" "0 0 8" "1 0 0" "1 0 1" "1 0 2" "1 0 4" "1 0 8"; do
  set -- $cfg
  PA_CONV3_TRI=$1 PA_CONV3_DBG=$3 python tools/bench_conv3_64.py 2>&1 | grep mode
done > gpurun_out/r4c_micro.txt 2>&1
cat gpurun_out/r4c_micro.txt
for cfg in "0 0 0" "0 128 0" "1 128 8" "0 128 8" "0 0 0" "0 128 0"; do
  set -- $cfg
  echo "== PA_CONV3_TRI=$1 PA_FIN_PROLOGUE=$2 PA_CONV3_DBG=$3" >> gpurun_out/r4c_ab.txt
  PA_CONV3_TRI=$1 PA_FIN_PROLOGUE=$2 PA_CONV3_DBG=$3 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-parity --no-traffic --no-floor 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['ms_per_step'], d['ms_per_step_median'], {k:v['ms_per_step'] for k,v in d['roofline']['classes'].items()})" >> gpurun_out/r4c_ab.txt
done
cat gpurun_out/r4c_ab.txt
