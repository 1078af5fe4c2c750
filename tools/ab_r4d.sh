cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "finalize or residual or c2_size or hourglass_forward") > gpurun_out/r4d_tests.txt 2>&1; tail -4 gpurun_out/r4d_tests.txt
run() { echo "== $*" >> gpurun_out/r4d_ab.txt; env "$@" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-parity --no-traffic --no-floor 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['ms_per_step'], d['ms_per_step_median'], {k:v['ms_per_step'] for k,v in d['roofline']['classes'].items()})" >> gpurun_out/r4d_ab.txt; }
for rep in 1 2; do
run PA_FIN_PROLOGUE=0
run PA_FIN_MASK=1
run PA_FIN_MASK=3
run PA_FIN_MASK=7
run PA_FIN_MASK=11
run PA_FIN_MASK=15
done
run PA_FIN_MASK=3 PA_WGRAD_MINPER=2
run PA_FIN_MASK=3 PA_WGRAD_MINPER=4
cat gpurun_out/r4d_ab.txt
