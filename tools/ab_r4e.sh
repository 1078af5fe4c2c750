cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PA_FIN_MASK=7 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r4e_trace -o bench -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-traffic --no-floor --no-roofline > gpurun_out/r4e_trace.log 2>&1
python tools/trace_gaps.py $(find gpurun_out/r4e_trace -name "*results.db" | head -1) 8 > gpurun_out/r4e_gaps.txt 2>&1
cat gpurun_out/r4e_gaps.txt
run() { echo "== $*" >> gpurun_out/r4e_ab.txt; env "$@" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-parity --no-traffic --no-floor 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['ms_per_step'], d['ms_per_step_median'], {k:v['ms_per_step'] for k,v in d['roofline']['classes'].items()})" >> gpurun_out/r4e_ab.txt; }
run PA_FIN_MASK=7
run PA_FIN_MASK=7 PA_CONV3_BN64=1
run PA_FIN_MASK=7 PA_FIN_PROLOGUE=96
run PA_FIN_MASK=7 PA_WGRAD_MINPER=2
run PA_FIN_MASK=7
cat gpurun_out/r4e_ab.txt | cut -c1-150
rm -rf gpurun_out/r4e_trace
