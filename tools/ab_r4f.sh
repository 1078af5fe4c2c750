cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PA_FIN_MASK=7 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r4f_trace -o bench -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-traffic --no-floor --no-roofline > gpurun_out/r4f_trace.log 2>&1
python tools/trace_gaps.py $(find gpurun_out/r4f_trace -name "*results.db" | head -1) 8 > gpurun_out/r4f_gaps.txt 2>&1
sed -n '/step boundary/,$p' gpurun_out/r4f_gaps.txt
rm -rf gpurun_out/r4f_trace
