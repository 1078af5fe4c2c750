#!/bin/bash
# HBM traffic per kernel from PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE); run via gpurun
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PA_BENCH_CHILD=1      # bench.py: no nested rocprofv3 child passes, no median pass (fixed step counts)
for C in FETCH_SIZE WRITE_SIZE; do
  PA_BENCH_SEQ_OUT=gpurun_out/pmc_${TAG}_seq.json rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-traffic --no-floor > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-120
done
python - $TAG > gpurun_out/pmc_summary_$TAG.txt <<'PY'
import csv, sys, glob, collections
tag = sys.argv[1]
tot = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_%s_%s/*counter_collection.csv' % (tag, C))
    if not f:
        print('no counter file for', C, glob.glob('gpurun_out/pmc_%s_%s/*' % (tag, C))); continue
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != C: continue
        k = r['Kernel_Name'][:70]
        per[k][0] += float(r['Counter_Value']); per[k][1] += 1
    tot[C] = per
steps = 7.0          # 1 warm-up + 3 timed steps + the 3 steps of the roofline leg
print('# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1   (4 steps + the 3 steps of its single-stream roofline leg)')
print('# units: KiB as reported (x1024 bytes); gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md): corrected column = 2x')
keys = sorted(set(tot.get('FETCH_SIZE', {})) | set(tot.get('WRITE_SIZE', {})), key=lambda k: -(tot.get('FETCH_SIZE', {}).get(k, [0])[0] * 2 + tot.get('WRITE_SIZE', {}).get(k, [0])[0]))
gf = gw = 0
print('%12s %12s %12s %8s  %s' % ('fetch_MB/st', 'fetch_x2', 'write_MB/st', 'calls/st', 'kernel'))
for k in keys[:45]:
    f = tot.get('FETCH_SIZE', {}).get(k, [0, 0]); w = tot.get('WRITE_SIZE', {}).get(k, [0, 0])
    fm, wm = f[0] * 1024 / 1e6 / steps, w[0] * 1024 / 1e6 / steps
    print('%12.1f %12.1f %12.1f %8.1f  %s' % (fm, 2 * fm, wm, max(f[1], w[1]) / steps, k))
for k in keys:
    gf += tot.get('FETCH_SIZE', {}).get(k, [0, 0])[0]; gw += tot.get('WRITE_SIZE', {}).get(k, [0, 0])[0]
print('# TOTAL per step: fetch %.1f MB (x2 = %.1f MB), write %.1f MB' % (gf * 1024 / 1e6 / steps, 2 * gf * 1024 / 1e6 / steps, gw * 1024 / 1e6 / steps))
PY
head -60 gpurun_out/pmc_summary_$TAG.txt
