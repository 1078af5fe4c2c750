#!/bin/bash
# Everything profiles/<round>_* is made from, in one gpurun call:  bash tools/prof_round.sh round3
#   d/e  kernel stats of the timed region (multi-stream) and of the roofline pass (single stream) + trace classes
#   f    PMC HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes)      g  MFMA busy        rccl  kernel names of one rank through RCCL
R=${1:-round4}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PA_BENCH_CHILD=1      # bench.py: no nested rocprofv3 child passes, no median pass (fixed step counts)
mkdir -p gpurun_out
env -u PA_BENCH_CHILD python bench.py --keep-profiles > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench_line.err
PA_BENCH_SEQ_OUT=gpurun_out/${R}_seq.json rocprofv3 --kernel-trace --stats --output-format csv rocpd -d gpurun_out/prof_${R} -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-traffic --no-floor > gpurun_out/${R}_prof_bench.log 2>&1
python - $R <<'PY'
import csv, glob, json, sqlite3, sys, collections
R = sys.argv[1]
db = glob.glob('gpurun_out/prof_%s/**/*results.db' % R, recursive=True)[0]
rows = list(sqlite3.connect(db).execute('select name, start, end, queue_id from kernels order by start'))
idx = [i for i, r in enumerate(rows) if 'rmsprop' in r[0]]
# bench.py: 5 warm-up + 20 timed steps (multi-stream), then 20 roofline steps (single stream)
def stats(lo, hi, title, path, steps):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, s, e, q in rows[lo:hi]:
        a = agg[name.split('(')[0][:90]]; a[0] += 1; a[1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    with open(path, 'w') as f:
        f.write('# %s\n# kernel time %.3f ms/step, %d launches/step\n' % (title, tot / 1e3 / steps, (hi - lo) // steps))
        f.write('%10s %8s %9s %6s  %s\n' % ('us/step', 'calls/st', 'avg us', '%', 'kernel'))
        for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
            f.write('%10.1f %8.1f %9.2f %6.2f  %s\n' % (us / steps, c / steps, us / c, 100 * us / tot, k))
cmd = 'rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity   (1x MI355X)'
stats(idx[4] + 1, idx[24] + 1, cmd + '\n# TIMED region (20 steps, engine on 4 hardware queues)', 'gpurun_out/%s_d_kernel_stats_multistream.txt' % R, 20)
stats(idx[24] + 1, idx[44] + 1, cmd + '\n# roofline pass (side streams OFF, HIP events around every MFMA launch): 20 steps', 'gpurun_out/%s_e_kernel_stats_single_stream.txt' % R, 20)
wall = (rows[idx[24]][2] - rows[idx[4]][2]) / 1e6 / 20
open('gpurun_out/%s_d_kernel_stats_multistream.txt' % R, 'a').write('# wall time of the region in this (profiled) run: %.3f ms/step\n' % wall)
print('profiled wall %.3f ms/step' % wall)
PY
python tools/trace_classes.py $(find gpurun_out/prof_${R} -name "*results.db" | head -1) gpurun_out/${R}_seq.json gpurun_out/${R}_trace_classes.json > gpurun_out/${R}_trace_classes.txt 2>&1
bash tools/prof_pmc.sh $R > /dev/null 2>&1; cp gpurun_out/pmc_summary_$R.txt gpurun_out/${R}_f_pmc_hbm_traffic.txt
python tools/pmc_to_json.py $R > gpurun_out/${R}_pmc_classes.txt 2>&1
bash tools/prof_mfma.sh $R > /dev/null 2>&1; cp gpurun_out/mfma_summary_$R.txt gpurun_out/${R}_g_mfma_busy.txt
# one rank through RCCL: kernel names
POSEADV_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_rccl -o rccl -- python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --overlap 1 > gpurun_out/${R}_rccl_bench.log 2>&1
f=$(find gpurun_out/prof_${R}_rccl -name "*kernel_stats.csv" | head -1)
python - "$f" > gpurun_out/${R}_rccl_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('# POSEADV_FORCE_DIST=1 WORLD_SIZE=1 rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 5 --warmup 2 --overlap 1   (one rank through RCCL: backend nccl)')
print('# kernels whose name says RCCL / collective:')
for r in rows:
    n = r['Name']
    if 'ccl' in n.lower() or 'AllReduce' in n or 'Broadcast' in n or 'ncclDev' in n:
        print('%6d calls %10.1f us avg  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, n[:140]))
PY
tail -1 gpurun_out/${R}_rccl_bench.log | cut -c1-200
cat gpurun_out/${R}_rccl_kernels.txt | head; cat gpurun_out/${R}_trace_classes.txt
