#!/bin/bash
# MFMA-busy per kernel over the bench step (PMC pass of its own, kernel-trace only); run via gpurun
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PA_BENCH_CHILD=1      # bench.py: no nested rocprofv3 child passes, no median pass (fixed step counts)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/mfma_$TAG -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/mfma_$TAG.log 2>&1
tail -1 gpurun_out/mfma_$TAG.log | cut -c1-120
python - $TAG > gpurun_out/mfma_summary_$TAG.txt <<'PY'
import csv, sys, glob, collections
tag = sys.argv[1]
f = glob.glob('gpurun_out/mfma_%s/*counter_collection.csv' % tag)
t = glob.glob('gpurun_out/mfma_%s/*kernel_trace.csv' % tag)
if not f or not t:
    print('no counter file', glob.glob('gpurun_out/mfma_%s/*' % tag)); sys.exit(0)
dur = {}
for r in csv.DictReader(open(t[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
per = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); secs = collections.defaultdict(float)
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:72]
    per[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_BUSY_CYCLES':
        calls[k] += 1; secs[k] += dur.get(r['Dispatch_Id'], 0.0)
CLK = 2.4e9; SIMDS = 1024
print('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 3 --warmup 1   (4 steps; kernels serialised by the counter pass)')
print('# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration from the same trace x 2.4 GHz); check: SQ_INSTS_MFMA x 16 cycles (16x16x32 bf16) / the same denominator')
print('# (GRBM_GUI_ACTIVE is summed over the 8 XCDs in this collection, so the gfx94x derived formula MfmaUtil = busy / (GUI_ACTIVE x CUs x 4) reads 8x low)')
print('%9s %9s %9s %9s %12s  %s' % ('busy%', 'insts%', 'calls/st', 'us/call', 'mfma/call', 'kernel'))
rows = []; tb = ts = 0.0
for k, c in per.items():
    if not calls[k] or secs[k] <= 0: continue
    den = SIMDS * secs[k] * CLK
    tb += c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); ts += secs[k]
    rows.append((secs[k], k, 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / den, 100 * 16 * c.get('SQ_INSTS_MFMA', 0.0) / den, calls[k] / 4.0,
                 1e6 * secs[k] / calls[k], c.get('SQ_INSTS_MFMA', 0.0) / calls[k]))
rows.sort(reverse=True)
for sec, k, b, i2, n, us, ins in rows[:40]:
    print('%9.1f %9.1f %9.1f %9.1f %12.0f  %s' % (b, i2, n, us, ins, k))
print('# all kernels together: MFMA busy %.1f %% of the kernel time (%.2f ms/step serialised)' % (100 * tb / (SIMDS * ts * CLK), 1e3 * ts / 4))
PY
head -30 gpurun_out/mfma_summary_$TAG.txt
