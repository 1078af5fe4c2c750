#!/bin/bash
# SQ counters for the conv micro-benchmarks
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PA_BENCH_CHILD=1      # bench.py: no nested rocprofv3 child passes, no median pass (fixed step counts)
cat > /tmp/one.py <<'PY'
import sys; sys.path.insert(0, '.')
import ctypes as C, torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
L.pa_conv2d_time.restype = C.c_int
L.pa_conv2d_time.argtypes = [C.c_int]*9 + [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
ws = torch.zeros(3 << 30, dtype=torch.uint8, device='cuda')
ms = C.c_float()
for mode, var in ((0, 3), (1, 3), (2, 9)):
    check(L.pa_conv2d_time(mode, var, 24, 128, 128, 64, 64, 3, 5, ptr(ws), C.byref(ms), stream()))
    print(mode, var, ms.value * 1e3, 'us')
for mode, var in ((0, 7), (1, 7), (2, 9)):
    check(L.pa_conv2d_time(mode, var, 24, 256, 128, 64, 64, 1, 5, ptr(ws), C.byref(ms), stream()))
    print(mode, var, ms.value * 1e3, 'us')
PY
rm -rf gpurun_out/sq_*
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $SET | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/sq_$tag -o c -- python /tmp/one.py > gpurun_out/sq_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob('gpurun_out/sq_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:64]
        if 'conv' not in k and 'wgrad' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    n = 8.0
    print(k)
    print('   ', {c: round(v / n) for c, v in sorted(d.items())})
PY
