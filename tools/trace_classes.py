#!/usr/bin/env python3
"""Per-class kernel durations from a rocprofv3 --kernel-trace of bench.py (trace-grade twin of bench.py's event numbers).

    PA_BENCH_SEQ_OUT=seq.json rocprofv3 --kernel-trace --stats -d out -o r -- python bench.py --steps K ...
    python tools/trace_classes.py out/r_results.db seq.json profiles/round2_trace_classes.json

bench.py's roofline pass runs K steps single-stream and the engine records the class of every MFMA-kernel launch in launch
order (pa_net_profile_classes); on one queue launch order is execution order, so the LAST len(sequence) MFMA-kernel dispatches
of the trace are that pass, one to one."""
import collections
import json
import sqlite3
import sys

MFMA_KERNELS = ('conv1x1_tile_kernel', 'conv1x1_oneshot_kernel', 'conv3x3_tile_kernel', 'conv_igemm_kernel', 'wgrad_tile_kernel', 'wgrad_group_kernel', 'conv_wgrad_kernel',
                'stem_conv', 'stem_wgrad')


def main(db, seq_path, out_path):
    seq = json.load(open(seq_path))
    rows = list(sqlite3.connect(db).execute('select name, start, end from kernels order by start'))
    mf = [r for r in rows if any(k in r[0] for k in MFMA_KERNELS) and 'reduce' not in r[0]]
    n = len(seq['sequence'])
    assert len(mf) >= n, (len(mf), n)
    mf = mf[-n:]
    agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for (name, t0, t1), cls in zip(mf, seq['sequence']):
        # a kernel missing from MFMA_KERNELS shifts the whole matching: weight-gradient classes must meet weight-gradient kernels
        assert ('wgrad' in seq['classes'][cls]) == ('wgrad' in name), ('launch order and class sequence disagree', seq['classes'][cls], name)
        a = agg[seq['classes'][cls]]
        a[0] += 1; a[1] += (t1 - t0) / 1e3; a[2][name.split('(')[0]] += 1
    out = {}
    for k, (cnt, us, names) in agg.items():
        out[k] = {'avg_us': round(us / cnt, 3), 'launches_per_step': cnt // seq['steps'], 'ms_per_step': round(us / 1e3 / seq['steps'], 4),
                  'kernels': dict(names.most_common(6)), 'source': 'rocprofv3 --kernel-trace of bench.py (%s)' % db.split('/')[-1]}
    json.dump(out, open(out_path, 'w'), indent=1)
    for k, v in out.items():
        print('%-16s %6.2f us x %3d /step = %.3f ms/step' % (k, v['avg_us'], v['launches_per_step'], v['ms_per_step']))


if __name__ == '__main__':
    main(*sys.argv[1:4])
