#!/usr/bin/env python3
"""Summary of tools/trace_dump.py's table of one step: forward / backward split, per-queue busy time, kernels per queue, the step's tail."""
import collections
import sys


def main(path, top=26):
    rows = []
    for l in open(path):
        if l.startswith('#'):
            continue
        t, d, q, g, w, name = l.rstrip('\n').split('\t')
        rows.append((float(t), float(d), int(q), int(g), int(w), name))
    end = max(r[0] + r[1] for r in rows)
    tb = [r[0] for r in rows if 'heat_grad' in r[5]][0]
    tf = [r[0] for r in rows if 'conv_igemm_kernel<128, 64, 0, 1, true' in r[5]][0]
    print('step wall %.1f us, launches %d; forward %.1f, backward %.1f' % (end, len(rows), tb - tf, end - tb))
    qs = sorted(set(r[2] for r in rows))
    mainq = max(qs, key=lambda q: sum(r[1] for r in rows if r[2] == q))
    for q in qs:
        rs = [r for r in rows if r[2] == q]
        print('queue', q, 'launches', len(rs), 'busy %.1f' % sum(r[1] for r in rs),
              'fwd busy %.1f bwd busy %.1f' % (sum(r[1] for r in rs if r[0] < tb), sum(r[1] for r in rs if r[0] >= tb)))
    for q in qs:
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            if r[2] == q:
                a = agg[r[5][:66]]; a[0] += 1; a[1] += r[1]
        print('--- queue', q)
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:(top if q == mainq else 10)]:
            print('  %3d %7.1f %6.1f  %s' % (a[0], a[1], a[1] / a[0], k))
    m = [r for r in rows if r[2] == mainq]
    lm = max(r[0] + r[1] for r in m if r[0] > tb and not any(s in r[5] for s in ('rmsprop', 'weight_prep', 'argmax', 'pck', 'reduce_kernel')))
    print('last main-queue backward kernel ends %.1f; step end %.1f' % (lm, end))
    for r in rows:
        if r[0] > lm - 60:
            print('   %8.1f %7.1f q%d g%-6d %s' % (r[0], r[1], r[2], r[3] // max(1, r[4]), r[5][:70]))


if __name__ == '__main__':
    main(sys.argv[1])
