"""rocprofv3 passes of bench.py started FROM bench.py (its `roofline.traffic` / `roofline.trace` measured in the same run), and the
parsers tools/pmc_to_json.py / tools/prof_pmc.sh share with it.

  traffic : two `rocprofv3 --pmc <C> --kernel-trace` passes (C = FETCH_SIZE, WRITE_SIZE -- separate passes, never combined with
            sys / runtime / hip / hsa traces) of `python bench.py --steps 3 --warmup 1 ...` (4 steps + the 3 steps of its roofline leg, whose
            launches are matched with the engine's class sequence).  Counter unit:
            KiB as reported; FETCH_SIZE x2 (gfx950 tallies a wide coalesced 128-byte request at 64 bytes,
            /opt/skills/guides/MI355X_MICROARCH.md section HBM).
  trace   : one `rocprofv3 --kernel-trace` pass whose roofline leg (single stream) is matched launch by launch with the engine's
            class sequence (tools/trace_classes.py).

Every pass is a child process with its own timeout; a failure returns None and the caller keeps the recorded numbers."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(name):
    """bench.py's MFMA-kernel class of a kernel name of this library (None: not an MFMA kernel)."""
    m = re.match(r'void conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\w+)>', name)
    if m:
        ld, taps, stem = int(m.group(3)), int(m.group(4)), m.group(5) == 'true'
        if stem:
            return 'stem_fwd_7x7'
        return ('conv_dgrad_' if ld == 2 else 'conv_fwd_') + ('3x3' if taps == 9 else '1x1')
    m = re.match(r'void conv3x3_(?:tile|tri)_kernel<(\d+), (\d+), (\d+)[,>]', name)
    if m:
        return 'conv_dgrad_3x3' if int(m.group(3)) == 2 else 'conv_fwd_3x3'
    m = re.match(r'void conv1x1_oneshot_kernel<(\d+), (\d+),', name)
    if m:
        return 'conv_dgrad_1x1' if int(m.group(2)) == 2 else 'conv_fwd_1x1'
    m = re.match(r'void conv1x1_tile_kernel<(\d+), (\d+), (\d+), (\d+)[,>]', name)
    if m:
        return 'conv_dgrad_1x1' if int(m.group(4)) == 2 else 'conv_fwd_1x1'
    if 'wgrad_group_kernel' in name:           # several layers in one launch (round 5): counted with the 3x3 class, whose job is the longest of a block's group
        return 'conv_wgrad_3x3'
    m = re.match(r'void wgrad_tile_kernel<(\d+),', name)
    if m:
        return 'conv_wgrad_3x3' if int(m.group(1)) in (3, 9) else 'conv_wgrad_1x1'
    m = re.match(r'void conv_wgrad_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\w+)>', name)
    if m:
        if m.group(6) == 'true':
            return 'stem_wgrad_7x7'
        return 'conv_wgrad_' + ('3x3' if int(m.group(5)) == 9 else '1x1')
    return None


def parse_pmc(fetch_csv, write_csv, steps):
    """-> {'classes': {cls: {...per launch}}, 'fetch_bytes_per_step_x2', 'write_bytes_per_step', 'kernels': {...}}"""
    out = collections.defaultdict(lambda: {'fetch_bytes': 0.0, 'write_bytes': 0.0, 'launches': 0})
    per_kernel = collections.defaultdict(lambda: [0.0, 0.0, 0])
    tot = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0}
    for C, path in (('FETCH_SIZE', fetch_csv), ('WRITE_SIZE', write_csv)):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != C:
                continue
            v = float(r['Counter_Value']) * 1024.0 * (2.0 if C == 'FETCH_SIZE' else 1.0)
            tot[C] += v
            pk = per_kernel[r['Kernel_Name'].split('(')[0][:90]]
            pk[0 if C == 'FETCH_SIZE' else 1] += v
            if C == 'FETCH_SIZE':
                pk[2] += 1
            k = classify(r['Kernel_Name'])
            if not k:
                continue
            if C == 'FETCH_SIZE':
                out[k]['fetch_bytes'] += v
                out[k]['launches'] += 1
            else:
                out[k]['write_bytes'] += v
    classes = {k: {'hbm_bytes_per_launch': (v['fetch_bytes'] + v['write_bytes']) / max(1, v['launches']),
                   'fetch_bytes_per_launch_x2': v['fetch_bytes'] / max(1, v['launches']),
                   'write_bytes_per_launch': v['write_bytes'] / max(1, v['launches']), 'launches_profiled': v['launches']} for k, v in out.items()}
    return {'classes': classes, 'fetch_bytes_per_step_x2': tot['FETCH_SIZE'] / steps, 'write_bytes_per_step': tot['WRITE_SIZE'] / steps,
            'kernels': {k: {'fetch_x2_MB_per_step': v[0] / 1e6 / steps, 'write_MB_per_step': v[1] / 1e6 / steps, 'calls_per_step': v[2] / steps}
                        for k, v in per_kernel.items()}}


MFMA_KERNELS = ('conv1x1_tile_kernel', 'conv1x1_oneshot_kernel', 'conv3x3_tile_kernel', 'conv_igemm_kernel', 'wgrad_tile_kernel', 'wgrad_group_kernel', 'conv_wgrad_kernel', 'stem_conv',
                'stem_wgrad')


def parse_pmc_sequence(fetch_csv, write_csv, seq_path, steps):
    """Per-class traffic with bench.py's OWN classes (they depend on the map size, which a kernel name does not show): the child pass
    also runs the roofline leg, whose MFMA-kernel launches are the last len(sequence) MFMA dispatches of the counter file, in launch
    order (single stream) -- the matching tools/trace_classes.py does for durations.  Whole-step totals over all `steps` steps."""
    seq = json.load(open(seq_path))
    names, order = seq['classes'], seq['sequence']
    out = collections.defaultdict(lambda: {'fetch_bytes': 0.0, 'write_bytes': 0.0, 'launches': 0})
    tot, fill = {}, {}
    for C, path in (('FETCH_SIZE', fetch_csv), ('WRITE_SIZE', write_csv)):
        rows = [(int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value']) * 1024.0 * (2.0 if C == 'FETCH_SIZE' else 1.0))
                for r in csv.DictReader(open(path)) if r['Counter_Name'] == C]
        rows.sort()
        # the runtime's one-time clear of the workspace (pa_net_bind -> hipMemset -> __amd_rocclr_fillBufferAligned) is not part of a step
        fill[C] = sum(v for _, k, v in rows if 'fillBuffer' in k)
        tot[C] = sum(v for _, k, v in rows if 'fillBuffer' not in k)
        mf = [r for r in rows if any(k in r[1] for k in MFMA_KERNELS) and 'reduce' not in r[1]]
        if len(mf) < len(order):
            raise ValueError('fewer MFMA dispatches than the class sequence')
        for (_, kname, v), c in zip(mf[-len(order):], order):
            if ('wgrad' in names[c]) != ('wgrad' in kname):      # (a kernel missing from MFMA_KERNELS shifts the whole matching)
                raise ValueError('launch order and class sequence disagree: %s / %s' % (names[c], kname))
            o = out[names[c]]
            if C == 'FETCH_SIZE':
                o['fetch_bytes'] += v; o['launches'] += 1
            else:
                o['write_bytes'] += v
    classes = {k: {'hbm_bytes_per_launch': (v['fetch_bytes'] + v['write_bytes']) / max(1, v['launches']),
                   'fetch_bytes_per_launch_x2': v['fetch_bytes'] / max(1, v['launches']),
                   'write_bytes_per_launch': v['write_bytes'] / max(1, v['launches']), 'launches_profiled': v['launches']} for k, v in out.items()}
    return {'classes': classes, 'fetch_bytes_per_step_x2': tot['FETCH_SIZE'] / steps, 'write_bytes_per_step': tot['WRITE_SIZE'] / steps,
            'excluded_one_time_fill_bytes': fill['FETCH_SIZE'] + fill['WRITE_SIZE']}


def _env():
    e = dict(os.environ, PA_BENCH_CHILD='1', TMPDIR='/tmp')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'POSEADV_DIST_INIT', 'POSEADV_FORCE_DIST'):
        e.pop(k, None)
    return e


def available():
    return shutil.which('rocprofv3') is not None and os.environ.get('PA_BENCH_CHILD') != '1'


def _workdir(keep):
    base = os.path.join(ROOT, 'gpurun_out') if keep and os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else None
    return tempfile.mkdtemp(prefix='inrun_prof_', dir=base)


def measure_traffic(bench_args, timeout=120, keep=False):
    """bench_args: the workload flags of the running bench.py (--bs / --stacks / ...).  -> parse_pmc() dict or None."""
    if not available():
        return None
    d = _workdir(keep)
    try:
        paths = {}
        seq = os.path.join(d, 'seq.json')
        for C in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(d, C)
            # 1 warm-up + 3 timed steps, then the roofline leg's 3 steps (single stream, class sequence recorded): 7 steps of the same work
            cmd = ['rocprofv3', '--pmc', C, '--kernel-trace', '--output-format', 'csv', '-d', out, '-o', 'bench', '--', sys.executable,
                   os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-parity', '--no-traffic',
                   '--no-floor'] + list(bench_args)
            r = subprocess.run(cmd, cwd=ROOT, env=dict(_env(), PA_BENCH_SEQ_OUT=seq), capture_output=True, text=True, timeout=timeout)
            f = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not f or not os.path.isfile(seq):
                return None
            paths[C] = f[0]
        return parse_pmc_sequence(paths['FETCH_SIZE'], paths['WRITE_SIZE'], seq, 7.0)
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)


def measure_trace(bench_args, steps=5, timeout=120, keep=False):
    """One --kernel-trace pass; -> {class: {'avg_us', 'launches_per_step', 'ms_per_step'}} of its roofline leg, or None."""
    if not available():
        return None
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import trace_classes
    d = _workdir(keep)
    try:
        seq = os.path.join(d, 'seq.json')
        cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', 'rocpd', '-d', os.path.join(d, 'trace'), '-o', 'bench', '--', sys.executable,
               os.path.join(ROOT, 'bench.py'), '--steps', str(steps), '--warmup', '2', '--no-cpu-baseline', '--no-parity', '--no-traffic', '--no-floor'] + list(bench_args)
        r = subprocess.run(cmd, cwd=ROOT, env=dict(_env(), PA_BENCH_SEQ_OUT=seq), capture_output=True, text=True, timeout=timeout)
        db = glob.glob(os.path.join(d, 'trace', '**', '*results.db'), recursive=True)
        if r.returncode != 0 or not db or not os.path.isfile(seq):
            return None
        outp = os.path.join(d, 'classes.json')
        with open(os.devnull, 'w') as null:
            so, sys.stdout = sys.stdout, null
            try:
                trace_classes.main(db[0], seq, outp)
            finally:
                sys.stdout = so
        return json.load(open(outp))
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError, AssertionError):
        return None
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)
