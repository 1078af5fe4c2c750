"""Turn the merged rocprofv3 PMC CSVs (tools/prof_pmc.sh) into profiles/<tag>_pmc.json: measured HBM bytes per
launch for each MFMA-kernel class of bench.py.  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced
read, /opt/skills/guides/MI355X_MICROARCH.md section HBM); counter units are KiB."""
import csv, glob, json, re, sys, collections
tag = sys.argv[1]
def cls(name):
    m = re.match(r'void conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\w+)>', name)
    if m:
        ld, taps, stem = int(m.group(3)), int(m.group(4)), m.group(5) == 'true'
        if stem: return 'stem_fwd_7x7'
        return ('conv_dgrad_' if ld == 2 else 'conv_fwd_') + ('3x3' if taps == 9 else '1x1')
    m = re.match(r'void conv3x3_tile_kernel<(\d+), (\d+), (\d+)[,>]', name)
    if m: return 'conv_dgrad_3x3' if int(m.group(3)) == 2 else 'conv_fwd_3x3'
    m = re.match(r'void conv1x1_tile_kernel<(\d+), (\d+), (\d+), (\d+)>', name)
    if m: return 'conv_dgrad_1x1' if int(m.group(4)) == 2 else 'conv_fwd_1x1'
    m = re.match(r'void wgrad_tile_kernel<(\d+),', name)
    if m: return 'conv_wgrad_3x3' if int(m.group(1)) == 9 else 'conv_wgrad_1x1'
    m = re.match(r'void conv_wgrad_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\w+)>', name)
    if m:
        if m.group(6) == 'true': return 'stem_wgrad_7x7'
        return 'conv_wgrad_' + ('3x3' if int(m.group(5)) == 9 else '1x1')
    return None
out = collections.defaultdict(lambda: {'fetch_bytes': 0.0, 'write_bytes': 0.0, 'launches': 0})
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_%s_%s/*counter_collection.csv' % (tag, C))[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != C: continue
        k = cls(r['Kernel_Name'])
        if not k: continue
        v = float(r['Counter_Value']) * 1024.0
        if C == 'FETCH_SIZE':
            out[k]['fetch_bytes'] += 2.0 * v; out[k]['launches'] += 1
        else:
            out[k]['write_bytes'] += v
res = {k: {'hbm_bytes_per_launch': (v['fetch_bytes'] + v['write_bytes']) / max(1, v['launches']),
           'fetch_bytes_per_launch_x2': v['fetch_bytes'] / max(1, v['launches']),
           'write_bytes_per_launch': v['write_bytes'] / max(1, v['launches']), 'launches_profiled': v['launches']} for k, v in out.items()}
res['_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_pmc.sh %s); FETCH_SIZE x2 per the gfx950 correction' % tag
json.dump(res, open('gpurun_out/%s_pmc.json' % tag, 'w'), indent=1, sort_keys=True)
print(json.dumps({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in res.items() if k[0] != '_'}))
