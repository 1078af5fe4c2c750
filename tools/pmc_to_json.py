"""Turn the merged rocprofv3 PMC CSVs (tools/prof_pmc.sh) into profiles/<tag>_pmc.json: measured HBM bytes per launch for each
MFMA-kernel class of bench.py (its own classes: the launches of the roofline leg matched with the engine's class sequence in launch
order).  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, /opt/skills/guides/MI355X_MICROARCH.md section HBM);
counter units are KiB.  Parser shared with bench.py's in-run traffic measurement: tools/inrun_prof.py."""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import inrun_prof
tag = sys.argv[1]
f = {C: glob.glob('gpurun_out/pmc_%s_%s/*counter_collection.csv' % (tag, C))[0] for C in ('FETCH_SIZE', 'WRITE_SIZE')}
p = inrun_prof.parse_pmc_sequence(f['FETCH_SIZE'], f['WRITE_SIZE'], 'gpurun_out/pmc_%s_seq.json' % tag, 7.0)
res = dict(p['classes'])
res['_whole_step'] = {'fetch_bytes_x2': p['fetch_bytes_per_step_x2'], 'write_bytes': p['write_bytes_per_step']}
res['_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_pmc.sh %s); FETCH_SIZE x2 per the gfx950 correction; classes by launch order' % tag
json.dump(res, open('gpurun_out/%s_pmc.json' % tag, 'w'), indent=1, sort_keys=True)
print(json.dumps({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in res.items() if k[0] != '_'}))
