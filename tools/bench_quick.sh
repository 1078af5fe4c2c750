#!/bin/bash
# bench.py one-liner for A/B experiments: prints img/s, ms/step and the per-class MFMA kernel times
python bench.py --no-cpu-baseline "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['classes'].items()})"
