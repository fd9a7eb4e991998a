#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/nondet_bisect.py base,pk1,pk1f8,pk1f9,pk1f10,pk1f8910 --S 2432 --reps 12 --entry dq --out nondet_bisect_inloop.json > gpurun_out/nondet6.log 2>&1
python - <<'P'
import json
d=json.load(open('gpurun_out/nondet_bisect_inloop.json'))
print({k:{e:(v[e]['differing_launches'],v[e]['of']) for e in v} for k,v in d.items()})
P
