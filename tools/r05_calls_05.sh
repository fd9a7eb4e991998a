#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/nondet_bisect.py base,pk1,pk1q1,pk1q2,pk1q4,pk1q8,pk1q16,pk1q32,pk1q63 --S 2432 --reps 12 --entry dq --out nondet_bisect_opq.json > gpurun_out/nondet8.log 2>&1
python - <<'P'
import json
d=json.load(open('gpurun_out/nondet_bisect_opq.json'))
print({k:{e:(v[e]['differing_launches'],v[e]['of']) for e in v} for k,v in d.items()})
P
