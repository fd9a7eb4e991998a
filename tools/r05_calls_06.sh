#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/nondet_bisect.py noslp,pk1noslp,pk1q1 --S 8576 --reps 16 --out nondet_bisect_noslp.json > gpurun_out/nondet9.log 2>&1
python - <<'P'
import json
d=json.load(open('gpurun_out/nondet_bisect_noslp.json'))
print({k:{e:(v[e]['differing_launches'],v[e]['of']) for e in v} for k,v in d.items()})
P
timeout 600 python tools/attn_var_bench.py base,noslp,pk1noslp --qk --hl 16 > gpurun_out/attn_noslp.log 2>&1; tail -1 gpurun_out/attn_noslp.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(k,v)
"
