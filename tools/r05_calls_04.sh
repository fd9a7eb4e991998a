#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/nondet_bisect.py base,pk1,pk1f8n3,pk1f8n7,pk1f8n15 --S 2432 --reps 12 --entry dq --out nondet_bisect_nops.json > gpurun_out/nondet7.log 2>&1
python - <<'P'
import json
d=json.load(open('gpurun_out/nondet_bisect_nops.json'))
print({k:{e:(v[e]['differing_launches'],v[e]['of']) for e in v} for k,v in d.items()})
P
cd tools/hazard_probe && timeout 120 ./pk_cvt_probe 2048 2000 | tee $GRAFT_REPO_ROOT/gpurun_out/pk_cvt_probe.jsonl | cut -c1-200
