#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/attn_onepass_bench.py --S 2432 --rounds 6 --libs st0,st1,st2,st17 2>&1 | grep "^S=" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l[l.index('{'):]); print(l[:l.index('{')], {k:(v['us_best'] if isinstance(v,dict) else v) for k,v in d.items()})
"
