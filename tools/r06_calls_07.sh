#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_attention_onepass_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/attn_onepass_bench.py --S 2432,8576,2432:24:2 --rounds 6 --libs a3,a31 2>&1 | grep "^S=" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l[l.index('{'):]); print(l[:l.index('{')], {k:(v['us_best'] if isinstance(v,dict) else v) for k,v in d.items()})
"
