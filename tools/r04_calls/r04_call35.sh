#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/attn_var_bench.py b1,b2,g1 --S 2432,8576 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if 'equal' in k: print(k,v)
"
