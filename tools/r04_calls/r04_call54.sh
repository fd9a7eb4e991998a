#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
for w in 8 4 8 4; do echo "waves $w"; QFX_ATTN_FWD_WAVES=$w timeout 300 python tools/attn_var_bench.py base --entry fwd --S 2432,3072,4096 2>&1 | tail -1; done
