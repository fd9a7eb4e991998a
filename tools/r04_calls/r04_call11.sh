#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 300 python tools/attn_var_bench.py base-,base --entry fwd --S 2432 --hl 16 2>&1 | tail -1
bash tools/r04_calls/r04_call8.sh
