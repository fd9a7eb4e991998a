#!/bin/bash
R=$(pwd); export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 300 python tools/attn_var_bench.py base-,base,hlnoload,hlnostore,hlnone --entry fwd --S 2432 --hl 16 2>&1 | tail -1
