#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
for v in vcur vnarrow vscalar; do echo "== $v"; QFX_LIB_PATH=$R/tools/_ab/libqfx_$v.so timeout 600 python tools/find_nondet.py --blocks 2 --passes 3 2>&1 | grep "^pass"; done
