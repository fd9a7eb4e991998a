#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
QFX_LIB_PATH=$R/tools/_ab/libqfx_vfix.so timeout 600 python tools/find_nondet.py --blocks 2 --passes 5 2>&1 | grep -v "^  " | tail -6
QFX_LIB_PATH=$R/tools/_ab/libqfx_vcur.so timeout 600 python tools/find_nondet.py --blocks 2 --passes 3 2>&1 | grep -v "^  " | tail -4
