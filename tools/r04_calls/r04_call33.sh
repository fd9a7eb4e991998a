#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/r04_lin.py 2>&1 | tail -6
QFX_FUSE_HEAD_LORA=0 timeout 600 python tools/r04_lin.py 2>&1 | tail -6
