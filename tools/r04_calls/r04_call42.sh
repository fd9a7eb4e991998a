#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_fullsize_cfgs_gpu.py -x -q -k "bit_reproducible" 2>&1 | grep -E "Error|assert|bad|^E" | head -12 )
