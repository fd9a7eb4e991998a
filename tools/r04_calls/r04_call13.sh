#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 900 python tools/step_without.py --steps 8 --rounds 3 2>&1 | tail -3
