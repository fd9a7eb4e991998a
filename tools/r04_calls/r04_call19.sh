#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/step_ab.py r3,prev,pk --layers 6 --reps 5 --only "attn,ln_" --out gpurun_out/step_ab_pk.json 2>&1 | tail -12
timeout 600 python tools/step_lib_ab.py r3,prev,pk --steps 20 --rounds 3 --out gpurun_out/step_lib_ab_pk.json 2>&1 | tail -3
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "ln or modulate" 2>&1 | tail -2 )
