#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( QFX_LIB_PATH=$R/tools/_ab/libqfx_pp2.so timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_cfgs_gpu.py -x -q -k "attention or attn" 2>&1 | tail -6 ) > gpurun_out/c4_attn_tests_pp2.log 2>&1
( timeout 300 python tools/attn_var_bench.py base,pp1,pp2 --entry fwd 2>&1 | tail -2 ) > gpurun_out/c4_attn_var.log 2>&1
( timeout 300 python tools/step_lib_ab.py base,pp1 --steps 20 --rounds 3 --out gpurun_out/c4_pp_lib_ab.json 2>&1 | tail -4 ) > gpurun_out/c4_pp_lib_ab.log 2>&1
cat gpurun_out/c4_attn_tests_pp2.log | tail -3; cat gpurun_out/c4_attn_var.log; cat gpurun_out/c4_pp_lib_ab.log
