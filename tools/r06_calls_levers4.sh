#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora_grad or bit_reproducible" 2>&1 | tail -4
timeout 1200 python tools/step_plan_ab.py base,QFX_GRAD_DET=0 --steps 20 --rounds 3 --out gpurun_out/r06_step_grad_det_v2.json 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -3
