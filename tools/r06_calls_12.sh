#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora_grad" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_flux_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_fulldepth_gpu.py tests/test_fullsize_cfgs_gpu.py -x -q -k "reproducible or fulldepth or full_depth or sixty or 60" 2>&1 | tail -4
timeout 1200 python tools/step_ablate.py --variants full,QFX_GRAD_DET=0 --steps 20 --rounds 4 --out gpurun_out/r06_step_grad_det.json 2>&1 | tail -3
