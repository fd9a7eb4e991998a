#!/bin/bash
# round 6: lora_grad's deterministic epilogue -- hand-off forms (QFX_GRAD_HANDOFF build lever) under the 60-block step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIB=qwen-image-finetune_amd/qflux_amd/libqfx.so
cp $LIB /tmp/libqfx_keep.so
for v in h8; do
  cp tools/_ab/libqfx_$v.so $LIB
  for i in 1 2 3; do echo "$v run $i: $(timeout 600 python -m pytest tests/test_fulldepth_gpu.py -x -q -k qwen_60 2>&1 | grep -E "AssertionError|passed|failed" | head -2 | tr '\n' ' ')"; done
  timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "lora_grad or bit_reproducible or model" 2>&1 | tail -2
done
cp /tmp/libqfx_keep.so $LIB
timeout 900 python tools/step_lib_ab.py h0,h8,h1 --steps 20 --rounds 3 --gflat-repro --out gpurun_out/r06_grad_handoff_b.json 2>&1 | tail -8
