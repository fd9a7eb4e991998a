#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "ln_down or lora_pack or tiny_step or loss_curve or three_image or ff_lora or head_lora" 2>&1 | tail -3 )
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed"
for i in 1 2; do
  QFX_LN_DOWN_FRAG=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frag   ', d['ms_per_step'], d['roofline']['frac'])"
  QFX_LN_DOWN_FRAG=0 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rowmaj ', d['ms_per_step'], d['roofline']['frac'])"
done
