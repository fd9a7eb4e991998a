#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "head_lora or out_projection_down or attention or tiny_step or loss_curve or lora" 2>&1 | tail -12 ) > gpurun_out/c8_tests.log 2>&1
cat gpurun_out/c8_tests.log | tail -12
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed"
for i in 1 2; do
  QFX_FUSE_HEAD_LORA=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused   ', d['ms_per_step'], d['roofline']['frac'], {k:(v['launches'],v['ms_per_step']) for k,v in d['hbm_kernels'].items() if 'lora' in k})"
  QFX_FUSE_HEAD_LORA=0 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('separate', d['ms_per_step'], d['roofline']['frac'], {k:(v['launches'],v['ms_per_step']) for k,v in d['hbm_kernels'].items() if 'lora' in k})"
done
