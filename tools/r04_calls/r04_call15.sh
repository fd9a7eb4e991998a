#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "attention or head_lora or out_projection or tiny_step or loss_curve" 2>&1 | tail -5 )
timeout 600 python tools/attn_var_bench.py r3,base --entry fwd --S 2432,8576 2>&1 | tail -1
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed"
for i in 1 2; do
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['roofline']['frac'])"
done
