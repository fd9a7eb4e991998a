#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "tiny_step or loss_curve or golden or hipgraph or three_image" 2>&1 | tail -3 )
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed"
for c in 0 6 12 0 6 3; do
  QFX_MODS_SIDE_CHUNK=$c timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk $c', d['ms_per_step'], d['roofline']['frac'])"
done
