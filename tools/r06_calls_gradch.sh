#!/bin/bash
# round 6: token-chunk size of lora_grad under the two-launch reduction (whole libraries swapped in: the scratch size follows the chunk size)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIB=qwen-image-finetune_amd/qflux_amd/libqfx.so
cp $LIB /tmp/libqfx_keep.so
for r in 1 2; do for v in ch512 ch256 ch1024; do
  cp tools/_ab/libqfx_$v.so $LIB
  echo "$v round $r: $(timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed --sustained-steps 0 --no-live-traffic 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
cp /tmp/libqfx_keep.so $LIB
