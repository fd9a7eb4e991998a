#!/bin/bash
# round 6: levers and the N = 2 bench path at the final HEAD
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
QFX_ATTN_BWD=1pass timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fulldepth_gpu.py tests/test_flux_gpu.py tests/test_attention_onepass_gpu.py -x -q 2>&1 | tail -3
QFX_GRAD_DET=0 timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -2
QFX_SHARE_GPU=1 QFX_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-600
