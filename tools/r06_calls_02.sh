#!/bin/bash
# round 6, call 2: first run of the one-pass attention backward (parity, reproducibility, timing) + the claim fix of the two-pass kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_attention_onepass_gpu.py -x -q -k "333 or 200 or 64 or 257 or argument" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_attention_onepass_gpu.py -x -q 2>&1 | tail -15
timeout 300 python tools/attn_onepass_bench.py --S 2432,8576,2432:24:2 2>&1 | tail -5
timeout 300 python tools/attn_var_bench.py base,r4 --qk --S 2432 2>&1 | tail -2
timeout 300 python -m pytest tests/test_accelerate_gpu.py -x -q 2>&1 | tail -4
