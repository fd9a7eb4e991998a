#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_attention_onepass_gpu.py tests/test_kernels_gpu.py -x -q -k "attention or onepass" 2>&1 | tail -4
QFX_ATTN_BWD=1pass timeout 900 python -m pytest tests/test_model_gpu.py tests/test_flux_gpu.py -x -q 2>&1 | tail -4
QFX_ATTN_BWD=1pass timeout 900 python -m pytest tests/test_fulldepth_gpu.py tests/test_fullsize_cfgs_gpu.py -x -q 2>&1 | tail -4
timeout 300 python tools/attn_onepass_bench.py --S 2432 --rounds 6 2>&1 | grep "^S=" | cut -c1-400
for m in 2pass 1pass 2pass 1pass; do echo "== $m"; QFX_ATTN_BWD=$m timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed --sustained-steps 0 2>&1 | tail -1 | cut -c1-220; done
