#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bit_reproducible or attention" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "head_lora" 2>&1 | tail -5
timeout 300 python tools/nondet_bisect.py base --S 8576 --reps 24 --out nondet_product.json 2>&1 | tail -1 | cut -c1-300
