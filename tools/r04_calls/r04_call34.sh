#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
for c in c17ed98 49c4f68 b1565e8 595fc9d; do
  echo "== $c"; QFX_LIB_PATH=$R/tools/_ab/libqfx_bis_$c.so timeout 600 python tools/r04_lin.py 2>&1 | grep "repeat\|^lin"
done
