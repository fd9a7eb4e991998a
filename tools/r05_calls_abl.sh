#!/bin/bash
R=$(pwd); OUT=$R/gpurun_out/ablprof; mkdir -p $OUT
timeout 300 python tools/step_ab.py base,v2v,v2vb,nocomp --layers 6 --only gemm 2>&1 | tail -21 > $OUT/gemm_v2v.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o p -- python $R/tools/step_ablate.py --variants full,no_lora_grad --steps 8 --rounds 2 > $OUT/abl.log 2>&1
python $R/tools/trace_steps.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) -o $OUT/lora_grad_vs_gemm.json
find $OUT/tr -name "*kernel_trace.csv" -delete
cat $OUT/gemm_v2v.txt
