#!/bin/bash
# round 6, call 1: accelerate test, GEMM vs vendor library at HEAD + Tensile kernel names, FLUX / cfg #5 step times, attention r4 vs HEAD,
# default bench (with the new sustained line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_accelerate_gpu.py -x -q 2>&1 | tail -15
timeout 400 python tools/gemm_vs_lib_r06.py 2>&1 | tail -14
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tensile -o t -- python $GRAFT_REPO_ROOT/tools/tensile_names.py > $GRAFT_REPO_ROOT/gpurun_out/tensile.log 2>&1 )
CSV=$(find gpurun_out/tensile -name "*kernel_trace.csv" | head -1)
python tools/tensile_names.py --parse $CSV 2>&1 | tail -12
find gpurun_out/tensile -name "*.csv" -size +2M -delete
timeout 400 python tools/flux_bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/r06_flux_shared.json
timeout 400 python tools/flux_bench.py --steps 10 --warmup 3 --multires 20x20,40x40 2>&1 | tail -1 | tee gpurun_out/r06_flux_multires.json
timeout 400 python tools/flux_bench.py --steps 10 --warmup 3 --multires 20x20,32x32 2>&1 | tail -1 | tee gpurun_out/r06_flux_multires_b.json
timeout 300 python tools/attn_var_bench.py base,r4 --qk --S 2432 2>&1 | tail -12 | tee gpurun_out/r06_attn_r4_vs_head.txt
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/r06_bench_call1.json | cut -c1-1500
