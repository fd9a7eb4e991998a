#!/bin/bash
set -u
R=$(pwd)
mkdir -p gpurun_out
export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -5 ) > gpurun_out/c3_gemm_tests.log 2>&1
( timeout 300 python tools/step_lib_ab.py r3,base --steps 20 --rounds 4 --out gpurun_out/c3_r3_lib_ab.json 2>&1 | tail -8 ) > gpurun_out/c3_r3_lib_ab.log 2>&1
( timeout 300 python tools/tiles_ab.py --steps 12 --rounds 3 --policies legacy,n160,n160only 2>&1 | tail -3 | cut -c1-300 ) > gpurun_out/c3_tiles_ab.log 2>&1
cp gpurun_out/tiles_ab.json gpurun_out/c3_tiles_ab.json
( timeout 300 python tools/tiles_ab.py --steps 10 --rounds 3 --batch 2 --policies legacy,n160 2>&1 | tail -3 | cut -c1-300 ) > gpurun_out/c3_tiles_ab_b2.log 2>&1
cp gpurun_out/tiles_ab.json gpurun_out/c3_tiles_ab_b2.json
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/c3_pmc_$TAG -o p -- python $R/tools/attn_pmc.py 2432 3 > $R/gpurun_out/c3_pmc_$TAG.log 2>&1
done
cd $R
find gpurun_out -name "*kernel_trace.csv" -path "*c3_pmc_*" -delete
python - <<'P'
import csv, glob, collections, json
out = {}
for f in glob.glob('gpurun_out/c3_pmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        if 'attn' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
json.dump(out, open('gpurun_out/c3_attn_pmc.json', 'w'), indent=1)
print(json.dumps(out)[:3000])
P
cat gpurun_out/c3_gemm_tests.log | tail -2; cat gpurun_out/c3_r3_lib_ab.log
