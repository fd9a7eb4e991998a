#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/c5_pmc_$TAG -o p -- python $R/tools/gemm_pmc.py > $R/gpurun_out/c5_pmc_$TAG.log 2>&1
done
cd $R
find gpurun_out -name "*kernel_trace.csv" -path "*c5_pmc_*" -delete
python - <<'P'
import csv, glob, collections, json
out = {}
for f in glob.glob('gpurun_out/c5_pmc_*/p_counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm256' not in k: continue
        k = k.split('gemm256_kernel')[1].split('(')[0][:24] + ' grid' + r['Grid_Size']
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[k]['_dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k, d in acc.items():
        out.setdefault(k, {}).update({c: sorted(v)[len(v)//2] for c, v in d.items()})
json.dump(out, open('gpurun_out/c5_gemm_pmc.json', 'w'), indent=1)
P
cat gpurun_out/c5_gemm_pmc.json | head -120
