#!/bin/bash
# rocprofv3 counter passes over the attention backward kernels (two-pass pair and the one-pass kernel), S = 2432 -> gpurun_out/attn_bwd1_pmc.json
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/b1pmc_$i -o p -- python $R/tools/attn_onepass_bench.py --S ${1:-2432} --rounds 1 --iters 2 ${2:-} > $R/gpurun_out/b1pmc_$i.log 2>&1
done
cd $R
python - <<'P'
import csv, glob, collections, json, re
out = {}
for f in glob.glob('gpurun_out/b1pmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r'attn_\w+', r['Kernel_Name'])
        if not m: continue
        acc[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
for f in glob.glob('gpurun_out/b1pmc_*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'attn_\w+', r['Kernel_Name'])
        if m: out.setdefault(m.group(0), {}).setdefault('_dur', []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in out:
    if '_dur' in out[k]:
        d = sorted(out[k].pop('_dur')); out[k]['_dur_us'] = d[len(d) // 2]
import os
json.dump(out, open('gpurun_out/attn_bwd1_pmc%s.json' % os.environ.get('PMC_TAG', ''), 'w'), indent=1)
for k, v in out.items(): print(k, json.dumps({a: (round(b) if isinstance(b, float) else b) for a, b in v.items()}))
P
rm -rf gpurun_out/b1pmc_?
