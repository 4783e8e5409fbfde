#!/bin/bash
# PMC passes over the causal prefill attention probe at S = 4096, one counter group per run (run on the GPU box).
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/pf_pmc; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 scripts/pf_probe.py 4096 > $out/$name.log 2>&1 || exit 1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob('gpurun_out/pf_pmc/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'attn_prefill_pp' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        res[k] = sum(v) / len(v)
        print(k, sum(v) / len(v), len(v))
json.dump(res, open('gpurun_out/pf_pmc/summary.json', 'w'), indent=1)
PY
