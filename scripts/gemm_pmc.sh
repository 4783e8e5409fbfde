#!/bin/bash
# PMC passes over the 512x8192x8192 GEMM probe, one counter group per run (run on the GPU box).
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/gemm_pmc; mkdir -p $out
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 scripts/gemm_probe.py 512x8192x8192 > $out/$name.log 2>&1 || exit 1; }
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/gemm_pmc/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm3' in r['Kernel_Name']:
            k = 'int4' if 'ILb1E' in r['Kernel_Name'] or '<true>' in r['Kernel_Name'] else 'int8'
            acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(k, sum(v) / len(v), len(v))
PY
