#!/bin/bash
# Counter passes (one counter group per rocprofv3 run) over a probe command; prints per-kernel averages.
#   scripts/r2_pmc.sh <tag> <kernel substring> -- <probe command...>
R="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1; kern=$2; shift 3
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- "${CMD[@]}" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }; }
CMD=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC
run sq3 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_I8
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- "${CMD[@]}" > $out/stats.log 2>&1
cd $R
python3 - "$out" "$kern" <<'PY'
import csv, glob, collections, sys, json
out, kern = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(out + '/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        res[k] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
for f in glob.glob(out + '/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if kern in r['Name']:
            res.setdefault("kernel_stats", []).append({"name": r['Name'][:120], "calls": int(r['Calls']), "avg_ns": float(r['AverageNs'])})
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
for k, v in res.items():
    print(k, v)
PY
tail -2 $out/stats.log
