#!/bin/bash
# Kernel timeline of the split decode step (two half-batches): scripts/split_timeline.sh <batch> <cache>
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/split_tl; rm -rf $out; mkdir -p $out
export CACTUS_HIP_STEP_SPLIT=2
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out/run -- python3 $R/scripts/decode_profile.py $1 $2 > $out/run.log 2>&1 || exit 1
f=$(ls $out/run/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-460:-230]       # one step from the middle of the last replays
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:70]:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f}  q{r.get("Queue_Id", "?")} {n}')
PY
