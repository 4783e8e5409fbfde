#!/usr/bin/env python3
"""Durations and gaps of the launches of one captured step from a rocprofv3 --kernel-trace CSV:
python3 scripts/trace_gaps.py <kernel_trace.csv> <launches per step> [out.json]
Takes the last whole step of the trace: per kernel name the count, mean duration and the mean idle time between its
end and the next launch's start; totals of busy and idle time."""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
per_step = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = rows[-per_step:]
by = defaultdict(lambda: [0, 0.0, 0.0])
busy = idle = 0.0
for a, b in zip(step, step[1:] + [None]):
    d = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 if b else 0.0
    name = a["Kernel_Name"].split("(")[0][:90]
    e = by[name]
    e[0] += 1; e[1] += d; e[2] += g
    busy += d; idle += g
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
out = {"launches": len(step), "span_us": round(span, 1), "kernel_us": round(busy, 1), "gaps_us": round(idle, 1),
       "mean_gap_us": round(idle / max(1, len(step) - 1), 2),
       "by_kernel": {k: {"count": v[0], "mean_us": round(v[1] / v[0], 2), "mean_gap_after_us": round(v[2] / v[0], 2)}
                     for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])}}
js = json.dumps(out, indent=1)
print(js)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(js)
