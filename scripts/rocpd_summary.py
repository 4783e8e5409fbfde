#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite, the default
output format of ROCm 7) as CSV: one row per kernel name with calls / total / average / min / max
duration, and a second table split by launch grid so that shapes of one kernel can be told apart.

    python scripts/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/<name>.csv
"""
import sqlite3
import sys


def main(path, pattern="%"):
    cur = sqlite3.connect(path).cursor()
    print("# per kernel (durations in microseconds)")
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct")
    rows = list(cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
        "from kernels where name like ? group by name order by 3 desc", (pattern,)))
    total = sum(r[2] for r in rows) or 1.0
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 160 else name[:157] + "..."
        print(f"\"{short}\",{n},{tot:.2f},{avg:.3f},{mn:.3f},{mx:.3f},{100 * tot / total:.2f}")
    print()
    print("# per kernel and launch shape")
    print("kernel,grid_x,grid_y,workgroup_x,lds_bytes,vgprs,calls,avg_us,min_us")
    for r in cur.execute(
            "select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, count(*), avg(duration)/1e3, "
            "min(duration)/1e3 from kernels where name like '%cactus_hip%' group by name, grid_x, grid_y "
            "order by name, grid_x, grid_y"):
        short = r[0] if len(r[0]) < 100 else r[0][:97] + "..."
        print(f"\"{short}\",{r[1]},{r[2]},{r[3]},{r[4]},{r[5]},{r[6]},{r[7]:.3f},{r[8]:.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
