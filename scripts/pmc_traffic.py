#!/usr/bin/env python3
"""Turn a `rocprofv3 --pmc FETCH_SIZE` (and optionally a second `--pmc WRITE_SIZE`) pass over
bench.py into HBM bytes per launch of the batched GEMV kernel.

FETCH_SIZE is reported in KiB and, on gfx950, tallies the 128-byte requests of wide coalesced
streaming reads at 64 bytes each: the read figure is doubled.  WRITE_SIZE needs no correction.

    python scripts/pmc_traffic.py fetch_counter_collection.csv [write_counter_collection.csv] > profiles/r01_pmc_traffic.json
"""
import csv
import json
import sys


def mean_counter(path, kernel, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main(fetch_csv, write_csv=None, kernel="skinny_batch"):
    f, nf = mean_counter(fetch_csv, kernel, "FETCH_SIZE")
    out = {"kernel": kernel, "launches": nf, "FETCH_SIZE_KiB_raw": f,
           "read_bytes_per_launch": None if f is None else int(f * 1024 * 2),
           "correction": "FETCH_SIZE x 1024 (KiB) x 2 (gfx950 tallies 128-B requests at 64 B)"}
    total = out["read_bytes_per_launch"]
    if write_csv:
        w, nw = mean_counter(write_csv, kernel, "WRITE_SIZE")
        out["WRITE_SIZE_KiB_raw"] = w
        out["write_bytes_per_launch"] = None if w is None else int(w * 1024)
        if w is not None and total is not None:
            total += out["write_bytes_per_launch"]
    out["hbm_bytes_per_launch"] = total
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
