#!/usr/bin/env python3
"""Print VGPR / AGPR / SGPR / scratch / LDS of every kernel in a HIP source, from the metadata
hipcc emits with -S (no GPU needed).   python scripts/kernel_regs.py cactus_amd/csrc/x.hip [filter]"""
import re
import subprocess
import sys
import tempfile


def main(src, flt=""):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                        "--cuda-device-only", "-o", f.name, src], check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    for blk in text.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        if flt in name:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} "
                  f"lds {g('group_segment_fixed_size'):>6}  {dem[:110]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
