#!/bin/bash
# rocprofv3 kernel stats of the LFM2-1.2B-shape decode step at 1024 sequences (run on the GPU box).
# Output: gpurun_out/lfm2prof/lfm2_b1024_kernel_stats.csv
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/lfm2prof; rm -rf $out; mkdir -p $out
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b1024 -- python3 $R/scripts/lfm2_profile.py 1024 512 > $out/b1024.log 2>&1 || exit 1
cp $(ls $out/b1024/*/*kernel_stats.csv | head -1) $out/lfm2_b1024_kernel_stats.csv
head -14 $out/lfm2_b1024_kernel_stats.csv | cut -c1-220
