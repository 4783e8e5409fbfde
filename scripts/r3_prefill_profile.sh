#!/bin/bash
# rocprofv3 kernel stats of the chunked 4096-token Qwen3-0.6B prefill (256-token chunks). Output: gpurun_out/r3prof/
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/r3prof; mkdir -p $out; rm -rf $out/prefill256
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prefill256 -- python3 $R/scripts/prefill_e2e.py 4096 256 > $out/prefill256.log 2>&1 || exit 1
cp $(ls $out/prefill256/*/*kernel_stats.csv | head -1) $out/prefill256_kernel_stats.csv
head -16 $out/prefill256_kernel_stats.csv | cut -c1-160
