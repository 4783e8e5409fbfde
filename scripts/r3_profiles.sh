#!/bin/bash
# Round-3 profile set (run on the GPU box): rocprofv3 kernel stats of the default bench command, of the batch-1 and
# batch-256 Qwen3 decode steps (with the launch-gap table of the batch-1 step), of the M = 512 GEMM probe.
# Outputs under gpurun_out/r3prof/.
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/r3prof; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1 || exit 1; cp $(ls $out/$name/*/*kernel_stats.csv | head -1) $out/${name}_kernel_stats.csv; }
run bench python3 $R/bench.py --no-extra --no-cpu --no-traffic
run decode_b1 python3 $R/scripts/decode_profile.py 1
python3 $R/scripts/trace_gaps.py $(ls $out/decode_b1/*/*kernel_trace.csv | head -1) ${DECODE_B1_LAUNCHES:-143} $out/decode_b1_gaps.json > /dev/null
run decode_b256 python3 $R/scripts/decode_profile.py 256
run gemm python3 $R/scripts/gemm_probe.py 512x8192x8192 512x4096x4096
head -8 $out/decode_b1_kernel_stats.csv | cut -c1-200
