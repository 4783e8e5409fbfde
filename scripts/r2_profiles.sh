#!/bin/bash
# Round-2 profile set (run on the GPU box): rocprofv3 kernel stats of the default bench command, of the
# batch-256 / batch-1 Qwen3 decode steps, of the M = 512 GEMM probe, the causal prefill attention and the FP16 matmul.  Outputs under gpurun_out/r2prof/.
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/r2prof; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1 || exit 1; cp $(ls $out/$name/*/*kernel_stats.csv | head -1) $out/${name}_kernel_stats.csv; }
run bench python3 $R/bench.py --no-extra --no-cpu --no-traffic
run decode_b256 python3 $R/scripts/decode_profile.py 256
run decode_b1 python3 $R/scripts/decode_profile.py 1
run gemm python3 $R/scripts/gemm_probe.py 512x8192x8192 512x4096x4096
run prefill_attn python3 $R/scripts/pf_probe.py 1024 4096
run matmul_f16 python3 $R/scripts/f16_probe.py
head -12 $out/decode_b256_kernel_stats.csv | cut -c1-220
