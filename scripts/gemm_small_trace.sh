#!/bin/bash
# Per-step phase cycles of the small-tile GEMMs of a 256-row step (library built by scripts/build_gemm_trace.sh)
set -e
cd "$(dirname "$0")/.."
export CACTUS_HIP_LIBRARY=$PWD/scripts/variants/libcactus_hip_gemmtrace.so
for s in 256x2048x1024 256x3072x1024 256x1024x4096 256x1024x6144; do
  echo "== $s"
  timeout -k 10 120 python scripts/gemm_trace.py $s
done
