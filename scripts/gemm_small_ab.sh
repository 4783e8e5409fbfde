#!/bin/bash
# A/B of the small-tile GEMM variants on the 256-row decode-step shapes (hot weights) and the batch-256 step (cold)
cd "$(dirname "$0")/.."
S="256x2048x1024 256x3072x1024 256x1024x4096 256x1024x6144"
for v in "" gemmold gemm_ring4_ahead gemm_deep_plain gemm_ring4_plain; do
  if [ -n "$v" ]; then export CACTUS_HIP_LIBRARY=$PWD/scripts/variants/libcactus_hip_$v.so; fi
  echo "== ${v:-current}"
  timeout -k 10 200 python scripts/gemm_probe.py $S 2>&1 | grep int8 || exit 1
  timeout -k 10 200 python scripts/decode_profile.py 256 2>&1 | tail -1 | cut -c1-60 || exit 1
done
