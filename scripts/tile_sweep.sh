#!/bin/bash
# Times decode-step GEMM shapes with the workgroup tile forced (variants built by
#   scripts/build_gemm_timing.sh r32c64 "-DG3_FORCE_ROWS=32 -DG3_FORCE_COLS=64" ...).
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "== $v"
  CACTUS_HIP_LIBRARY=scripts/variants/libcactus_hip_$v.so timeout -k 10 120 python scripts/gemm_probe.py $SHAPES 2>&1 | grep "int8\|int4"
done
