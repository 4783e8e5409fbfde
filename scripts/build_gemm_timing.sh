#!/bin/bash
# Timing-experiment variants of the GEMM (wrong results by design):
#   scripts/build_gemm_timing.sh name "-DG3_TIMING=2" (bits: cactus_amd/csrc/matmul_gemm.hip) [name2 "flags2" ...]
# -> scripts/variants/libcactus_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
python cactus_amd/build.py > /dev/null
mkdir -p scripts/variants
objs=$(ls cactus_amd/build/*.hip.o cactus_amd/build/*.cpp.o | grep -v matmul_gemm | grep -v gemv_bench)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $flags -c cactus_amd/csrc/matmul_gemm.hip -o scripts/variants/gemm_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/libcactus_hip_$name.so $objs scripts/variants/gemm_$name.o
  echo built $name
done
