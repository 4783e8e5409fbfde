#!/bin/bash
# Library variant with the GEMM phase stamps: scripts/variants/libcactus_hip_gemmtrace.so
set -e
cd "$(dirname "$0")/.."
python cactus_amd/build.py > /dev/null
mkdir -p scripts/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DGEMM_TRACE -c cactus_amd/csrc/matmul_gemm.hip -o scripts/variants/gemm_trace.o
objs=$(ls cactus_amd/build/*.hip.o cactus_amd/build/*.cpp.o | grep -v matmul_gemm | grep -v gemv_bench)
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/libcactus_hip_gemmtrace.so $objs scripts/variants/gemm_trace.o
echo built
