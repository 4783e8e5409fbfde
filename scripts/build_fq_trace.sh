#!/bin/bash
# Library variant with the phase stamps of the fused-quantization GEMV body: scripts/variants/libcactus_hip_fqtrace.so
set -e
cd "$(dirname "$0")/.."
python cactus_amd/build.py > /dev/null
mkdir -p scripts/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DFQ_TRACE -c cactus_amd/csrc/matmul_skinny.hip -o scripts/variants/skinny_fqtrace.o
objs=$(ls cactus_amd/build/*.hip.o cactus_amd/build/*.cpp.o | grep -v matmul_skinny | grep -v gemv_bench)
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/libcactus_hip_fqtrace.so $objs scripts/variants/skinny_fqtrace.o -ldl
echo built
