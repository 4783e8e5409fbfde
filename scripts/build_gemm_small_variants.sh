#!/bin/bash
# Library variants of the small-tile GEMM loop: ring depth and stage-ahead fragment reads switched separately.
set -e
cd "$(dirname "$0")/.."
python cactus_amd/build.py > /dev/null
mkdir -p scripts/variants
objs=$(ls cactus_amd/build/*.hip.o cactus_amd/build/*.cpp.o | grep -v matmul_gemm | grep -v gemv_bench)
build() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $2 -c cactus_amd/csrc/matmul_gemm.hip -o scripts/variants/gemm_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/libcactus_hip_gemm_$1.so $objs scripts/variants/gemm_$1.o -ldl
}
build ring4_ahead "-DG3_RING_DEEP=0 -DG3_STAGE_AHEAD=1" &
build deep_plain "-DG3_RING_DEEP=1 -DG3_STAGE_AHEAD=0" &
build ring4_plain "-DG3_RING_DEEP=0 -DG3_STAGE_AHEAD=0" &
wait
echo built
