#!/bin/bash
# Builds scripts/gemv_bench (timing harness) against the kernel objects of cactus_amd/build.
set -e
cd "$(dirname "$0")/.."
python cactus_amd/build.py > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c scripts/gemv_bench.hip -o cactus_amd/build/gemv_bench.o
hipcc --offload-arch=gfx950 -o scripts/gemv_bench cactus_amd/build/gemv_bench.o cactus_amd/build/matmul_skinny.hip.o
echo built scripts/gemv_bench
