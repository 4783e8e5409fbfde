#!/bin/bash
# Builds experiment variants of the library that differ in -D flags of matmul_skinny.hip:
#   scripts/build_variants.sh name "-DSK_WPE=6 -DSK_BATCH_P=2" [name2 "flags2" ...]
# -> scripts/variants/libcactus_hip_<name>.so ; run with CACTUS_HIP_LIBRARY=<path>.
set -e
cd "$(dirname "$0")/.."
python cactus_amd/build.py > /dev/null
mkdir -p scripts/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $flags -c cactus_amd/csrc/matmul_skinny.hip -o scripts/variants/skinny_$name.o
  objs=$(ls cactus_amd/build/*.hip.o cactus_amd/build/*.cpp.o | grep -v matmul_skinny | grep -v gemv_bench)
  hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/libcactus_hip_$name.so $objs scripts/variants/skinny_$name.o
  echo built scripts/variants/libcactus_hip_$name.so
done
