#!/bin/bash
# Builds experiment variants of the library that differ in -D flags of attention_decode.hip:
#   scripts/build_attn_variant.sh name "-DDC_WPE128=3" [name2 "flags2" ...]
# -> scripts/variants/libcactus_hip_<name>.so ; run with CACTUS_HIP_LIBRARY=<path>.
set -e
cd "$(dirname "$0")/.."
python cactus_amd/build.py > /dev/null
mkdir -p scripts/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c cactus_amd/csrc/attention_decode.hip -o scripts/variants/attn_$name.o
  objs=$(ls cactus_amd/build/*.hip.o cactus_amd/build/*.cpp.o | grep -v attention_decode | grep -v gemv_bench)
  hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/libcactus_hip_$name.so $objs scripts/variants/attn_$name.o
  echo built scripts/variants/libcactus_hip_$name.so
done
