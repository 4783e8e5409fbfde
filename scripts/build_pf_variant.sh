#!/bin/bash
# Experiment variants of the library that differ in -D flags of attention_prefill.hip:
#   scripts/build_pf_variant.sh name "-DPF_SPLIT_TARGET=512" [name2 "flags2" ...]
# -> scripts/variants/libcactus_hip_<name>.so ; run with CACTUS_HIP_LIBRARY=<path>.
cd "$(dirname "$0")/.."
mkdir -p scripts/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c cactus_amd/csrc/attention_prefill.hip -o scripts/variants/pf_$name.o || exit 1
  objs=$(ls cactus_amd/build/*.hip.o cactus_amd/build/*.cpp.o | grep -v attention_prefill)
  hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/libcactus_hip_$name.so $objs scripts/variants/pf_$name.o || exit 1
  echo built scripts/variants/libcactus_hip_$name.so
done
