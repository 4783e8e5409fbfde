#!/bin/bash
# Times the M = 512 INT8 GEMM on the default library and on each timing variant
# (scripts/variants/libcactus_hip_<name>.so; their results are wrong by design).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
shape=${SHAPE:-512x8192x8192}
echo "base: $(timeout -k 10 120 python scripts/gemm_probe.py $shape 2>&1 | grep int8)" | tee gpurun_out/gemm_ablate.txt
for v in "$@"; do
  echo "$v: $(CACTUS_HIP_LIBRARY=scripts/variants/libcactus_hip_$v.so timeout -k 10 120 python scripts/gemm_probe.py $shape 2>&1 | grep int8)" | tee -a gpurun_out/gemm_ablate.txt
done
