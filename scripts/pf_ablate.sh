#!/bin/bash
# Times the causal prefill attention on the default library and on each timing variant
# (scripts/variants/libcactus_hip_<name>.so; their results are wrong by design).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "base: $(timeout -k 10 120 python scripts/pf_probe.py 4096 2>&1 | grep prefill)" | tee gpurun_out/pf_ablate.txt
for v in "$@"; do
  echo "$v: $(CACTUS_HIP_LIBRARY=scripts/variants/libcactus_hip_$v.so timeout -k 10 120 python scripts/pf_probe.py 4096 2>&1 | grep prefill)" | tee -a gpurun_out/pf_ablate.txt
done
