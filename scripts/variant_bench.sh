#!/bin/bash
# Runs bench.py --no-extra --no-cpu on the default library and on each named variant library
# (scripts/variants/libcactus_hip_<name>.so), printing value / ms_per_step / roofline.frac.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])"; }
timeout -k 10 120 python bench.py --no-extra --no-cpu > gpurun_out/vb_base.json 2>gpurun_out/vb_base.err || exit 1
show gpurun_out/vb_base.json base
for v in "$@"; do
  CACTUS_HIP_LIBRARY=scripts/variants/libcactus_hip_$v.so timeout -k 10 120 python bench.py --no-extra --no-cpu > gpurun_out/vb_$v.json 2>gpurun_out/vb_$v.err || exit 1
  show gpurun_out/vb_$v.json $v
done
