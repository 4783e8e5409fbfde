set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "f16act or fused or batch" --timeout 120 --timeout-method thread > gpurun_out/ct_base_tests.log 2>&1
CACTUS_HIP_LIBRARY=scripts/variants/libcactus_hip_ct2.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ct2_tests.log 2>&1
timeout -k 10 120 python bench.py --no-extra --no-cpu > gpurun_out/ct_base.json 2>gpurun_out/ct_base.err
for v in ct2 ct2p2 ct2w2 ct2w8 ct2p3; do
  CACTUS_HIP_LIBRARY=scripts/variants/libcactus_hip_$v.so timeout -k 10 120 python bench.py --no-extra --no-cpu > gpurun_out/ct_$v.json 2>gpurun_out/ct_$v.err
done
tail -2 gpurun_out/ct_base_tests.log gpurun_out/ct2_tests.log
for f in gpurun_out/ct_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
