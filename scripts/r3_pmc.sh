#!/bin/bash
# Round-3 counter passes (run on the GPU box; each pass is its own rocprofv3 run):
#   the headline batched GEMV on tile-major twins, INT8 and (probe) INT4: kernel stats + FETCH_SIZE / WRITE_SIZE
#   gemm3_kernel at M = 512, 8192^2: SQ issue / busy / wait counters
R="$(cd "$(dirname "$0")/.." && pwd)"
bash $R/scripts/r2_pmc.sh r3_bench skinny_batch_sub_kernel -- python3 $R/bench.py --no-extra --no-cpu --no-traffic > $R/gpurun_out/r3_pmc_bench.txt 2>&1 || exit 1
bash $R/scripts/r2_pmc.sh r3_gemm gemm3_kernel -- python3 $R/scripts/gemm_probe.py 512x8192x8192 > $R/gpurun_out/r3_pmc_gemm.txt 2>&1 || exit 1
bash $R/scripts/r2_pmc.sh r3_int4 skinny_batch_sub_kernel -- python3 $R/scripts/int4_sweep_probe.py > $R/gpurun_out/r3_pmc_int4.txt 2>&1 || exit 1
tail -4 $R/gpurun_out/r3_pmc_bench.txt
