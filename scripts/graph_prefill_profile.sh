#!/bin/bash
# rocprofv3 kernel stats of a 4096-token prompt as one chunk through the graph surface from the C++ host
# (examples/qwen_graph_decode.cpp).  Run on the GPU box.  Output: gpurun_out/graphpf/
R="$(cd "$(dirname "$0")/.." && pwd)"
N=${1:-4096}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/graphpf; rm -rf $out; mkdir -p $out
python3 $R/scripts/write_qwen3_folder.py /tmp/qwen3_folder > $out/write.log 2>&1 || exit 1
prompt=$(python3 -c "import numpy as np; print(' '.join(str(t) for t in np.random.default_rng(4).integers(0, 151936, $N).tolist()))")
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/run -- $R/cactus_amd/lib/qwen_graph_decode /tmp/qwen3_folder 2 1 $prompt > $out/run.log 2>&1 || exit 1
cp $(ls $out/run/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
tail -1 $out/run.log | cut -c1-400
head -25 $out/kernel_stats.csv | cut -c1-180
rm -rf /tmp/qwen3_folder
