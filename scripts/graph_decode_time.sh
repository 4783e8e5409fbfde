#!/bin/bash
# Batch-1 decode through the graph surface from the C++ host (examples/qwen_graph_decode.cpp): 512-token prompt, 40 tokens;
# prints the program's JSON line (ms per token and its build / execute / update / reset parts).  Run on the GPU box.
R="$(cd "$(dirname "$0")/.." && pwd)"
python3 $R/scripts/write_qwen3_folder.py /tmp/qwen3_folder > /dev/null 2>&1 || exit 1
prompt=$(python3 -c "import numpy as np; print(' '.join(str(t) for t in np.random.default_rng(3).integers(0, 151936, 512).tolist()))")
for i in 1 2; do $R/cactus_amd/lib/qwen_graph_decode /tmp/qwen3_folder 42 1 $prompt | tail -1 | cut -c1-40,400-700; done
rm -rf /tmp/qwen3_folder
