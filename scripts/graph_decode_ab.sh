#!/bin/bash
# A/B of library builds on batch-1 decode through the graph surface from the C++ host: scripts/graph_decode_ab.sh dirA dirB ...
# (each directory holds a libcactus_hip.so; the host program is cactus_amd/lib/qwen_graph_decode).  Run on the GPU box.
R="$(cd "$(dirname "$0")/.." && pwd)"
python3 $R/scripts/write_qwen3_folder.py /tmp/qwen3_folder > /dev/null 2>&1 || exit 1
prompt=$(python3 -c "import numpy as np; print(' '.join(str(t) for t in np.random.default_rng(3).integers(0, 151936, 512).tolist()))")
for rep in 1 2; do
  for d in "$@"; do
    echo -n "$d: "
    LD_LIBRARY_PATH=$R/$d:$LD_LIBRARY_PATH $R/cactus_amd/lib/qwen_graph_decode /tmp/qwen3_folder 42 1 $prompt | tail -1 | grep -o '"ms_per_token": [0-9.]*, "build_ms": [0-9.]*, "execute_ms": [0-9.]*, "cache_update_ms": [0-9.]*, "soft_reset_ms": [0-9.]*'
  done
done
rm -rf /tmp/qwen3_folder
