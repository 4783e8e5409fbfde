#!/bin/bash
# rocprofv3 kernel trace of batch-1 decode through the graph surface from the C++ host (examples/qwen_graph_decode.cpp):
# 512-token prompt, then 40 tokens.  Run on the GPU box.  Output: gpurun_out/graphdec/ (kernel_stats.csv, per-token
# kernel time and gaps from the trace).
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/graphdec; rm -rf $out; mkdir -p $out
python3 $R/scripts/write_qwen3_folder.py /tmp/qwen3_folder > $out/write.log 2>&1 || exit 1
prompt=$(python3 -c "import numpy as np; print(' '.join(str(t) for t in np.random.default_rng(3).integers(0, 151936, 512).tolist()))")
$R/cactus_amd/lib/qwen_graph_decode /tmp/qwen3_folder 42 1 $prompt > $out/plain.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/run -- $R/cactus_amd/lib/qwen_graph_decode /tmp/qwen3_folder 42 1 $prompt > $out/run.log 2>&1 || exit 1
cp $(ls $out/run/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 - "$(ls $out/run/*/*kernel_trace.csv | head -1)" > $out/gaps.txt <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
# the decode tokens are the tail of the trace: the last 30 sampler launches bound 29 whole tokens
idx = [i for i, r in enumerate(rows) if "argmax" in r[2] or "sample" in r[2]]
idx = idx[-30:]
tok = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a + 1:b + 1]
    busy = sum(e - s for s, e, _ in seg)
    span = seg[-1][1] - rows[a][1]
    tok.append((span, busy, len(seg)))
n = len(tok)
print("tokens %d: span %.1f us, kernel time %.1f us, launches %.1f" % (n, sum(t[0] for t in tok) / n / 1e3, sum(t[1] for t in tok) / n / 1e3, sum(t[2] for t in tok) / n))
PY
tail -1 $out/plain.log | cut -c1-300
cat $out/gaps.txt
head -16 $out/kernel_stats.csv | cut -c1-200
rm -rf /tmp/qwen3_folder
