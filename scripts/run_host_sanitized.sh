#!/bin/bash
# Runs the host-sanitized C++ examples (scripts/build_host_sanitized.sh) on the GPU box against a small Qwen3-shaped
# model folder: graph decode (one graph per token, then a chunked prompt), the batched step driver, prompt + decode, and the host-pointer family.
# Output: gpurun_out/san/*.log ; any sanitizer report makes the run fail.  (The sanitized examples leave
# with _Exit after flushing: the ROCm ASan runtime can trip over its own device allocator while the HIP runtime unloads,
# after main has returned -- "CHECK failed: sanitizer_allocator_device.h ... dev_runtime_unloaded_".)
R="$(cd "$(dirname "$0")/.." && pwd)"
out=$R/gpurun_out/san; rm -rf $out; mkdir -p $out
S=$R/scripts/variants/san
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python3 $R/scripts/write_qwen3_folder.py /tmp/qwen3_small small > $out/write.log 2>&1 || exit 1
prompt=$(python3 -c "import numpy as np; print(' '.join(str(t) for t in np.random.default_rng(3).integers(0, 2000, 300).tolist()))")
timeout -k 10 120 $S/qwen_graph_decode /tmp/qwen3_small 10 1 $prompt > $out/graph_fused.log 2>&1 || { echo graph_fused failed; tail -30 $out/graph_fused.log; exit 1; }
timeout -k 10 120 $S/qwen_graph_decode /tmp/qwen3_small 10 0 $prompt > $out/graph_plain.log 2>&1 || { echo graph_plain failed; tail -30 $out/graph_plain.log; exit 1; }
CACTUS_PREFILL_CHUNK=64 timeout -k 10 120 $S/qwen_graph_decode /tmp/qwen3_small 6 1 $prompt > $out/graph_chunked.log 2>&1 || { echo graph_chunked failed; tail -30 $out/graph_chunked.log; exit 1; }
timeout -k 10 120 $S/qwen_batched_decode /tmp/qwen3_small 8 6 1 2 3 4 5 6 7 8 > $out/batched.log 2>&1 || { echo batched failed; tail -30 $out/batched.log; exit 1; }
timeout -k 10 120 $S/qwen_prompt_decode /tmp/qwen3_small 6 64 2 $prompt > $out/prompt.log 2>&1 || { echo prompt failed; tail -30 $out/prompt.log; exit 1; }
timeout -k 10 120 $S/host_pointer_calls 3 > $out/host_pointers.log 2>&1 || { echo host_pointer_calls failed; tail -30 $out/host_pointers.log; exit 1; }
rm -rf /tmp/qwen3_small
grep -l "ERROR: AddressSanitizer\|runtime error" $out/*.log && exit 1
echo "sanitized host runs clean"; for f in $out/graph_fused.log $out/batched.log $out/prompt.log $out/host_pointers.log; do tail -1 $f | cut -c1-160; done
