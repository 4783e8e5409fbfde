#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the library and the C++ host examples (device code is untouched:
# -Xarch_host puts the sanitizers on the host compilation only).  Output: scripts/variants/san/{libcactus_hip.so,
# qwen_graph_decode, qwen_batched_decode, qwen_prompt_decode, host_pointer_calls}.  Run the examples on the GPU box as usual.
set -e
cd "$(dirname "$0")/.."
out=scripts/variants/san; mkdir -p $out
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -g"
objs=""
for s in quant_kernels.hip matmul_skinny.hip matmul_gemm.hip matmul_generic.hip matmul_f16.hip attention.hip attention_decode.hip attention_prefill.hip block_ops.hip conv_ops.hip sampler.hip comm.hip api.hip graph.cpp steps.cpp; do
  extra=""; lang=""
  [ $s = matmul_gemm.hip ] && extra="-fno-slp-vectorize"
  [ $s = attention_decode.hip -o $s = matmul_skinny.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  case $s in *.cpp) lang="-x hip";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $SAN $extra $lang -c cactus_amd/csrc/$s -o $out/$s.o &
  objs="$objs $out/$s.o"
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fsanitize=undefined -o $out/libcactus_hip.so $objs -ldl
for e in qwen_graph_decode qwen_batched_decode qwen_prompt_decode host_pointer_calls; do
  hipcc -O1 -g -std=c++17 -DEXIT_WITHOUT_TEARDOWN -fsanitize=address -fsanitize=undefined -I include examples/$e.cpp -L $out -lcactus_hip '-Wl,-rpath,$ORIGIN' -o $out/$e
done
ls -la $out | grep -v "\.o$"
