#!/usr/bin/env python3
"""Phase cycle totals of the pipelined GEMM (library built with -DGEMM_TRACE: scripts/build_gemm_trace.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
M, K, N = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512x8192x8192").split("x"))
B, Bs = make_weights(K, N, False, gen)
x = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
q, s = H.quantize_activations(x)
for _ in range(3):
    H.matmul_int8(q, s, B, Bs, 128)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
H.lib.cactus_hip_debug_gemm_trace.argtypes = [C.c_void_p]
H.lib.cactus_hip_debug_gemm_trace(buf)
G = K // 128
for wg in (0, 3):
    for wv in range(8):
        slot = wg * 8 + wv
        cyc, ticks, pa, bar, pb = (buf[slot * 8 + i] for i in range(5))
        print(f"wg {wg} wave {wv}: loop {cyc / G:.0f} cycles/step, {ticks * 10.0 / G:.0f} ns/step (clock {cyc / max(ticks, 1) * 100:.0f} MHz); "
              f"first slab {pa / G:.0f}, barrier {bar / G:.0f}, second slab {pb / G:.0f}")
