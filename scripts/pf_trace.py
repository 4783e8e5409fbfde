#!/usr/bin/env python3
"""Phase cycle totals of the two-group prefill attention (library built with -DPF_TRACE:
scripts/build_pf_timing.sh pftrace "-DPF_TRACE"; run with CACTUS_HIP_LIBRARY pointing at it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = (torch.rand((1, S, 16, 128), device="cuda", generator=gen) * 2 - 1).half()
k = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
v = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
for _ in range(3):
    H.attention_f16(q, k, v, 0.0, is_causal=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
H.lib.cactus_hip_debug_pf_trace.argtypes = [C.c_void_p]
H.lib.cactus_hip_debug_pf_trace(buf)
for wg in (0, 1):
    for wv in range(8):
        slot = wg * 8 + wv
        cyc, ticks, mf, b1, st, sm, b2, n = (buf[slot * 8 + i] for i in range(8))
        n = max(n, 1)
        print(f"wg {wg} wave {wv}: {n} iterations, {cyc / n:.0f} cycles each, {ticks * 10.0 / n:.0f} ns (clock {cyc / max(ticks, 1) * 100:.0f} MHz); "
              f"MFMA phase {mf / n:.0f}, barrier {b1 / n:.0f}, staging {st / n:.0f}, softmax {sm / n:.0f}, barrier {b2 / n:.0f}")
