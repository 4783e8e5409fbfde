#!/usr/bin/env python3
"""Times causal FP16 attention under a cutting window (Gemma-3 local layer: 4 query heads, 1 kv head, head_dim 256,
window 512; and Qwen3-shaped heads with a window) against the same call without a window."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
for (S, Hq, Hkv, D, win) in [(2048, 4, 1, 256, 512), (4096, 4, 1, 256, 512), (4096, 16, 8, 128, 512), (4096, 16, 8, 64, 512)]:
    q = (torch.rand((1, S, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
    k = (torch.rand((1, S, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    v = (torch.rand((1, S, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    o = torch.empty_like(q)
    for w in (win, 0):
        ms = time_graph(lambda i: H.attention_f16(q, k, v, 0.0, window_size=w, is_causal=True, out=o), 4)
        pairs = sum(min(i + 1, w + 1) if w else i + 1 for i in range(S))
        fl = 4.0 * pairs * Hq * D
        print(f"S={S} Hq={Hq} Hkv={Hkv} D={D} window={w}: {ms * 1e3:8.1f} us  {fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
