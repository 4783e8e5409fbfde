#!/usr/bin/env python3
"""Times the causal FP16 prefill attention (Qwen3-0.6B heads) at a few lengths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
lens = [int(v) for v in sys.argv[1:]] or [1024, 4096]
for S in lens:
    q = (torch.rand((1, S, 16, 128), device="cuda", generator=gen) * 2 - 1).half()
    k = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
    v = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
    o = torch.empty_like(q)
    ms = time_graph(lambda i: H.attention_f16(q, k, v, 0.0, is_causal=True, out=o), 4)
    fl = 4.0 * S * S / 2 * 16 * 128
    print(f"prefill attention S={S}: {ms * 1e3:8.1f} us  {fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s")
