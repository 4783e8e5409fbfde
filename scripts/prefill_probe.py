#!/usr/bin/env python3
"""Times FP16 (prefill) attention and hybrid prefill chunks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_events
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
for S, Hq, Hkv, D in ((256, 16, 8, 128), (1024, 16, 8, 128), (4096, 16, 8, 128), (512, 4, 1, 256), (4096, 4, 1, 256), (4096, 8, 4, 256)):
    q = (torch.rand((1, S, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
    k = (torch.rand((1, S, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    v = (torch.rand((1, S, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    ms = time_events(lambda i: H.attention_f16(q, k, v, 0.0), 5, warm=2)
    fl = 4.0 * S * S / 2 * Hq * D
    print(f"f16 causal prefill S={S} Hq={Hq} D={D}: {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.2f} TFLOP/s (causal half counted)")
# a 256-token chunk against an INT8 cache (the engine's chunked prefill after chunk 0)
for cache, Hq, Hkv, D in ((256, 16, 8, 128), (2048, 16, 8, 128), (3840, 16, 8, 128), (3840, 4, 1, 256), (3840, 8, 4, 256)):
    S = 256
    q = (torch.rand((1, S, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
    kn = (torch.rand((1, S, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    vn = (torch.rand((1, S, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    kc = torch.randint(-127, 128, (cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vc = torch.randint(-127, 128, (cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((cache, Hkv, (D + 127) // 128), device="cuda", generator=gen) * 0.02
    vs = torch.rand((cache, Hkv, (D + 127) // 128), device="cuda", generator=gen) * 0.02
    ms = time_events(lambda i: H.attention_hybrid(q, kc, vc, ks, vs, kn, vn, 0.0, position_offset=cache), 5, warm=2)
    fl = 4.0 * S * (cache + S / 2) * Hq * D
    print(f"hybrid chunk S=256 cache={cache} Hq={Hq} D={D}: {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.2f} TFLOP/s")
