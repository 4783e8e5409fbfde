#!/usr/bin/env python3
"""Times a 256-token prefill chunk's attention against INT8 caches of a few lengths (Qwen3-0.6B heads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
S = int(os.environ.get("CHUNK", "256"))
for cache in [int(v) for v in sys.argv[1:]] or [256, 2048, 3840]:
    Hq_, Hkv_, D_ = 16, 8, 128
    q = (torch.rand((1, S, Hq_, D_), device="cuda", generator=gen) * 2 - 1).half()
    kn = (torch.rand((1, S, Hkv_, D_), device="cuda", generator=gen) * 2 - 1).half()
    vn = (torch.rand((1, S, Hkv_, D_), device="cuda", generator=gen) * 2 - 1).half()
    kc = torch.randint(-127, 128, (cache, Hkv_, D_), dtype=torch.int8, device="cuda", generator=gen)
    vc = torch.randint(-127, 128, (cache, Hkv_, D_), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((cache, Hkv_, 1), device="cuda", generator=gen) * 0.02
    vs = torch.rand((cache, Hkv_, 1), device="cuda", generator=gen) * 0.02
    H.attention_hybrid(q, kc, vc, ks, vs, kn, vn, 0.0, position_offset=cache)
    ms = time_graph(lambda i: H.attention_hybrid(q, kc, vc, ks, vs, kn, vn, 0.0, position_offset=cache), 4)
    fl = 4.0 * S * (cache + S / 2) * Hq_ * D_
    print(f"chunk {S} over cache {cache}: {ms * 1e3:7.1f} us  {fl / (ms * 1e-3) / 1e12:6.1f} TFLOP/s")
