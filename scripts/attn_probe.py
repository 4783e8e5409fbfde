#!/usr/bin/env python3
"""Times decode attention over the INT8 KV cache (graph-captured launches, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
cases = [("qwen3_c512", 16, 8, 128, 512, 1), ("qwen3_c4096", 16, 8, 128, 4096, 1), ("8b_c32768", 32, 8, 128, 32768, 1),
         ("qwen3_c4096_b64", 16, 8, 128, 4096, 64), ("qwen3_c512_b256", 16, 8, 128, 512, 256),
         ("gemma_c2048", 4, 1, 256, 2048, 1), ("lfm2_c4096_b16", 32, 8, 64, 4096, 16)]
for name, Hq, Hkv, D, cache, Bn in cases:
    kq = torch.randint(-127, 128, (Bn, cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vq = torch.randint(-127, 128, (Bn, cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    G = (D + 127) // 128
    ks = torch.rand((Bn, cache, Hkv, G), device="cuda", generator=gen) * 0.02
    vs = torch.rand((Bn, cache, Hkv, G), device="cuda", generator=gen) * 0.02
    qh = (torch.rand((Bn, 1, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
    kn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    vn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    lens = torch.full((Bn,), cache, dtype=torch.int32, device="cuda")
    ms = time_graph(lambda i: H.attention_hybrid_batched(qh, kq, vq, ks, vs, kn, vn, 0.0, lens, lens), 8)
    byts = Bn * (2 * cache * Hkv * (D + 4 * G) + 2 * Hkv * D * 2 + 2 * Hq * D * 2)
    print(f"{name:18s} {byts / 1e6:8.1f} MB {ms * 1e3:8.2f} us {byts / ms / 1e6:8.1f} GB/s")
