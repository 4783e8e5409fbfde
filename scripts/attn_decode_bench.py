#!/usr/bin/env python3
"""Decode attention over the INT8 KV cache: time and HBM rate per shape (HIP graph of 8 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
shapes = [("qwen3_c512_b1", 16, 8, 128, 512, 1), ("qwen3_c4096_b1", 16, 8, 128, 4096, 1), ("8b_c32768_b1", 32, 8, 128, 32768, 1),
          ("qwen3_c4096_b64", 16, 8, 128, 4096, 64), ("qwen3_c512_b256", 16, 8, 128, 512, 256),
          ("lfm2_c512_b1024", 32, 8, 64, 512, 1024), ("lfm2_c512_b128", 32, 8, 64, 512, 128), ("8b_c4096_b32", 32, 8, 128, 4096, 32),
          ("qwen3_c2048_b4", 16, 8, 128, 2048, 4), ("qwen3_c4096_b8", 16, 8, 128, 4096, 8), ("qwen3_c4096_b16", 16, 8, 128, 4096, 16),
          ("qwen3_c1024_b32", 16, 8, 128, 1024, 32), ("lfm2_c4096_b16", 32, 8, 64, 4096, 16),
          ("gemma270m_c512_b1024", 4, 1, 256, 512, 1024), ("gemma270m_c4096_b256", 4, 1, 256, 4096, 256),
          ("gemma270m_c4096_b1", 4, 1, 256, 4096, 1), ("gemma4b_c4096_b64", 8, 4, 256, 4096, 64)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] in sys.argv[1:]]
# ROTATE=n: n cache sets used in turn (more than the 256 MB Infinity Cache between two uses of a set), as the layers of
# a decode step do; default 1 = the same cache every launch
rot = int(os.environ.get("ROTATE", "1"))
for name, Hq, Hkv, D, cache, Bn in shapes:
    ng = (D + 127) // 128      # scale groups per row
    sets = []
    for _ in range(rot):
        kq = torch.randint(-127, 128, (Bn, cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
        vq = torch.randint(-127, 128, (Bn, cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
        ks = torch.rand((Bn, cache, Hkv, ng), device="cuda", generator=gen) * 0.02
        vs = torch.rand((Bn, cache, Hkv, ng), device="cuda", generator=gen) * 0.02
        sets.append((kq, vq, ks, vs))
    qh = (torch.rand((Bn, 1, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
    kn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    vn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
    lens = torch.full((Bn,), cache, dtype=torch.int32, device="cuda")
    out = torch.empty_like(qh)
    ms = time_graph(lambda i: H.attention_hybrid_batched(qh, *sets[i % rot], kn, vn, 0.0, lens, lens, out=out), max(8, 2 * rot))
    byts = Bn * (2 * cache * Hkv * (D + 4 * ng) + 2 * Hkv * D * 2 + 2 * Hq * D * 2)
    print(f"{name:20s} {ms * 1e3:9.2f} us  {byts / (ms * 1e-3) / 1e9:8.1f} GB/s  {byts / (ms * 1e-3) / 8e12:.3f} of 8 TB/s")
    del sets
    torch.cuda.empty_cache()
