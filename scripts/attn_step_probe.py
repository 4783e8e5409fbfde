#!/usr/bin/env python3
"""Batch decode attention as the step runs it, on caches that no cache level holds (several cache sets in rotation):
the plain batched launch, the fused launch (norm + RoPE + append in front) and the fused launch that also leaves the
output rows quantized.  python3 scripts/attn_step_probe.py [batch] [cache] [sets]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cache = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sets = int(sys.argv[3]) if len(sys.argv) > 3 else 4
Hq, Hkv, D = 16, 8, 128
slots = cache + 8
S = []
for _ in range(sets):
    kq = torch.randint(-127, 128, (Bn, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vq = torch.randint(-127, 128, (Bn, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((Bn, slots, Hkv, 1), device="cuda", generator=gen) * 0.02
    vs = torch.rand((Bn, slots, Hkv, 1), device="cuda", generator=gen) * 0.02
    S.append((kq, vq, ks, vs))
qh = (torch.rand((Bn, 1, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
kn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
vn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
qw = (torch.rand((D,), device="cuda", generator=gen) + 0.5).half()
kw = (torch.rand((D,), device="cuda", generator=gen) + 0.5).half()
lens = torch.full((Bn,), cache, dtype=torch.int32, device="cuda")
out = torch.empty_like(qh)
oq = torch.empty((Bn, Hq * D), dtype=torch.int8, device="cuda")
osc = torch.empty((Bn,), dtype=torch.float32, device="cuda")
H.rope_reserve(D, 1e6, slots + 1) if hasattr(H, "rope_reserve") else None
byts = Bn * (2 * cache * Hkv * (D + 4) + 2 * Hkv * D * 2 + 2 * Hq * D * 2)
def show(name, ms):
    print(f"{name:28s} {ms * 1e3:8.2f} us  {byts / (ms * 1e-3) / 1e9:8.1f} GB/s  {byts / (ms * 1e-3) / 8e12:.3f} of 8 TB/s")
show("plain batched", time_graph(lambda i: H.attention_hybrid_batched(qh, *S[i % sets], kn, vn, 0.0, lens, lens, out=out), 8))
show("fused", time_graph(lambda i: H.attention_decode_fused(qh, kn, vn, qw, kw, *S[i % sets], lens, lens, slots, 1e-6, 1e6, out=out), 8))
show("fused + quantized rows", time_graph(lambda i: H.attention_decode_fused(qh, kn, vn, qw, kw, *S[i % sets], lens, lens, slots, 1e-6, 1e6, out=out, out_q=oq, out_scales=osc), 8))
