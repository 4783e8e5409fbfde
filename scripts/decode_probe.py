#!/usr/bin/env python3
"""Runs only the Qwen3-0.6B decode-step measurement of bench.py (batch 1 and 256, cache 512)
and times the block ops on their own (HIP-graph captured, event timed)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import qwen3_decode_step, time_graph

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
for b in (1, 256):
    print(json.dumps(qwen3_decode_step(H, gen, b, 512)))
half = lambda *s: (torch.rand(s, device="cuda", generator=gen) * 2 - 1).half()
rows = {}
for name, shape in (("rms_norm 1x1024", (1, 1024)), ("rms_norm 4096x1024", (4096, 1024)), ("rms_norm 65536x128", (65536, 128))):
    x, w, o = half(*shape), half(shape[1]), half(*shape)
    ms = time_graph(lambda i: H.rms_norm(x, w, 1e-6, out=o), 8)
    rows[name] = (ms * 1e3, 4 * x.numel() / ms / 1e6)
for name, shape in (("rope 1x1x16x128", (1, 1, 16, 128)), ("rope 1x4096x16x128", (1, 4096, 16, 128))):
    x, o = half(*shape), half(*shape)
    H.rope_reserve(128, 1e6, 8192)
    ms = time_graph(lambda i: H.rope(x, 1e6, 0, out=o), 8)
    rows[name] = (ms * 1e3, 4 * x.numel() / ms / 1e6)
for n in (3072, 4096 * 3072):
    x, y, o = half(n), half(n), half(n)
    rows[f"silu {n}"] = (time_graph(lambda i: H.silu(x, out=o), 8) * 1e3,)
    rows[f"silu {n}"] += (4 * n / rows[f"silu {n}"][0] / 1e3,)
    rows[f"multiply {n}"] = (time_graph(lambda i: H.multiply(x, y, out=o), 8) * 1e3,)
    rows[f"multiply {n}"] += (6 * n / rows[f"multiply {n}"][0] / 1e3,)
for k, v in rows.items():
    print(f"{k:28s} {v[0]:9.2f} us  {v[1]:9.1f} GB/s")
