#!/usr/bin/env python3
"""Times the stochastic sampler (engine defaults: temperature 0.6, top-p 0.95, top-k 20) at the Qwen3 vocabulary."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
for batch in (1, 256):
    for top_k in (20, 100):
        lg = (torch.randn((batch, 151936), device="cuda", generator=gen) * 3).half()
        dr = torch.rand((batch,), device="cuda", generator=gen)
        smp = H.Sampler(batch, 151936)
        tok = torch.empty((batch,), dtype=torch.int32, device="cuda")
        smp.sample(lg, 0.6, 0.95, top_k, dr, out=tok)
        ms = time_graph(lambda i: smp.sample(lg, 0.6, 0.95, top_k, dr, out=tok, check=False), 4)
        print(f"sample rows={batch} top_k={top_k}: {ms * 1e3:7.1f} us")
