#!/usr/bin/env python3
"""A few launches of FP16 causal prefill attention at one size (for rocprofv3 passes):
python3 scripts/prefill_one.py [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
H.init(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gen = torch.Generator(device="cuda").manual_seed(5)
q = (torch.rand((1, S, 16, 128), device="cuda", generator=gen) * 2 - 1).half()
k = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
v = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
for _ in range(5):
    H.attention_f16(q, k, v, 0.0)
torch.cuda.synchronize()
