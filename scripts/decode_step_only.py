#!/usr/bin/env python3
"""Only the Qwen3-0.6B decode-step rows of bench.py (batch 1 and 256, cache 512) and the LFM2 one (batch 1024)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import qwen3_decode_step

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
for b in [int(a) for a in sys.argv[1:]] or [1, 256]:
    r = qwen3_decode_step(H, gen, b, 512)
    print(b, r["ms_per_step"])
