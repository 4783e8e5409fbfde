#!/usr/bin/env python3
"""The Qwen3-0.6B decode step at batch 1 (and 4) only, as bench.py measures it."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import qwen3_decode_step
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
for b in (1, 4):
    r = qwen3_decode_step(H, gen, b, 512)
    print(b, r["ms_per_step"])
