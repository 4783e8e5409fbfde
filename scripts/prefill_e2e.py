#!/usr/bin/env python3
"""The chunked Qwen3-0.6B prefill measurement of bench.py on its own (for profiling):
python3 scripts/prefill_e2e.py [tokens] [chunk] [chunks in flight] [lfm2]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import qwen3_prefill

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
print(json.dumps(qwen3_prefill(H, gen, int(sys.argv[1]) if len(sys.argv) > 1 else 4096,
                               int(sys.argv[2]) if len(sys.argv) > 2 else 256,
                               lanes=int(sys.argv[3]) if len(sys.argv) > 3 else 1,
                               lfm2=len(sys.argv) > 4 and sys.argv[4] == "lfm2")))
