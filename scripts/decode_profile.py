#!/usr/bin/env python3
"""One batch size of the Qwen3-0.6B decode-step measurement, for rocprofv3 --kernel-trace --stats:
python3 scripts/decode_profile.py <batch> [cache_len] [mixed]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import qwen3_decode_step

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
print(json.dumps(qwen3_decode_step(H, gen, int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 512,
                                   mixed=len(sys.argv) > 3 and sys.argv[3] == "mixed")))
