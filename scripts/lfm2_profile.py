#!/usr/bin/env python3
"""One batch size of the LFM2-1.2B-shape decode-step measurement, for rocprofv3 --kernel-trace --stats:
python3 scripts/lfm2_profile.py <batch> [cache_len]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import lfm2_decode_step

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(5)
print(json.dumps(lfm2_decode_step(H, gen, int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 512)))
