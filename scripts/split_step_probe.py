#!/usr/bin/env python3
"""Whole decode steps with and without the two-half-batch split (cactus_hip_decode_step_split):
python3 scripts/split_step_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from cactus_amd import steps as ST
from bench import qwen3_decode_step, lfm2_decode_step
H.init(0)
for name, fn, args in (("qwen3 b256", qwen3_decode_step, (256, 512)), ("qwen3 b128", qwen3_decode_step, (128, 512)),
                       ("qwen3 b64", qwen3_decode_step, (64, 512)), ("qwen3 b256 c2048", qwen3_decode_step, (256, 2048)),
                       ("lfm2 b1024", lfm2_decode_step, (1024, 512)), ("lfm2 b128", lfm2_decode_step, (128, 512))):
    row = {}
    for split in (0, 2):
        ST.decode_step_split(split)
        gen = torch.Generator(device="cuda").manual_seed(7)
        row["split" if split else "single"] = fn(H, gen, *args)["ms_per_step"]
        torch.cuda.empty_cache()
    print(name, json.dumps(row), flush=True)
