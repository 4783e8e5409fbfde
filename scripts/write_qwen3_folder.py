#!/usr/bin/env python3
"""Writes a synthetic Qwen3-0.6B-shape model folder in the reference's layout:  python3 scripts/write_qwen3_folder.py <dir> [small]
(small: 4 blocks, a vocabulary of 2048 -- for runs under a sanitizer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from cactus_amd import steps as ST, cact
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
cfg = dict(ST.QWEN3_0P6B, layers=4, vocab=2048) if len(sys.argv) > 2 and sys.argv[2] == "small" else ST.QWEN3_0P6B
w = ST.qwen3_synthetic_weights(gen, cfg)
host = {"layers": [{k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v.cpu().numpy()) for k, v in L.items()} for L in w["layers"]],
        "emb": tuple(t.cpu().numpy() for t in w["emb"]), "final_norm": w["final_norm"].cpu().numpy()}
os.makedirs(sys.argv[1], exist_ok=True)
cact.write_qwen_folder(sys.argv[1], host, cfg)
print("written", sys.argv[1])
