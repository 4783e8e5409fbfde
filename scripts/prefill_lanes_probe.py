#!/usr/bin/env python3
"""A 4096-token prefill of the Qwen3-0.6B op sequence through cactus_hip_prefill_dev as ONE captured graph, by chunk
size and by the number of chunks in flight: python3 scripts/prefill_lanes_probe.py [tokens]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from cactus_amd import steps as ST
H.init(0)
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = dict(ST.QWEN3_0P6B)
gen = torch.Generator(device="cuda").manual_seed(3)
w = ST.qwen3_synthetic_weights(gen, cfg)
for chunk in (256, 512, 1024):
    row = {}
    for lanes in (1, 2, 4, 6, 8):
        pf = ST.Qwen3Prefill(w, cfg, tokens, chunk)
        pf.toks.copy_(torch.randint(0, cfg["vocab"], pf.toks.shape, dtype=torch.int32, device="cuda", generator=gen))
        pf.run(lanes); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pf.run(lanes)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        row[lanes] = round(e0.elapsed_time(e1) / 3, 2)
        del pf, g
        torch.cuda.empty_cache()
    print(f"chunk {chunk}: ms by chunks in flight {json.dumps(row)}", flush=True)
