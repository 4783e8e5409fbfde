#!/usr/bin/env python3
"""4096-token Qwen3-0.6B prefill at several chunk sizes (Model::prefill's chunk_size argument, engine_model.cpp:170)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
import bench
H.init(0)
for chunk in (256, 512, 1024, 2048, 4096):
    gen = torch.Generator(device="cuda").manual_seed(1)
    r = bench.qwen3_prefill(H, gen, 4096, chunk)
    print(chunk, r["ms"], r["tokens_per_s"], flush=True)
    torch.cuda.empty_cache()
