#!/usr/bin/env python3
"""M = 1 fused-quantization GEMV, one launch per shape in a HIP graph of 8: time and HBM rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph, make_weights
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(7)
shapes = [(1024, 1024), (4096, 4096), (8192, 8192), (1024, 151936), (8192, 32768), (4096, 32768), (2048, 65536)]
for int4 in (False, True):
    for K, N in shapes:
        nb = max(2, min(16, int(600e6 // (N * K // (2 if int4 else 1))) + 1))
        ws = [make_weights(K, N, int4, gen) for _ in range(nb)]
        x = (torch.rand((1, K), device="cuda", generator=gen) * 2 - 1).half()
        o = torch.empty((1, N), dtype=torch.float16, device="cuda")
        ms = time_graph(lambda i: H.matmul_f16act(x, ws[i % nb][0], ws[i % nb][1], 128, int4=int4, out=o), 8)
        byts = N * K // (2 if int4 else 1) + 2 * N * K // 128 + 2 * K + 2 * N
        print(f"{'int4' if int4 else 'int8'} K{K} N{N}: {ms*1e3:8.2f} us {byts/(ms*1e-3)/1e9:8.1f} GB/s {byts/(ms*1e-3)/8e12:.3f}")
        del ws
