#!/usr/bin/env python3
"""Pre-quantized activations, few rows: the streaming kernel (M <= 16) against the tile kernel (M = 17) on rotating
weight buffers (> 600 MB in all, so that weights come from HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights, time_graph

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
shapes = [(1024, 1024), (2048, 2048), (4096, 4096), (8192, 8192), (2048, 8192), (8192, 2048), (1024, 151936), (4096, 12288)]
Ms = [int(v) for v in sys.argv[1:]] or [8, 16, 17, 32]
for int4 in (False, True):
    for K, N in shapes:
        nb = min(64, max(2, int(600e6 // max(1, (N * K // (2 if int4 else 1)))) + 1))
        ws = [make_weights(K, N, int4, gen, pack=os.environ.get('PACK', '1') == '1') for _ in range(nb)]
        row = []
        for M in Ms:
            x = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
            q, s = H.quantize_activations(x)
            o = torch.empty((M, N), dtype=torch.float16, device="cuda")
            fn = H.matmul_int4 if int4 else H.matmul_int8
            ms = time_graph(lambda i: fn(q, s, ws[i % nb][0], ws[i % nb][1], 128, out=o), max(nb, 16))
            row.append(f"M={M}: {ms * 1e3:6.1f} us")
        print(f"{'int4' if int4 else 'int8'} K={K:5d} N={N:6d}  " + "  ".join(row))
        del ws
