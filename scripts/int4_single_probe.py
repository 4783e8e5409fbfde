#!/usr/bin/env python3
"""Single-launch M = 1 matmuls on packed INT4 (and INT8) weights with tile-major twins, graph-timed on rotating buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights, gemv_bytes, time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(11)
for int4 in (True, False):
    for K, N in ((8192, 8192), (4096, 4096), (1024, 151936), (2048, 8192)):
        nb = max(2, min(16, int(600e6 // (N * K // (2 if int4 else 1))) + 1))
        ws = [make_weights(K, N, int4, gen, pack=True) for _ in range(nb)]
        x = (torch.rand((1, K), device="cuda", generator=gen) * 2 - 1).half()
        o = torch.empty((1, N), dtype=torch.float16, device="cuda")
        ms = time_graph(lambda i: H.matmul_f16act(x, ws[i % nb][0], ws[i % nb][1], 128, int4=int4, out=o), 2 * nb)
        b = gemv_bytes(1, K, N, int4)
        print(f"{'int4' if int4 else 'int8'} K={K} N={N}: {ms * 1e3:.2f} us  {b / (ms * 1e-3) / 8e12:.3f} of 8 TB/s")
        del ws
        torch.cuda.empty_cache()
