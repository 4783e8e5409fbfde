#!/usr/bin/env python3
"""Times the M > 16 quantized GEMM (HIP events over graph-captured launches) and checks one
shape against the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cactus_amd.hip as H
from bench import make_weights, time_graph

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
shapes = [(512, 4096, 4096), (512, 8192, 8192), (256, 1024, 3072), (2048, 4096, 4096), (4096, 4096, 12288)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for int4 in (False, True):
    for M, K, N in shapes:
        B, Bs = make_weights(K, N, int4, gen)
        x = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
        q, s = H.quantize_activations(x)
        o = torch.empty((M, N), dtype=torch.float16, device="cuda")
        fn = H.matmul_int4 if int4 else H.matmul_int8
        ms = time_graph(lambda i: fn(q, s, B, Bs, 128, out=o), 8)
        tops = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        print(f"{'int4' if int4 else 'int8'} M={M} K={K} N={N}: {ms * 1e3:8.1f} us  {tops:7.1f} TOPS  {tops / 50:.1f}% of 5 POPS")
