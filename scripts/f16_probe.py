#!/usr/bin/env python3
"""Times cactus_hip_matmul_f16_dev (FP16 x FP16^T, fp32 accumulate) on a few shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
shapes = [(1, 4096, 4096), (16, 4096, 4096), (512, 4096, 4096), (512, 8192, 8192), (4096, 1024, 4096), (256, 1024, 151936)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for M, K, N in shapes:
    a = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
    b = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).half()
    o = torch.empty((M, N), dtype=torch.float16, device="cuda")
    ms = time_graph(lambda i: H.matmul_f16(a, b, out=o), 8)
    fl = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    gb = (M * K + N * K + M * N) * 2 / (ms * 1e-3) / 1e9
    print(f"f16 M={M} K={K} N={N}: {ms * 1e3:8.1f} us  {fl:7.1f} TFLOP/s  {gb:7.1f} GB/s")
