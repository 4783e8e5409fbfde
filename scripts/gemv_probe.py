#!/usr/bin/env python3
"""One kind of M = 1 matmul launched repeatedly on rotating weights (for rocprofv3 --pmc / --kernel-trace):
python3 scripts/gemv_probe.py int4|int8 K N [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights
H.init(0)
int4 = sys.argv[1] == "int4"
K, N = int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 1
gen = torch.Generator(device="cuda").manual_seed(7)
nb = max(2, min(64, int(600e6 // (N * K // (2 if int4 else 1))) + 1))
ws = [make_weights(K, N, int4, gen) for _ in range(nb)]
x = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
o = torch.empty((M, N), dtype=torch.float16, device="cuda")
for i in range(3 * nb):
    H.matmul_f16act(x, ws[i % nb][0], ws[i % nb][1], 128, int4=int4, out=o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(4 * nb):
    H.matmul_f16act(x, ws[i % nb][0], ws[i % nb][1], 128, int4=int4, out=o)
e1.record()
torch.cuda.synchronize()
print(f"{sys.argv[1]} M={M} K={K} N={N}: {e0.elapsed_time(e1) / (4 * nb) * 1e3:.2f} us per launch (stream of launches, no graph)")
