#!/usr/bin/env python3
"""A few launches of one M > 16 quantized GEMM, for counter passes: gemm_one.py MxKxN int8|int4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
M, K, N = (int(v) for v in sys.argv[1].split("x"))
int4 = len(sys.argv) > 2 and sys.argv[2] == "int4"
B, Bs = make_weights(K, N, int4, gen)
x = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
q, s = H.quantize_activations(x)
o = torch.empty((M, N), dtype=torch.float16, device="cuda")
for _ in range(6):
    (H.matmul_int4 if int4 else H.matmul_int8)(q, s, B, Bs, 128, out=o)
torch.cuda.synchronize()
