#!/usr/bin/env python3
"""Times the batched FP16-activation GEMV launch on subsets of the M=1 sweep (HIP events, two
alternating weight sets so nothing is cache resident)."""
import itertools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights, gemv_bytes, time_events, SIZES

H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
sets = [{(K, N): make_weights(K, N, False, gen) for K in SIZES for N in SIZES} for _ in range(3)]
acts = {K: (torch.rand((1, K), device="cuda", generator=gen) * 2 - 1).half() for K in SIZES}
outs = {(K, N): torch.empty((1, N), dtype=torch.float16, device="cuda") for K in SIZES for N in SIZES}


def probe(name, shapes):
    batches = [H.MatmulBatch([(acts[K], ws[(K, N)][0], ws[(K, N)][1], outs[(K, N)]) for K, N in shapes], 128) for ws in sets]
    ms = time_events(lambda i: batches[i % 3].run(), 60, warm=6)
    b = sum(gemv_bytes(1, K, N) for K, N in shapes)
    print(f"{name:28s} {len(shapes):2d} problems {b / 1e6:7.1f} MB  {ms * 1e3:7.2f} us  {b / ms / 1e6:7.1f} GB/s")


for K in SIZES:
    probe(f"K={K} all N", [(K, N) for N in SIZES])
allshapes = sorted(itertools.product(SIZES, SIZES), key=lambda kn: -kn[0] * kn[1])
probe("all, big first", allshapes)
probe("all, small first", allshapes[::-1])
probe("all, by K desc", [(K, N) for K in SIZES[::-1] for N in SIZES[::-1]])
probe("K>=4096", [(K, N) for K in (8192, 4096) for N in SIZES])
probe("K<=2048", [(K, N) for K in (2048, 1024) for N in SIZES])
probe("8192x8192 only", [(8192, 8192)])
probe("8192x8192 x1 + small", [(8192, 8192)] + [(1024, N) for N in SIZES])
