#!/usr/bin/env python3
"""The headline sweep with packed INT4 weights (tile-major twins), one batched launch per step, for the counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights, SIZES, gemv_bytes, time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(42)
NSETS = int(os.environ.get("SETS", "6"))     # 122 MB each: two would both sit in the 256 MB Infinity Cache
sets = [{(K, N): make_weights(K, N, True, gen, pack=True) for K in SIZES for N in SIZES} for _ in range(NSETS)]
acts = {K: (torch.rand((1, K), device="cuda", generator=gen) * 2 - 1).half() for K in SIZES}
outs = {(K, N): torch.empty((1, N), dtype=torch.float16, device="cuda") for K in SIZES for N in SIZES}
order = sorted(((K, N) for K in SIZES for N in SIZES), key=lambda kn: -kn[0] * kn[1])
batches = [H.MatmulBatch([(acts[K], ws[(K, N)][0], ws[(K, N)][1], outs[(K, N)]) for K, N in order], 128, int4=True) for ws in sets]
ms = time_graph(lambda i: batches[i % NSETS].run(), 2 * NSETS)
b = sum(gemv_bytes(1, K, N, True) for K in SIZES for N in SIZES)
print(f"int4 sweep: {ms*1e3:.2f} us per launch, {b} bytes, {b/(ms*1e-3)/1e9:.1f} GB/s = {b/(ms*1e-3)/8e12:.3f} of 8 TB/s")
