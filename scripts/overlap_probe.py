#!/usr/bin/env python3
"""Do a KV-streaming decode attention launch and a chain of small GEMMs overlap when a HIP graph holds them on two
branches?  Times, per graph replay: the attention alone (batch B sequences), the GEMM chain alone, both on two
streams.  python3 scripts/overlap_probe.py [batch] [cache]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
from bench import make_weights
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cache = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Hq, Hkv, D = 16, 8, 128
slots = cache + 8
sets = []
for _ in range(6):
    kq = torch.randint(-127, 128, (Bn, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vq = torch.randint(-127, 128, (Bn, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((Bn, slots, Hkv, 1), device="cuda", generator=gen) * 0.02
    vs = torch.rand((Bn, slots, Hkv, 1), device="cuda", generator=gen) * 0.02
    sets.append((kq, vq, ks, vs))
qh = (torch.rand((Bn, 1, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
kn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
vn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
qw = (torch.rand((D,), device="cuda", generator=gen) + 0.5).half()
kw = (torch.rand((D,), device="cuda", generator=gen) + 0.5).half()
lens = torch.full((Bn,), cache, dtype=torch.int32, device="cuda")
out = torch.empty_like(qh)
oq = torch.empty((Bn, Hq * D), dtype=torch.int8, device="cuda")
osc = torch.empty((Bn,), dtype=torch.float32, device="cuda")
H.rope_reserve(D, 1e6, slots + 1)
# the GEMM chain of a block at M = Bn
shapes = [(1024, 4096), (2048, 1024), (1024, 6144), (3072, 1024)]
W = [make_weights(K, N, False, gen) for K, N in shapes]
X = [H.quantize_activations((torch.rand((Bn, K), device="cuda", generator=gen) * 2 - 1).half()) for K, N in shapes]
O = [torch.empty((Bn, N), dtype=torch.float16, device="cuda") for K, N in shapes]

def attn(i):
    H.attention_decode_fused(qh, kn, vn, qw, kw, *sets[i % 6], lens, lens, slots, 1e-6, 1e6, out=out, out_q=oq, out_scales=osc)

def chain():
    for (q, s), (B, Bs), o in zip(X, W, O):
        H.matmul_int8(q, s, B, Bs, 128, out=o)

def timed(build, reps=20):
    build(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

N = 6
def only_attn():
    for i in range(N): attn(i)
def only_chain():
    for i in range(N): chain()
side = torch.cuda.Stream()
def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    for i in range(N): attn(i)
    with torch.cuda.stream(side):
        for i in range(N): chain()
    main.wait_stream(side)
a, c, b = timed(only_attn), timed(only_chain), timed(both)
print(f"batch {Bn} cache {cache}: {N} attention launches {a:.1f} us, {N} chains of 4 GEMMs {c:.1f} us, both on two branches {b:.1f} us (sum {a + c:.1f})")
