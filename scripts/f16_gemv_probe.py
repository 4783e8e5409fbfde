import sys, torch
sys.path.insert(0, ".")
import cactus_amd.hip as H
from bench import time_graph
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
for (M, K, N) in [(1, 4096, 4096), (4, 4096, 4096), (1, 8192, 8192), (1, 1024, 151936), (1, 2048, 2048)]:
    nb = max(2, int(600e6 // (2 * N * K)) + 1)
    bs = [(torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).half() for _ in range(nb)]
    a = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
    o = torch.empty((M, N), dtype=torch.float16, device="cuda")
    for tag in ("row-major", "twins"):
        if tag == "twins":
            for b in bs: H.pack_weights(b)
        ms = time_graph(lambda i: H.matmul_f16(a, bs[i % nb], out=o), max(nb, 16))
        gb = (2.0 * N * K) / (ms * 1e-3) / 1e9
        print(f"M{M} K{K} N{N} {tag}: {ms*1e3:.2f} us {gb:.0f} GB/s {gb/8000:.3f}", flush=True)
    del bs
