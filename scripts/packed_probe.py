"""Tile-major twins: bit-equality with the row-major launches and timing of the headline sweep (one batched launch)
and of single launches, HIP events over graphs of rotating weight sets."""
import sys, json, torch
sys.path.insert(0, ".")
import cactus_amd.hip as H
H.init(0)
gen = torch.Generator(device="cuda"); gen.manual_seed(42)

def mat(N, K, int4=False):
    if int4:
        B = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device="cuda", generator=gen)
    else:
        B = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=gen)
    Bs = (0.01 + torch.rand((N, K // 128), device="cuda", generator=gen) * 0.05).half()
    return B, Bs

def timed(fn, iters=20, reps=5):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(iters): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

res = {}
# parity: assorted shapes incl. ragged N, M 1..4, INT4, and M up to 16 through the pre-quantized path
for (M, K, N, i4) in [(1, 1024, 1024, False), (3, 2048, 1000, False), (4, 8192, 4096 + 5, False), (2, 1024, 151936, False),
                      (1, 4096, 4096, True), (4, 2048, 333, True), (1, 1024, 151936, True), (1, 8192, 8192, True),
                      (16, 4096, 1024, False), (9, 2048, 77, True)]:
    B, Bs = mat(N, K, i4)
    X = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
    if M <= 4:
        ref = H.matmul_f16act(X, B, Bs, int4=i4).clone()
        H.pack_weights(B)
        got = H.matmul_f16act(X, B, Bs, int4=i4)
    else:
        q, s = H.quantize_activations(X)
        f = H.matmul_int4 if i4 else H.matmul_int8
        ref = f(q, s, B, Bs).clone()
        H.pack_weights(B)
        got = f(q, s, B, Bs)
    torch.cuda.synchronize()
    same = bool(torch.equal(ref.view(torch.int16), got.view(torch.int16)))
    print("parity", M, K, N, "int4" if i4 else "int8", "SAME" if same else "DIFFERENT", flush=True)
    assert same
    H.release_packed(B)

KN = [1024, 2048, 4096, 8192]
for i4 in (False, True):
    sets = []
    for s_ in range(2):
        ws = [(K, N) + mat(N, K, i4) for K in KN for N in KN]
        sets.append(ws)
    X = {K: (torch.rand((1, K), device="cuda", generator=gen) * 2 - 1).half() for K in KN}
    outs = {N: torch.empty((1, N), dtype=torch.float16, device="cuda") for N in KN}
    alg = sum(N * K // (2 if i4 else 1) + 2 * N * K // 128 + 2 * K + 2 * N for K in KN for N in KN)
    for packed in (False, True):
        if packed:
            for ws in sets:
                for (_, _, B, _) in ws: H.pack_weights(B)
        batches = [H.MatmulBatch([(X[K], B, Bs, outs[N]) for (K, N, B, Bs) in ws], 128, int4=i4) for ws in sets]
        us = timed(lambda i=0: batches[i % 2].run())
        print(f"sweep {'int4' if i4 else 'int8'} packed={packed}: {us:.2f} us  {alg/us/1e6:.3f} TB/s  {alg/us/8e6:.3f}", flush=True)
        res[f"sweep_{'int4' if i4 else 'int8'}_{'packed' if packed else 'rowmajor'}"] = us
        # single launches
        for (K, N) in [(8192, 8192), (4096, 4096), (1024, 1024)]:
            sel = [[(B, Bs) for (k, n, B, Bs) in ws if (k, n) == (K, N)][0] for ws in sets]
            a1 = N * K // (2 if i4 else 1) + 2 * N * K // 128 + 2 * K + 2 * N
            us = timed(lambda i=0: H.matmul_f16act(X[K], sel[i % 2][0], sel[i % 2][1], int4=i4, out=outs[N]))
            print(f"  single {K}x{N} packed={packed}: {us:.2f} us {a1/us/8e6:.3f}", flush=True)
        # separate launches of the 16
        def sep(i=0):
            for (K, N, B, Bs) in sets[i % 2]:
                H.matmul_f16act(X[K], B, Bs, int4=i4, out=outs[N])
        us = timed(sep, iters=4)
        print(f"  16 separate launches packed={packed}: {us:.2f} us", flush=True)
    H.release_packed(None)
    del sets, batches
# lm_head
for i4 in (False, True):
    ws = [mat(151936, 1024, i4) for _ in range(3)]
    X = (torch.rand((1, 1024), device="cuda", generator=gen) * 2 - 1).half()
    out = torch.empty((1, 151936), dtype=torch.float16, device="cuda")
    a1 = 151936 * 1024 // (2 if i4 else 1) + 2 * 151936 * 8 + 2048 + 2 * 151936
    for packed in (False, True):
        if packed:
            for B, _ in ws: H.pack_weights(B)
        us = timed(lambda i=0: H.matmul_f16act(X, ws[i % 3][0], ws[i % 3][1], int4=i4, out=out))
        print(f"lm_head {'int4' if i4 else 'int8'} packed={packed}: {us:.2f} us {a1/us/8e6:.3f}", flush=True)
    H.release_packed(None)
print(json.dumps(res))
