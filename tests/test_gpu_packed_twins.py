"""Tile-major twins of resident weights (cactus_hip_weights_pack_dev): the streaming matmuls (M <= 16) read a
second copy of the weight in which each 16-row x 128-byte unit is one contiguous 2 KB run.  Only addresses change:
every result must equal the CPU oracle's (kernel_gemm.cpp:128-378 restated) bit for bit, exactly as on the
row-major matrix, through every entry point that can take the twin, for ragged N, several rows, INT4, and
launches that mix packed and unpacked problems."""
import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import rand_weights_int8, rand_weights_int4, rand_act_f16

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def H():
    import cactus_amd.hip as hip
    hip.init(0)
    return hip


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


def oracle_f16act(x, B, Bs, int4):
    q, s = O.quantize_activations(x)
    return (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)


def acts(rng, M, K):
    x = rand_act_f16(rng, M, K)
    if M > 1:
        x[1] = 0          # all-zero row: scale floor
    if M > 2:
        x[2] *= 30.0      # rows with very different scales
    return x


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N", [(1, 1024, 1024), (1, 8192, 512), (2, 2048, 100), (3, 4096, 48), (4, 8192, 256 + 3),
                                   (4, 1024, 3072), (1, 256, 16), (1, 768, 33), (1, 12288, 64), (1, 128, 1),
                                   (1, 1024, 151936), (3, 1024, 32768 + 40), (2, 2048, 33000), (1, 4096, 32768)])
def test_f16act_on_a_twin_bit_exact(H, M, K, N, int4):
    """cactus_hip_matmul_int{8,4}_f16act_dev, every kernel variant it chooses among (waves per tile by row length,
    one wave per tile for thousands of short rows), reading the twin; then the same call after the release."""
    if int4 and K % 256:
        pytest.skip("packed units hold two groups")
    rng = np.random.default_rng(M * 31 + K + N + int4)
    x = acts(rng, M, K)
    B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
    want = oracle_f16act(x, B, Bs, int4)
    dB, dBs, dx = dev(B), dev(Bs), dev(x)
    assert not H.is_packed(dB)
    H.pack_weights(dB)
    assert H.is_packed(dB)
    got = bits(H.matmul_f16act(dx, dB, dBs, 128, int4=int4))
    assert (got == want).all()
    H.release_packed(dB)
    assert not H.is_packed(dB)
    assert (bits(H.matmul_f16act(dx, dB, dBs, 128, int4=int4)) == want).all()


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M", [1, 5, 9, 16])
def test_prequantized_rows_on_a_twin_bit_exact(H, M, int4):
    """cactus_hip_matmul_int{8,4}_dev with up to 16 pre-quantized rows and the INT32 group sums, on the twin."""
    rng = np.random.default_rng(900 + M + int4)
    for (K, N) in [(1024, 64), (4096, 1000), (2048, 16 * 7 + 1)]:
        x = acts(rng, M, K)
        B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
        q, s = O.quantize_activations(x)
        want, want_sums = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128, want_group_sums=True)
        dB = H.pack_weights(dev(B))
        f = H.matmul_int4 if int4 else H.matmul_int8
        got = bits(f(dev(q), dev(s), dB, dev(Bs), 128))
        assert (got == want).all(), (K, N)
        sums = H.matmul_group_sums(dev(q), dB, 128, int4=int4).cpu().numpy()
        assert (sums == want_sums).all(), (K, N)


@pytest.mark.parametrize("int4", [False, True])
def test_batched_launch_mixing_twins_and_row_major_bit_exact(H, int4):
    """One batched launch (cactus_hip_matmul_int{8,4}_f16act_batch_dev) whose problems alternate between weights
    with a twin and weights without: the layout is a per-problem flag inside the launch."""
    rng = np.random.default_rng(31 + int4)
    shapes = [(1, 1024, 1024), (1, 1024, 2048), (1, 1024, 40), (2, 2048, 512), (4, 4096, 128), (1, 8192, 272),
              (1, 512, 16), (3, 1024, 1000), (1, 2048, 8192), (1, 256, 7)] + [(1, 1024, 64)] * 6
    mk = rand_weights_int4 if int4 else rand_weights_int8
    entries, wants = [], []
    for i, (M, K, N) in enumerate(shapes):
        x = acts(rng, M, K)
        B, Bs = mk(rng, N, K)
        wants.append(oracle_f16act(x, B, Bs, int4))
        dB = dev(B)
        if i % 2 == 0:
            H.pack_weights(dB)
        entries.append((dev(x), dB, dev(Bs), torch.zeros((M, N), dtype=torch.float16, device="cuda")))
    outs = H.MatmulBatch(entries, 128, int4=int4).run()
    for o, w, shp in zip(outs, wants, shapes):
        assert (bits(o) == w).all(), shp
    # the wave-per-tile launch (thousands of short rows), all packed
    entries, wants = [], []
    for (M, K, N) in [(1, 1024, 20000), (2, 1024, 17000 + 8), (4, 1024, 16384)]:
        x = acts(rng, M, K)
        B, Bs = mk(rng, N, K)
        wants.append(oracle_f16act(x, B, Bs, int4))
        entries.append((dev(x), H.pack_weights(dev(B)), dev(Bs), torch.zeros((M, N), dtype=torch.float16, device="cuda")))
    outs = H.MatmulBatch(entries, 128, int4=int4).run()
    for o, w in zip(outs, wants):
        assert (bits(o) == w).all()


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M", [1, 3])
def test_produced_rows_on_twins_bit_exact(H, M, int4):
    """The launches that derive their activation rows in the prologue (cactus_hip_matmul_int{8,4}_norm / _silu /
    _gelu_f16act_dev), weights packed: the oracle's norm / activation followed by its matmul."""
    rng = np.random.default_rng(64 + M + int4)
    mk = rand_weights_int4 if int4 else rand_weights_int8
    K = 1024
    x, add = acts(rng, M, K), rand_act_f16(rng, M, K)
    w = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float16)
    summed = O.binary("add", x.view(np.uint16), add.view(np.uint16))
    rows = O.rms_norm(summed, w, 1e-6)
    ws = [mk(rng, N, K) for N in (2048, 1024, 1000)]
    outs = [torch.zeros((M, B.shape[0]), dtype=torch.float16, device="cuda") for B, _ in ws]
    mp = H.MatmulProduced([(H.pack_weights(dev(B)), dev(Bs), o) for (B, Bs), o in zip(ws, outs)], 128, int4=int4)
    out_sum = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    mp.run_norm(dev(x), dev(add), dev(w), 1e-6, out_sum)
    assert (bits(out_sum) == summed).all()
    for (B, Bs), o in zip(ws, outs):
        assert (bits(o) == oracle_f16act(rows.view(np.float16), B, Bs, int4)).all()
    for act in ("silu", "gelu"):
        K2 = 3072
        g, u = rand_act_f16(rng, M, K2, -3, 3), rand_act_f16(rng, M, K2)
        prod = O.binary("multiply", (O.silu if act == "silu" else O.gelu)(g.view(np.uint16)), u.view(np.uint16))
        B, Bs = mk(rng, 1024, K2)
        o = torch.zeros((M, 1024), dtype=torch.float16, device="cuda")
        H.MatmulProduced([(H.pack_weights(dev(B)), dev(Bs), o)], 128, int4=int4).run_act(dev(g), dev(u), act)
        assert (bits(o) == oracle_f16act(prod.view(np.float16), B, Bs, int4)).all(), act


def test_twin_follows_a_repack_and_ignores_other_shapes(H):
    """The twin is a copy: after B is rewritten in place a second pack call refreshes it; a call that names B with
    another N (a row slice starting at the same address) does not match the twin and reads B itself."""
    rng = np.random.default_rng(4)
    K, N = 2048, 512
    x = acts(rng, 1, K)
    B, Bs = rand_weights_int8(rng, N, K)
    B2, _ = rand_weights_int8(rng, N, K)
    dB, dBs, dx = dev(B), dev(Bs), dev(x)
    H.pack_weights(dB)
    assert (bits(H.matmul_f16act(dx, dB, dBs)) == oracle_f16act(x, B, Bs, False)).all()
    dB.copy_(dev(B2))
    torch.cuda.synchronize()
    H.pack_weights(dB)
    assert (bits(H.matmul_f16act(dx, dB, dBs)) == oracle_f16act(x, B2, Bs, False)).all()
    half = dB[:N // 2]                       # same address, half the rows: no twin of that shape
    assert (bits(H.matmul_f16act(dx, half, dBs[:N // 2].contiguous())) == oracle_f16act(x, B2[:N // 2], Bs[:N // 2], False)).all()


@pytest.mark.parametrize("M,K,N", [(1, 4096, 4096), (1, 64, 1), (3, 1024, 1000), (16, 1024, 333), (2, 8192, 48), (1, 1024, 151936)])
def test_f16_gemv_streams_and_its_twin_gives_the_same_bits(H, M, K, N):
    """cactus_hip_matmul_f16_dev with up to 16 rows takes the streaming kernel (fp32 accumulation on the matrix
    cores; the reference accumulates in fp16 lanes, kernel_gemm.cpp:17-126): within the reference's own distance
    from the exact product plus one ulp, and the same bits from the tile-major twin of the FP16 weight."""
    rng = np.random.default_rng(M + K + N)
    a = rng.uniform(-1, 1, (M, K)).astype(np.float16)
    b = rng.uniform(-1, 1, (N, K)).astype(np.float16)
    da, db = dev(a), dev(b)
    got = H.matmul_f16(da, db)
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    err = np.abs(got.cpu().numpy().astype(np.float64) - exact)
    if N <= 4096:
        ref = O.matmul_f16(a, b).view(np.float16).astype(np.float64)
        ref_err = np.abs(ref - exact)
        ulp = np.maximum(np.abs(exact), 2.0 ** -14) * 2.0 ** -10
        assert (err <= ref_err.max() + ulp).all()
    assert (err <= 2e-3 + 1e-3 * np.abs(exact)).all()            # fp32 accumulation, one rounding to fp16
    H.pack_weights(db)
    assert H.is_packed(db)
    again = H.matmul_f16(da, db)
    assert torch.equal(got.view(torch.int16), again.view(torch.int16))
    H.release_packed(db)


def test_pack_argument_errors(H):
    import ctypes as C
    from cactus_amd.hip import lib
    t = torch.zeros((16, 192), dtype=torch.int8, device="cuda")
    assert lib.cactus_hip_weights_pack_dev(C.c_void_p(t.data_ptr()), 16, 192, 0, None) == -1      # K % 128
    assert b"multiple of 128" in lib.cactus_hip_last_error()
    t4 = torch.zeros((16, 64), dtype=torch.uint8, device="cuda")
    assert lib.cactus_hip_weights_pack_dev(C.c_void_p(t4.data_ptr()), 16, 128, 1, None) == -1     # INT4: K % 256
    assert lib.cactus_hip_weights_pack_dev(None, 16, 128, 0, None) == -1
    ok = torch.zeros((16, 256), dtype=torch.int8, device="cuda")
    assert lib.cactus_hip_weights_pack_dev(C.c_void_p(ok.data_ptr()), 16, 256, 0, None) == 0
    assert lib.cactus_hip_weights_pack_dev(C.c_void_p(ok.data_ptr()), 8, 256, 0, None) == -1      # other shape, same address
    assert b"release it first" in lib.cactus_hip_last_error()
    assert lib.cactus_hip_weights_release_packed_dev(C.c_void_p(ok.data_ptr())) == 0
    assert lib.cactus_hip_weights_release_packed_dev(C.c_void_p(ok.data_ptr())) == 0              # nothing to drop: fine
    lib.cactus_hip_clear_error()


def test_full_size_sweep_shapes_twin_equals_row_major(H):
    """The headline shapes at full size (K, N up to 8192; the 151936 x 1024 vocabulary projection): the launch on the
    twin equals the launch on the row-major matrix bit for bit (which the other tests tie to the oracle at sizes it
    finishes in seconds), and a checksum of the outputs against a float64 restatement of the same arithmetic."""
    gen = torch.Generator(device="cuda").manual_seed(7)
    for int4 in (False, True):
        for (K, N) in [(8192, 8192), (1024, 8192), (8192, 1024), (4096, 4096), (1024, 151936)]:
            if int4:
                B = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device="cuda", generator=gen)
            else:
                B = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=gen)
            Bs = (0.01 + torch.rand((N, K // 128), device="cuda", generator=gen) * 0.05).half()
            x = (torch.rand((2, K), device="cuda", generator=gen) * 2 - 1).half()
            ref = H.matmul_f16act(x, B, Bs, 128, int4=int4).clone()
            H.pack_weights(B)
            got = H.matmul_f16act(x, B, Bs, 128, int4=int4)
            assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (K, N, int4)
            H.release_packed(B)
