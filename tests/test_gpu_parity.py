"""GPU parity tests: the HIP kernels, called through the C ABI (cactus_amd.hip -> libcactus_hip.so),
against the CPU oracle on the same seeded inputs.

Bars: bit-exact for integer / byte / index work (quantized values, scales, INT32 group sums, INT4
unpack) and -- because the kernels keep the reference's fp32 operation order -- bit-exact fp16
outputs for the quantized matmuls too.  Attention and the FP16 matmul are floating-point kernels
with a different summation order and are held to the tolerances written in each test.
"""
import os

import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import (parse_cact, f16, rand_weights_int8, rand_weights_int4, rand_act_f16,
                           ulp_diff_f16)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def H():
    import cactus_amd.hip as hip
    hip.init(0)
    return hip


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def bits(t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


# ------------------------------------------------- FP16 activations, quantized in the GEMV
@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N", [(1, 1024, 1024), (1, 8192, 512), (2, 2048, 100), (3, 4096, 48), (4, 8192, 256),
                                   (4, 1024, 3072), (1, 256, 16), (1, 768, 33), (5, 1024, 64), (16, 2048, 128),
                                   (40, 512, 64), (1, 12288, 64)])
def test_matmul_f16_activations_fused_quant_bit_exact(H, M, K, N, int4):
    """graph_ops_nn.cpp:166-177 / :196-207: quantize rows, then the grouped matmul.  The fused
    kernel (M <= 4) and the two-launch fallback both equal the oracle doing the two steps."""
    if int4 and K % 256:
        pytest.skip("packed units hold two groups")
    rng = np.random.default_rng(M * 31 + K + N + int4)
    x = rand_act_f16(rng, M, K)
    x[0, :8] = np.array([0.25, -0.25, 0.125, 0.375, -0.375, 0.5, -0.5, 1.0], np.float16)
    if M > 1:
        x[1] = 0          # all-zero row: scale floor
    if M > 2:
        x[2] *= 30.0      # rows with very different scales
    B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
    q, s = O.quantize_activations(x)
    want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
    got = bits(H.matmul_f16act(dev(x), dev(B), dev(Bs), 128, int4=int4))
    assert (got == want).all()


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N", [(1, 1024, 151936), (3, 1024, 32768 + 40), (4, 1024, 40000), (2, 2048, 33000),
                                   (1, 4096, 32768)])
def test_matmul_f16_activations_many_short_rows_bit_exact(H, M, K, N, int4):
    """2048 tiles or more of short rows: one wave per 16-row tile with the fp32 chain in registers (the
    vocabulary projection's launch; INT8 up to K = 1024, INT4 up to 4096 - the other shapes here take the
    split-row kernel at the same sizes).  Ragged N, several rows, rows of very different scales."""
    rng = np.random.default_rng(M * 31 + K + N + int4)
    x = rand_act_f16(rng, M, K)
    if M > 1:
        x[1] = 0
    if M > 2:
        x[2] *= 30.0
    B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
    q, s = O.quantize_activations(x)
    want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
    got = bits(H.matmul_f16act(dev(x), dev(B), dev(Bs), 128, int4=int4))
    assert (got == want).all()


@pytest.mark.parametrize("int4", [False, True])
def test_matmul_f16_activations_batch_of_short_rows_bit_exact(H, int4):
    """The batched launch over the wave-per-tile kernel: three vocabulary-sized problems with their own rows."""
    rng = np.random.default_rng(5 + int4)
    mk = rand_weights_int4 if int4 else rand_weights_int8
    entries, wants = [], []
    for (M, K, N) in [(1, 1024, 20000), (2, 1024, 17000 + 8), (4, 1024, 16384)]:
        x = rand_act_f16(rng, M, K)
        B, Bs = mk(rng, N, K)
        q, s_ = O.quantize_activations(x)
        wants.append((O.matmul_int4 if int4 else O.matmul_int8)(q, s_, B, Bs, 128))
        entries.append((dev(x), dev(B), dev(Bs), torch.zeros((M, N), dtype=torch.float16, device="cuda")))
    outs = H.MatmulBatch(entries, 128, int4=int4).run()
    for o, w in zip(outs, wants):
        assert (bits(o) == w).all()


@pytest.mark.parametrize("int4", [False, True])
def test_matmul_f16_activations_batch_equals_single_calls(H, int4):
    """One batched launch of mixed shapes (shared and distinct activation rows, ragged N, one
    entry that is not fusable and runs on its own) == one call per entry == the oracle."""
    rng = np.random.default_rng(77 + int4)
    shapes = [(1, 1024, 1024), (1, 1024, 2048), (1, 1024, 40), (2, 2048, 512), (4, 4096, 128), (1, 8192, 272),
              (1, 512, 16), (3, 1024, 1000), (1, 2048, 8192), (1, 256, 7), (16, 1024, 64)] + [(1, 1024, 64)] * 8
    mk = rand_weights_int4 if int4 else rand_weights_int8
    xs = {}
    entries, wants = [], []
    for (M, K, N) in shapes:
        if (M, K) not in xs:
            xs[(M, K)] = rand_act_f16(rng, M, K)
        x = xs[(M, K)]
        B, Bs = mk(rng, N, K)
        q, s_ = O.quantize_activations(x)
        wants.append((O.matmul_int4 if int4 else O.matmul_int8)(q, s_, B, Bs, 128))
        entries.append((dev(x), dev(B), dev(Bs), torch.zeros((M, N), dtype=torch.float16, device="cuda")))
    outs = H.MatmulBatch(entries, 128, int4=int4).run()
    for o, w, shp in zip(outs, wants, shapes):
        assert (bits(o) == w).all(), shp


# ------------------------------------------------------------------ quantization
@pytest.mark.parametrize("M,K", [(1, 1024), (3, 3072), (16, 4096), (5, 264), (2, 100), (1, 12288), (40, 128)])
def test_quantize_activations_bit_exact(H, M, K):
    rng = np.random.default_rng(M * 1000 + K)
    x = rand_act_f16(rng, M, K)
    if M > 1:
        x[1] = 0                       # all-zero row: scale floor 1e-10
    x[0, :8] = np.array([0.25, -0.25, 0.125, 0.375, -0.375, 0.5, -0.5, 1.0], np.float16)  # ties
    q, s = H.quantize_activations(dev(x))
    oq, os_ = O.quantize_activations(x)
    assert (host(s).view(np.uint32) == os_.view(np.uint32)).all()
    assert (host(q) == oq).all()


def test_quantize_activations_ties_to_even(H):
    # scale = 2^-9 exactly (absmax 127*2^-9): x*inv lands on .5 ties
    K = 256
    x = (np.arange(K, dtype=np.float32) - 128) * 2.0 ** -10
    x[0] = 127 * 2.0 ** -9
    x = x.astype(np.float16).reshape(1, K)
    q, s = H.quantize_activations(dev(x))
    oq, os_ = O.quantize_activations(x)
    assert float(os_[0]) == 2.0 ** -9
    assert (host(q) == oq).all() and (host(s) == os_).all()


@pytest.mark.parametrize("count", [8, 12, 1000, 1003])
def test_fp16_to_int8_and_max_abs(H, count):
    rng = np.random.default_rng(count)
    x = rng.uniform(-3, 3, count).astype(np.float16)
    x[:4] = np.array([0.5, 1.5, 2.5, -2.5], np.float16)
    x[-4:] = np.array([0.5, 1.5, 2.5, -2.5], np.float16)   # tail elements round half away
    got = host(H.fp16_to_int8(dev(x), 1.0))
    assert (got == O.fp16_to_int8(x, 1.0)).all()
    got = host(H.fp16_to_int8(dev(x), 0.02))
    assert (got == O.fp16_to_int8(x, 0.02)).all()
    assert float(host(H.fp16_max_abs(dev(x)))[0]) == O.fp16_max_abs(x)


@pytest.mark.parametrize("S,Hk,D,gs", [(1, 8, 128, 128), (256, 8, 128, 128), (7, 1, 256, 128), (5, 8, 64, 128),
                                       (3, 2, 72, 128), (3, 2, 132, 128), (4, 2, 128, 64), (2, 3, 20, 128)])
def test_quantize_kv_bit_exact(H, S, Hk, D, gs):
    rng = np.random.default_rng(S * 100 + D)
    x = rng.uniform(-2, 2, (S, Hk, D)).astype(np.float16)
    x[0, 0, :] = 0
    q, s = H.quantize_kv(dev(x), gs)
    oq, os_ = O.quantize_kv(x, gs)
    assert (host(s).view(np.uint32) == os_.view(np.uint32)).all()
    assert (host(q) == oq).all()


@pytest.mark.parametrize("count", [32, 64, 1000, 4096, 33, 95, 511])
def test_unpack_int4_bit_exact(H, count):
    rng = np.random.default_rng(count)
    packed = rng.integers(0, 256, (count + 1) // 2, dtype=np.uint8)
    got = host(H.unpack_int4(dev(packed), count))
    assert (got == O.unpack_int4(packed, count)).all()


# ------------------------------------------------------------------ quantized matmul
SHAPES = [
    # M, K, N
    (1, 1024, 1024), (1, 3072, 1000), (2, 4096, 48), (5, 256, 13), (16, 1024, 2048), (8, 8192, 64),
    (1, 5120, 16), (3, 128, 7),                                      # skinny path (M <= 16)
    (17, 1024, 256), (64, 512, 130), (200, 1024, 384), (256, 3072, 128),   # LDS/MFMA tile path
    (33, 256, 16), (128, 256, 128), (129, 768, 129), (300, 4096, 272),     # one / two / three K steps, ragged tiles
    (40, 12288, 48),                                                        # 96 groups: largest scale panel in LDS
    (24, 25600, 32),                                                        # 200 groups: falls back to the plain tile kernel
    (512, 384, 8192), (520, 256, 8200),                                     # 256 or more 128 x 128 tiles
    (512, 256, 2048), (1000, 384, 2100),                                    # smaller tiles while a launch has fewer workgroups than CUs
    # the tile kernel's four-stage ring (loader waves run three tiles ahead): K steps around its depth, and long runs
    (48, 640, 256), (100, 896, 640), (70, 1408, 400), (256, 640, 4224), (256, 768, 4224), (256, 1152, 4224),
    (256, 2048, 4224), (200, 8192, 256),
    # a single K step (no steady state at all) at each tile height: 32-, 64- and 128-row tiles
    (40, 128, 300), (512, 128, 2048), (512, 128, 8192),
    # 12 to 16 rows against a wide matrix: the tile kernel with 64-column tiles (INT8; INT4 stays on the streaming kernel)
    (16, 1024, 12288 + 16), (12, 256, 12300), (11, 256, 12300),
    # the tile is chosen by workgroup count: 32 x 128 needs few rows against a very wide matrix
    (24, 640, 32768), (20, 2048, 32768 + 64),
]


def _check_matmul(H, M, K, N, gs, int4, seed):
    rng = np.random.default_rng(seed)
    B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K, gs)
    A16 = rand_act_f16(rng, M, K)
    A, As = O.quantize_activations(A16)
    ofn = O.matmul_int4 if int4 else O.matmul_int8
    oc, og = ofn(A, As, B, Bs, gs, True)
    dA, dAs, dB, dBs = dev(A), dev(As), dev(B), dev(Bs)
    g = host(H.matmul_group_sums(dA, dB, gs, int4=int4))
    assert (g == og).all(), "INT32 group sums differ"
    c = bits((H.matmul_int4 if int4 else H.matmul_int8)(dA, dAs, dB, dBs, gs))
    assert (c == oc).all(), f"fp16 outputs differ, max ulp {ulp_diff_f16(c, oc).max()}"
    # against the non-contracted evaluation of the same statements: FP16 tolerance
    O.set_fma(0)
    oc2 = ofn(A, As, B, Bs, gs)
    O.set_fma(1)
    ref = f16(oc2)
    assert np.abs(f16(c) - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())
    # one fp16 ulp except where the group terms cancel (a result far below the row's magnitude
    # keeps the absolute fp32 rounding of the larger partial sums): then the absolute bound above
    d = ulp_diff_f16(c, oc2)
    assert (d <= 1).mean() >= 0.999
    if M * N <= (1 << 17):      # with hundreds of thousands of outputs some cancellation exceeds any fixed ulp count
        assert d.max() <= 8


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_matmul_int8_bit_exact(H, M, K, N):
    _check_matmul(H, M, K, N, 128, False, M + K + N)


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_matmul_int4_bit_exact(H, M, K, N):
    _check_matmul(H, M, K, N, 128, True, M + K + N + 1)


# Deep-K shapes for every row-tile variant of the pipelined tile kernel, so that its steady-state loop
# (the branch-free body with counted waits, entered once there are at least 5 K steps) runs under test:
#   128-row tiles (256 or more of them), 64-row tiles (fewer), 32-row tiles (fewer than 128 64-row tiles).
DEEP_K = [
    (512, 1024, 8192),     # 128-row tiles, 8 K steps
    (512, 8192, 8192),     # the benchmark's GEMM shape, 64 K steps
    (256, 1024, 151936),   # lm_head of a batch-256 decode step: 128-row tiles, ragged last n tile
    (512, 1024, 2048),     # 64-row tiles, 8 K steps
    (200, 2048, 3072),     # 64-row tiles, ragged m
    (256, 3072, 1024),     # 32-row tiles, 24 K steps (down projection of a prefill chunk)
    (48, 4096, 640),       # 32-row tiles, ragged m tile
]


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N", DEEP_K)
def test_matmul_tile_kernel_steady_state_bit_exact(H, M, K, N, int4):
    rng = np.random.default_rng(M + K + N + 3 * int4)
    B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
    A, As = O.quantize_activations(rand_act_f16(rng, M, K))
    want = (O.matmul_int4 if int4 else O.matmul_int8)(A, As, B, Bs, 128)
    got = bits((H.matmul_int4 if int4 else H.matmul_int8)(dev(A), dev(As), dev(B), dev(Bs), 128))
    assert (got == want).all(), f"max ulp {ulp_diff_f16(got, want).max()}"


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N", [(512, 2048, 8192), (512, 1024, 2048), (96, 1024, 512)])
def test_matmul_tile_kernel_group_sums_through_outputs(H, M, K, N, int4):
    """The INT32 group sums of the pipelined tile kernel itself (the group-sum entry point runs the
    plain tile kernel for M > 16): with activations in {-1, 0, 1} and small weights every group sum
    is an integer that binary16 holds exactly, so a scale panel that is 1 in one group and 0
    elsewhere makes the fp16 output equal that group's INT32 sum."""
    rng = np.random.default_rng(M + N + int4)
    G = K // 128
    A = rng.integers(-1, 2, (M, K), dtype=np.int8)
    As = np.ones((M,), np.float32)
    if int4:
        B = rng.integers(0, 256, (N, K // 2), dtype=np.uint8)
        lo = (B & 0xF).astype(np.int16); hi = (B >> 4).astype(np.int16)
        lo = np.where(lo > 7, lo - 16, lo); hi = np.where(hi > 7, hi - 16, hi)
        Bu = np.empty((N, K), np.int16); Bu[:, 0::2] = lo; Bu[:, 1::2] = hi
    else:
        B = rng.integers(-15, 16, (N, K), dtype=np.int8)
        Bu = B.astype(np.int16)
    dA, dAs, dB = dev(A), dev(As), dev(B)
    fn = H.matmul_int4 if int4 else H.matmul_int8
    for g in sorted({0, 1, G // 2, G - 2, G - 1}):
        Bs = np.zeros((N, G), np.float16)
        Bs[:, g] = 16.0 if int4 else 1.0     # INT4: the kernels fold an exact 1/16 into the scale
        got = fn(dA, dAs, dB, dev(Bs), 128).cpu().numpy().astype(np.float64)
        sl = slice(g * 128, (g + 1) * 128)
        want = A[:, sl].astype(np.int64) @ Bu[:, sl].astype(np.int64).T * (16 if int4 else 1)
        assert np.abs(want).max() <= 2048 * (16 if int4 else 1)
        assert np.array_equal(got, want.astype(np.float64)), g


@pytest.mark.parametrize("int4", [False, True])
def test_matmul_shared_activations_128_row_tiles(H, int4):
    """q/k/v-style shared launch large enough for 128-row tiles in steady state (16 K steps)."""
    rng = np.random.default_rng(31 + int4)
    M, K, Ns = 512, 2048, (4096, 2048, 2048 + 40)
    q, s = O.quantize_activations(rand_act_f16(rng, M, K))
    ws = [(rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K) for N in Ns]
    outs = [torch.full((M, N), 7.0, dtype=torch.float16, device="cuda") for N in Ns]
    sh = H.MatmulShared([(dev(B), dev(Bs.view(np.float16)), o) for (B, Bs), o in zip(ws, outs)], 128, int4)
    sh.run(dev(q), dev(s))
    for (B, Bs), o in zip(ws, outs):
        want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
        assert np.array_equal(bits(o), want)


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N,gs", [(2, 128, 4, 64), (3, 256, 9, 64), (2, 512, 5, 256), (1, 192, 3, 128)])
def test_matmul_other_group_sizes(H, M, K, N, gs, int4):
    """group 64 is what the reference's own tests use (tests/test_kernel.cpp:217); K = 192 with
    group 128 has a dropped K tail (kernel_gemm.cpp:147)."""
    _check_matmul(H, M, K, N, gs, int4, 77)


def test_matmul_zero_sized(H):
    A = torch.zeros((0, 128), dtype=torch.int8, device="cuda")
    As = torch.zeros((0,), dtype=torch.float32, device="cuda")
    B = torch.zeros((4, 128), dtype=torch.int8, device="cuda")
    Bs = torch.ones((4, 1), dtype=torch.float16, device="cuda")
    assert H.matmul_int8(A, As, B, Bs).shape == (0, 4)


@pytest.mark.parametrize("name", ["tensor_io_a.npz", "tensor_io_b.npz", "tensor_io_pad.npz"])
def test_matmul_on_reference_written_weights(H, name):
    """Weights quantized and written by the reference's tensor_io.py (committed fixture):
    packed INT4 straight from the file == INT8 on unpacked values == oracle."""
    z = np.load(os.path.join(GOLD, name))
    rng = np.random.default_rng(9)
    for key, int4 in (("file_int8", False), ("file_int4", True)):
        f = parse_cact(z[key])
        N, K = f["shape"]
        A, As = O.quantize_activations(rand_act_f16(rng, 4, K))
        Bs = f["scales"].view(np.float16).reshape(N, K // 128)
        if int4:
            Bp = f["data"].reshape(N, K // 2)
            c = bits(H.matmul_int4(dev(A), dev(As), dev(Bp), dev(Bs)))
            assert (c == O.matmul_int4(A, As, Bp, Bs, 128)).all()
            un = host(H.unpack_int4(dev(f["data"]), N * K)).reshape(N, K)
            c8 = bits(H.matmul_int8(dev(A), dev(As), dev(un), dev(Bs)))
            assert (c8 == c).all()
        else:
            Bq = f["data"].view(np.int8).reshape(N, K)
            c = bits(H.matmul_int8(dev(A), dev(As), dev(Bq), dev(Bs)))
            assert (c == O.matmul_int8(A, As, Bq, Bs, 128)).all()


def test_matmul_full_size_properties(H):
    """BASELINE sweep's largest shape (K = N = 8192), too big for the scalar oracle to be the
    only check: size-independent properties instead.
      * INT32 group sums are linear in A (exact integers);
      * packed INT4 == INT8 on the unpacked nibbles, bit for bit;
      * the M<=16 streaming kernel and the LDS-tiled kernel give identical bits;
      * a sampled set of rows against the oracle."""
    K = N = 8192
    g = torch.Generator(device="cuda").manual_seed(1)
    B = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    Bs = (0.01 + torch.rand((N, K // 128), device="cuda", generator=g) * 0.05).half()
    A1 = torch.randint(-60, 61, (16, K), dtype=torch.int8, device="cuda", generator=g)
    A2 = torch.randint(-60, 61, (16, K), dtype=torch.int8, device="cuda", generator=g)
    s1 = H.matmul_group_sums(A1, B)
    s2 = H.matmul_group_sums(A2, B)
    s12 = H.matmul_group_sums((A1 + A2), B)
    assert torch.equal(s1 + s2, s12)
    As = torch.rand((16,), device="cuda", generator=g) * 0.01
    c16 = H.matmul_int8(A1, As, B, Bs)
    A17 = torch.cat([A1, A2[:1]]); As17 = torch.cat([As, As[:1]])
    c17 = H.matmul_int8(A17, As17, B, Bs)            # tiled kernel
    assert torch.equal(c16.view(torch.int16), c17[:16].view(torch.int16))
    c1 = H.matmul_int8(A1[:1].contiguous(), As[:1].contiguous(), B, Bs)   # M = 1
    assert torch.equal(c1.view(torch.int16), c16[:1].view(torch.int16))
    # INT4: pack random nibbles, compare with INT8 on the unpacked values
    Bp = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device="cuda", generator=g)
    Bu = H.unpack_int4(Bp.view(-1), N * K).view(N, K)
    assert torch.equal(H.matmul_int4(A1, As, Bp, Bs).view(torch.int16),
                       H.matmul_int8(A1, As, Bu, Bs).view(torch.int16))
    # sampled rows against the oracle
    rows = np.array([0, 1, 17, 4095, 8191])
    oc = O.matmul_int8(host(A1), host(As), host(B[rows]), bits(Bs[rows]), 128)
    assert (bits(c16[:, rows].contiguous()) == oc).all()


def test_host_pointer_family_matches_device_family(H):
    rng = np.random.default_rng(5)
    M, K, N = 3, 1024, 256
    B, Bs = rand_weights_int8(rng, N, K)
    A, As = O.quantize_activations(rand_act_f16(rng, M, K))
    Cc = np.zeros((M, N), np.uint16)
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    H.lib.cactus_hip_clear_error()
    for _ in range(2):   # second call hits the resident-weight cache
        H.lib.cactus_hip_matmul_int8(p(A), p(As), p(B), p(Bs), p(Cc), M, K, N, 128)
    assert H.lib.cactus_hip_last_error() == b""
    assert (Cc == O.matmul_int8(A, As, B, Bs, 128)).all()
    x = rand_act_f16(rng, 1, 1000)
    assert H.lib.cactus_hip_fp16_max_abs(p(x), x.size) == O.fp16_max_abs(x)
    kv = rng.uniform(-2, 2, (3, 2, 128)).astype(np.float16)
    q = np.zeros(kv.shape, np.int8); s = np.zeros((3, 2, 1), np.float32)
    H.lib.cactus_hip_quantize_kv_fp16_to_int8(p(kv), p(q), p(s), 3, 2, 128, 128)
    oq, os_ = O.quantize_kv(kv)
    assert (q == oq).all() and (s == os_).all()
    H.lib.cactus_hip_release_weights(None)


def test_host_pointer_drop_ins_against_the_oracle(H):
    """Every host-pointer entry with the reference's signature (kernel.h:45-50, 87-104, 148, 163), numpy
    buffers in and out, against the oracle: INT4 matmul, FP16 matmul, both attentions, fp16_to_int8,
    unpack_int4_to_int8."""
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L = H.lib
    L.cactus_hip_clear_error()
    rng = np.random.default_rng(12)
    # cactus_matmul_int4
    M, K, N = 3, 1024, 200
    Bp, Bs = rand_weights_int4(rng, N, K)
    A, As = O.quantize_activations(rand_act_f16(rng, M, K))
    Cc = np.zeros((M, N), np.uint16)
    L.cactus_hip_matmul_int4(p(A), p(As), p(Bp), p(Bs), p(Cc), M, K, N, 128)
    assert (Cc == O.matmul_int4(A, As, Bp, Bs, 128)).all()
    # cactus_matmul_f16
    a = rng.uniform(-1, 1, (5, 256)).astype(np.float16); b = rng.uniform(-1, 1, (33, 256)).astype(np.float16)
    c = np.zeros((5, 33), np.uint16)
    L.cactus_hip_matmul_f16(p(a), p(b), p(c), 5, 256, 33)
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    assert np.abs(f16(c) - exact).max() <= np.abs(f16(O.matmul_f16(a, b)) - exact).max() + np.abs(exact).max() * 2.0 ** -10
    # cactus_attention_f16 (window + offset) and cactus_attention_hybrid_int8_fp16
    q = rng.uniform(-1, 1, (1, 9, 4, 64)).astype(np.float16)
    k = rng.uniform(-1, 1, (1, 50, 2, 64)).astype(np.float16); v = rng.uniform(-1, 1, (1, 50, 2, 64)).astype(np.float16)
    out = np.zeros(q.shape, np.uint16)
    L.cactus_hip_attention_f16(p(q), p(k), p(v), p(out), 1, 9, 50, 4, 2, 64, 0.0, None, 41, 16, 1)
    want = O.attention_f16(q, k, v, 0.0, position_offset=41, window_size=16)
    assert (np.abs(f16(out) - f16(want)) <= 2e-3 + 4e-3 * np.abs(f16(want))).all()
    q, kq, vq, ks, vs, kn, vn = _hybrid_case(rng, 1, 2, 16, 8, 128, 300, 2)
    out = np.zeros(q.shape, np.uint16)
    L.cactus_hip_attention_hybrid_int8_fp16(p(q), p(kq), p(vq), p(ks), p(vs), p(kn), p(vn), p(out), 1, 2, 300, 2, 16, 8,
                                            128, 0.0, 300, 1, 128)
    want = O.attention_hybrid(q, kq, vq, ks, vs, kn, vn, 0.0, position_offset=300, cache_len=300)
    assert (np.abs(f16(out) - f16(want)) <= 2e-3 + 4e-3 * np.abs(f16(want))).all()
    # cactus_fp16_to_int8 (vector body half-to-even, tail half-away) and cactus_unpack_int4_to_int8
    x = rng.uniform(-3, 3, 1003).astype(np.float16)
    x[-3:] = np.array([0.5, 1.5, -2.5], np.float16)
    d = np.zeros(1003, np.int8)
    L.cactus_hip_fp16_to_int8(p(x), p(d), 1003, C.c_float(1.0))
    assert (d == O.fp16_to_int8(x, 1.0)).all()
    packed = rng.integers(0, 256, 517, dtype=np.uint8)
    un = np.full(1033, 99, np.int8)
    L.cactus_hip_unpack_int4_to_int8(p(packed), p(un), 1033)
    assert (un == O.unpack_int4(packed, 1033)).all()
    assert L.cactus_hip_last_error() == b""
    L.cactus_hip_release_weights(None)


def test_host_pointer_matmuls_see_new_contents_at_an_old_address(H):
    """The reference's callers refill buffers in place: the FP16 matmul's b_transposed is a reused
    transpose scratch (graph_ops_nn.cpp:222-228), and a heap buffer holding quantized weights can be
    freed and handed out again.  A second call from the same address with other contents must
    multiply the new contents."""
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L = H.lib
    rng = np.random.default_rng(13)
    a = rng.uniform(-1, 1, (4, 128)).astype(np.float16)
    b = np.zeros((16, 128), np.float16)
    c = np.zeros((4, 16), np.uint16)
    for rep in range(3):
        b[...] = rng.uniform(-1, 1, b.shape).astype(np.float16)      # same address, new data
        L.cactus_hip_matmul_f16(p(a), p(b), p(c), 4, 128, 16)
        exact = a.astype(np.float64) @ b.astype(np.float64).T
        assert np.abs(f16(c) - exact).max() <= 0.02, rep
    M, K, N = 2, 512, 64
    A, As = O.quantize_activations(rand_act_f16(rng, M, K))
    B = np.zeros((N, K), np.int8); Bs = np.zeros((N, K // 128), np.float16)
    Cc = np.zeros((M, N), np.uint16)
    for rep in range(3):
        nb, nbs = rand_weights_int8(rng, N, K)
        B[...] = nb; Bs[...] = nbs
        L.cactus_hip_matmul_int8(p(A), p(As), p(B), p(Bs), p(Cc), M, K, N, 128)
        assert (Cc == O.matmul_int8(A, As, B, Bs, 128)).all(), rep
    L.cactus_hip_release_weights(None)


def test_matmul_weights_of_two_gib_take_the_tile_kernel(H):
    """The streaming (M <= 16) kernels address weights with 32-bit offsets below 2 GiB; a larger matrix
    must be routed to a kernel that can address it, not read as zeros.  262144 x 8192 INT8 = 2 GiB:
    sampled weight rows across the whole range, the last ones included, against the oracle, for
    pre-quantized and for FP16 activations."""
    N, K = 262144, 8192
    g = torch.Generator(device="cuda").manual_seed(3)
    B = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    Bs = (0.01 + torch.rand((N, K // 128), device="cuda", generator=g) * 0.05).half()
    rng = np.random.default_rng(4)
    x = rand_act_f16(rng, 1, K)
    A, As = O.quantize_activations(x)
    c1 = H.matmul_int8(dev(A), dev(As), B, Bs)
    c2 = H.matmul_f16act(dev(x), B, Bs)
    rows = np.array([0, 5, 131071, 131072, 200001, N - 17, N - 1])
    want = O.matmul_int8(A, As, host(B[rows]), bits(Bs[rows]), 128)
    assert (bits(c1[:, rows].contiguous()) == want).all()
    assert (bits(c2[:, rows].contiguous()) == want).all()
    assert torch.equal(c1.view(torch.int16), c2.view(torch.int16))
    del B, Bs
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ FP16 matmul
@pytest.mark.parametrize("M,K,N", [(1, 1024, 64), (4, 100, 33), (70, 256, 130), (2, 3, 2), (256, 1024, 256),
                                   (40, 128, 300), (300, 64, 520), (130, 192, 1000), (17, 2048, 1200), (64, 320, 2050)])
def test_matmul_f16_tolerance(H, M, K, N):
    """fp32-accumulating MFMA vs the reference's fp16-lane accumulation: both are compared with
    the fp64 product; the GPU error must not exceed the reference's own error (+1 fp16 ulp)."""
    rng = np.random.default_rng(M + K + N)
    a = rng.uniform(-1, 1, (M, K)).astype(np.float16)
    b = rng.uniform(-1, 1, (N, K)).astype(np.float16)
    got = f16(bits(H.matmul_f16(dev(a), dev(b))))
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    ref_err = np.abs(f16(O.matmul_f16(a, b)) - exact).max()
    ulp = np.abs(exact).max() * 2.0 ** -10
    assert np.abs(got - exact).max() <= max(ref_err, 0.0) + ulp


@pytest.mark.parametrize("M,K,N", [(512, 4096, 4096), (1, 4096, 12288), (33, 8192, 520), (1000, 1024, 777),
                                   (128, 2048, 151936 // 8), (70, 4160, 136), (5, 72, 19)])
def test_matmul_f16_close_to_fp32_product(H, M, K, N):
    """Larger shapes than the CPU restatement finishes quickly, every tile of the pipelined kernel and the plain kernel
    (K not a multiple of 64): within one fp16 ulp of the fp32 product rounded to fp16 (tolerance: 2^-10 relative to the
    exact value plus the fp32 accumulation error K * 2^-24 * sum |a||b| bound, taken here as 2^-10 of the row's
    absolute product sum)."""
    gen = torch.Generator(device="cuda").manual_seed(M + K + N)
    a = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
    b = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).half()
    got = H.matmul_f16(a, b).double()
    exact = a.double() @ b.double().T
    bound = exact.abs() * 2.0 ** -10 + (a.double().abs() @ b.double().abs().T) * (K * 2.0 ** -24) + 2.0 ** -24
    assert bool(((got - exact).abs() <= bound).all())


def test_matmul_f16_reference_kat(H):
    a = np.array([[1, 2, 3], [4, 5, 6]], np.float16)
    b = np.array([[1, 2], [3, 4], [5, 6]], np.float16)
    got = host(H.matmul_f16(dev(a), dev(np.ascontiguousarray(b.T))))
    assert (got == np.array([[22, 28], [49, 64]], np.float16)).all()


# ------------------------------------------------------------------ attention
def _hybrid_case(rng, B, S, Hq, Hkv, D, cache, new, gs=128):
    q = rng.uniform(-1, 1, (B, S, Hq, D)).astype(np.float16)
    kq, ks = O.quantize_kv(rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16), gs)
    vq, vs = O.quantize_kv(rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16), gs)
    kq = np.broadcast_to(kq, (B,) + kq.shape).copy()
    vq = np.broadcast_to(vq, (B,) + vq.shape).copy()
    if B > 1:   # different cache content per batch row (scales are shared: reference quirk)
        kq[1:] = rng.integers(-127, 128, kq[1:].shape, dtype=np.int8)
    kn = rng.uniform(-2, 2, (B, new, Hkv, D)).astype(np.float16)
    vn = rng.uniform(-2, 2, (B, new, Hkv, D)).astype(np.float16)
    return q, kq, vq, ks, vs, kn, vn


HYBRID = [
    # B, S, Hq, Hkv, D, cache, new, offset(None -> cache)
    (1, 1, 16, 8, 128, 512, 1, None),     # Qwen3-0.6B decode
    (1, 1, 16, 8, 128, 70, 1, None),
    (1, 1, 4, 1, 256, 100, 1, None),      # Gemma-3-270m heads
    (1, 1, 32, 8, 64, 333, 1, None),      # LFM2-1.2B heads
    (1, 1, 32, 8, 128, 4096, 1, None),    # long context, split-KV path
    (1, 5, 4, 2, 128, 70, 5, None),       # prefill chunk over a cache
    (1, 37, 4, 2, 64, 33, 37, None),
    (1, 4, 4, 2, 128, 0, 4, 0),           # empty cache
    (2, 3, 6, 2, 128, 40, 3, None),       # GQA ratio 3
    (1, 3, 4, 4, 72, 21, 3, None),        # head_dim not a multiple of 16
    (1, 2, 2, 2, 128, 64, 2, 700),        # window has slid: offset > cache_len (no masking inside chunk)
    (1, 4, 2, 1, 128, 50, 4, 10),         # offset < cache_len: causal cut inside the cache
    (1, 3, 4, 2, 20, 45, 3, None),        # head_dim not a multiple of 8: any-shape kernel
    (2, 2, 2, 1, 300, 37, 2, None),       # head_dim above 256, three scale groups
    # many queries, head_dim 64 / 128: the matrix-core chunk kernel (key ranges split over workgroups)
    (1, 256, 16, 8, 128, 512, 256, None),   # Qwen3 prefill chunk 2
    (1, 256, 4, 2, 128, 1000, 256, None),   # cache not a multiple of 32
    (1, 40, 4, 2, 64, 333, 40, None),
    (2, 64, 6, 2, 128, 200, 64, None),      # two sequences (shared scales, as the reference indexes them)
    (1, 130, 2, 1, 128, 0, 130, 0),         # empty cache: only the chunk's own keys
    (1, 64, 2, 2, 64, 300, 64, 120),        # offset < cache_len: causal cut inside the cache
    (1, 33, 4, 1, 128, 90, 33, 700),        # offset beyond the chunk: nothing masked
    (1, 256, 4, 1, 256, 700, 256, None),    # Gemma heads (head_dim 256, two scale groups per cached row)
    (2, 70, 8, 2, 256, 333, 70, None),
    (1, 64, 4, 1, 256, 0, 64, 0),           # ... empty cache
    (1, 40, 4, 2, 256, 300, 40, 150),       # ... causal cut inside the cache
    # one query per sequence: the matrix-core decode kernel, one wave per (sequence, kv head, key range)
    (5, 1, 6, 3, 128, 77, 1, None),         # 15 waves of work: the last workgroup is part empty
    (3, 1, 16, 1, 64, 100, 1, None),        # 16 query heads on one kv head
    (2, 1, 8, 4, 128, 40, 3, 42),           # three tokens of the step itself, all visible
    (2, 1, 8, 4, 64, 40, 3, 41),            # ... the last one masked
    (1, 1, 4, 2, 128, 31, 1, None),         # less than one block
    (1, 1, 4, 2, 64, 2048, 1, 1000),        # causal cut inside the cache, key ranges split
    (520, 1, 16, 8, 64, 40, 1, None),       # 4160 waves of work: the 8-wave workgroups
    (300, 1, 16, 8, 128, 65, 1, None),      # 2400 waves of work, no split
    # head_dim 256 (Gemma): two scale groups per cached row, two-wave workgroups
    (3, 1, 8, 2, 256, 700, 1, None),
    (1, 1, 4, 1, 256, 4096, 1, None),       # key ranges split
    (2, 1, 16, 1, 256, 45, 3, 46),          # 16 query heads, three tokens of the step itself, the last one masked
    (1, 1, 8, 4, 256, 1500, 1, 900),        # causal cut inside the cache
    (257, 1, 4, 1, 256, 33, 1, None),       # an odd number of waves: the last workgroup is half empty
]


@pytest.mark.parametrize("B,S,Hq,Hkv,D,cache,new,off", HYBRID)
def test_attention_hybrid_vs_oracle(H, B, S, Hq, Hkv, D, cache, new, off):
    """fp32 softmax with a different summation order and __expf: |diff| <= 2e-3 + 4e-3*|ref|
    (2e-3 is about 2 fp16 ulps at 1.0; outputs here have magnitude <= 2)."""
    rng = np.random.default_rng(B * 7 + S + D + cache)
    q, kq, vq, ks, vs, kn, vn = _hybrid_case(rng, B, S, Hq, Hkv, D, cache, new)
    off = cache if off is None else off
    want = O.attention_hybrid(q, kq, vq, ks, vs, kn, vn, 0.0, position_offset=off, cache_len=cache)
    got = bits(H.attention_hybrid(dev(q), dev(kq), dev(vq), dev(ks), dev(vs), dev(kn), dev(vn), 0.0,
                                  position_offset=off, cache_len=cache))
    Da = D // 8 * 8    # the reference leaves output dims past the last whole 8 unwritten (:464-474)
    got, want = f16(got)[..., :Da], f16(want)[..., :Da]
    d = np.abs(got - want)
    assert (d <= 2e-3 + 4e-3 * np.abs(want)).all(), d.max()


@pytest.mark.parametrize("B,S,Hq,Hkv,D,cache,new,off", [
    (1, 256, 16, 8, 128, 512, 256, None),    # Qwen3 prefill chunk: splits merged per token, quantized in the same launch
    (1, 256, 16, 8, 128, 3840, 256, None),   # the last chunk of a 4096-token prompt
    (2, 64, 6, 2, 128, 200, 64, None),       # 768 outputs per token: part of the workgroup idle
    (1, 70, 40, 8, 128, 333, 70, None),      # 5120 outputs per token: three passes per thread
    (1, 256, 4, 1, 256, 700, 256, None),     # Gemma heads
    (1, 40, 4, 2, 64, 333, 40, None),
    (1, 5, 4, 2, 128, 70, 5, None),          # few queries: attention, then the row quantization kernel
    (3, 1, 16, 8, 128, 100, 1, None),        # decode shape
    (1, 40, 80, 8, 128, 333, 40, None),      # 10240 outputs per token: more than the merge kernel holds
])
def test_attention_hybrid_quant_equals_attention_then_quantize(H, B, S, Hq, Hkv, D, cache, new, off):
    """cactus_hip_attention_hybrid_int8_fp16_quant_dev: the quantized rows and scales are exactly what
    cactus_hip_quantize_activations_dev makes of ITS output, and that output is within the hybrid attention's
    tolerance of the oracle."""
    rng = np.random.default_rng(B * 7 + S + D + cache + 1)
    q, kq, vq, ks, vs, kn, vn = _hybrid_case(rng, B, S, Hq, Hkv, D, cache, new)
    off = cache if off is None else off
    oq = torch.empty((B * S, Hq * D), dtype=torch.int8, device="cuda")
    osc = torch.empty((B * S,), dtype=torch.float32, device="cuda")
    out = H.attention_hybrid(dev(q), dev(kq), dev(vq), dev(ks), dev(vs), dev(kn), dev(vn), 0.0, position_offset=off,
                             cache_len=cache, out_q=oq, out_scales=osc)
    want_q, want_s = H.quantize_activations(out.reshape(B * S, Hq * D))
    assert torch.equal(oq, want_q)
    assert torch.equal(osc, want_s)
    want = O.attention_hybrid(q, kq, vq, ks, vs, kn, vn, 0.0, position_offset=off, cache_len=cache)
    d = np.abs(f16(bits(out)) - f16(want))
    assert (d <= 2e-3 + 4e-3 * np.abs(f16(want))).all(), d.max()


@pytest.mark.parametrize("B,Hq,Hkv,D,cache", [(1, 16, 8, 128, 512), (4, 16, 8, 128, 4096), (1, 32, 8, 64, 333),
                                              (64, 32, 8, 64, 100), (1, 32, 8, 128, 20000), (2, 4, 1, 256, 512),
                                              (1, 8, 4, 256, 6000)])
def test_attention_decode_rounds_like_the_oracle(H, B, Hq, Hkv, D, cache):
    """The matrix-core decode kernel against the oracle, output by output.  Both sum in fp32, in different
    orders (the kernel: exact fp16 products, weights as two binary16 parts, the MFMA's order), so they
    differ before the final rounding by about 1e-5 of a head's largest output: more than 95 % of the fp16
    outputs are bit-identical (observed 96.5 - 97.3 %: these random heads average hundreds of values of
    both signs, which is the worst case for relative error) and no output is further off than one fp16 ulp
    of the largest."""
    rng = np.random.default_rng(B + Hq + D + cache)
    q, kq, vq, ks, vs, kn, vn = _hybrid_case(rng, B, 1, Hq, Hkv, D, cache, 1)
    want = O.attention_hybrid(q, kq, vq, ks, vs, kn, vn, 0.0, position_offset=cache, cache_len=cache)
    got = bits(H.attention_hybrid(dev(q), dev(kq), dev(vq), dev(ks), dev(vs), dev(kn), dev(vn), 0.0,
                                  position_offset=cache, cache_len=cache))
    same = (got.view(np.uint16) == want.view(np.uint16)).mean()
    g, w = f16(got).astype(np.float64), f16(want).astype(np.float64)
    assert same > 0.95, same
    assert np.abs(g - w).max() <= 2.0 ** -10 * np.abs(w).max(), (np.abs(g - w).max(), np.abs(w).max())


def test_attention_hybrid_noncausal_and_scale(H):
    rng = np.random.default_rng(3)
    q, kq, vq, ks, vs, kn, vn = _hybrid_case(rng, 1, 3, 4, 2, 128, 45, 3)
    for causal, scale in ((False, 0.0), (True, 0.05)):
        want = O.attention_hybrid(q, kq, vq, ks, vs, kn, vn, scale, position_offset=45, is_causal=causal)
        got = bits(H.attention_hybrid(dev(q), dev(kq), dev(vq), dev(ks), dev(vs), dev(kn), dev(vn), scale,
                                      position_offset=45, is_causal=causal))
        assert np.abs(f16(got) - f16(want)).max() <= 2e-3


def test_attention_hybrid_batched_matches_single(H):
    """The batched-decode entry point on sequences of different lengths == one call per sequence."""
    rng = np.random.default_rng(8)
    Bn, Hq, Hkv, D, maxc = 3, 16, 8, 128, 96
    lens = np.array([96, 10, 57], np.int32)
    q = rng.uniform(-1, 1, (Bn, 1, Hq, D)).astype(np.float16)
    kn = rng.uniform(-2, 2, (Bn, 1, Hkv, D)).astype(np.float16)
    vn = rng.uniform(-2, 2, (Bn, 1, Hkv, D)).astype(np.float16)
    kq = np.zeros((Bn, maxc, Hkv, D), np.int8); vq = np.zeros_like(kq)
    ks = np.ones((Bn, maxc, Hkv, 1), np.float32); vs = np.ones_like(ks)
    for b in range(Bn):
        a, s_ = O.quantize_kv(rng.uniform(-2, 2, (lens[b], Hkv, D)).astype(np.float16))
        kq[b, :lens[b]] = a; ks[b, :lens[b]] = s_
        a, s_ = O.quantize_kv(rng.uniform(-2, 2, (lens[b], Hkv, D)).astype(np.float16))
        vq[b, :lens[b]] = a; vs[b, :lens[b]] = s_
    got = bits(H.attention_hybrid_batched(dev(q), dev(kq), dev(vq), dev(ks), dev(vs), dev(kn), dev(vn), 0.0,
                                          dev(lens), dev(lens)))
    for b in range(Bn):
        L = int(lens[b])
        want = O.attention_hybrid(q[b:b + 1], kq[b, :L], vq[b, :L], ks[b, :L], vs[b, :L], kn[b:b + 1],
                                  vn[b:b + 1], 0.0, position_offset=L, cache_len=L)
        assert np.abs(f16(got[b:b + 1]) - f16(want)).max() <= 2e-3


F16ATT = [
    # B, S, KV, Hq, Hkv, D, offset, window, causal, mask
    (1, 64, 64, 4, 2, 64, 0, 0, True, False),
    (2, 37, 37, 4, 2, 128, 0, 0, True, False),
    (1, 256, 256, 16, 8, 128, 0, 0, True, False),    # Qwen3 prefill chunk 0
    (1, 40, 40, 4, 1, 256, 0, 16, True, False),      # Gemma local layer window
    (1, 9, 50, 2, 2, 64, 41, 0, True, False),        # offset
    (1, 7, 30, 4, 4, 64, 0, 0, False, False),        # non-causal (encoders)
    (1, 6, 35, 2, 1, 64, 0, 0, False, True),         # explicit mask
    (1, 1, 131, 4, 2, 128, 130, 0, True, False),     # single query, kv tail of 3 uses expf
    (1, 2, 2, 1, 1, 4, 0, 0, True, False),           # tests/test_graph.cpp:220-240 shape, any-shape kernel
    (1, 35, 35, 2, 2, 20, 0, 8, True, False),
    (1, 5, 70, 3, 1, 36, 65, 0, False, True),
    # many queries, head_dim 64 / 128, no window, no mask: the matrix-core prefill kernel
    (1, 300, 300, 4, 2, 128, 0, 0, True, False),     # ragged query tile and key block
    (1, 130, 200, 4, 2, 64, 70, 0, True, False),     # chunk with an offset into a longer K/V
    (2, 40, 75, 2, 1, 128, 0, 0, False, False),      # non-causal, 75 = 2 blocks + 11 (tail of 3 uses expf)
    (1, 33, 131, 8, 8, 64, 98, 0, True, False),
    (1, 1024, 1024, 2, 1, 128, 0, 0, True, False),
    (3, 64, 64, 6, 3, 64, 0, 0, True, False),
    (1, 200, 200, 4, 1, 128, 0, 512, True, False),   # a window longer than the prompt cuts nothing: matrix-core kernel
    (1, 64, 300, 4, 1, 128, 236, 512, True, False),  # ... with an offset, still inside the window
    (1, 64, 600, 4, 1, 128, 536, 512, True, False),  # the window cuts: every query's blocks start at its own first key
    (1, 700, 700, 4, 1, 256, 0, 512, True, False),   # Gemma local layer, the window starts to cut inside a workgroup
    (1, 300, 300, 4, 2, 128, 0, 100, True, False),   # a window that is no multiple of the block
    (2, 200, 331, 4, 2, 64, 131, 50, True, False),   # offset; the last block is cut by the K/V length (expf tail of 3)
    (1, 130, 130, 2, 1, 64, 0, 7, True, False),      # a window shorter than a block
    (1, 96, 400, 2, 1, 128, 1000, 512, True, False), # every query lies further than the window behind the last key: zeros
    (1, 64, 300, 2, 1, 64, 500, 400, True, False),   # a window longer than K/V masks by distance only (blocks from key 0)
    (1, 64, 200, 2, 1, 128, 300, 64, False, False),  # non-causal with a window: the VALU kernel
    (2, 300, 300, 8, 2, 128, 0, 0, True, False),     # two sequences, heads x sequences a multiple of 8 (heads pinned to XCDs)
    (2, 200, 260, 4, 2, 64, 60, 0, True, False),
    (1, 128, 96, 8, 8, 128, 0, 0, False, False),     # 3 key blocks: the two wave groups take 2 and 1
    (1, 50, 32, 2, 1, 64, 0, 0, False, False),       # 1 key block: the second group has nothing
    (1, 2048, 2048, 2, 1, 128, 0, 0, True, False),   # 32 iterations per group on the last tile
    (4, 130, 130, 32, 8, 128, 0, 0, True, False),    # 256 tiles of 128 queries: the 512-thread workgroups (fewer take 64-query ones)
    (2, 200, 260, 64, 8, 64, 60, 0, True, False),    # ... head_dim 64, offset
    (1, 300, 300, 4, 1, 256, 0, 512, True, False),   # Gemma heads, a local layer's window longer than the prompt
    (2, 130, 200, 8, 4, 256, 70, 0, True, False),    # ... a global layer, offset
    (1, 75, 75, 4, 1, 256, 0, 0, False, False),      # ... non-causal, tail of 3
]


@pytest.mark.parametrize("B,S,KV,Hq,Hkv,D,off,win,causal,use_mask", F16ATT)
def test_attention_f16_vs_oracle(H, B, S, KV, Hq, Hkv, D, off, win, causal, use_mask):
    """Same per-block polynomial weights as the reference; only fp32 summation order and
    expf-vs-__expf differ: |diff| <= 2e-3 + 4e-3*|ref|."""
    rng = np.random.default_rng(S * 3 + KV + D)
    q = rng.uniform(-1, 1, (B, S, Hq, D)).astype(np.float16)
    k = rng.uniform(-1, 1, (B, KV, Hkv, D)).astype(np.float16)
    v = rng.uniform(-1, 1, (B, KV, Hkv, D)).astype(np.float16)
    mask = None
    if use_mask:
        mask = (rng.random((B, S, KV)) > 0.3).astype(np.float16)
        mask[:, :, 0] = 1
    want = O.attention_f16(q, k, v, 0.0, mask=mask, position_offset=off, window_size=win, is_causal=causal)
    got = bits(H.attention_f16(dev(q), dev(k), dev(v), 0.0, mask=None if mask is None else dev(mask),
                               position_offset=off, window_size=win, is_causal=causal))
    d = np.abs(f16(got) - f16(want))
    assert (d <= 2e-3 + 4e-3 * np.abs(f16(want))).all(), d.max()


def test_attention_f16_full_size(H):
    """A 4096-token causal prefill with Qwen3-0.6B heads: (a) query heads 0 and 9 against the oracle (their kv heads
    only), same tolerance as above; (b) the first 2048 queries see only the first 2048 keys, so they equal a 2048-token
    call - which runs 64-query workgroups, so the cut between the two wave groups' key ranges falls elsewhere and the
    fp32 sums differ in order: at most one fp16 ulp apart, and equal in 99 % of the outputs."""
    S, Hq, Hkv, D = 4096, 16, 8, 128
    rng = np.random.default_rng(4096)
    q = rng.uniform(-1, 1, (1, S, Hq, D)).astype(np.float16)
    k = rng.uniform(-1, 1, (1, S, Hkv, D)).astype(np.float16)
    v = rng.uniform(-1, 1, (1, S, Hkv, D)).astype(np.float16)
    got = bits(H.attention_f16(dev(q), dev(k), dev(v), 0.0, is_causal=True))
    for h in (0, 9):
        kvh = h // (Hq // Hkv)
        want = O.attention_f16(q[:, :, h:h + 1], k[:, :, kvh:kvh + 1], v[:, :, kvh:kvh + 1], 0.0, is_causal=True)
        d = np.abs(f16(got[:, :, h:h + 1]) - f16(want))
        assert (d <= 2e-3 + 4e-3 * np.abs(f16(want))).all(), d.max()
    half = bits(H.attention_f16(dev(q[:, :2048]), dev(k[:, :2048]), dev(v[:, :2048]), 0.0, is_causal=True))
    assert ulp_diff_f16(half, got[:, :2048]).max() <= 1
    assert (half == got[:, :2048]).mean() >= 0.99


def test_attention_long_context_properties(H):
    """32k-token INT8 cache (config 5 shape: Hq 32, Hkv 8, D 128): the oracle is too slow for the
    whole thing, so (a) all 32 heads against the oracle on 2 of the 8 kv heads, and (b) the
    property that duplicating the cache (K,V repeated twice) leaves the output unchanged when
    the new token is masked out of neither copy."""
    rng = np.random.default_rng(11)
    cache, Hq, Hkv, D = 32768, 32, 8, 128
    q = rng.uniform(-1, 1, (1, 1, Hq, D)).astype(np.float16)
    k16 = torch.empty((cache, Hkv, D), device="cuda").uniform_(-2, 2).half()
    v16 = torch.empty((cache, Hkv, D), device="cuda").uniform_(-2, 2).half()
    kq, ks = H.quantize_kv(k16); vq, vs = H.quantize_kv(v16)
    kn = rng.uniform(-2, 2, (1, 1, Hkv, D)).astype(np.float16)
    vn = rng.uniform(-2, 2, (1, 1, Hkv, D)).astype(np.float16)
    got = bits(H.attention_hybrid(dev(q), kq, vq, ks, vs, dev(kn), dev(vn), 0.0, position_offset=cache))
    # (a) kv heads 0 and 5 -> q heads 0-3 and 20-23
    for kvh in (0, 5):
        sl = slice(kvh, kvh + 1)
        want = O.attention_hybrid(q[:, :, kvh * 4:(kvh + 1) * 4], host(kq[:, sl]), host(vq[:, sl]),
                                  host(ks[:, sl]), host(vs[:, sl]), kn[:, :, sl], vn[:, :, sl], 0.0,
                                  position_offset=cache)
        assert np.abs(f16(got[:, :, kvh * 4:(kvh + 1) * 4]) - f16(want)).max() <= 2e-3
    # (b) softmax over two copies of the same keys == over one copy (new token excluded: new_len 0)
    empty = torch.empty((1, 0, Hkv, D), dtype=torch.float16, device="cuda")
    one = H.attention_hybrid(dev(q), kq[:8192].contiguous(), vq[:8192].contiguous(), ks[:8192].contiguous(),
                             vs[:8192].contiguous(), empty, empty, 0.0, is_causal=False)
    two = H.attention_hybrid(dev(q), torch.cat([kq[:8192]] * 2), torch.cat([vq[:8192]] * 2),
                             torch.cat([ks[:8192]] * 2), torch.cat([vs[:8192]] * 2), empty, empty, 0.0,
                             is_causal=False)
    assert np.abs(f16(bits(one)) - f16(bits(two))).max() <= 1e-3


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,Ns", [(256, 1024, (2048, 1024, 1024)), (40, 512, (100, 272)), (300, 1024, (3072, 3072)),
                                    (8, 1024, (256, 128)), (33, 768, (64,) * 9)])
def test_matmul_shared_activations_one_launch_bit_exact(H, M, K, Ns, int4):
    """q/k/v or gate/up on one quantized activation matrix (model_qwen.cpp:55-57, 112-113): the
    batched launch equals one matmul per weight."""
    if int4 and K % 256:
        pytest.skip("packed units hold two groups")
    rng = np.random.default_rng(M + K + sum(Ns) + int4)
    x = rand_act_f16(rng, M, K)
    q, s = O.quantize_activations(x)
    ws = [(rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K) for N in Ns]
    outs = [torch.full((M, N), 7.0, dtype=torch.float16, device="cuda") for N in Ns]
    sh = H.MatmulShared([(dev(B), dev(Bs.view(np.float16)), o) for (B, Bs), o in zip(ws, outs)], 128, int4)
    sh.run(dev(q), dev(s))
    for (B, Bs), o in zip(ws, outs):
        want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
        assert np.array_equal(bits(o), want)


# --------------------------------------------------------------------------- matmuls with produced activations
@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,Ns,with_add", [
    (1, 1024, (2048, 1024, 1024), True),      # Qwen3 q/k/v behind the input norm
    (1, 1024, (3072, 3072), True),            # gate/up behind the post-attention norm
    (1, 1024, (2048,), False),                # plain norm (first layer), one weight
    (4, 1024, (1000, 40), True),              # four rows, ragged N
    (2, 3072, (1024,), True),
    (3, 256, (64, 24), False),
    (1, 8192, (512,), True),
    (5, 1024, (2048, 1024), True),            # M > 4: producer + quantization launch, then the matmuls
    (16, 512, (256,), False),
    (40, 1024, (1024, 512), True),
])
def test_matmul_norm_f16act_equals_norm_then_matmuls(H, M, K, Ns, with_add, int4):
    """Residual add + RMS norm + activation quantization + the matmuls that read the normalized rows
    (model_qwen.cpp:51-57, 109-113), one launch for M <= 4: sum and every output bit-exact against
    the oracle's add, rms_norm, activation quantization and matmul."""
    if int4 and K % 256:
        pytest.skip("packed units hold two groups")
    rng = np.random.default_rng(M + K + sum(Ns) + int4 + with_add)
    a = (rng.standard_normal((M, K)) * 1.5).astype(np.float16)
    b = (rng.standard_normal((M, K)) * 1.5).astype(np.float16)
    w = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float16)
    ws = [(rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K) for N in Ns]
    outs = [torch.full((M, N), 7.0, dtype=torch.float16, device="cuda") for N in Ns]
    mp = H.MatmulProduced([(dev(B), dev(Bs.view(np.float16)), o) for (B, Bs), o in zip(ws, outs)], 128, int4)
    out_sum = torch.full((M, K), 3.0, dtype=torch.float16, device="cuda") if with_add else None
    mp.run_norm(dev(a), dev(b) if with_add else None, dev(w), 1e-6, out_sum)
    want_sum = O.binary("add", a, b) if with_add else a.view(np.uint16)
    if with_add:
        assert np.array_equal(bits(out_sum), want_sum)
    q, s = O.quantize_activations(O.rms_norm(want_sum, w, 1e-6))
    for (B, Bs), o in zip(ws, outs):
        want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
        assert np.array_equal(bits(o), want)


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("act", ["silu", "gelu"])
@pytest.mark.parametrize("M,K,Ns", [(1, 3072, (1024,)), (4, 3072, (1024,)), (2, 2048, (640, 24)), (1, 256, (16,)),
                                    (1, 12288, (256,)), (7, 3072, (1024,)), (33, 512, (128, 64))])
def test_matmul_activation_f16act_equals_activation_then_matmul(H, M, K, Ns, act, int4):
    """act(gate) * up + activation quantization + the down projection (model_qwen.cpp:114-117), one
    launch for M <= 4."""
    if int4 and K % 256:
        pytest.skip("packed units hold two groups")
    rng = np.random.default_rng(M + K + sum(Ns) + int4)
    g = (rng.standard_normal((M, K)) * 3).astype(np.float16)
    u = rng.standard_normal((M, K)).astype(np.float16)
    ws = [(rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K) for N in Ns]
    outs = [torch.full((M, N), 7.0, dtype=torch.float16, device="cuda") for N in Ns]
    mp = H.MatmulProduced([(dev(B), dev(Bs.view(np.float16)), o) for (B, Bs), o in zip(ws, outs)], 128, int4)
    mp.run_act(dev(g), dev(u), act)
    rows = O.binary("multiply", getattr(O, act)(g.reshape(-1)).view(np.float16), u.reshape(-1)).reshape(M, K)
    q, s = O.quantize_activations(rows)
    for (B, Bs), o in zip(ws, outs):
        want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
        assert np.array_equal(bits(o), want)


def test_matmul_produced_argument_errors(H):
    import ctypes as C
    x = dev(np.zeros((1, 128), np.float16))
    B = dev(np.zeros((16, 128), np.int8)); Bs = dev(np.ones((16, 1), np.float16))
    o = torch.empty((1, 16), dtype=torch.float16, device="cuda")
    arr = (H.MatmulWeights * 1)(H.MatmulWeights(B.data_ptr(), Bs.data_ptr(), o.data_ptr(), 16))
    p = lambda t: C.c_void_p(t.data_ptr())
    fn = H.lib.cactus_hip_matmul_int8_norm_f16act_dev
    assert fn(p(x), p(x), p(x), None, 1e-6, arr, 1, 1, 128, 128, None) != 0          # addend without sum_output
    assert fn(p(x), None, None, None, 1e-6, arr, 1, 1, 128, 128, None) != 0          # no norm weight
    assert fn(None, None, p(x), None, 1e-6, arr, 1, 1, 128, 128, None) != 0
    assert H.lib.cactus_hip_matmul_int8_silu_f16act_dev(p(x), None, arr, 1, 1, 128, 128, None) != 0
    assert fn(None, None, None, None, 1e-6, None, 0, 0, 128, 128, None) == 0         # nothing to do
