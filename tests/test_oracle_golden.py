"""CPU tests: the oracle (oracle/cactus_oracle.c) against every known-answer
test and fixture the reference holds for the hot path, plus self-consistency
against float64 numpy.  No GPU needed."""
import os

import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import parse_cact, f16, rand_weights_int8, rand_weights_int4, rand_act_f16

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = ["tensor_io_a.npz", "tensor_io_b.npz", "tensor_io_pad.npz"]


# --------------------------------------------------------------------------
# Fixtures written by the reference's python/src/tensor_io.py
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", FIXTURES)
def test_tensor_io_int8_layout(name):
    z = np.load(os.path.join(GOLD, name))
    w = z["weights"]
    f = parse_cact(z["file_int8"])
    N, K = f["shape"]
    assert f["precision"] == 0 and f["group_size"] == 128 and K % 128 == 0
    assert N == w.shape[0] and K >= w.shape[1] and f["num_groups"] == K // 128
    q = f["data"].view(np.int8).reshape(N, K)
    s = f["scales"].view(np.float16).astype(np.float32).reshape(N, K // 128)
    wp = np.zeros((N, K), np.float32)
    wp[:, :w.shape[1]] = w
    deq = q.astype(np.float32).reshape(N, -1, 128) * s[:, :, None]
    # scale = absmax/127 stored as fp16: dequant error <= half a step (+ fp16 scale rounding)
    err = np.abs(deq.reshape(N, K) - wp)
    assert (err <= 0.51 * np.repeat(s, 128, axis=1) + 1e-3 * np.abs(wp)).all()


@pytest.mark.parametrize("name", FIXTURES)
def test_tensor_io_int4_nibble_rule(name):
    """The packed file the reference wrote, decoded by the oracle's unpack
    (kernel_quants.cpp:295-351 rule), reproduces the reference quantizer's
    integers clip(round(w/scale),-8,7) (tensor_io.py:213-229)."""
    z = np.load(os.path.join(GOLD, name))
    w = z["weights"]
    f = parse_cact(z["file_int4"])
    N, K = f["shape"]
    assert f["precision"] == 3 and f["data"].size == N * K // 2
    wp = np.zeros((N, K), np.float32)
    wp[:, :w.shape[1]] = w
    g = wp.reshape(N, K // 128, 128)
    scales = np.maximum((np.abs(g).max(axis=2) / 7.0).astype(np.float32), 1e-10)
    q_expect = np.clip(np.round(g / scales[:, :, None]), -8, 7).astype(np.int8).reshape(N, K)
    q = O.unpack_int4(f["data"], N * K).reshape(N, K)
    assert (q == q_expect).all()
    assert (f["scales"].view(np.float16) == scales.astype(np.float16).ravel()).all()


@pytest.mark.parametrize("name", FIXTURES)
def test_int4_packed_equals_unpacked_int8(name):
    """Packed INT4 matmul == INT8 matmul on the unpacked values, bit for bit
    (what MappedFile::data() relies on, graph_io.cpp:435-463)."""
    z = np.load(os.path.join(GOLD, name))
    f = parse_cact(z["file_int4"])
    N, K = f["shape"]
    rng = np.random.default_rng(5)
    A16 = rand_act_f16(rng, 3, K)
    A, As = O.quantize_activations(A16)
    packed = f["data"].reshape(N, K // 2)
    unpacked = O.unpack_int4(f["data"], N * K).reshape(N, K)
    c4, g4 = O.matmul_int4(A, As, packed, f["scales"].reshape(N, -1), 128, True)
    c8, g8 = O.matmul_int8(A, As, unpacked, f["scales"].reshape(N, -1), 128, True)
    assert (g4 == g8).all() and (c4 == c8).all()


# --------------------------------------------------------------------------
# The reference's own known-answer tests, re-run with the oracle as the kernel
# --------------------------------------------------------------------------
def _glibc_rand_stream():
    """rand() as the reference's test binary sees it on Linux: unseeded (= srand(1)), and the
    INT8 test is the first rand() user in tests/test_kernel.cpp's main()."""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    return libc.rand, 2147483647


def _ref_test_formula(A16, B, Bs, gs, half_away=True):
    """Scalar formula of tests/test_kernel.cpp:260-294 (divide + std::round).  The kernel it
    checks rounds ties to even, so on tie-heavy data (scale = 2^-9 happens with these inputs)
    the formula and the kernel legitimately differ by an LSB per tie; half_away=False is the
    same formula with the kernel's tie rule."""
    M, K = A16.shape
    N = B.shape[0]
    out = np.zeros((M, N), np.float32)
    for m in range(M):
        a = A16[m].astype(np.float32)
        a_scale = max(np.abs(a).max() / np.float32(127.0), np.float32(1e-10))
        v = a / a_scale
        r = np.sign(v) * np.floor(np.abs(v) + 0.5) if half_away else np.rint(v)
        aq = np.clip(r, -128, 127).astype(np.int32)
        for n in range(N):
            acc = np.float32(0)
            for g in range(K // gs):
                gsum = int((aq[g * gs:(g + 1) * gs] * B[n, g * gs:(g + 1) * gs].astype(np.int32)).sum())
                acc += np.float32(gsum) * (a_scale * np.float32(Bs[n, g]))
            out[m, n] = acc
    return out


def test_reference_kat_matmul_int8_glibc_rand():
    """tests/test_kernel.cpp:214-303 with the very inputs its binary draws on Linux."""
    rand, RAND_MAX = _glibc_rand_stream()
    M, K, N, gs = 2, 128, 4, 64
    A16 = np.array([(np.float32(rand()) / np.float32(RAND_MAX) - np.float32(0.5)) * np.float32(0.5)
                    for _ in range(M * K)], np.float32).astype(np.float16).reshape(M, K)
    B = np.array([(rand() % 128) - 64 for _ in range(N * K)], np.int8).reshape(N, K)
    amax = np.abs(B.reshape(N, K // gs, gs).astype(np.float32)).max(axis=2)
    Bs = np.maximum(amax / 127.0, 1e-6).astype(np.float16)
    A, As = O.quantize_activations(A16)
    C = O.matmul_int8(A, As, B, Bs, gs)
    ref = _ref_test_formula(A16, B, Bs.astype(np.float32), gs)
    assert np.abs(f16(C) - ref).max() < 0.1
    # ... and the INT4 test that follows it in the same process (:305-405)
    A16 = np.array([(np.float32(rand()) / np.float32(RAND_MAX) - np.float32(0.5)) * np.float32(0.5)
                    for _ in range(M * K)], np.float32).astype(np.float16).reshape(M, K)
    Bu = np.array([(rand() % 16) - 8 for _ in range(N * K)], np.int8).reshape(N, K)
    Bp = ((Bu[:, 0::2] & 0x0F) | ((Bu[:, 1::2] & 0x0F) << 4)).astype(np.uint8)
    amax = np.abs(Bu.reshape(N, K // gs, gs).astype(np.float32)).max(axis=2)
    Bs = np.maximum(amax / 7.0, 1e-6).astype(np.float16)
    A, As = O.quantize_activations(A16)
    C = O.matmul_int4(A, As, Bp, Bs, gs)
    ref = _ref_test_formula(A16, Bu, Bs.astype(np.float32), gs)
    assert np.abs(f16(C) - ref).max() < 0.1


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reference_kat_matmul_int8(seed):
    """tests/test_kernel.cpp:214-303 on other seeds: M=2,K=128,N=4, group 64, tolerance 0.1
    (with the kernel's tie rule in the formula, see _ref_test_formula)."""
    rng = np.random.default_rng(seed)
    M, K, N, gs = 2, 128, 4, 64
    A16 = ((rng.random((M, K)) - 0.5) * 0.5).astype(np.float16)
    B = (rng.integers(0, 128, (N, K)) - 64).astype(np.int8)
    amax = np.abs(B.reshape(N, K // gs, gs).astype(np.float32)).max(axis=2)
    Bs = np.maximum(amax / 127.0, 1e-6).astype(np.float16)
    A, As = O.quantize_activations(A16)
    C = O.matmul_int8(A, As, B, Bs, gs)
    ref = _ref_test_formula(A16, B, Bs.astype(np.float32), gs, half_away=False)
    assert np.abs(f16(C) - ref).max() < 0.1


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reference_kat_matmul_int4(seed):
    """tests/test_kernel.cpp:305-405: packing lo nibble = even element (:323-329)."""
    rng = np.random.default_rng(seed)
    M, K, N, gs = 2, 128, 4, 64
    A16 = ((rng.random((M, K)) - 0.5) * 0.5).astype(np.float16)
    Bu = (rng.integers(0, 16, (N, K)) - 8).astype(np.int8)
    Bp = ((Bu[:, 0::2] & 0x0F) | ((Bu[:, 1::2] & 0x0F) << 4)).astype(np.uint8)
    amax = np.abs(Bu.reshape(N, K // gs, gs).astype(np.float32)).max(axis=2)
    Bs = np.maximum(amax / 7.0, 1e-6).astype(np.float16)
    A, As = O.quantize_activations(A16)
    C = O.matmul_int4(A, As, Bp, Bs, gs)
    ref = _ref_test_formula(A16, Bu, Bs.astype(np.float32), gs, half_away=False)
    assert np.abs(f16(C) - ref).max() < 0.1


def test_reference_kat_matmul_f16():
    """tests/test_graph.cpp:75-90: [[1,2,3],[4,5,6]] x [[1,2],[3,4],[5,6]] = [[22,28],[49,64]]."""
    a = np.array([[1, 2, 3], [4, 5, 6]], np.float16)
    b = np.array([[1, 2], [3, 4], [5, 6]], np.float16)
    c = O.matmul_f16(a, np.ascontiguousarray(b.T))
    assert (c.view(np.float16) == np.array([[22, 28], [49, 64]], np.float16)).all()


def test_reference_kat_kv_sink_value():
    """tests/test_kv_cache.cpp:108-124: a constant-1.0 key quantizes to within 2 LSB of
    round(1.0/scale)."""
    src = np.full((4, 8, 64), 1.0, np.float16)
    q, s = O.quantize_kv(src)
    expect = int(round(1.0 / float(s[0, 0, 0])))
    assert np.abs(q.astype(np.int32) - expect).max() <= 2
    assert s.shape == (4, 8, 1)


# --------------------------------------------------------------------------
# Self-consistency of the restatement against float64 numpy
# --------------------------------------------------------------------------
def test_quantize_activations_rules():
    rng = np.random.default_rng(3)
    x = rand_act_f16(rng, 4, 264)                       # 264 % 8 == 0
    x[1] = 0                                            # all-zero row -> scale 1e-10, q = 0
    q, s = O.quantize_activations(x)
    xf = x.astype(np.float32)
    sc = np.maximum(np.abs(xf).max(axis=1) / np.float32(127.0), np.float32(1e-10))
    assert (s == sc).all() and s[1] == np.float32(1e-10) and (q[1] == 0).all()
    inv = (np.float32(1.0) / sc)[:, None]
    expect = np.clip(np.rint(np.clip(xf * inv, -128, 127)), -128, 127).astype(np.int8)
    assert (q == expect).all()
    # the count % 8 tail rounds half away from zero (kernel_quants.cpp:189-192)
    t = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 0.5, 1.5, 2.5, 0.5, 1.5, 2.5, -2.5], np.float16)
    got = O.fp16_to_int8(t, 1.0)
    assert got.tolist() == [0, 2, 2, 0, -2, 0, 2, 2, 1, 2, 3, -3]


def test_kv_quant_groups_and_tail():
    rng = np.random.default_rng(4)
    for D in (64, 128, 256, 72, 132):
        x = rng.uniform(-2, 2, (3, 2, D)).astype(np.float16)
        q, s = O.quantize_kv(x)
        G = (D + 127) // 128
        assert s.shape == (3, 2, G)
        for g in range(G):
            seg = x[:, :, g * 128:(g + 1) * 128].astype(np.float32)
            sc = np.maximum(np.abs(seg).max(axis=2) / np.float32(127.0), np.float32(1e-10))
            assert (s[:, :, g] == sc).all()
            deq = q[:, :, g * 128:(g + 1) * 128].astype(np.float32) * sc[:, :, None]
            assert np.abs(deq - seg).max() <= 0.5 * sc.max() * 1.001


@pytest.mark.parametrize("shape", [(1, 256, 8, 128), (3, 384, 13, 128), (5, 128, 4, 64)])
def test_matmul_group_sums_exact(shape):
    M, K, N, gs = shape
    rng = np.random.default_rng(11)
    B, Bs = rand_weights_int8(rng, N, K, gs)
    A, As = O.quantize_activations(rand_act_f16(rng, M, K))
    C, g = O.matmul_int8(A, As, B, Bs, gs, True)
    gref = np.einsum("mgk,ngk->mng", A.reshape(M, K // gs, gs).astype(np.int64),
                     B.reshape(N, K // gs, gs).astype(np.int64))
    assert (g == gref).all()
    ref = (gref * Bs.astype(np.float64)[None]).sum(-1) * As[:, None].astype(np.float64)
    assert np.abs(f16(C) - ref).max() <= 2e-3 * np.abs(ref).max()
    # contracted vs non-contracted scale-accumulate: same to within fp16 rounding
    O.set_fma(0)
    C2 = O.matmul_int8(A, As, B, Bs, gs)
    O.set_fma(1)
    assert np.abs(f16(C) - f16(C2)).max() <= 2e-3 * np.abs(ref).max()


def test_matmul_int4_group_sums_exact():
    M, K, N, gs = 3, 256, 9, 128
    rng = np.random.default_rng(12)
    Bp, Bs = rand_weights_int4(rng, N, K, gs)
    A, As = O.quantize_activations(rand_act_f16(rng, M, K))
    C, g = O.matmul_int4(A, As, Bp, Bs, gs, True)
    lo = (Bp & 0x0F).astype(np.int16); hi = (Bp >> 4).astype(np.int16)
    lo = np.where(lo > 7, lo - 16, lo); hi = np.where(hi > 7, hi - 16, hi)
    Bu = np.empty((N, K), np.int64); Bu[:, 0::2] = lo; Bu[:, 1::2] = hi
    gref = np.einsum("mgk,ngk->mng", A.reshape(M, K // gs, gs).astype(np.int64),
                     Bu.reshape(N, K // gs, gs))
    assert (g == gref).all()


def test_matmul_zero_dims_and_k_tail():
    A = np.zeros((0, 128), np.int8)
    C = O.matmul_int8(A, np.zeros(0, np.float32), np.zeros((4, 128), np.int8),
                      np.ones((4, 1), np.float16), 128)
    assert C.shape == (0, 4)
    # K tail beyond the last whole group is dropped (kernel_gemm.cpp:147)
    rng = np.random.default_rng(2)
    A = rng.integers(-127, 128, (1, 192), dtype=np.int8)
    B = rng.integers(-127, 128, (2, 192), dtype=np.int8)
    Bs = np.ones((2, 1), np.float16)
    C = O.matmul_int8(A, np.ones(1, np.float32) * 1e-3, B, Bs, 128)
    ref = (A[:, :128].astype(np.int64) @ B[:, :128].astype(np.int64).T) * 1e-3
    assert np.abs(f16(C) - ref).max() <= 2e-3 * np.abs(ref).max()


def test_matmul_f16_fp16_accumulation():
    """fp16 lane accumulation loses precision against fp64 but stays close; K tails run."""
    rng = np.random.default_rng(8)
    for K in (16, 24, 29, 64):
        a = rng.uniform(-1, 1, (3, K)).astype(np.float16)
        b = rng.uniform(-1, 1, (5, K)).astype(np.float16)
        c = f16(O.matmul_f16(a, b))
        ref = a.astype(np.float64) @ b.astype(np.float64).T
        assert np.abs(c - ref).max() < 0.02 * max(1.0, np.abs(ref).max())


def _softmax_attention_ref(q, k, v, scale, causal, offset, window=0):
    """float64 softmax attention on [B,S,H,D] / [B,KV,Hkv,D] arrays."""
    B, S, Hq, D = q.shape
    KV, Hkv = k.shape[1], k.shape[2]
    out = np.zeros((B, S, Hq, D))
    for b in range(B):
        for h in range(Hq):
            kh = h // (Hq // Hkv)
            for s in range(S):
                sc = (k[b, :, kh, :] @ q[b, s, h, :]) * scale
                qa = offset + s
                pos = np.arange(KV)
                m = np.ones(KV, bool)
                if causal:
                    m &= pos <= qa
                if 0 < window < KV:
                    m &= ~((pos < qa) & (qa - pos > window))
                if not m.any():
                    continue
                sc = np.where(m, sc, -np.inf)
                p = np.exp(sc - sc.max()); p /= p.sum()
                out[b, s, h] = p @ v[b, :, kh, :]
    return out


@pytest.mark.parametrize("D,cache,new,S", [(128, 70, 1, 1), (64, 70, 5, 5), (256, 33, 3, 3), (128, 0, 4, 4)])
def test_attention_hybrid_vs_float64(D, cache, new, S):
    rng = np.random.default_rng(21)
    B, Hq, Hkv = 1, 4, 2
    q = rng.uniform(-1, 1, (B, S, Hq, D)).astype(np.float16)
    kc16 = rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16)
    vc16 = rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16)
    kq, ks = O.quantize_kv(kc16); vq, vs = O.quantize_kv(vc16)
    kn = rng.uniform(-2, 2, (B, new, Hkv, D)).astype(np.float16)
    vn = rng.uniform(-2, 2, (B, new, Hkv, D)).astype(np.float16)
    scale = 1.0 / np.sqrt(D)
    out = f16(O.attention_hybrid(q, kq, vq, ks, vs, kn, vn, scale, position_offset=cache,
                                 cache_len=cache))
    G = ks.shape[-1]
    kd = kq.astype(np.float64) * np.repeat(ks.astype(np.float64), D // G, axis=-1).reshape(cache, Hkv, D)
    vd = vq.astype(np.float64) * np.repeat(vs.astype(np.float64), D // G, axis=-1).reshape(cache, Hkv, D)
    kall = np.concatenate([kd[None], kn.astype(np.float64)], axis=1)
    vall = np.concatenate([vd[None], vn.astype(np.float64)], axis=1)
    ref = _softmax_attention_ref(q.astype(np.float64), kall, vall, scale, True, cache)
    assert np.abs(out - ref).max() < 2e-3


@pytest.mark.parametrize("window", [0, 16])
def test_attention_f16_vs_float64(window):
    """The polynomial exp (kernel_attention.cpp:174-195) is a few percent off true exp per
    weight, so the comparison with true softmax is loose; exact behaviour is what the GPU
    kernel is compared against."""
    rng = np.random.default_rng(22)
    B, S, Hq, Hkv, D = 2, 37, 4, 2, 64
    q = rng.uniform(-1, 1, (B, S, Hq, D)).astype(np.float16)
    k = rng.uniform(-1, 1, (B, S, Hkv, D)).astype(np.float16)
    v = rng.uniform(-1, 1, (B, S, Hkv, D)).astype(np.float16)
    out = f16(O.attention_f16(q, k, v, 0.0, window_size=window))
    ref = _softmax_attention_ref(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64),
                                 1.0 / np.sqrt(D), True, 0, window)
    assert np.abs(out - ref).max() < 0.05
    # first query row attends to one key only: output is exactly v[0]
    assert (O.attention_f16(q, k, v, 0.0)[:, 0] ==
            np.repeat(v[:, 0], Hq // Hkv, axis=1).view(np.uint16)).all()


def test_attention_threads_do_not_change_results():
    rng = np.random.default_rng(23)
    q = rng.uniform(-1, 1, (1, 9, 4, 64)).astype(np.float16)
    k = rng.uniform(-1, 1, (1, 9, 2, 64)).astype(np.float16)
    v = rng.uniform(-1, 1, (1, 9, 2, 64)).astype(np.float16)
    a = O.attention_f16(q, k, v, 0.0)
    O.set_threads(4)
    b = O.attention_f16(q, k, v, 0.0)
    O.set_threads(1)
    assert (a == b).all()
