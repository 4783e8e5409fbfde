"""GPU parity tests of the ops between the hot kernels of a decoder block (RMS norm, RoPE, SiLU,
GELU, element-wise fp16 binaries, clipped add, embedding), called through the C ABI and compared
with the CPU oracle on the same seeded inputs.

Bar: bit-exact.  The kernels keep the reference's evaluation order (fp32 lane accumulators,
fp16 products, the polynomial exp / rational tanh of the 8-wide paths); the RoPE tables are built
by the host's libm in both the library and the oracle.  Only the scalar tails of SiLU / GELU
(the last n % 8 elements, which the reference evaluates with libm expf / tanhf) go through the
GPU's own expf / tanhf and are held to 1 fp16 ulp.
"""
import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import ulp_diff_f16

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


def halves(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float16)


# ------------------------------------------------------------------------------ RMS norm
@pytest.mark.parametrize("rows,dims", [(1, 1024), (7, 128), (3, 20), (5, 8), (49, 64), (2, 4096), (1, 7),
                                       (33, 136), (4, 2560), (1, 8192), (3, 20000), (5, 16384), (19, 1023)])
@pytest.mark.parametrize("eps", [1e-5, 1e-6])
def test_rms_norm_bit_exact(H, rows, dims, eps):
    """kernel_attention.cpp:484-563: 16 fp32 lane accumulators, fp16 products, fp32 scalar tail."""
    rng = np.random.default_rng(rows * 131 + dims)
    x = halves(rng, (rows, dims), 2.0)
    if rows > 1:
        x[1] = 0                      # zero row: rms = sqrt(eps)
    if rows > 2:
        x[2] *= 40.0                  # large row
    w = (1.0 + 0.1 * rng.standard_normal(dims)).astype(np.float16)
    want = O.rms_norm(x, w, eps)
    got = bits(H.rms_norm(dev(x), dev(w), eps))
    assert np.array_equal(got, want)


def test_rms_norm_reference_known_answer(H):
    """tests/test_graph.cpp:176-200: input 1..8, unit weight, eps 1e-5, tolerance 0.01 there."""
    x = np.arange(1, 9, dtype=np.float16)[None, :]
    w = np.ones(8, np.float16)
    got = H.rms_norm(dev(x), dev(w), 1e-5).cpu().numpy().astype(np.float32)
    inv = 1.0 / np.sqrt(np.float32((x.astype(np.float32) ** 2).sum() / 8 + 1e-5))
    assert np.abs(got - x.astype(np.float32) * inv).max() < 0.01


# ---------------------------------------------------------------------------------- RoPE
@pytest.mark.parametrize("shape,start,theta", [((1, 1, 16, 128), 0, 1e6), ((1, 1, 16, 128), 37, 1e6),
                                               ((2, 3, 4, 64), 5, 10000.0), ((1, 5, 2, 20), 2, 10000.0),
                                               ((1, 2, 1, 4), 0, 10000.0), ((1, 7, 8, 256), 1500, 1e6),
                                               ((1, 300, 2, 64), 0, 10000.0)])
def test_rope_bit_exact(H, shape, start, theta):
    """kernel_attention.cpp:614-676 with the tables of :594-604 (positions start .. start+seq-1);
    head_dim 20 and 4 exercise the scalar tail (half_dim % 8 != 0)."""
    rng = np.random.default_rng(sum(shape) + start)
    x = halves(rng, shape, 1.5)
    want = O.rope(x, theta, start)
    got = bits(H.rope(dev(x), theta, start))
    assert np.array_equal(got, want)


def test_rope_table_grows_and_keeps_rows(H):
    """A later call at a far position grows the cached table; earlier positions give the same bits."""
    rng = np.random.default_rng(9)
    x = halves(rng, (1, 4, 2, 64))
    first = bits(H.rope(dev(x), 5e5, 3))
    far = bits(H.rope(dev(x), 5e5, 40000))
    again = bits(H.rope(dev(x), 5e5, 3))
    assert np.array_equal(first, again)
    assert np.array_equal(far, O.rope(x, 5e5, 40000))
    H.rope_reserve(64, 5e5, 70000)
    assert np.array_equal(bits(H.rope(dev(x), 5e5, 69990)), O.rope(x, 5e5, 69990))


# ---------------------------------------------------------------------------- SiLU / GELU
@pytest.mark.parametrize("name", ["silu", "gelu"])
@pytest.mark.parametrize("n", [8, 4096, 13, 3, 100003, 1 << 20])
def test_unary_vector_part_bit_exact_tail_one_ulp(H, name, n):
    """kernel_nn.cpp:12-91: groups of 8 use the polynomial exp (kernel_utils.h:108-141) / rational
    tanh (:143-163); the last n % 8 elements use expf / tanhf."""
    rng = np.random.default_rng(n)
    x = halves(rng, n, 4.0)
    x[: min(n, 6)] = np.array([0.0, -0.0, 100.0, -100.0, 4.5, -4.6], np.float16)[: min(n, 6)]
    want = getattr(O, name)(x)
    got = bits(getattr(H, name)(dev(x)))
    vend = (n // 8) * 8
    assert np.array_equal(got[:vend], want[:vend])
    if vend < n:
        assert ulp_diff_f16(got[vend:], want[vend:]).max() <= 1


# ------------------------------------------------------------------------ binaries, clipped add
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide"])
@pytest.mark.parametrize("n", [5, 1000, 65536 + 3])
def test_binary_bit_exact(H, op, n):
    """kernel_blas.cpp:187-343: one fp16 operation per element."""
    rng = np.random.default_rng(n + len(op))
    a, b = halves(rng, n, 8.0), halves(rng, n, 0.5)
    sp = np.array([0.0, 65504.0, -65504.0, 6e-8, 6.1e-5, 1.0], np.float16)
    a[: min(n, 5)] = sp[: min(n, 5)]
    b[: min(n, 5)] = sp[::-1][: min(n, 5)]
    want = O.binary(op, a, b)
    got = bits(getattr(H, op)(dev(a), dev(b)))
    nan_w = np.isnan(want.view(np.float16)); nan_g = np.isnan(got.view(np.float16))
    assert np.array_equal(nan_w, nan_g)
    assert np.array_equal(got[~nan_w], want[~nan_w])


def test_binary_reference_known_answers(H):
    """tests/test_graph.cpp:35-72."""
    f = lambda v: dev(np.array(v, np.float16))
    assert H.add(f([1, 2, 3, 4]), f([5, 6, 7, 8])).cpu().tolist() == [6, 8, 10, 12]
    assert H.subtract(f([10, 8, 6, 4]), f([2, 3, 1, 2])).cpu().tolist() == [8, 5, 5, 2]
    assert H.multiply(f([2, 3, 4, 5]), f([3, 4, 2, 2])).cpu().tolist() == [6, 12, 8, 10]
    assert H.divide(f([12, 15, 8, 9]), f([3, 5, 2, 3])).cpu().tolist() == [4, 3, 4, 3]


@pytest.mark.parametrize("n", [7, 4096 + 5])
def test_add_clipped_bit_exact(H, n):
    """kernel_blas.cpp:224-262: fp32 sum clamped to +-65500 (so never infinite)."""
    rng = np.random.default_rng(n)
    a, b = halves(rng, n, 30000.0), halves(rng, n, 30000.0)
    a[:3] = np.array([65504, -65504, 40000], np.float16)
    b[:3] = np.array([65504, -65504, 30000], np.float16)
    want = O.add_clipped(a, b)
    got = bits(H.add_clipped(dev(a), dev(b)))
    assert np.array_equal(got, want)
    assert np.isfinite(got.view(np.float16)).all()


# ------------------------------------------------------------------------------ embedding
def test_embedding_int8_grouped_bit_exact(H):
    """graph_ops_tensor.cpp:243-254: half(float(q) * float(scale[row, group]))."""
    rng = np.random.default_rng(5)
    vocab, hidden = 300, 512
    table = rng.integers(-127, 128, (vocab, hidden), dtype=np.int8)
    scales = (0.01 + np.abs(rng.standard_normal((vocab, hidden // 128))) * 0.05).astype(np.float16)
    idx = rng.integers(0, vocab, 37).astype(np.int32)
    idx[:2] = [0, vocab - 1]
    want = O.embedding(table, idx, scales, 128)
    got = bits(H.embedding(dev(table), dev(idx), dev(scales), 128))
    assert np.array_equal(got, want)


def test_embedding_f16_and_bounds(H):
    """graph_ops_tensor.cpp:301-316 (copy) and :261-263 (index >= vocab is an error)."""
    rng = np.random.default_rng(6)
    table = halves(rng, (50, 96))
    idx = np.array([3, 49, 0, 3], np.int32)
    got = bits(H.embedding(dev(table), dev(idx)))
    assert np.array_equal(got, table.view(np.uint16)[idx])
    with pytest.raises(H.CactusHipError):
        H.embedding(dev(table), dev(np.array([1, 50], np.int32)))
    with pytest.raises(IndexError):
        O.embedding(table, np.array([1, 50], np.int32))


def test_host_pointer_forms_match_device_forms(H):
    """The kernel.h-shaped host-pointer entries stage through the device and give the same bits."""
    import ctypes as C
    rng = np.random.default_rng(8)
    x = halves(rng, (3, 64)); w = halves(rng, 64)
    out = np.zeros_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    H.lib.cactus_hip_rms_norm_f16(p(x), p(w), p(out), 3, 64, 1e-6)
    assert np.array_equal(out.view(np.uint16), O.rms_norm(x, w, 1e-6))
    q = halves(rng, (1, 2, 2, 32)); qo = np.zeros_like(q)
    H.lib.cactus_hip_rope_f16(p(q), p(qo), 1, 2, 2, 32, 11, 10000.0)
    assert np.array_equal(qo.view(np.uint16), O.rope(q, 10000.0, 11))
    a, b = halves(rng, 40), halves(rng, 40)
    o = np.zeros_like(a)
    H.lib.cactus_hip_silu_f16(p(a), p(o), 40)
    assert np.array_equal(o.view(np.uint16), O.silu(a))
    H.lib.cactus_hip_multiply_f16(p(a), p(b), p(o), 40)
    assert np.array_equal(o.view(np.uint16), O.binary("multiply", a, b))
    H.lib.cactus_hip_add_f16_clipped(p(a), p(b), p(o), 40)
    assert np.array_equal(o.view(np.uint16), O.add_clipped(a, b))


# ------------------------------------------------------------- sequences fused into one launch
@pytest.mark.parametrize("rows,dims", [(1, 1024), (5, 136), (3, 20), (33, 2560), (2, 20000), (18, 1023)])
def test_add_rms_norm_equals_the_two_calls(H, rows, dims):
    """model_qwen.cpp:125-126: residual add, then the norm of the sum."""
    rng = np.random.default_rng(rows + dims)
    a, b = halves(rng, (rows, dims), 2.0), halves(rng, (rows, dims), 2.0)
    w = (1.0 + 0.1 * rng.standard_normal(dims)).astype(np.float16)
    s, n = H.add_rms_norm(dev(a), dev(b), dev(w), 1e-6)
    want_s = O.binary("add", a, b)
    assert np.array_equal(bits(s), want_s)
    assert np.array_equal(bits(n), O.rms_norm(want_s, w, 1e-6))


@pytest.mark.parametrize("B,S,Hq,Hkv,D,start", [(1, 1, 16, 8, 128, 512), (1, 7, 4, 2, 64, 0), (2, 3, 2, 1, 20, 9),
                                                (1, 300, 16, 8, 128, 100), (1, 1, 8, 4, 256, 3)])
def test_qk_norm_rope_equals_the_four_calls(H, B, S, Hq, Hkv, D, start):
    """model_qwen.cpp:63-81: per-head RMS norm of q and k, then RoPE of both."""
    rng = np.random.default_rng(B + S + D)
    q, k = halves(rng, (B, S, Hq, D), 3.0), halves(rng, (B, S, Hkv, D), 3.0)
    qw = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float16)
    kw = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float16)
    oq, ok = H.qk_norm_rope(dev(q), dev(k), dev(qw), dev(kw), 1e-6, 1e6, start)
    want_q = O.rope(O.rms_norm(q.reshape(-1, D), qw, 1e-6).reshape(q.shape), 1e6, start)
    want_k = O.rope(O.rms_norm(k.reshape(-1, D), kw, 1e-6).reshape(k.shape), 1e6, start)
    assert np.array_equal(bits(oq), want_q)
    assert np.array_equal(bits(ok), want_k)


@pytest.mark.parametrize("name", ["silu", "gelu"])
@pytest.mark.parametrize("n", [3072, 13, 4096 * 5 + 3])
def test_activation_multiply_equals_the_two_calls(H, name, n):
    """model_qwen.cpp:114-115 (silu) and the Gemma MLP (gelu): activation, rounded, times up."""
    rng = np.random.default_rng(n)
    g, u = halves(rng, n, 3.0), halves(rng, n, 1.0)
    got = bits(getattr(H, name + "_multiply")(dev(g), dev(u)))
    act = bits(getattr(H, name)(dev(g)))                       # the device activation (tail: its own expf / tanhf)
    assert np.array_equal(got, O.binary("multiply", act.view(np.float16), u))
    vend = (n // 8) * 8
    assert np.array_equal(got[:vend], O.binary("multiply", getattr(O, name)(g).view(np.float16), u)[:vend])


@pytest.mark.parametrize("rows,dims", [(1, 1024), (16, 1024), (256, 1024), (5, 640), (3, 100), (2, 7), (4, 4096)])
@pytest.mark.parametrize("with_add", [True, False])
def test_add_rms_norm_quant_equals_norm_then_activation_quant(H, rows, dims, with_add):
    """The norm that feeds q/k/v or gate/up and the row quantization of that matmul's input
    (graph_ops_nn.cpp:35-66), one launch: sum, norm, int8 row and scale all bit-exact."""
    rng = np.random.default_rng(rows * 7 + dims + with_add)
    a, b = halves(rng, (rows, dims), 2.0), halves(rng, (rows, dims), 2.0)
    w = (1.0 + 0.1 * rng.standard_normal(dims)).astype(np.float16)
    if rows > 1:
        a[1] = 0; b[1] = 0                                    # an all-zero row: scale floor 1e-10
    s, n, q, sc = H.add_rms_norm_quant(dev(a), dev(b) if with_add else None, dev(w), 1e-6, want_norm=True)
    want_s = O.binary("add", a, b) if with_add else a.view(np.uint16)
    want_n = O.rms_norm(want_s, w, 1e-6)
    wq, ws = O.quantize_activations(want_n)
    if with_add:
        assert np.array_equal(bits(s), want_s)
    else:
        assert s is None
    assert np.array_equal(bits(n), want_n)
    assert np.array_equal(q.cpu().numpy(), wq) and np.array_equal(sc.cpu().numpy(), ws)
    # the norm output is optional
    _, n2, q2, sc2 = H.add_rms_norm_quant(dev(a), dev(b) if with_add else None, dev(w), 1e-6)
    assert n2 is None and torch.equal(q2, q) and torch.equal(sc2, sc)


@pytest.mark.parametrize("name", ["silu", "gelu"])
@pytest.mark.parametrize("rows,dims", [(1, 3072), (16, 3072), (256, 3072), (3, 2048), (5, 13), (2, 100)])
def test_activation_multiply_quant_equals_the_three_calls(H, name, rows, dims):
    """act(gate) * up and the row quantization of the down projection's input, one launch."""
    rng = np.random.default_rng(rows + dims)
    g, u = halves(rng, (rows, dims), 3.0), halves(rng, (rows, dims), 1.0)
    act, q, sc = getattr(H, name + "_multiply_quant")(dev(g), dev(u), want_act=True)
    want_act = bits(getattr(H, name + "_multiply")(dev(g), dev(u)))      # verified against the oracle above
    assert np.array_equal(bits(act), want_act)
    wq, ws = O.quantize_activations(want_act)
    assert np.array_equal(q.cpu().numpy(), wq) and np.array_equal(sc.cpu().numpy(), ws)
    if (rows * dims) % 8 == 0:                                            # no libm tail: the oracle end to end
        oa = O.binary("multiply", getattr(O, name)(g.reshape(-1)).view(np.float16), u.reshape(-1)).reshape(rows, dims)
        assert np.array_equal(want_act, oa)
    act2, q2, _ = getattr(H, name + "_multiply_quant")(dev(g), dev(u))
    assert act2 is None and torch.equal(q2, q)


def test_row_quant_argument_errors(H):
    import ctypes as C
    x = dev(np.zeros((1, 8), np.float16))
    q = torch.empty((1, 8), dtype=torch.int8, device="cuda")
    sc = torch.empty((1,), dtype=torch.float32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    # an addend needs somewhere to put the sum
    assert H.lib.cactus_hip_add_rms_norm_quant_f16_dev(p(x), p(x), p(x), None, None, p(q), p(sc), 1, 8, 1e-6, None) != 0
    assert H.lib.cactus_hip_add_rms_norm_quant_f16_dev(p(x), None, p(x), None, None, None, p(sc), 1, 8, 1e-6, None) != 0
    assert H.lib.cactus_hip_silu_multiply_quant_f16_dev(p(x), None, None, p(q), p(sc), 1, 8, None) != 0
    assert H.lib.cactus_hip_add_rms_norm_quant_f16_dev(p(x), None, p(x), None, None, p(q), p(sc), 1, 20000, 1e-6, None) != 0
    assert H.lib.cactus_hip_add_rms_norm_quant_f16_dev(None, None, None, None, None, None, None, 0, 8, 1e-6, None) == 0


def test_quantize_kv_pair_equals_two_calls(H):
    """engine_cache.cpp:253-263: K then V, written straight into cache-slot views."""
    rng = np.random.default_rng(4)
    k, v = halves(rng, (3, 8, 128), 2.0), halves(rng, (3, 8, 128), 2.0)
    kc = torch.zeros((10, 8, 128), dtype=torch.int8, device="cuda"); vc = torch.zeros_like(kc)
    ks = torch.zeros((10, 8, 1), dtype=torch.float32, device="cuda"); vs = torch.zeros_like(ks)
    H.quantize_kv_pair(dev(k), dev(v), kc[4:7], vc[4:7], ks[4:7], vs[4:7])
    okq, oks = O.quantize_kv(k); ovq, ovs = O.quantize_kv(v)
    assert np.array_equal(kc[4:7].cpu().numpy(), okq) and np.array_equal(vc[4:7].cpu().numpy(), ovq)
    assert np.array_equal(ks[4:7].cpu().numpy(), oks) and np.array_equal(vs[4:7].cpu().numpy(), ovs)
    assert not kc[:4].any() and not kc[7:].any()


# --------------------------------------------------------------------------- greedy sampler
@pytest.mark.parametrize("rows,vocab", [(1, 151936), (3, 1000), (5, 1), (2, 1025), (256, 4096), (3, 20000),
                                        (64, 16384), (65, 16384)])
def test_sample_greedy_first_maximum(H, rows, vocab):
    """kernel_nn.cpp:450-466: strict '>' scan from element 0 -- ties to the lowest index."""
    rng = np.random.default_rng(rows + vocab)
    x = halves(rng, (rows, vocab), 3.0)
    if vocab >= 1000:
        x[0, 700] = x[0, 77] = np.float16(60000.0)         # tie: index 77 wins
        if vocab >= 20000:
            x[0, 19000] = np.float16(60000.0)              # a tie in another workgroup's chunk
        x[-1, 5] = np.float16(np.nan)                      # NaNs are never greater
        if rows > 1:
            x[1, :] = np.float16(-np.inf)                  # nothing exceeds element 0
    got = H.sample_greedy(dev(x)).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, O.sample_greedy(x))
    if vocab >= 1000:
        assert got[0] == 77


def test_sample_greedy_nan_first_element(H):
    x = np.array([[np.nan, 5.0, 7.0, 1.0]], np.float16)
    assert H.sample_greedy(dev(x)).cpu().tolist() == [0] and O.sample_greedy(x).tolist() == [0]


def test_sample_greedy_twice_in_a_row_and_all_equal(H):
    """The split form cleans its scratch for the next call; equal values and signed zeros keep index 0."""
    x = np.zeros((2, 40000), np.float16)
    x[1, 1::2] = np.float16(-0.0)
    for _ in range(3):
        assert H.sample_greedy(dev(x)).cpu().tolist() == [0, 0]
    x[0, 39999] = 1.0
    assert H.sample_greedy(dev(x)).cpu().tolist() == [39999, 0]
    x[:, 0] = np.float16(np.nan)
    assert H.sample_greedy(dev(x)).cpu().tolist() == [0, 0]


def test_kv_append_batched_equals_per_sequence_quantization(H):
    """engine_cache.cpp:234-267 per sequence: the new token's K and V land in that sequence's slot."""
    rng = np.random.default_rng(12)
    B, slots, Hkv, D = 5, 9, 2, 128
    k, v = halves(rng, (B, Hkv, D), 2.0), halves(rng, (B, Hkv, D), 2.0)
    kc = torch.zeros((B, slots, Hkv, D), dtype=torch.int8, device="cuda"); vc = torch.zeros_like(kc)
    ks = torch.zeros((B, slots, Hkv, 1), dtype=torch.float32, device="cuda"); vs = torch.zeros_like(ks)
    where = np.array([0, 8, 3, 9, 5], np.int32)            # 9 is out of range: nothing stored
    H.kv_append_batched(dev(k), dev(v), kc, vc, ks, vs, dev(where))
    okq, oks = O.quantize_kv(k); ovq, ovs = O.quantize_kv(v)
    kc, vc, ks, vs = kc.cpu().numpy(), vc.cpu().numpy(), ks.cpu().numpy(), vs.cpu().numpy()
    for b in range(B):
        if where[b] < slots:
            assert np.array_equal(kc[b, where[b]], okq[b]) and np.array_equal(vc[b, where[b]], ovq[b])
            assert np.array_equal(ks[b, where[b]], oks[b]) and np.array_equal(vs[b, where[b]], ovs[b])
            kc[b, where[b]] = 0; vc[b, where[b]] = 0; ks[b, where[b]] = 0; vs[b, where[b]] = 0
    assert not kc.any() and not vc.any() and not ks.any() and not vs.any()


# --------------------------------------------------------------------------- q/k norm + RoPE + cache append
@pytest.mark.parametrize("B,S,Hq,Hkv,D,gs,start", [(1, 1, 16, 8, 128, 128, 512), (3, 1, 16, 8, 128, 128, 9),
                                                   (2, 3, 4, 2, 64, 128, 0), (1, 1, 4, 1, 256, 128, 77),
                                                   (2, 2, 2, 1, 20, 128, 5), (1, 5, 3, 3, 192, 64, 1),
                                                   (256, 1, 16, 8, 128, 128, 512)])
def test_qk_norm_rope_kv_append_equals_the_separate_launches(H, B, S, Hq, Hkv, D, gs, start):
    """model_qwen.cpp:63-81 followed by the cache append of engine_cache.cpp:234-267: q / k outputs as
    qk_norm_rope, cache slots as cactus_quantize_kv_fp16_to_int8 of the rotated k and of v."""
    rng = np.random.default_rng(B * 31 + S + D + gs)
    q, k = halves(rng, (B, S, Hq, D), 3.0), halves(rng, (B, S, Hkv, D), 3.0)
    v = halves(rng, (B, S, Hkv, D), 2.0)
    qw = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float16)
    kw = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float16)
    G = (D + gs - 1) // gs
    max_slots = 12
    slots = rng.integers(0, max_slots - S + 1, B).astype(np.int32)
    if B > 2:
        slots[1] = max_slots        # out of range: nothing is stored for that sequence
        slots[2] = -1
    kc = torch.full((B, max_slots, Hkv, D), 99, dtype=torch.int8, device="cuda"); vc = torch.full_like(kc, 99)
    ks = torch.full((B, max_slots, Hkv, G), -1.0, dtype=torch.float32, device="cuda"); vs = torch.full_like(ks, -1.0)
    oq, ok = H.qk_norm_rope_kv_append(dev(q), dev(k), dev(v), dev(qw), dev(kw), 1e-6, 1e6, start, kc, vc, ks, vs,
                                      dev(slots), gs)
    want_q = O.rope(O.rms_norm(q.reshape(-1, D), qw, 1e-6).reshape(q.shape), 1e6, start)
    want_k = O.rope(O.rms_norm(k.reshape(-1, D), kw, 1e-6).reshape(k.shape), 1e6, start)
    assert np.array_equal(bits(oq), want_q) and np.array_equal(bits(ok), want_k)
    kc, vc, ks, vs = kc.cpu().numpy(), vc.cpu().numpy(), ks.cpu().numpy(), vs.cpu().numpy()
    for b in range(B):
        wkq, wks = O.quantize_kv(want_k[b].view(np.float16), gs)
        wvq, wvs = O.quantize_kv(v[b], gs)
        ok_slot = 0 <= slots[b] and slots[b] + S <= max_slots
        for s in range(max_slots):
            inside = ok_slot and slots[b] <= s < slots[b] + S
            if inside:
                j = s - slots[b]
                assert np.array_equal(kc[b, s], wkq[j]) and np.array_equal(vc[b, s], wvq[j]), (b, s)
                assert np.array_equal(ks[b, s], wks[j].reshape(Hkv, G)) and np.array_equal(vs[b, s], wvs[j].reshape(Hkv, G))
            else:
                assert (kc[b, s] == 99).all() and (vc[b, s] == 99).all() and (ks[b, s] == -1).all(), (b, s)


def test_qk_norm_rope_kv_append_row_for_row(H):
    """slots NULL: the quantized rows land row for row (the quantize_kv_pair layout)."""
    rng = np.random.default_rng(12)
    B, S, Hq, Hkv, D = 1, 4, 4, 2, 128
    q, k, v = halves(rng, (B, S, Hq, D), 3.0), halves(rng, (B, S, Hkv, D), 3.0), halves(rng, (B, S, Hkv, D), 2.0)
    qw = np.ones(D, np.float16); kw = np.ones(D, np.float16)
    kc = torch.zeros((B * S, Hkv, D), dtype=torch.int8, device="cuda"); vc = torch.zeros_like(kc)
    ks = torch.zeros((B * S, Hkv, 1), dtype=torch.float32, device="cuda"); vs = torch.zeros_like(ks)
    _, ok = H.qk_norm_rope_kv_append(dev(q), dev(k), dev(v), dev(qw), dev(kw), 1e-6, 1e4, 3, kc, vc, ks, vs)
    wkq, wks = O.quantize_kv(bits(ok).reshape(S, Hkv, D).view(np.float16))
    wvq, wvs = O.quantize_kv(v.reshape(S, Hkv, D))
    assert np.array_equal(kc.cpu().numpy(), wkq) and np.array_equal(vc.cpu().numpy(), wvq)
    assert np.array_equal(ks.cpu().numpy().reshape(-1), wks.reshape(-1)) and np.array_equal(vs.cpu().numpy().reshape(-1), wvs.reshape(-1))


# ------------------------------------------------------------------------------ scalar ops
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide"])
@pytest.mark.parametrize("n", [1, 7, 8, 4099, 640 * 33])
def test_scalar_op_bit_exact(H, op, n):
    """kernel_scalar.cpp:7-154: scalar rounded to half, one fp16 operation per element."""
    rng = np.random.default_rng(n)
    x = halves(rng, n, 6.0)
    for value in (25.298221, -0.3333, 2.0):
        assert np.array_equal(bits(H.scalar_op(op, dev(x), value)), O.scalar_op(op, x, value)), (op, value)


def test_scalar_op_graph_known_answers(H):
    """tests/test_graph.cpp:10-33, 120-150 through the graph nodes; the transcendental scalar ops
    are not on the device."""
    import cactus_amd.graph as G
    g = G.CactusGraph()
    a, b = g.input((2, 3), G.FP16), g.input((2, 3), G.FP16)
    r = g.scalar_multiply(g.multiply(g.add(a, b), a), 2.0)
    c = g.input((4,), G.FP16)
    r2 = g.scalar_multiply(g.scalar_add(c, 5.0), 2.0)
    r3, r4 = g.scalar_subtract(c, 2.0), g.scalar_divide(c, 2.0)
    da, db = np.array([[1, 2, 3], [4, 5, 6]], np.float16), np.array([[2, 3, 4], [5, 6, 7]], np.float16)
    g.set_input(a, da); g.set_input(b, db); g.set_input(c, np.array([10, 8, 6, 4], np.float16))
    g.execute()
    assert np.array_equal(g.get_output(r), ((da + db) * da * np.float16(2)).astype(np.float16))
    assert g.get_output(r2).tolist() == [30, 26, 22, 18]
    assert g.get_output(r3).tolist() == [8, 6, 4, 2] and g.get_output(r4).tolist() == [5, 4, 3, 2]
    g.close()
    with pytest.raises(H.CactusHipError):
        H.scalar_op("exp", dev(np.ones(8, np.float16)), 0.0)
