"""ctypes binding of the CPU restatement (oracle/libcactus_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by cactus_amd/.  Arrays are numpy; halves
travel as uint16 bit patterns (use .view(np.float16) to look at them).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcactus_oracle.so")


def build():
    """Compile the restatement with gcc (oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    sz, vp = C.c_size_t, C.c_void_p
    lib.oracle_h2f.restype = C.c_float
    lib.oracle_h2f.argtypes = [C.c_uint16]
    lib.oracle_f2h.restype = C.c_uint16
    lib.oracle_f2h.argtypes = [C.c_float]
    lib.oracle_set_fma.argtypes = [C.c_int]
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_matmul_int8.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz, vp]
    lib.oracle_matmul_int4.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz, vp]
    lib.oracle_matmul_f16.argtypes = [vp, vp, vp, sz, sz, sz]
    lib.oracle_fp16_max_abs.restype = C.c_float
    lib.oracle_fp16_max_abs.argtypes = [vp, sz]
    lib.oracle_fp16_to_int8.argtypes = [vp, vp, sz, C.c_float]
    lib.oracle_quantize_activations.argtypes = [vp, vp, vp, sz, sz]
    lib.oracle_quantize_kv_fp16_to_int8.argtypes = [vp, vp, vp, sz, sz, sz, sz]
    lib.oracle_unpack_int4_to_int8.argtypes = [vp, vp, sz]
    lib.oracle_attention_f16.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz, sz, sz, C.c_float,
                                         vp, sz, sz, C.c_int]
    lib.oracle_attention_hybrid_int8_fp16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp,
                                                      sz, sz, sz, sz, sz, sz, sz, C.c_float,
                                                      sz, C.c_int, sz]
    lib.oracle_rms_norm_f16.argtypes = [vp, vp, vp, sz, sz, C.c_float]
    lib.oracle_rope_tables.argtypes = [vp, vp, sz, sz, sz, C.c_float]
    lib.oracle_rope_f16.argtypes = [vp, vp, sz, sz, sz, sz, sz, C.c_float]
    lib.oracle_silu_f16.argtypes = [vp, vp, sz, sz]
    lib.oracle_gelu_f16.argtypes = [vp, vp, sz, sz]
    lib.oracle_binary_f16.argtypes = [C.c_int, vp, vp, vp, sz]
    lib.oracle_add_f16_clipped.argtypes = [vp, vp, vp, sz]
    lib.oracle_embedding_int8.restype = C.c_int
    lib.oracle_embedding_int8.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz]
    lib.oracle_embedding_f16.restype = C.c_int
    lib.oracle_embedding_f16.argtypes = [vp, vp, vp, sz, sz, sz]
    lib.oracle_sample_greedy_f16.restype = C.c_uint32
    lib.oracle_sample_greedy_f16.argtypes = [vp, sz]
    lib.oracle_sample_f16.restype = C.c_uint32
    lib.oracle_sample_f16.argtypes = [vp, sz, C.c_float, C.c_float, sz, C.c_float, vp, vp]
    lib.oracle_sample_uniform.restype = C.c_float
    lib.oracle_sample_uniform.argtypes = [C.c_uint32]
    lib.oracle_logits_entropy_f16.restype = C.c_float
    lib.oracle_logits_entropy_f16.argtypes = [vp, sz]
    lib.oracle_scalar_op_f16.argtypes = [vp, vp, sz, C.c_float, C.c_int]
    lib.oracle_conv1d_causal_depthwise_f16.argtypes = [vp, vp, vp, sz, sz, sz, sz, sz]
    lib.oracle_lfm2_conv_gate_f16.argtypes = [vp, vp, vp, vp, vp, vp, sz, sz, sz, sz]
    return lib


lib = _load()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16 and dt == np.uint16:
        a = a.view(np.uint16)
    assert a.dtype == dt, (a.dtype, dt)
    return a


def set_fma(fused):
    lib.oracle_set_fma(int(bool(fused)))


def set_threads(n):
    lib.oracle_set_threads(int(n))


def h2f(bits):
    return lib.oracle_h2f(int(bits))


def f2h(x):
    return lib.oracle_f2h(float(x))


def matmul_int8(A, A_scales, B, B_scales, group_size, want_group_sums=False):
    A = _c(A, np.int8); B = _c(B, np.int8)
    As = _c(A_scales, np.float32); Bs = _c(B_scales, np.uint16)
    M, K = A.shape; N = B.shape[0]
    Cout = np.zeros((M, N), np.uint16)
    gs = np.zeros((M, N, K // group_size), np.int32) if want_group_sums else None
    lib.oracle_matmul_int8(_p(A), _p(As), _p(B), _p(Bs), _p(Cout), M, K, N, group_size, _p(gs))
    return (Cout, gs) if want_group_sums else Cout


def matmul_int4(A, A_scales, B_packed, B_scales, group_size, want_group_sums=False):
    A = _c(A, np.int8); Bp = _c(B_packed, np.uint8)
    As = _c(A_scales, np.float32); Bs = _c(B_scales, np.uint16)
    M, K = A.shape; N = Bp.shape[0]
    assert Bp.shape[1] * 2 == K
    Cout = np.zeros((M, N), np.uint16)
    gs = np.zeros((M, N, K // group_size), np.int32) if want_group_sums else None
    lib.oracle_matmul_int4(_p(A), _p(As), _p(Bp), _p(Bs), _p(Cout), M, K, N, group_size, _p(gs))
    return (Cout, gs) if want_group_sums else Cout


def matmul_f16(a, bT):
    a = _c(a, np.uint16); bT = _c(bT, np.uint16)
    M, K = a.shape; N = bT.shape[0]
    c = np.zeros((M, N), np.uint16)
    lib.oracle_matmul_f16(_p(a), _p(bT), _p(c), M, K, N)
    return c


def fp16_max_abs(src):
    src = _c(src, np.uint16)
    return lib.oracle_fp16_max_abs(_p(src), src.size)


def fp16_to_int8(src, scale):
    src = _c(src, np.uint16)
    dst = np.zeros(src.shape, np.int8)
    lib.oracle_fp16_to_int8(_p(src), _p(dst), src.size, float(scale))
    return dst


def quantize_activations(src):
    src = _c(src, np.uint16)
    M, K = src.shape
    dst = np.zeros((M, K), np.int8)
    scales = np.zeros(M, np.float32)
    lib.oracle_quantize_activations(_p(src), _p(dst), _p(scales), M, K)
    return dst, scales


def quantize_kv(src, group_size=128):
    src = _c(src, np.uint16)
    S, H, D = src.shape
    G = (D + group_size - 1) // group_size
    dst = np.zeros((S, H, D), np.int8)
    scales = np.zeros((S, H, G), np.float32)
    lib.oracle_quantize_kv_fp16_to_int8(_p(src), _p(dst), _p(scales), S, H, D, group_size)
    return dst, scales


def unpack_int4(packed, count):
    packed = _c(packed, np.uint8)
    out = np.zeros(count, np.int8)
    lib.oracle_unpack_int4_to_int8(_p(packed), _p(out), count)
    return out


def attention_f16(q, k, v, scale, mask=None, position_offset=0, window_size=0, is_causal=True):
    q = _c(q, np.uint16); k = _c(k, np.uint16); v = _c(v, np.uint16)
    B, S, Hq, D = q.shape
    KV, Hkv = k.shape[1], k.shape[2]
    m = None if mask is None else _c(mask, np.uint16)
    out = np.zeros(q.shape, np.uint16)
    lib.oracle_attention_f16(_p(q), _p(k), _p(v), _p(out), B, S, KV, Hq, Hkv, D, float(scale),
                             _p(m), position_offset, window_size, int(bool(is_causal)))
    return out


def attention_hybrid(q, k_cached, v_cached, k_scales, v_scales, k_new, v_new, scale,
                     position_offset=0, is_causal=True, group_size=128, cache_len=None):
    q = _c(q, np.uint16); kn = _c(k_new, np.uint16); vn = _c(v_new, np.uint16)
    kc = _c(k_cached, np.int8); vc = _c(v_cached, np.int8)
    ks = _c(k_scales, np.float32); vs = _c(v_scales, np.float32)
    B, S, Hq, D = q.shape
    new_len, Hkv = kn.shape[1], kn.shape[2]
    if cache_len is None:
        cache_len = kc.shape[-3]
    out = np.zeros(q.shape, np.uint16)
    lib.oracle_attention_hybrid_int8_fp16(_p(q), _p(kc), _p(vc), _p(ks), _p(vs), _p(kn), _p(vn),
                                          _p(out), B, S, cache_len, new_len, Hq, Hkv, D,
                                          float(scale), position_offset, int(bool(is_causal)),
                                          group_size)
    return out


def rms_norm(x, weight, eps=1e-5):
    x = _c(x, np.uint16); w = _c(weight, np.uint16)
    rows, dims = x.shape
    out = np.zeros(x.shape, np.uint16)
    lib.oracle_rms_norm_f16(_p(x), _p(w), _p(out), rows, dims, float(eps))
    return out


def rope_tables(pos0, npos, head_dim, theta):
    c = np.zeros((npos, head_dim // 2), np.uint16)
    s = np.zeros((npos, head_dim // 2), np.uint16)
    lib.oracle_rope_tables(_p(c), _p(s), pos0, npos, head_dim, float(theta))
    return c, s


def rope(x, theta, start_pos=0):
    x = _c(x, np.uint16)
    B, S, H, D = x.shape
    out = np.zeros(x.shape, np.uint16)
    lib.oracle_rope_f16(_p(x), _p(out), B, S, H, D, start_pos, float(theta))
    return out


def silu(x, ranges=1):
    x = _c(x, np.uint16)
    out = np.zeros(x.shape, np.uint16)
    lib.oracle_silu_f16(_p(x), _p(out), x.size, ranges)
    return out


def gelu(x, ranges=1):
    x = _c(x, np.uint16)
    out = np.zeros(x.shape, np.uint16)
    lib.oracle_gelu_f16(_p(x), _p(out), x.size, ranges)
    return out


BINARY_OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3}


def binary(op, a, b):
    a = _c(a, np.uint16); b = _c(b, np.uint16)
    out = np.zeros(a.shape, np.uint16)
    lib.oracle_binary_f16(BINARY_OPS[op], _p(a), _p(b), _p(out), a.size)
    return out


def add_clipped(a, b):
    a = _c(a, np.uint16); b = _c(b, np.uint16)
    out = np.zeros(a.shape, np.uint16)
    lib.oracle_add_f16_clipped(_p(a), _p(b), _p(out), a.size)
    return out


def embedding(table, indices, scales=None, group_size=128):
    idx = _c(indices, np.int32)
    vocab, hidden = table.shape
    out = np.zeros((idx.size, hidden), np.uint16)
    if table.dtype == np.int8:
        rc = lib.oracle_embedding_int8(_p(_c(table, np.int8)), _p(_c(scales, np.uint16)), _p(idx), _p(out),
                                       idx.size, vocab, hidden, group_size)
    else:
        rc = lib.oracle_embedding_f16(_p(_c(table, np.uint16)), _p(idx), _p(out), idx.size, vocab, hidden)
    if rc != 0:
        raise IndexError("Embedding index out of bounds")
    return out


def sample_greedy(logits):
    logits = _c(logits, np.uint16)
    return np.array([lib.oracle_sample_greedy_f16(_p(r), r.size) for r in logits.reshape(-1, logits.shape[-1])],
                    np.uint32)


def sample(logits, temperature, top_p, top_k, draw, history=None):
    """One row.  history: list of earlier tokens (the reference's process-wide token_history); the
    updated list is returned beside the token."""
    logits = _c(logits, np.uint16).reshape(-1)
    h = np.zeros(128, np.uint32)
    n = 0 if history is None else len(history)
    h[:n] = history if n else []
    cnt = C.c_size_t(n)
    tok = lib.oracle_sample_f16(_p(logits), logits.size, temperature, top_p, top_k, draw, _p(h), C.byref(cnt))
    return int(tok), h[:cnt.value].tolist()


def sample_uniform(seed):
    return float(lib.oracle_sample_uniform(seed & 0xFFFFFFFF))


def logits_entropy(logits):
    logits = _c(logits, np.uint16).reshape(-1)
    return float(lib.oracle_logits_entropy_f16(_p(logits), logits.size))


def conv1d_causal_depthwise(x, weight, dilation=1):
    x = _c(x, np.uint16); w = _c(weight, np.uint16)
    N, L, Cc = x.shape
    K = w.shape[-1]
    out = np.zeros(x.shape, np.uint16)
    lib.oracle_conv1d_causal_depthwise_f16(_p(x), _p(w), _p(out), N, L, Cc, K, dilation)
    return out


class Lfm2ConvState:
    """One sequence's ConvCache layer: circular rows [W, C], head, count."""

    def __init__(self, W, Cc):
        self.rows = np.zeros((max(W, 1), Cc), np.uint16)
        self.W = W
        self.head = C.c_size_t(0)
        self.count = C.c_size_t(0)


def lfm2_conv_gate(in_proj, weight, state):
    """in_proj [L, 3*C] of one sequence, weight [C, K]; state: Lfm2ConvState (updated)."""
    ip = _c(in_proj, np.uint16); w = _c(weight, np.uint16)
    L, C3 = ip.shape
    Cc = C3 // 3
    out = np.zeros((L, Cc), np.uint16)
    lib.oracle_lfm2_conv_gate_f16(_p(ip), _p(w), _p(state.rows), C.byref(state.head), C.byref(state.count),
                                  _p(out), L, Cc, w.shape[-1], state.W)
    return out


def scalar_op(op, x, value):
    x = _c(x, np.uint16)
    out = np.zeros(x.shape, np.uint16)
    lib.oracle_scalar_op_f16(_p(x), _p(out), x.size, float(value), BINARY_OPS[op])
    return out
