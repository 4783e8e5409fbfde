"""A Qwen3-style decoder block (models/model_qwen.cpp:51-129) built from graph nodes and run on
the device: prefill over 7 tokens into the INT8 KV cache, then one decode step that reads it.
The CPU oracle runs the same op sequence.  Everything up to the attention inputs is bit-exact;
attention sums in a different order, so what depends on it is held to an fp16 tolerance.
"""
import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import rand_weights_int8

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

H, HQ, HKV, D, FF, VOCAB = 256, 4, 2, 64, 512, 97
EPS, THETA = 1e-6, 1e6


@pytest.fixture(scope="module")
def G():
    import cactus_amd.hip as hip
    hip.init(0)
    import cactus_amd.graph as graph
    return graph


def f16(bits):
    return np.ascontiguousarray(bits).view(np.float16)


def make_weights(rng):
    w = {}
    for name, (n, k) in {"q": (HQ * D, H), "k": (HKV * D, H), "v": (HKV * D, H), "o": (H, HQ * D),
                         "gate": (FF, H), "up": (FF, H), "down": (H, FF)}.items():
        B, Bs = rand_weights_int8(rng, n, k)
        w[name] = (B, (f16(Bs).astype(np.float32) * 0.2).astype(np.float16).view(np.uint16))
    for name, n in {"in_norm": H, "post_norm": H, "q_norm": D, "k_norm": D}.items():
        w[name] = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float16)
    w["emb"] = rng.integers(-127, 128, (VOCAB, H), dtype=np.int8)
    w["emb_s"] = (0.01 + 0.02 * np.abs(rng.standard_normal((VOCAB, H // 128)))).astype(np.float16)
    return w


def oracle_mm(x_bits, wt):
    q, s = O.quantize_activations(x_bits)
    return O.matmul_int8(q, s, wt[0], wt[1], 128)


def oracle_block(w, tokens, pos, cache):
    """One block on `tokens` at positions pos..; cache = None (prefill, attention_f16) or the
    quantized K/V of earlier tokens.  Returns the named intermediate tensors (uint16 bits)."""
    S = len(tokens)
    t = {}
    t["hidden"] = O.embedding(w["emb"], np.asarray(tokens, np.int32), w["emb_s"], 128)
    t["normed"] = O.rms_norm(t["hidden"], w["in_norm"], EPS)
    q = oracle_mm(t["normed"], w["q"]); k = oracle_mm(t["normed"], w["k"]); v = oracle_mm(t["normed"], w["v"])
    q = O.rms_norm(q.reshape(S * HQ, D), w["q_norm"], EPS).reshape(1, S, HQ, D)
    k = O.rms_norm(k.reshape(S * HKV, D), w["k_norm"], EPS).reshape(1, S, HKV, D)
    v = v.reshape(1, S, HKV, D)
    t["q_rope"], t["k_rope"], t["v"] = O.rope(q, THETA, pos), O.rope(k, THETA, pos), v
    scale = 1.0 / np.sqrt(D)
    if cache is None:
        attn = O.attention_f16(t["q_rope"], t["k_rope"], v, scale, position_offset=pos)
    else:
        kq, ks, vq, vs = cache
        attn = O.attention_hybrid(t["q_rope"], kq[None], vq[None], ks[None], vs[None], t["k_rope"], v, scale,
                                  position_offset=pos)
    t["attn"] = attn
    o = oracle_mm(attn.reshape(S, HQ * D), w["o"])
    t["after_attn"] = O.binary("add", t["hidden"], o)
    n2 = O.rms_norm(t["after_attn"], w["post_norm"], EPS)
    gate, up = oracle_mm(n2, w["gate"]), oracle_mm(n2, w["up"])
    t["gated"] = O.binary("multiply", O.silu(gate), up)
    t["out"] = O.binary("add", t["after_attn"], oracle_mm(t["gated"], w["down"]))
    return t


def graph_block(G, g, wn, tokens, pos, cache):
    S = len(tokens)
    t = {}
    idx = g.input((S,), G.FP32)
    g.set_input(idx, np.asarray(tokens, np.float32))
    t["hidden"] = g.embedding(wn["emb"], idx)
    t["normed"] = g.rms_norm(t["hidden"], wn["in_norm"], EPS)
    q = g.matmul(t["normed"], wn["q"], True); k = g.matmul(t["normed"], wn["k"], True)
    v = g.matmul(t["normed"], wn["v"], True)
    q = g.reshape(g.rms_norm(g.reshape(q, (S * HQ, D)), wn["q_norm"], EPS), (1, S, HQ, D))
    k = g.reshape(g.rms_norm(g.reshape(k, (S * HKV, D)), wn["k_norm"], EPS), (1, S, HKV, D))
    t["v"] = g.reshape(v, (1, S, HKV, D))
    t["q_rope"], t["k_rope"] = g.rope(q, THETA, pos), g.rope(k, THETA, pos)
    scale = 1.0 / np.sqrt(D)
    if cache is None or cache.is_empty():
        t["attn"] = g.attention(t["q_rope"], t["k_rope"], t["v"], scale, pos)
    else:
        t["attn"] = g.attention_int8_hybrid(t["q_rope"], t["k_rope"], t["v"], scale, pos, cache.get_keys_int8(0),
                                            cache.get_values_int8(0), cache.get_key_scales(0),
                                            cache.get_value_scales(0), cache.get_effective_seq_len(), HKV, D)
    o = g.matmul(g.reshape(t["attn"], (S, HQ * D)), wn["o"], True)
    t["after_attn"] = g.add(t["hidden"], o)
    n2 = g.rms_norm(t["after_attn"], wn["post_norm"], EPS)
    gate, up = g.matmul(n2, wn["gate"], True), g.matmul(n2, wn["up"], True)
    t["gated"] = g.multiply(g.silu(gate), up)
    t["out"] = g.add(t["after_attn"], g.matmul(t["gated"], wn["down"], True))
    return t


def bind_weights(G, g, w):
    wn = {}
    for name in ("q", "k", "v", "o", "gate", "up", "down"):
        B, Bs = w[name]
        n = g.input(B.shape, G.INT8)
        g.set_external_input(n, B)
        g.set_grouped_scales(n, 128, B.shape[1] // 128, Bs)
        wn[name] = n
    for name in ("in_norm", "post_norm", "q_norm", "k_norm"):
        n = g.input(w[name].shape, G.FP16)
        g.set_external_input(n, w[name])
        wn[name] = n
    n = g.input(w["emb"].shape, G.INT8)
    g.set_external_input(n, w["emb"])
    g.set_grouped_scales(n, 128, H // 128, w["emb_s"])
    wn["emb"] = n
    return wn


def close(got, ref_bits, what):
    ref = f16(ref_bits).astype(np.float64).reshape(got.shape)
    err = np.abs(got.astype(np.float64) - ref).max()
    assert err <= 4e-3 + 8e-3 * np.abs(ref).max(), (what, err)


def test_decoder_block_prefill_then_decode(G):
    rng = np.random.default_rng(11)
    w = make_weights(rng)
    g = G.CactusGraph()
    wn = bind_weights(G, g, w)
    cache = G.KVCache(1, 512, HKV, D)

    # ---- prefill: 7 tokens, attention over the FP16 K/V of the same call
    tokens = [5, 17, 96, 0, 33, 33, 2]
    t = graph_block(G, g, wn, tokens, 0, cache)
    g.execute()
    ref = oracle_block(w, tokens, 0, None)
    for name in ("hidden", "normed", "q_rope", "k_rope", "v"):     # no attention upstream: bit-exact
        assert np.array_equal(g.get_output(t[name]).view(np.uint16).ravel(), ref[name].ravel()), name
    for name in ("attn", "after_attn", "gated", "out"):
        close(g.get_output(t[name]), ref[name], name)
    cache.update_from_graph(g, [t["k_rope"]], [t["v"]], len(tokens))
    kq, ks, vq, vs = cache.download(0)
    okq, oks = O.quantize_kv(ref["k_rope"].reshape(len(tokens), HKV, D))
    ovq, ovs = O.quantize_kv(ref["v"].reshape(len(tokens), HKV, D))
    assert np.array_equal(kq, okq) and np.array_equal(vq, ovq)

    # ---- decode: one token at position 7 over the INT8 cache
    g.soft_reset()
    t = graph_block(G, g, wn, [41], len(tokens), cache)
    g.execute()
    ref = oracle_block(w, [41], len(tokens), (okq, oks, ovq, ovs))
    for name in ("hidden", "normed", "q_rope", "k_rope", "v"):
        assert np.array_equal(g.get_output(t[name]).view(np.uint16).ravel(), ref[name].ravel()), name
    for name in ("attn", "after_attn", "gated", "out"):
        close(g.get_output(t[name]), ref[name], name)
    cache.update_from_graph(g, [t["k_rope"]], [t["v"]], 1)
    assert cache.get_effective_seq_len() == 8 and cache.get_total_seq_len() == 8
    g.close()
    cache.close()


def test_block_op_node_errors(G):
    """Messages and conditions of graph_ops_nn.cpp:237-247, 261-268, graph_ops_math.cpp:114-116,
    161-163, graph_ops_tensor.cpp:261-263, graph_builder.cpp:528-530."""
    g = G.CactusGraph()
    x3 = g.input((2, 3, 8), G.FP16)
    w = g.input((8,), G.FP16)
    g.set_input(x3, np.zeros((2, 3, 8), np.float16)); g.set_input(w, np.ones(8, np.float16))
    bad = g.rms_norm(x3, w)
    with pytest.raises(G.CactusHipError, match="RMS normalization requires 2D"):
        g.execute()
    g.hard_reset()
    x2 = g.input((2, 8), G.FP16)
    g.set_input(x2, np.zeros((2, 8), np.float16))
    g.rope(x2, 10000.0)
    with pytest.raises(G.CactusHipError, match="RoPE operation requires 4D"):
        g.execute()
    g.hard_reset()
    a = g.input((2, 8), G.FP16); b = g.input((1, 8), G.FP16)
    with pytest.raises(G.CactusHipError, match="equal shapes"):
        g.add(a, b)
    i8 = g.input((2, 8), G.INT8)
    g.set_input(i8, np.zeros((2, 8), np.int8))
    g.silu(i8)
    g.set_input(a, np.zeros((2, 8), np.float16)); g.set_input(b, np.zeros((1, 8), np.float16))
    with pytest.raises(G.CactusHipError, match="Activation operations only support FP16"):
        g.execute()
    g.hard_reset()
    emb = g.input((4, 3), G.FP16)
    idx = g.input((2,), G.FP32)
    e = g.embedding(emb, idx)
    g.set_input(emb, np.arange(12, dtype=np.float16).reshape(4, 3))
    g.set_input(idx, np.array([3, 1], np.float32))
    g.execute()
    assert g.get_output(e).tolist() == [[9, 10, 11], [3, 4, 5]]
    g.set_input(idx, np.array([1, 4], np.float32))
    with pytest.raises(G.CactusHipError, match="Embedding index out of bounds"):
        g.execute()
    with pytest.raises(G.CactusHipError, match="must be 2D"):
        g.embedding(idx, idx)
    g.close()


def test_reshape_is_a_view_and_binary_kats(G):
    """tests/test_graph.cpp:35-72 through graph nodes; reshape keeps the bytes."""
    g = G.CactusGraph()
    a = g.input((4,), G.FP16); b = g.input((4,), G.FP16)
    s, d, m, q = g.add(a, b), g.subtract(a, b), g.multiply(a, b), g.divide(a, b)
    r = g.reshape(m, (2, 2))
    g.set_input(a, np.array([12, 15, 8, 9], np.float16)); g.set_input(b, np.array([3, 5, 2, 3], np.float16))
    g.execute()
    assert g.get_output(s).tolist() == [15, 20, 10, 12]
    assert g.get_output(d).tolist() == [9, 10, 6, 6]
    assert g.get_output(m).tolist() == [36, 75, 16, 27]
    assert g.get_output(q).tolist() == [4, 3, 4, 3]
    assert g.get_output(r).tolist() == [[36, 75], [16, 27]]
    with pytest.raises(G.CactusHipError, match="Reshape"):
        g.reshape(m, (3, 2))
    g.close()
