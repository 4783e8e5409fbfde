"""GPU tests of the graph / KV-cache mirror (cactus_amd/csrc/graph.cpp through the C ABI of
include/cactus_hip_graph.h): the reference's own graph and KV-cache tests restated
(tests/test_graph.cpp:75-90, 220-240; tests/test_kv_cache.cpp:21-254) plus parity of every
dispatch branch of the matmul node (graph_ops_nn.cpp:129-231) against the CPU oracle.
"""
import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import rand_weights_int8, rand_weights_int4, rand_act_f16

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def G():
    import cactus_amd.hip as hip
    hip.init(0)
    import cactus_amd.graph as graph
    return graph


def u16(a):
    return np.ascontiguousarray(a).view(np.uint16)


# ---------------------------------------------------------------------------- matmul node
def test_matrix_multiplication_kat(G):
    """tests/test_graph.cpp:75-90: [2,3] x [3,2], rhs not pretransposed."""
    g = G.CactusGraph()
    a = g.input((2, 3), G.FP16)
    b = g.input((3, 2), G.FP16)
    c = g.matmul(a, b, False)
    g.set_input(a, np.array([1, 2, 3, 4, 5, 6], np.float16))
    g.set_input(b, np.array([1, 2, 3, 4, 5, 6], np.float16))
    g.execute()
    assert g.get_output(c).ravel().tolist() == [22, 28, 49, 64]
    g.close()


@pytest.mark.parametrize("M,K,N", [(1, 1024, 1024), (4, 2048, 512), (16, 1024, 3072), (33, 512, 256)])
@pytest.mark.parametrize("int4", [False, True])
def test_matmul_node_quantized_weights_fp16_activations(G, M, K, N, int4):
    """FP16 activations are quantized per row on the fly, then the grouped kernel runs
    (graph_ops_nn.cpp:139-155, 169-185); fp16 outputs bit-exact with the oracle doing the same."""
    rng = np.random.default_rng(M + K + N + int4)
    x = rand_act_f16(rng, M, K)
    g = G.CactusGraph()
    xin = g.input((M, K), G.FP16)
    w = g.input((N, K), G.INT8)
    if int4:
        Bp, Bs = rand_weights_int4(rng, N, K)
        g.set_packed_int4(w, Bp)
    else:
        B, Bs = rand_weights_int8(rng, N, K)
        g.set_external_input(w, B)
    g.set_grouped_scales(w, 128, K // 128, Bs)
    y = g.matmul(xin, w, True)
    g.set_input(xin, x)
    g.execute()
    got = u16(g.get_output(y))
    q, s = O.quantize_activations(x)
    ref = O.matmul_int4(q, s, Bp, Bs, 128) if int4 else O.matmul_int8(q, s, B, Bs, 128)
    assert (got == ref).all()
    g.close()


def test_matmul_node_prequantized_activations_shared_by_three_weights(G):
    """quantize_activations once, q/k/v projections read the INT8 node and its row scales
    (graph_ops_nn.cpp:78-127, 141-144)."""
    rng = np.random.default_rng(7)
    M, K = 3, 1024
    x = rand_act_f16(rng, M, K)
    g = G.CactusGraph()
    xin = g.input((M, K), G.FP16)
    xq = g.quantize_activations(xin)
    outs, refs = [], []
    q, s = O.quantize_activations(x)
    for N in (1024, 256, 256):
        B, Bs = rand_weights_int8(rng, N, K)
        w = g.input((N, K), G.INT8)
        g.set_external_input(w, B)
        g.set_grouped_scales(w, 128, K // 128, Bs)
        outs.append(g.matmul(xq, w, True))
        refs.append(O.matmul_int8(q, s, B, Bs, 128))
    g.set_input(xin, x)
    g.execute()
    assert (g.get_output(xq) == q).all()
    for o, r in zip(outs, refs):
        assert (u16(g.get_output(o)) == r).all()
    g.close()


@pytest.mark.parametrize("int4", [False, True])
def test_decode_projections_share_one_launch(G, int4):
    """q/k/v-shaped nodes on one FP16 activation row, then a node that reads one of them: the
    independent run is batched into a single launch, the dependent node runs after it, and
    every output equals the oracle's quantize + matmul."""
    rng = np.random.default_rng(21 + int4)
    K = 1024
    x = rand_act_f16(rng, 1, K)
    g = G.CactusGraph()
    xin = g.input((1, K), G.FP16)
    mk = rand_weights_int4 if int4 else rand_weights_int8
    ws, outs = [], []
    for N in (2048, 1024, 1024):
        B, Bs = mk(rng, N, K)
        w = g.input((N, K), G.INT8)
        (g.set_packed_int4 if int4 else g.set_external_input)(w, B)
        g.set_grouped_scales(w, 128, K // 128, Bs)
        ws.append((B, Bs))
        outs.append(g.matmul(xin, w, True))
    Bo, Bso = mk(rng, 512, 2048)                     # o-projection shaped: reads the first output
    wo = g.input((512, 2048), G.INT8)
    (g.set_packed_int4 if int4 else g.set_external_input)(wo, Bo)
    g.set_grouped_scales(wo, 128, 2048 // 128, Bso)
    y = g.matmul(outs[0], wo, True)
    g.set_input(xin, x)
    g.execute()
    mm = O.matmul_int4 if int4 else O.matmul_int8
    q, s = O.quantize_activations(x)
    refs = [mm(q, s, B, Bs, 128) for B, Bs in ws]
    for o, r in zip(outs, refs):
        assert (u16(g.get_output(o)) == r).all()
    q2, s2 = O.quantize_activations(refs[0].view(np.float16))
    assert (u16(g.get_output(y)) == mm(q2, s2, Bo, Bso, 128)).all()
    g.close()


def test_matmul_node_soft_reset_keeps_weights_and_requantizes(G):
    """Decode loop shape: weights stay resident over soft_reset(); the activation-quantization
    cache must not outlive a step (new activations at the same address give new results)."""
    rng = np.random.default_rng(11)
    K, N = 1024, 512
    B, Bs = rand_weights_int8(rng, N, K)
    g = G.CactusGraph()
    w = g.input((N, K), G.INT8)
    g.set_external_input(w, B)
    g.set_grouped_scales(w, 128, K // 128, Bs)
    x = np.empty((1, K), np.float16)
    for step in range(3):
        x[:] = rand_act_f16(rng, 1, K)          # same host buffer, new contents
        xin = g.input((1, K), G.FP16)
        y = g.matmul(xin, w, True)
        g.set_input(xin, x)
        g.execute()
        q, s = O.quantize_activations(x)
        assert (u16(g.get_output(y)) == O.matmul_int8(q, s, B, Bs, 128)).all()
        g.soft_reset()
        assert g.get_node_count() == 1
    g.hard_reset()
    assert g.get_node_count() == 0
    g.close()


def test_matmul_node_fp16_pretransposed(G):
    rng = np.random.default_rng(5)
    M, K, N = 7, 200, 50
    a = rand_act_f16(rng, M, K)
    bt = rand_act_f16(rng, N, K)
    g = G.CactusGraph()
    an = g.input((M, K), G.FP16)
    bn = g.input((N, K), G.FP16)
    c = g.matmul(an, bn, True)
    g.set_input(an, a)
    g.set_input(bn, bt)
    g.execute()
    got = g.get_output(c).astype(np.float64)
    ref = a.astype(np.float64) @ bt.astype(np.float64).T
    # fp32 accumulation, one fp16 rounding of the result
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.abs(ref).max() + 1e-3
    g.close()


def test_matmul_node_errors(G):
    """Messages of graph_ops_nn.cpp:160-167, 190-197."""
    rng = np.random.default_rng(3)
    B, Bs = rand_weights_int8(rng, 128, 128)
    g = G.CactusGraph()
    x = g.input((1, 128), G.FP16)
    w = g.input((128, 128), G.INT8)
    g.set_external_input(w, B)
    g.set_grouped_scales(w, 128, 1, Bs)
    g.matmul(x, w, False)
    g.set_input(x, np.zeros((1, 128), np.float16))
    with pytest.raises(Exception, match="Group-wise INT8 matmul requires pretransposed weights"):
        g.execute()
    g.hard_reset()
    x = g.input((1, 128), G.FP32)
    w = g.input((128, 128), G.INT8)
    g.set_external_input(w, B)
    g.set_grouped_scales(w, 128, 1, Bs)
    g.matmul(x, w, True)
    g.set_input(x, np.zeros((1, 128), np.float32))
    with pytest.raises(Exception, match="INT8 matmul requires INT8 .pre-quantized. or FP16 activations"):
        g.execute()
    g.hard_reset()
    x = g.input((1, 128), G.FP16)
    y = g.input((1, 128), G.FP16)
    g.matmul(x, y, True)
    with pytest.raises(Exception, match="has no data"):
        g.execute()
    g.close()


# ------------------------------------------------------------------------- attention nodes
def test_attention_node_kat(G):
    """tests/test_graph.cpp:220-240.  The reference only runs it: with head_dim 4 its 8-wide
    output loop never executes and the scalar tail stores zeros (kernel_attention.cpp:233-247),
    so the result is all zeros, here as there."""
    g = G.CactusGraph()
    q = g.input((1, 2, 1, 4), G.FP16)
    k = g.input((1, 2, 1, 4), G.FP16)
    v = g.input((1, 2, 1, 4), G.FP16)
    o = g.attention(q, k, v, 0.5)
    qd = np.array([1, 0, 0, 0, 0, 1, 0, 0], np.float16).reshape(1, 2, 1, 4)
    vd = np.array([1, 2, 3, 4, 5, 6, 7, 8], np.float16).reshape(1, 2, 1, 4)
    g.set_input(q, qd)
    g.set_input(k, qd)
    g.set_input(v, vd)
    g.execute()
    got = g.get_output(o)
    ref = O.attention_f16(qd, qd, vd, 0.5).view(np.float16)
    assert (ref == 0).all() and (got == 0).all()
    g.close()


def test_attention_node_prefill(G):
    """Causal FP16 attention node with offset and window overloads (graph_ops_nn.cpp:302-338)."""
    rng = np.random.default_rng(9)
    B, S, Hq, Hkv, D = 1, 40, 4, 2, 64
    qd = rng.uniform(-1, 1, (B, S, Hq, D)).astype(np.float16)
    kd = rng.uniform(-1, 1, (B, S, Hkv, D)).astype(np.float16)
    vd = rng.uniform(-1, 1, (B, S, Hkv, D)).astype(np.float16)
    for kwargs in ({}, {"window_size": 16}, {"is_causal": False}):
        g = G.CactusGraph()
        q = g.input(qd.shape, G.FP16)
        k = g.input(kd.shape, G.FP16)
        v = g.input(vd.shape, G.FP16)
        o = g.attention(q, k, v, 0.125, **kwargs)
        g.set_input(q, qd)
        g.set_input(k, kd)
        g.set_input(v, vd)
        g.execute()
        got = g.get_output(o).astype(np.float64)
        ref = O.attention_f16(qd, kd, vd, 0.125, **kwargs).view(np.float16).astype(np.float64)
        assert np.abs(got - ref).max() <= 2e-3 + 4e-3 * np.abs(ref).max()
        g.close()


def _fill_cache(G, g, cache, k, v, seq_len=None):
    """Append FP16 [S, Hkv, D] per layer through graph input nodes, as Model::forward does."""
    kn, vn = [], []
    for l in range(cache.num_layers):
        a = g.input(k[l].shape, G.FP16)
        b = g.input(v[l].shape, G.FP16)
        g.set_input(a, k[l])
        g.set_input(b, v[l])
        kn.append(a)
        vn.append(b)
    g.execute()
    cache.update_from_graph(g, kn, vn, k[0].shape[0] if seq_len is None else seq_len)


@pytest.mark.parametrize("D", [64, 128, 256])
def test_hybrid_attention_node_over_kv_cache(G, D):
    """Prefill into the INT8 cache, then one decode step with attention_int8_hybrid reading the
    cache's device pointers (model_qwen.cpp:106-116 shape of the call)."""
    rng = np.random.default_rng(D)
    Hq, Hkv, S = 8, 2, 77
    cache = G.KVCache(1, 2048, Hkv, D)
    g = G.CactusGraph()
    k = rng.normal(0, 1, (S, Hkv, D)).astype(np.float16)
    v = rng.normal(0, 1, (S, Hkv, D)).astype(np.float16)
    _fill_cache(G, g, cache, [k], [v])
    assert cache.get_effective_seq_len() == S and cache.get_total_seq_len() == S
    kq, ks, vq, vs = cache.download(0)
    okq, oks = O.quantize_kv(k)
    ovq, ovs = O.quantize_kv(v)
    assert (kq == okq).all() and (vq == ovq).all()
    assert (ks.view(np.uint32) == oks.view(np.uint32)).all() and (vs.view(np.uint32) == ovs.view(np.uint32)).all()

    g.soft_reset()
    qd = rng.normal(0, 1, (1, 1, Hq, D)).astype(np.float16)
    knew = rng.normal(0, 1, (1, 1, Hkv, D)).astype(np.float16)
    vnew = rng.normal(0, 1, (1, 1, Hkv, D)).astype(np.float16)
    qn = g.input(qd.shape, G.FP16)
    kn = g.input(knew.shape, G.FP16)
    vn = g.input(vnew.shape, G.FP16)
    scale = 1.0 / np.sqrt(D)
    o = g.attention_int8_hybrid(qn, kn, vn, scale, S, cache.get_keys_int8(0), cache.get_values_int8(0),
                                cache.get_key_scales(0), cache.get_value_scales(0), S, Hkv, D)
    g.set_input(qn, qd)
    g.set_input(kn, knew)
    g.set_input(vn, vnew)
    g.execute()
    got = g.get_output(o).astype(np.float64)
    ref = O.attention_hybrid(qd, okq[None], ovq[None], oks[None], ovs[None], knew, vnew, scale,
                             position_offset=S).view(np.float16).astype(np.float64)
    assert np.abs(got - ref).max() <= 2e-3 + 4e-3 * np.abs(ref).max()
    g.close()
    cache.close()


# ------------------------------------------------------------------------------- KV cache
def _const(S, H, D, val):
    return np.full((S, H, D), val, np.float16)


def test_sliding_window_basic(G):
    """tests/test_kv_cache.cpp:21-130."""
    L, Hkv, D, W, sink = 2, 8, 64, 16, 4
    cache = G.KVCache(L, 2048, Hkv, D)
    cache.set_window_size(W, sink)
    g = G.CactusGraph()
    _fill_cache(G, g, cache, [_const(10, Hkv, D, l + 1.0) for l in range(L)],
                [_const(10, Hkv, D, l + 2.0) for l in range(L)])
    assert cache.get_effective_seq_len() == 10
    _fill_cache(G, g, cache, [_const(6, Hkv, D, l + 10.0) for l in range(L)],
                [_const(6, Hkv, D, l + 20.0) for l in range(L)])
    assert cache.get_effective_seq_len() == W
    _fill_cache(G, g, cache, [_const(10, Hkv, D, 100.0)] * L, [_const(10, Hkv, D, 100.0)] * L)
    assert cache.get_effective_seq_len() == W
    assert cache.get_total_seq_len() == 26
    kq, ks, vq, vs = cache.download(0)
    expected_sink_q = int(round(1.0 / ks[0, 0, 0]))
    assert (np.abs(kq[:sink].astype(int) - expected_sink_q) <= 2).all()
    # what the slots hold after the slide: 4 sink tokens, 2 tokens of the second chunk, 10 new ones
    deq = kq.astype(np.float64) * ks          # one scale group per head at D = 64
    assert np.allclose(deq[:sink], 1.0, atol=0.02)
    assert np.allclose(deq[sink:sink + 2], 10.0, atol=0.1)
    assert np.allclose(deq[sink + 2:], 100.0, atol=1.0)
    g.close()
    cache.close()


def test_sliding_window_matches_host_model(G):
    """Token-by-token and chunked appends against a plain list model of the same policy
    (keep the first `sink` tokens and the most recent window - sink)."""
    rng = np.random.default_rng(0)
    Hkv, D, W, sink = 2, 128, 12, 3
    cache = G.KVCache(1, 2048, Hkv, D)
    cache.set_window_size(W, sink)
    g = G.CactusGraph()
    held = []
    for step, n in enumerate([5, 1, 4, 4, 1, 1, 7, 1, 9, 1]):
        k = rng.normal(0, 1, (n, Hkv, D)).astype(np.float16)
        v = rng.normal(0, 1, (n, Hkv, D)).astype(np.float16)
        _fill_cache(G, g, cache, [k], [v])
        held += [(k[i], v[i]) for i in range(n)]
        if len(held) > W:
            held = held[:sink] + held[len(held) - (W - sink):]
        assert cache.get_effective_seq_len() == len(held)
        kq, ks, vq, vs = cache.download(0)
        okq, oks = O.quantize_kv(np.stack([h[0] for h in held]))
        ovq, ovs = O.quantize_kv(np.stack([h[1] for h in held]))
        assert (kq == okq).all() and (vq == ovq).all(), f"step {step}"
        assert (ks == oks).all() and (vs == ovs).all(), f"step {step}"
        g.soft_reset()
    assert cache.get_total_seq_len() == 34
    g.close()
    cache.close()


def test_ring_window_matches_sliding_window(G):
    """Ring mode holds the same tokens as the sliding cache, rotated by ring_head, with no move
    per append (the reference memmoves the window per layer and token, engine_cache.cpp:276-303);
    hybrid attention over the rotated cache agrees with attention over the ordered one."""
    rng = np.random.default_rng(5)
    Hq, Hkv, D, W, sink = 4, 2, 128, 40, 3
    R = W - sink
    slide = G.KVCache(1, 2048, Hkv, D)
    ring = G.KVCache(1, 2048, Hkv, D)
    slide.set_window_size(W, sink)
    ring.set_window_size(W, sink)
    ring.set_ring(True)
    g = G.CactusGraph()
    for step, n in enumerate([30, 1, 7, 5, 1, 1, 36, 1, 1, 37, 20, 1]):
        k = rng.normal(0, 1, (n, Hkv, D)).astype(np.float16)
        v = rng.normal(0, 1, (n, Hkv, D)).astype(np.float16)
        _fill_cache(G, g, slide, [k], [v])
        g.soft_reset()
        _fill_cache(G, g, ring, [k], [v])
        g.soft_reset()
        L = slide.get_effective_seq_len()
        assert ring.get_effective_seq_len() == L and ring.get_total_seq_len() == slide.get_total_seq_len()
        h = ring.ring_head()
        assert (h == 0) if L < W else (0 <= h < R)
        order = np.arange(L)
        if L == W:
            order = np.concatenate([np.arange(sink), sink + (h + np.arange(R)) % R])
        for a, b in zip(slide.download(0), ring.download(0)):
            assert (a == b[order]).all(), f"step {step}"

        qd = rng.normal(0, 1, (1, 1, Hq, D)).astype(np.float16)
        knew = rng.normal(0, 1, (1, 1, Hkv, D)).astype(np.float16)
        vnew = rng.normal(0, 1, (1, 1, Hkv, D)).astype(np.float16)
        outs = []
        for cache in (slide, ring):
            qn = g.input(qd.shape, G.FP16)
            kn = g.input(knew.shape, G.FP16)
            vn = g.input(vnew.shape, G.FP16)
            o = g.attention_int8_hybrid(qn, kn, vn, 0.0, cache.get_total_seq_len(), cache.get_keys_int8(0),
                                        cache.get_values_int8(0), cache.get_key_scales(0),
                                        cache.get_value_scales(0), L, Hkv, D)
            g.set_input(qn, qd)
            g.set_input(kn, knew)
            g.set_input(vn, vnew)
            g.execute()
            outs.append(g.get_output(o).astype(np.float64))
            g.soft_reset()
        assert np.abs(outs[0] - outs[1]).max() <= 2e-3 + 4e-3 * np.abs(outs[0]).max(), f"step {step}"
    ring.reset()
    assert ring.ring_head() == 0 and ring.is_empty()
    with pytest.raises(Exception):
        slide.set_ring(True)        # not while tokens are held
    g.close()
    slide.close()
    ring.close()


def test_incremental_updates(G):
    """tests/test_kv_cache.cpp:132-178."""
    Hkv, D, W = 4, 32, 8
    cache = G.KVCache(1, 2048, Hkv, D)
    cache.set_window_size(W, 2)
    g = G.CactusGraph()
    for token in range(12):
        _fill_cache(G, g, cache, [_const(1, Hkv, D, token + 1.0)], [_const(1, Hkv, D, token + 101.0)])
        assert cache.get_effective_seq_len() == min(token + 1, W)
    assert cache.get_total_seq_len() == 12
    assert cache.get_keys_int8(0) and cache.get_key_scales(0)
    kq, ks, _, _ = cache.download(0)
    deq = kq[:, 0, 0] * ks[:, 0, 0]
    assert np.allclose(deq, [1, 2, 7, 8, 9, 10, 11, 12], rtol=0.01)
    g.close()
    cache.close()


def test_reset_functionality(G):
    """tests/test_kv_cache.cpp:180-215."""
    cache = G.KVCache(2, 1024, 8, 64)
    cache.set_window_size(16, 4)
    g = G.CactusGraph()
    _fill_cache(G, g, cache, [_const(10, 8, 64, 1.0)] * 2, [_const(10, 8, 64, 1.0)] * 2)
    assert cache.get_effective_seq_len() == 10 and cache.get_total_seq_len() == 10
    cache.reset()
    assert cache.get_effective_seq_len() == 0 and cache.get_total_seq_len() == 0
    assert not cache.get_keys_int8(0) and not cache.get_values_int8(0)
    g.close()
    cache.close()


def test_large_window(G):
    """tests/test_kv_cache.cpp:217-254: 600 tokens into a fresh 512-token window in one update
    (the whole-sequence branch).  The reference has sized the layer to the window in
    set_window_size, so it does not take this for the first slide and leaves the sink slots at
    their initial zeros with scale 1.0 (engine_cache.cpp:35-43, 169-198); same here."""
    L, Hkv, D, W, S = 4, 8, 64, 512, 600
    cache = G.KVCache(L, 2048, Hkv, D)
    cache.set_window_size(W, 4)
    g = G.CactusGraph()
    rng = np.random.default_rng(1)
    ks_in = [rng.normal(0, 1, (S, Hkv, D)).astype(np.float16) for _ in range(L)]
    vs_in = [rng.normal(0, 1, (S, Hkv, D)).astype(np.float16) for _ in range(L)]
    _fill_cache(G, g, cache, ks_in, vs_in)
    assert cache.get_effective_seq_len() == W
    assert cache.get_total_seq_len() == S
    for l in (0, L - 1):
        kq, ks, vq, vs = cache.download(l)
        assert (kq[:4] == 0).all() and (ks[:4] == 1.0).all()
        okq, oks = O.quantize_kv(ks_in[l][S - (W - 4):])
        ovq, ovs = O.quantize_kv(vs_in[l][S - (W - 4):])
        assert (kq[4:] == okq).all() and (ks[4:] == oks).all()
        assert (vq[4:] == ovq).all() and (vs[4:] == ovs).all()
    g.close()
    cache.close()


def test_whole_sequence_updates(G):
    """K/V nodes that hold every token so far (engine_cache.cpp:136-233): plain append without a
    window, and a first slide that writes the sink tokens and the most recent window - sink."""
    Hkv, D, W, S, sink = 2, 64, 64, 100, 4
    g = G.CactusGraph()
    rng = np.random.default_rng(2)
    k = rng.normal(0, 1, (S, Hkv, D)).astype(np.float16)
    v = rng.normal(0, 1, (S, Hkv, D)).astype(np.float16)
    okq, oks = O.quantize_kv(k)
    ovq, ovs = O.quantize_kv(v)

    plain = G.KVCache(1, 2048, Hkv, D)             # no window
    _fill_cache(G, g, plain, [k[:30]], [v[:30]])
    _fill_cache(G, g, plain, [k], [v], seq_len=S - 30)
    assert plain.get_effective_seq_len() == S and plain.get_total_seq_len() == S
    kq, ks, vq, vs = plain.download(0)
    assert (kq == okq).all() and (ks == oks).all() and (vq == ovq).all() and (vs == ovs).all()

    cache = G.KVCache(1, 2048, Hkv, D)
    cache.set_window_size(W, sink)
    _fill_cache(G, g, cache, [k[:10]], [v[:10]])   # logical size 10: the next update is a first slide
    _fill_cache(G, g, cache, [k], [v], seq_len=S - 10)
    assert cache.get_effective_seq_len() == W and cache.get_total_seq_len() == S
    kq, ks, vq, vs = cache.download(0)
    keep = np.r_[0:sink, S - (W - sink):S]
    assert (kq == okq[keep]).all() and (ks == oks[keep]).all()
    assert (vq == ovq[keep]).all() and (vs == ovs[keep]).all()
    g.close()
    cache.close()
    plain.close()


def test_appended_chunk_larger_than_window_is_rejected(G):
    cache = G.KVCache(1, 2048, 2, 64)
    cache.set_window_size(8, 2)
    g = G.CactusGraph()
    _fill_cache(G, g, cache, [_const(4, 2, 64, 1.0)], [_const(4, 2, 64, 1.0)])
    with pytest.raises(Exception, match="does not fit the window"):
        _fill_cache(G, g, cache, [_const(7, 2, 64, 1.0)], [_const(7, 2, 64, 1.0)])
    assert cache.get_effective_seq_len() == 4 and cache.get_total_seq_len() == 4
    g.close()
    cache.close()


# ------------------------------------------------------------------------- weight files
@pytest.mark.parametrize("name", ["tensor_io_a.npz", "tensor_io_b.npz", "tensor_io_pad.npz"])
@pytest.mark.parametrize("kind", ["file_int8", "file_int4", "file_fp16"])
def test_mmap_weights_of_reference_written_files(G, name, kind, tmp_path):
    """graph_io.cpp:48-72, 470-566: a file written by the reference's python/src/tensor_io.py
    (tests/golden holds the file bytes) becomes a weight node, and a matmul against it equals the
    oracle on the payload an independent parser extracts."""
    import os
    from tests.helpers import parse_cact
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    raw = d[kind].tobytes()
    path = tmp_path / (kind + ".weights")
    path.write_bytes(raw)
    ref = parse_cact(raw)
    N, K = ref["shape"]
    g = G.CactusGraph()
    w = g.mmap_weights(path)
    assert g.mmap_weights(path) == w                       # one node per file name
    info = g.node_info(w)
    assert info["shape"] == (N, K)
    assert info["packed_int4"] == (kind == "file_int4")
    assert info["precision"] == (G.FP16 if kind == "file_fp16" else G.INT8)
    rng = np.random.default_rng(N + K)
    M = 3
    x = rand_act_f16(rng, M, K)
    xin = g.input((M, K), G.FP16)
    y = g.matmul(xin, w, True)
    g.set_input(xin, x)
    g.execute()
    got = u16(g.get_output(y))
    if kind == "file_fp16":
        want = O.matmul_f16(x, ref["data"].view(np.uint16).reshape(N, K))
        gf, wf = got.view(np.float16).astype(np.float64), want.view(np.float16).astype(np.float64)
        assert np.abs(gf - wf).max() <= 2e-2 * max(1.0, np.abs(wf).max())
    else:
        assert info["group_size"] == 128 and info["num_groups"] == K // 128
        q, s = O.quantize_activations(x)
        Bs = ref["scales"].view(np.uint16).reshape(N, K // 128)
        if kind == "file_int4":
            want = O.matmul_int4(q, s, ref["data"].reshape(N, K // 2), Bs, 128)
        else:
            want = O.matmul_int8(q, s, ref["data"].view(np.int8).reshape(N, K), Bs, 128)
        assert (got == want).all()
    g.soft_reset()                                          # weights loaded from files survive
    assert g.node_info(w)["shape"] == (N, K)
    g.close()


def test_mmap_weights_errors(G, tmp_path):
    g = G.CactusGraph()
    with pytest.raises(G.CactusHipError, match="Cannot open file for mapping"):
        g.mmap_weights(tmp_path / "missing.weights")
    bad = tmp_path / "bad.weights"
    bad.write_bytes(b"NOPE" + bytes(200))
    with pytest.raises(G.CactusHipError, match="missing CACT magic"):
        g.mmap_weights(bad)
    g.close()


def test_ring_whole_sequence_update_restarts_the_ring(G):
    """A whole-sequence update of a ring cache rewrites the window in temporal order (ring head 0);
    appends after it wrap again."""
    rng = np.random.default_rng(6)
    Hkv, D, W, sink = 2, 64, 16, 2
    cache = G.KVCache(1, 2048, Hkv, D)
    cache.set_window_size(W, sink)
    cache.set_ring(True)
    g = G.CactusGraph()
    k = rng.normal(0, 1, (40, Hkv, D)).astype(np.float16)
    v = rng.normal(0, 1, (40, Hkv, D)).astype(np.float16)
    _fill_cache(G, g, cache, [k[:10]], [v[:10]])
    g.soft_reset()
    _fill_cache(G, g, cache, [k[10:19]], [v[10:19]])          # fills and wraps: 19 tokens into 16 slots
    assert cache.ring_head() == 3
    g.soft_reset()
    # whole-sequence node: the 16 tokens held (temporal order) followed by 11 new ones
    held = np.concatenate([np.arange(sink), np.arange(19 - (W - sink), 19)])
    allk = np.concatenate([k[held], k[19:30]])
    allv = np.concatenate([v[held], v[19:30]])
    _fill_cache(G, g, cache, [allk], [allv], seq_len=11)
    assert cache.ring_head() == 0 and cache.get_total_seq_len() == 30
    kq, ks, _, _ = cache.download(0)
    okq, oks = O.quantize_kv(np.concatenate([k[:sink], k[30 - (W - sink):30]]))
    assert (kq == okq).all() and (ks == oks).all()
    g.soft_reset()
    _fill_cache(G, g, cache, [k[30:35]], [v[30:35]])
    assert cache.ring_head() == 5
    kq, ks, _, _ = cache.download(0)
    R = W - sink
    order = np.concatenate([np.arange(sink), sink + (5 + np.arange(R)) % R])
    okq, oks = O.quantize_kv(np.concatenate([k[:sink], k[35 - R:35]]))
    assert (kq[order] == okq).all() and (ks[order] == oks).all()
    g.close()
    cache.close()


# ---------------------------------------------------------------------------- re-execution and the buffer pool
@pytest.mark.parametrize("int4", [False, True])
def test_chained_decode_matmuls_reexecuted_with_new_input(G, int4):
    """x -> matmul -> matmul executed twice with different inputs and no reset in between: the second
    matmul reads the first one's output of the SAME pass (it must not join the first one's batched
    launch on the strength of last pass's validity flags)."""
    rng = np.random.default_rng(91 + int4)
    K, N1, N2 = 1024, 2048, 512
    mk = rand_weights_int4 if int4 else rand_weights_int8
    om = O.matmul_int4 if int4 else O.matmul_int8
    g = G.CactusGraph()
    xin = g.input((1, K), G.FP16)
    B1, S1 = mk(rng, N1, K)
    B2, S2 = mk(rng, N2, N1)
    w1 = g.input((N1, K), G.INT8); w2 = g.input((N2, N1), G.INT8)
    (g.set_packed_int4 if int4 else g.set_external_input)(w1, B1)
    (g.set_packed_int4 if int4 else g.set_external_input)(w2, B2)
    g.set_grouped_scales(w1, 128, K // 128, S1)
    g.set_grouped_scales(w2, 128, N1 // 128, S2)
    h = g.matmul(xin, w1, True)
    y = g.matmul(h, w2, True)
    for rep in range(3):
        x = rand_act_f16(rng, 1, K) * (rep + 1)
        g.set_input(xin, x)
        g.execute()
        q, s = O.quantize_activations(x)
        h_ref = om(q, s, B1, S1, 128)
        q2, s2 = O.quantize_activations(h_ref.view(np.float16))
        assert (u16(g.get_output(h)) == h_ref).all(), rep
        assert (u16(g.get_output(y)) == om(q2, s2, B2, S2, 128)).all(), rep
    g.close()


def test_buffer_pool_survives_soft_reset(G):
    """A model rebuilds its graph every token (model_qwen.cpp:142: soft_reset, then the same nodes):
    after the first step the node buffers and the activation-quantization scratch come from the
    pool, so the number of device allocations stops growing, and results stay right."""
    rng = np.random.default_rng(17)
    M, K, N = 8, 1024, 512      # M > 4: the quantized rows go through the graph's scratch
    B, Bs = rand_weights_int8(rng, N, K)
    g = G.CactusGraph()
    w = g.input((N, K), G.INT8)
    g.set_external_input(w, B)
    g.set_grouped_scales(w, 128, K // 128, Bs)
    counts = []
    for step in range(4):
        x = rand_act_f16(rng, M, K)
        xin = g.input((M, K), G.FP16)
        y = g.matmul(xin, w, True)
        z = g.add(y, y)
        g.set_input(xin, x)
        g.execute()
        q, s = O.quantize_activations(x)
        ref = O.matmul_int8(q, s, B, Bs, 128)
        assert (u16(g.get_output(y)) == ref).all()
        assert (u16(g.get_output(z)) == O.binary("add", ref, ref)).all()
        counts.append(g.pool_allocations())
        g.soft_reset()
        assert g.get_node_count() == 1     # the weight survives
    assert counts[0] > 0 and counts[1] == counts[0] and counts[3] == counts[0], counts
    g.close()


def test_buffer_pool_is_shared_by_prompts_of_different_lengths_and_bounded(G):
    """Variable-length serving: passes over 300, 290, 310, 257 ... rows land in the same power-of-two buckets (the
    reference's BufferPool rounds the same way, graph_core.cpp:5-19), so the allocation count stops growing after
    the first pass; the idle bytes kept across soft resets never exceed the cap, and lowering the cap frees them."""
    rng = np.random.default_rng(23)
    K, N = 1024, 512
    B, Bs = rand_weights_int8(rng, N, K)
    g = G.CactusGraph()
    w = g.input((N, K), G.INT8)
    g.set_external_input(w, B)
    g.set_grouped_scales(w, 128, K // 128, Bs)
    counts, idle = [], []
    for M in (300, 290, 310, 257, 300, 512):
        x = rand_act_f16(rng, M, K)
        xin = g.input((M, K), G.FP16)
        y = g.matmul(xin, w, True)
        z = g.add(y, y)
        g.set_input(xin, x)
        g.execute()
        q, s = O.quantize_activations(x)
        ref = O.matmul_int8(q, s, B, Bs, 128)
        assert (u16(g.get_output(y)) == ref).all()
        assert (u16(g.get_output(z)) == O.binary("add", ref, ref)).all()
        g.soft_reset()
        counts.append(g.pool_allocations())
        idle.append(g.pool_idle_bytes())
    assert counts[0] > 0 and counts[-1] == counts[0], counts          # 257..512 rows: the same buckets
    assert idle[-1] == idle[0] and idle[0] > 0, idle
    # twice as many rows: new buckets join, the old ones stay idle within the cap
    xin = g.input((1024, K), G.FP16)
    y = g.matmul(xin, w, True)
    g.set_input(xin, rand_act_f16(rng, 1024, K))
    g.execute()
    g.soft_reset()
    assert g.pool_allocations() > counts[0] and g.pool_idle_bytes() > idle[0]
    g.set_pool_max_idle_bytes(idle[0] // 2)             # a lower cap frees idle buffers, largest first
    assert g.pool_idle_bytes() <= idle[0] // 2
    x = rand_act_f16(rng, 300, K)                        # and the graph keeps working
    xin = g.input((300, K), G.FP16)
    y = g.matmul(xin, w, True)
    g.set_input(xin, x)
    g.execute()
    q, s = O.quantize_activations(x)
    assert (u16(g.get_output(y)) == O.matmul_int8(q, s, B, Bs, 128)).all()
    g.soft_reset()
    assert g.pool_idle_bytes() <= idle[0] // 2
    g.close()


def test_graph_weights_stream_from_twins_same_bits(G):
    """Weights bound to a graph get a tile-major twin by default (decode rows stream it); with twins off the same
    graph gives the same bits."""
    rng = np.random.default_rng(29)
    K, N = 2048, 1000
    B, Bs = rand_weights_int8(rng, N, K)
    Bp, Bps = rand_weights_int4(rng, N, K)
    x = rand_act_f16(rng, 1, K)
    outs = []
    for twins in (True, False):
        g = G.CactusGraph()
        g.set_streaming_twins(twins)
        w = g.input((N, K), G.INT8)
        g.set_external_input(w, B)
        g.set_grouped_scales(w, 128, K // 128, Bs)
        w4 = g.input((N, K), G.INT8)
        g.set_packed_int4(w4, Bp)
        g.set_grouped_scales(w4, 128, K // 128, Bps)
        xin = g.input((1, K), G.FP16)
        y, y4 = g.matmul(xin, w, True), g.matmul(xin, w4, True)
        g.set_input(xin, x)
        g.execute()
        outs.append((u16(g.get_output(y)).copy(), u16(g.get_output(y4)).copy()))
        g.close()
    q, s = O.quantize_activations(x)
    assert (outs[0][0] == O.matmul_int8(q, s, B, Bs, 128)).all() and (outs[0][1] == O.matmul_int4(q, s, Bp, Bps, 128)).all()
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


# ---------------------------------------------------------------------------- fused runs
def _decoder_mlp_block(G, g, rng, M, HID, FF, mixed, act):
    """h2 = h + o;  n = rms_norm(h2);  out = h2 + down(act(gate(n)) * up(n))  plus a second norm without an add
    in front -- the node order a model builds (gate, activation, up, multiply)."""
    def weight(n, k, int4):
        node = g.input((n, k), G.INT8)
        if int4:
            Bp, Bs = rand_weights_int4(rng, n, k)
            g.set_packed_int4(node, Bp)
        else:
            B, Bs = rand_weights_int8(rng, n, k)
            g.set_external_input(node, B)
        g.set_grouped_scales(node, 128, k // 128, (Bs.astype(np.float32) * 0.05).astype(np.float16))
        return node
    h = g.input((M, HID), G.FP16)
    o = g.input((M, HID), G.FP16)
    g.set_input(h, rand_act_f16(rng, M, HID))
    g.set_input(o, rand_act_f16(rng, M, HID))
    nw1, nw2 = g.input((HID,), G.FP16), g.input((HID,), G.FP16)
    g.set_input(nw1, (1.0 + 0.1 * rng.standard_normal(HID)).astype(np.float16))
    g.set_input(nw2, (1.0 + 0.1 * rng.standard_normal(HID)).astype(np.float16))
    wg, wu, wd = weight(FF, HID, False), weight(FF, HID, mixed), weight(HID, FF, mixed)
    wq = weight(HID, HID, False)
    h2 = g.add(h, o)
    n = g.rms_norm(h2, nw1, 1e-6)
    gate = g.matmul(n, wg, True)
    a = g.silu(gate) if act == "silu" else g.gelu(gate)
    up = g.matmul(n, wu, True)
    prod = g.multiply(a, up)
    out = g.add(h2, g.matmul(prod, wd, True))
    n2 = g.rms_norm(out, nw2, 1e-6)        # no add in front that has not run: norm + matmul only
    q = g.matmul(n2, wq, True)
    return dict(h2=h2, n=n, gate=gate, a=a, up=up, prod=prod, out=out, n2=n2, q=q)


@pytest.mark.parametrize("M,mixed,act", [(1, False, "silu"), (3, True, "silu"), (2, False, "gelu"), (24, False, "silu"),
                                         (24, True, "gelu")])
def test_fused_runs_equal_node_by_node_execution(G, M, mixed, act):
    """execute() with the fused launches against execute() node by node: every node's output identical, the
    ones a fused launch never wrote included (computed on request)."""
    HID, FF = 1024, 3072
    outs = {}
    for fusion in (False, True):
        rng = np.random.default_rng(M + mixed)
        g = G.CactusGraph()
        g.set_fusion(fusion)
        nodes = _decoder_mlp_block(G, g, rng, M, HID, FF, mixed, act)
        g.execute()
        if fusion:
            # add+norm+(gate, up): one launch per weight precision; act+multiply+down: one; norm+q: one
            assert g.fused_launches() == (4 if mixed else 3)
        else:
            assert g.fused_launches() == 0
        outs[fusion] = {k: u16(g.get_output(v)).copy() for k, v in nodes.items()}
        g.close()
    for k in outs[True]:
        assert np.array_equal(outs[True][k], outs[False][k]), k


def _decode_attention_run(G, g, rng, Hq, Hkv, D, cache_len, pos, dev_cache, share_q=False):
    """q / k / v projections' rows -> per-head norm -> RoPE -> attention over an INT8 cache, as a decoder builds it."""
    def inp(shape, data):
        n = g.input(shape, G.FP16)
        g.set_input(n, data)
        return n
    half = lambda *sh: (rng.standard_normal(sh) * 2).astype(np.float16)
    qp, kp, vp = inp((1, Hq * D), half(1, Hq * D)), inp((1, Hkv * D), half(1, Hkv * D)), inp((1, Hkv * D), half(1, Hkv * D))
    qw, kw = inp((D,), (1 + 0.1 * rng.standard_normal(D)).astype(np.float16)), inp((D,), (1 + 0.1 * rng.standard_normal(D)).astype(np.float16))
    # the projections pass through an op first, so that they are computed nodes as in a model
    qp, kp, vp = g.scalar_multiply(qp, 1.0), g.scalar_multiply(kp, 1.0), g.scalar_multiply(vp, 1.0)
    qn = g.rms_norm(g.reshape(qp, (Hq, D)), qw, 1e-6)
    kn = g.rms_norm(g.reshape(kp, (Hkv, D)), kw, 1e-6)
    qr = g.rope(g.reshape(qn, (1, 1, Hq, D)), 1e4, pos)
    kr = g.rope(g.reshape(kn, (1, 1, Hkv, D)), 1e4, pos)
    v4 = g.reshape(vp, (1, 1, Hkv, D))
    kc, vc, ks, vs = dev_cache
    att = g.attention_int8_hybrid(qr, kr, v4, 1.0 / np.sqrt(D), pos, kc.data_ptr(), vc.data_ptr(), ks.data_ptr(), vs.data_ptr(),
                                  cache_len, Hkv, D)
    nodes = dict(att=att, kr=kr, qr=qr, qn=qn, kn=kn)
    if share_q:
        nodes["extra"] = g.scalar_multiply(qr, 2.0)     # a second reader of the rotated queries
    return nodes


@pytest.mark.parametrize("Hq,Hkv,D,cache_len,pos", [(16, 8, 128, 512, 512), (4, 1, 256, 77, 300), (32, 8, 64, 2500, 2500),
                                                    (6, 3, 128, 1, 1)])
def test_fused_decode_attention_equals_node_by_node_execution(G, Hq, Hkv, D, cache_len, pos):
    """rms_norm -> rope on q and k -> attention_int8_hybrid of one token as ONE launch: the attention rows, the
    rotated key rows (written by that launch) and the nodes it skipped (computed on request) are identical to
    node-by-node execution."""
    gen = torch.Generator(device="cuda").manual_seed(D + cache_len)
    NG = (D + 127) // 128
    kc = torch.randint(-127, 128, (cache_len, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vc = torch.randint(-127, 128, (cache_len, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((cache_len, Hkv, NG), device="cuda", generator=gen) * 0.02 + 1e-3
    vs = torch.rand((cache_len, Hkv, NG), device="cuda", generator=gen) * 0.02 + 1e-3
    outs = {}
    for fusion in (False, True):
        rng = np.random.default_rng(Hq + D)
        g = G.CactusGraph()
        g.set_fusion(fusion)
        nodes = _decode_attention_run(G, g, rng, Hq, Hkv, D, cache_len, pos, (kc, vc, ks, vs))
        g.execute()
        assert g.fused_launches() == (1 if fusion else 0)
        outs[fusion] = {k: u16(g.get_output(v)).copy() for k, v in nodes.items()}
        g.close()
    for k in outs[True]:
        assert np.array_equal(outs[True][k], outs[False][k]), k


def _prompt_attention_run(G, g, rng, S, Hq, Hkv, D, pos, dev_cache, cache_len):
    """The same run for S new tokens; dev_cache None: plain attention (a prompt against nothing), else the INT8 hybrid."""
    def inp(shape, data):
        n = g.input(shape, G.FP16)
        g.set_input(n, data)
        return n
    half = lambda *sh: (rng.standard_normal(sh) * 2).astype(np.float16)
    qp, kp, vp = inp((S, Hq * D), half(S, Hq * D)), inp((S, Hkv * D), half(S, Hkv * D)), inp((S, Hkv * D), half(S, Hkv * D))
    qw, kw = inp((D,), (1 + 0.1 * rng.standard_normal(D)).astype(np.float16)), inp((D,), (1 + 0.1 * rng.standard_normal(D)).astype(np.float16))
    qp, kp, vp = g.scalar_multiply(qp, 1.0), g.scalar_multiply(kp, 1.0), g.scalar_multiply(vp, 1.0)
    qn = g.rms_norm(g.reshape(qp, (S * Hq, D)), qw, 1e-6)
    kn = g.rms_norm(g.reshape(kp, (S * Hkv, D)), kw, 1e-6)
    qr = g.rope(g.reshape(qn, (1, S, Hq, D)), 1e4, pos)
    kr = g.rope(g.reshape(kn, (1, S, Hkv, D)), 1e4, pos)
    v4 = g.reshape(vp, (1, S, Hkv, D))
    if dev_cache is None:
        att = g.attention(qr, kr, v4, 1.0 / np.sqrt(D), pos)
    else:
        kc, vc, ks, vs = dev_cache
        att = g.attention_int8_hybrid(qr, kr, v4, 1.0 / np.sqrt(D), pos, kc.data_ptr(), vc.data_ptr(), ks.data_ptr(), vs.data_ptr(),
                                      cache_len, Hkv, D)
    return dict(att=att, kr=kr, qr=qr, qn=qn, kn=kn)


@pytest.mark.parametrize("S,Hq,Hkv,D,cache_len", [(40, 16, 8, 128, 0), (256, 16, 8, 128, 300), (7, 32, 8, 64, 0), (33, 4, 1, 256, 50)])
def test_fused_prompt_norm_rope_equals_node_by_node_execution(G, S, Hq, Hkv, D, cache_len):
    """rms_norm -> rope on q and k in front of an attention over several new tokens (a prompt, or a chunk against the
    INT8 cache) as ONE launch: the attention rows, both rotated nodes and the norms it skipped (computed on request) are
    identical to node-by-node execution."""
    gen = torch.Generator(device="cuda").manual_seed(D + S)
    NG = (D + 127) // 128
    cache = None
    if cache_len:
        cache = (torch.randint(-127, 128, (cache_len, Hkv, D), dtype=torch.int8, device="cuda", generator=gen),
                 torch.randint(-127, 128, (cache_len, Hkv, D), dtype=torch.int8, device="cuda", generator=gen),
                 torch.rand((cache_len, Hkv, NG), device="cuda", generator=gen) * 0.02 + 1e-3,
                 torch.rand((cache_len, Hkv, NG), device="cuda", generator=gen) * 0.02 + 1e-3)
    outs = {}
    for fusion in (False, True):
        rng = np.random.default_rng(S + D)
        g = G.CactusGraph()
        g.set_fusion(fusion)
        nodes = _prompt_attention_run(G, g, rng, S, Hq, Hkv, D, cache_len, cache, cache_len)
        g.execute()
        assert g.fused_launches() == (1 if fusion else 0)
        outs[fusion] = {k: u16(g.get_output(v)).copy() for k, v in nodes.items()}
        g.close()
    for k in outs[True]:
        assert np.array_equal(outs[True][k], outs[False][k]), k


def test_decode_attention_run_with_a_second_reader_is_left_alone(G):
    gen = torch.Generator(device="cuda").manual_seed(1)
    Hq, Hkv, D, cache_len = 16, 8, 128, 100
    kc = torch.randint(-127, 128, (cache_len, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vc = torch.randint(-127, 128, (cache_len, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((cache_len, Hkv, 1), device="cuda", generator=gen) * 0.02 + 1e-3
    vs = torch.rand((cache_len, Hkv, 1), device="cuda", generator=gen) * 0.02 + 1e-3
    outs = {}
    for share in (False, True):
        g = G.CactusGraph()
        nodes = _decode_attention_run(G, g, np.random.default_rng(2), Hq, Hkv, D, cache_len, cache_len, (kc, vc, ks, vs), share_q=share)
        g.execute()
        assert g.fused_launches() == (0 if share else 1)
        outs[share] = {k: u16(g.get_output(v)).copy() for k, v in nodes.items()}
        g.close()
    for k in outs[False]:
        assert np.array_equal(outs[True][k], outs[False][k]), k
    qr = outs[True]["qr"].view(np.float16).astype(np.float32)
    assert np.array_equal(outs[True]["extra"], (qr * 2).astype(np.float16).view(np.uint16))


def test_fusion_leaves_open_runs_alone(G):
    """A norm that something other than a matmul reads, and an activation read twice, are computed as nodes."""
    rng = np.random.default_rng(4)
    g = G.CactusGraph()
    HID = 512
    x = g.input((2, HID), G.FP16)
    g.set_input(x, rand_act_f16(rng, 2, HID))
    nw = g.input((HID,), G.FP16)
    g.set_input(nw, np.ones(HID, np.float16))
    w = g.input((256, HID), G.INT8)
    B, Bs = rand_weights_int8(rng, 256, HID)
    g.set_external_input(w, B)
    g.set_grouped_scales(w, 128, HID // 128, Bs)
    n = g.rms_norm(x, nw, 1e-6)
    y = g.matmul(n, w, True)
    z = g.add(n, x)                      # a second reader of the norm
    a = g.silu(x)
    p = g.multiply(a, x)
    p2 = g.multiply(a, p)                # the activation read twice
    y2 = g.matmul(p2, w, True)           # multiply(a, p) is not act * up of a closed run; p2's first operand is `a` though
    g.execute()
    xs = u16(rand_act_f16(np.random.default_rng(4), 2, HID))
    want_n = O.rms_norm(xs, np.ones(HID, np.float16), 1e-6)
    assert np.array_equal(u16(g.get_output(n)), want_n)
    q, s = O.quantize_activations(want_n)
    assert np.array_equal(u16(g.get_output(y)), O.matmul_int8(q, s, B, u16(Bs), 128))
    assert np.array_equal(u16(g.get_output(z)), O.binary("add", want_n, xs))
    sa = O.silu(xs)
    want_p2 = O.binary("multiply", sa, O.binary("multiply", sa, xs))
    assert np.array_equal(u16(g.get_output(p2)), want_p2)
    q, s = O.quantize_activations(want_p2)
    assert np.array_equal(u16(g.get_output(y2)), O.matmul_int8(q, s, B, u16(Bs), 128))
    g.close()
