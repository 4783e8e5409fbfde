"""GPU parity tests of the causal depthwise convolution and the LFM2 short-convolution block
(kernel_conv.cpp:15-110, model_lfm2.cpp:123-207, engine_cache.cpp:494-597), through the C ABI against
the CPU oracle on the same seeded inputs.  Bar: bit-exact (the kernels keep the reference's tap
order, its four fp32 lanes for whole groups of 8 taps and its fp16 products)."""
import numpy as np
import pytest

from oracle import binding as O

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


@pytest.mark.parametrize("N,L,C,K,dil", [(1, 1, 8, 3, 1), (2, 9, 5, 3, 1), (1, 40, 16, 8, 1), (1, 33, 7, 11, 2),
                                          (3, 17, 4, 16, 1), (1, 6, 3, 1, 1), (1, 5, 2, 9, 3), (1, 256, 2048, 3, 1),
                                          (4, 64, 96, 19, 1)])
def test_conv1d_causal_depthwise_bit_exact(H, N, L, C, K, dil):
    rng = np.random.default_rng(L * 100 + K)
    x, w = halves(rng, (N, L, C)), halves(rng, (C, K), 0.5)
    want = O.conv1d_causal_depthwise(x, w, dil)
    got = bits(H.conv1d_causal_depthwise(dev(x), dev(w), dil))
    assert np.array_equal(got, want)


def test_conv1d_empty_and_invalid(H):
    x = torch.empty((0, 4, 8), dtype=torch.float16, device="cuda")
    w = dev(np.ones((8, 3), np.float16))
    assert H.conv1d_causal_depthwise(x, w).numel() == 0
    with pytest.raises(H.CactusHipError):
        H.conv1d_causal_depthwise(dev(np.ones((1, 4, 8), np.float16)), w, dilation=0)


@pytest.mark.parametrize("C,K,W", [(2048, 3, 2), (24, 3, 2), (13, 3, 2), (64, 4, 3), (40, 9, 8), (7, 12, 11),
                                   (16, 3, 5), (8, 2, 0)])
def test_lfm2_conv_gate_steps_bit_exact(H, C, K, W):
    """Several sequences advanced together, prefill chunks and single tokens, against one oracle
    ConvCache per sequence; the state rows are compared in temporal order after every call."""
    rng = np.random.default_rng(C + 7 * K + W)
    nseq = 3
    w = halves(rng, (C, K), 0.5)
    state = torch.zeros((nseq, W, C), dtype=torch.float16, device="cuda") if W else None
    caches = [O.Lfm2ConvState(W, C) for _ in range(nseq)]
    for L in [5, 1, 1, 40, 1, 2, 1]:
        ip = halves(rng, (nseq, L, 3 * C))
        got = bits(H.lfm2_conv_gate(dev(ip), dev(w), state))
        for s in range(nseq):
            want = O.lfm2_conv_gate(ip[s], w, caches[s])
            assert np.array_equal(got[s], want), (L, s)
            if W:
                c = caches[s]
                n, h = c.count.value, c.head.value
                rows = c.rows[:n] if n < W else np.concatenate([c.rows[h:], c.rows[:h]])
                assert np.array_equal(bits(state[s])[W - n:], rows), (L, s)


def test_lfm2_conv_gate_long_prefill_uses_row_chunks(H):
    """One sequence, 300 rows: the launch is cut into row chunks and the state written by a second
    launch; same bits as the oracle, and the next token continues from the right state."""
    rng = np.random.default_rng(11)
    C, K, W, L = 256, 3, 2, 300
    w = halves(rng, (C, K), 0.5)
    state = torch.zeros((1, W, C), dtype=torch.float16, device="cuda")
    cache = O.Lfm2ConvState(W, C)
    for n in (L, 1, 1):
        ip = halves(rng, (1, n, 3 * C))
        got = bits(H.lfm2_conv_gate(dev(ip), dev(w), state))
        assert np.array_equal(got[0], O.lfm2_conv_gate(ip[0], w, cache))


def test_lfm2_conv_gate_decode_batch_full_size(H):
    """LFM2-1.2B decode shape, 256 sequences of one token: checked on a sample of sequences."""
    rng = np.random.default_rng(12)
    nseq, C, K, W = 256, 2048, 3, 2
    w = halves(rng, (C, K), 0.5)
    st0 = halves(rng, (nseq, W, C))
    state = dev(st0)
    ip = halves(rng, (nseq, 1, 3 * C))
    got = bits(H.lfm2_conv_gate(dev(ip), dev(w), state))
    for s in (0, 17, 255):
        c = O.Lfm2ConvState(W, C)
        c.rows[:] = st0[s].view(np.uint16)
        c.count.value = W
        assert np.array_equal(got[s], O.lfm2_conv_gate(ip[s], w, c))
        assert np.array_equal(bits(state[s]), np.concatenate([c.rows[c.head.value:], c.rows[:c.head.value]]))


def test_graph_conv1d_causal_node(H):
    """CactusGraph::conv1d_causal (graph_builder.cpp:318-321, graph_ops_nn.cpp:459-530) inside the
    node sequence of LFM2Model::build_conv1d without a cache (model_lfm2.cpp:146-204): multiply,
    conv1d_causal, multiply -- against the oracle and against the one-launch block op."""
    import cactus_amd.graph as G
    rng = np.random.default_rng(21)
    L, C, K = 37, 48, 3
    b, cg, x = halves(rng, (L, C)), halves(rng, (L, C)), halves(rng, (L, C))
    w = halves(rng, (C, 1, K), 0.5)
    g = G.CactusGraph()
    nb, ncg, nx = (g.input((L, C), G.FP16) for _ in range(3))
    nw = g.input((C, 1, K), G.FP16)
    bx = g.multiply(nb, nx)
    y = g.conv1d_causal(g.reshape(bx, (1, L, C)), nw, K, 1)
    gated = g.multiply(ncg, g.reshape(y, (L, C)))
    for node, arr in ((nb, b), (ncg, cg), (nx, x), (nw, w)):
        g.set_input(node, arr)
    g.execute()
    got = g.get_output(gated).view(np.uint16)
    ip = np.concatenate([b, cg, x], axis=1)
    want = O.lfm2_conv_gate(ip, w.reshape(C, K), O.Lfm2ConvState(0, C))
    assert np.array_equal(got, want)
    fused = bits(H.lfm2_conv_gate(dev(ip[None]), dev(w.reshape(C, K)), None))[0]
    assert np.array_equal(fused, want)
    bad = g.input((L, C), G.FP16)
    g.conv1d_causal(bad, nw, K, 1)
    g.set_input(bad, b)
    with pytest.raises(Exception, match="3D input"):
        g.execute()
    g.close()


def test_lfm2_conv_decoder_block_bit_exact(H):
    """A whole LFM2 convolution block (model_lfm2.cpp:123-207, 267-299): input norm, in-projection
    (INT8 weights, row-quantized activations), short convolution with its carried rows,
    out-projection, residual add, norm, gate / up, SiLU x up, down, residual add -- a 9-token
    prefill followed by three single-token steps, on the device and in the oracle.  No attention in
    this block, so every intermediate is compared bit for bit."""
    from tests.helpers import rand_weights_int8
    rng = np.random.default_rng(31)
    Hd, F, K, W, eps = 256, 512, 3, 2, 1e-5
    scale_down = lambda Bs: (Bs.astype(np.float32) * 0.2).astype(np.float16)
    wt = {}
    for name, (n, k) in {"in_proj": (3 * Hd, Hd), "out_proj": (Hd, Hd), "gate": (F, Hd), "up": (F, Hd),
                         "down": (Hd, F)}.items():
        B, Bs = rand_weights_int8(rng, n, k)
        wt[name] = (B, scale_down(Bs))
    in_norm = (1.0 + 0.1 * rng.standard_normal(Hd)).astype(np.float16)
    post_norm = (1.0 + 0.1 * rng.standard_normal(Hd)).astype(np.float16)
    conv_w = halves(rng, (Hd, K), 0.5)
    dw = {k: (dev(v[0]), dev(v[1])) for k, v in wt.items()}
    d_in_norm, d_post_norm, d_conv_w = dev(in_norm), dev(post_norm), dev(conv_w)
    state = torch.zeros((1, W, Hd), dtype=torch.float16, device="cuda")
    cache = O.Lfm2ConvState(W, Hd)

    def omm(x_bits, w):
        q, s = O.quantize_activations(x_bits)
        return O.matmul_int8(q, s, w[0], w[1].view(np.uint16), 128)

    for L in (9, 1, 1, 1):
        hidden = halves(rng, (L, Hd))
        # oracle
        normed = O.rms_norm(hidden, in_norm, eps)
        ipj = omm(normed, wt["in_proj"])
        gated = O.lfm2_conv_gate(ipj, conv_w, cache)
        proj = omm(gated, wt["out_proj"])
        after = O.binary("add", hidden, proj)
        normed2 = O.rms_norm(after, post_norm, eps)
        act = O.binary("multiply", O.silu(omm(normed2, wt["gate"])), omm(normed2, wt["up"]))
        want = O.binary("add", after, omm(act, wt["down"]))
        # device
        dh = dev(hidden)
        g_ipj = H.matmul_f16act(H.rms_norm(dh, d_in_norm, eps), *dw["in_proj"], 128)
        assert np.array_equal(bits(g_ipj), ipj), L
        g_gated = H.lfm2_conv_gate(g_ipj.view(1, L, 3 * Hd), d_conv_w, state).view(L, Hd)
        assert np.array_equal(bits(g_gated), gated), L
        g_after = H.add(dh, H.matmul_f16act(g_gated, *dw["out_proj"], 128))
        assert np.array_equal(bits(g_after), after), L
        g_n2 = H.rms_norm(g_after, d_post_norm, eps)
        g_act = H.silu_multiply(H.matmul_f16act(g_n2, *dw["gate"], 128), H.matmul_f16act(g_n2, *dw["up"], 128))
        got = H.add(g_after, H.matmul_f16act(g_act, *dw["down"], 128))
        assert np.array_equal(bits(got), want), L


def test_graph_slice_concat_index_nodes(H):
    """CactusGraph::slice / concat / index (graph_builder.cpp:125-160, 348-375, 499-522) against numpy,
    along the first axis (views) and inner axes (strided copies)."""
    import cactus_amd.graph as G
    rng = np.random.default_rng(40)
    a, b = halves(rng, (5, 3, 8)), halves(rng, (5, 2, 8))
    g = G.CactusGraph()
    na, nb = g.input(a.shape, G.FP16), g.input(b.shape, G.FP16)
    nodes = {"s0": g.slice(na, 0, 1, 3), "s1": g.slice(na, 1, 1, 2), "s2": g.slice(na, 2, 3, 4),
             "c1": g.concat(na, nb, 1), "c0": g.concat(na, na, 0), "cm": g.concat(na, nb, -2),
             "i0": g.index(na, 4, 0), "i1": g.index(na, 2, 1), "i2": g.index(na, 7, -1)}
    nodes["chain"] = g.add(g.slice(nodes["c1"], 1, 3, 2), nb)          # a slice of a concat feeds a kernel
    g.set_input(na, a)
    g.set_input(nb, b)
    g.execute()
    want = {"s0": a[1:4], "s1": a[:, 1:3], "s2": a[:, :, 3:7], "c1": np.concatenate([a, b], 1),
            "c0": np.concatenate([a, a], 0), "cm": np.concatenate([a, b], 1), "i0": a[4], "i1": a[:, 2],
            "i2": a[:, :, 7], "chain": b + b}
    for k, v in want.items():
        got = g.get_output(nodes[k])
        assert got.shape == v.shape and np.array_equal(got, v), k
    for bad in (lambda: g.slice(na, 1, 2, 2), lambda: g.concat(na, nb, 0), lambda: g.index(na, 3, 1)):
        with pytest.raises(Exception):
            bad()
    g.close()


def test_graph_lfm2_conv_block_as_the_reference_builds_it(H):
    """LFM2Model::build_conv1d node by node (model_lfm2.cpp:136-204): reshape to [L, 3, C], three slices,
    multiply, the cache window as input nodes concatenated in front, conv1d_causal, slice of the last
    L rows, multiply -- two calls with the ConvCache rows carried on the host as the reference carries
    them; equal to the oracle and to the one-launch block op."""
    import cactus_amd.graph as G
    rng = np.random.default_rng(41)
    C, K, W = 64, 3, 2
    w = halves(rng, (C, 1, K), 0.5)
    state = torch.zeros((1, W, C), dtype=torch.float16, device="cuda")
    cache = O.Lfm2ConvState(W, C)
    window = np.zeros((0, C), np.float16)                       # rows held, oldest first
    g = G.CactusGraph()
    for L in (5, 1, 1):
        ip = halves(rng, (L, 3 * C))
        nin, nw = g.input((L, 3 * C), G.FP16), g.input((C, 1, K), G.FP16)
        trip = g.reshape(nin, (L, 3, C))
        Bn, Cg, X = (g.reshape(g.slice(trip, 1, i, 1), (L, C)) for i in range(3))
        bx = g.multiply(Bn, X)
        conv_in = bx
        if len(window):
            wn = g.input(window.shape, G.FP16)
            g.set_input(wn, window)
            conv_in = g.concat(wn, bx, 0)
        total = len(window) + L
        y = g.conv1d_causal(g.reshape(conv_in, (1, total, C)), nw, K, 1)
        y = g.reshape(g.slice(y, 1, total - L, L), (L, C))
        gated = g.multiply(Cg, y)
        g.set_input(nin, ip)
        g.set_input(nw, w)
        g.execute()
        got = g.get_output(gated).view(np.uint16)
        want = O.lfm2_conv_gate(ip, w.reshape(C, K), cache)
        assert np.array_equal(got, want), L
        fused = bits(H.lfm2_conv_gate(dev(ip[None]), dev(w.reshape(C, K)), state))[0]
        assert np.array_equal(fused, want), L
        window = np.concatenate([window, g.get_output(bx)])[-W:]
        g.soft_reset()
    g.close()


def test_lfm2_conv_gate_empty_and_invalid(H):
    C, K, W = 16, 3, 2
    w = dev(np.ones((C, K), np.float16))
    state = torch.zeros((1, W, C), dtype=torch.float16, device="cuda")
    empty = torch.empty((1, 0, 3 * C), dtype=torch.float16, device="cuda")
    assert H.lfm2_conv_gate(empty, w, state).numel() == 0              # no rows: nothing launched
    assert float(state.abs().sum()) == 0.0
    ip = dev(np.ones((1, 2, 3 * C), np.float16))
    out = torch.empty((1, 2, C), dtype=torch.float16, device="cuda")
    rc = H.lib.cactus_hip_lfm2_conv_gate_f16_dev(ip.data_ptr(), w.data_ptr(), None, out.data_ptr(), 1, 2, C, K, W, None)
    assert rc != 0 and b"null operand" in H.lib.cactus_hip_last_error()   # carried rows asked for, no state given
    rc = H.lib.cactus_hip_lfm2_conv_gate_f16_dev(ip.data_ptr(), w.data_ptr(), state.data_ptr(), out.data_ptr(), 1, 2, C, 0, W,
                                                 None)
    assert rc != 0


@pytest.mark.parametrize("C,K,W", [(2048, 3, 2), (512, 2, 1), (1536, 4, 3), (1024, 3, 0)])
def test_lfm2_conv_gate_quant_equals_separate_calls(H, C, K, W):
    """The one-launch conv block + activation quantization gives the gated rows, the carried state and the
    INT8 rows / row scales of the two separate calls, bit for bit (and the oracle's quantization)."""
    rng = np.random.default_rng(C + K)
    nseq = 3
    w = dev(halves(rng, (C, K), 0.5))
    st_a = torch.zeros((nseq, W, C), dtype=torch.float16, device="cuda") if W else None
    st_b = torch.zeros((nseq, W, C), dtype=torch.float16, device="cuda") if W else None
    for L in (6, 1, 1, 70, 1):
        ip = dev(halves(rng, (nseq, L, 3 * C), 2.0))
        if L == 6:
            ip[1, 2] = 0                                     # a zero row: scale floor 1e-10
        want = H.lfm2_conv_gate(ip, w, st_a)
        wq, ws = H.quantize_activations(want.view(nseq * L, C))
        q, sc, got = H.lfm2_conv_gate_quant(ip, w, st_b, want_gated=True)
        assert torch.equal(got, want) and torch.equal(q, wq), L
        assert torch.equal(sc.view(torch.int32), ws.view(torch.int32)), L
        if W:
            assert torch.equal(st_a, st_b), L
        oq, osc = O.quantize_activations(bits(want.view(nseq * L, C)))
        assert np.array_equal(q.cpu().numpy(), oq) and np.array_equal(sc.cpu().numpy().view(np.uint32), osc.view(np.uint32))
    q2, sc2 = H.lfm2_conv_gate_quant(ip, w, st_b)            # gated rows not wanted
    assert q2.shape == (nseq, C) and sc2.shape == (nseq,)
    with pytest.raises(H.CactusHipError):
        H.lfm2_conv_gate_quant(dev(halves(rng, (1, 2, 3 * 24))), dev(halves(rng, (24, 3))), None)
