"""Greedy decoding of a Qwen3-style model (models/model_qwen.cpp:51-129, 143-172; Model::decode,
engine_model.cpp:221-253) on the device against the CPU oracle: the same prompt, then token after
token with the INT8 KV cache, and the two token streams must be identical.

Dimensions are Qwen3-0.6B's (hidden 1024, 16 q / 8 kv heads of 128, ffn 3072, eps 1e-6, theta 1e6,
tied INT8 embedding / lm_head; engine.h:44-59): two cut-down cases (4 layers, 8192 entries) and the
whole shape (28 layers, vocabulary 151936, 64 steps).  The weights are synthetic (no model files exist
offline).  Logits are held to an fp16 tolerance -- attention sums in another order on the device -- and
the argmax must agree at every step outside near ties of the oracle's own logits."""
import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import rand_weights_int8

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HID, HQ, HKV, D, FF = 1024, 16, 8, 128, 3072
EPS, THETA = 1e-6, 1e6


@pytest.fixture(scope="module")
def G():
    import cactus_amd.hip as hip
    hip.init(0)
    import cactus_amd.graph as graph
    return graph


def make_model(rng, LAYERS, VOCAB, gain):
    # weight = int8 ~ U{-127..127} times a group scale of (0.01 + |U(-1,1)| * 0.05) * gain: standard
    # deviation about 2.6 * gain
    def mat(n, k, gain):
        B, Bs = rand_weights_int8(rng, n, k)
        return B, (Bs.astype(np.float32) * gain).astype(np.float16)
    # The branches are kept small against the residual stream (embedding rows have standard deviation 2.6;
    # an attention or MLP branch adds about a tenth of that), as in a trained network: v / o / down take a
    # tenth of `gain`, gate / up a fifth.  With every matrix at the same gain each block multiplies the
    # stream by 25 or more, the stream forgets its past, and the model is a chaotic map whose logits move
    # by several per cent when one INT8 value anywhere flips - a test of nothing.
    rel = {"q": 1.0, "k": 1.0, "v": 0.1, "o": 0.1, "gate": 0.2, "up": 0.2, "down": 0.1}
    layers = []
    for _ in range(LAYERS):
        L = {name: mat(n, k, gain * rel[name]) for name, (n, k) in {"q": (HQ * D, HID), "k": (HKV * D, HID), "v": (HKV * D, HID),
                                                                    "o": (HID, HQ * D), "gate": (FF, HID), "up": (FF, HID),
                                                                    "down": (HID, FF)}.items()}
        for name, n in {"in_norm": HID, "post_norm": HID, "q_norm": D, "k_norm": D}.items():
            L[name] = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float16)
        layers.append(L)
    emb = mat(VOCAB, HID, 1.0)
    return {"layers": layers, "emb": emb, "final_norm": (1.0 + 0.1 * rng.standard_normal(HID)).astype(np.float16)}


def omm(x_bits, wt):
    q, s = O.quantize_activations(x_bits)
    return O.matmul_int8(q, s, wt[0], wt[1].view(np.uint16), 128)


def oracle_forward(m, tokens, pos, caches):
    """Logits (uint16 bits) of the last of `tokens`; caches: per layer [kq, ks, vq, vs] or None, extended."""
    S = len(tokens)
    h = O.embedding(m["emb"][0], np.asarray(tokens, np.int32), m["emb"][1].view(np.uint16), 128)
    for li, L in enumerate(m["layers"]):
        n1 = O.rms_norm(h, L["in_norm"], EPS)
        q = O.rms_norm(omm(n1, L["q"]).reshape(S * HQ, D), L["q_norm"], EPS).reshape(1, S, HQ, D)
        k = O.rms_norm(omm(n1, L["k"]).reshape(S * HKV, D), L["k_norm"], EPS).reshape(1, S, HKV, D)
        v = omm(n1, L["v"]).reshape(1, S, HKV, D)
        q, k = O.rope(q, THETA, pos), O.rope(k, THETA, pos)
        scale = 1.0 / np.sqrt(D)
        if caches[li] is None:
            attn = O.attention_f16(q, k, v, scale, position_offset=pos)
        else:
            kq, ks, vq, vs = caches[li]
            attn = O.attention_hybrid(q, kq[None], vq[None], ks[None], vs[None], k, v, scale, position_offset=pos)
        nkq, nks = O.quantize_kv(k.reshape(S, HKV, D))
        nvq, nvs = O.quantize_kv(v.reshape(S, HKV, D))
        if caches[li] is None:
            caches[li] = [nkq, nks, nvq, nvs]
        else:
            caches[li] = [np.concatenate([a, b]) for a, b in zip(caches[li], (nkq, nks, nvq, nvs))]
        h = O.binary("add", h, omm(attn.reshape(S, HQ * D), L["o"]))
        n2 = O.rms_norm(h, L["post_norm"], EPS)
        act = O.binary("multiply", O.silu(omm(n2, L["gate"])), omm(n2, L["up"]))
        h = O.binary("add", h, omm(act, L["down"]))
    last = O.rms_norm(h[S - 1:S], m["final_norm"], EPS)
    return omm(last, m.get("head", m["emb"]))[0]     # "head": an lm_head of its own instead of the tied embedding


def bind(G, g, m):
    def weight(pair):
        n = g.input(pair[0].shape, G.INT8)
        g.set_external_input(n, pair[0])
        g.set_grouped_scales(n, 128, pair[0].shape[1] // 128, pair[1])
        return n

    def vec(a):
        n = g.input(a.shape, G.FP16)
        g.set_external_input(n, a)
        return n
    wn = {"layers": [], "emb": weight(m["emb"]), "final_norm": vec(m["final_norm"])}
    wn["head"] = weight(m["head"]) if "head" in m else wn["emb"]
    for L in m["layers"]:
        wn["layers"].append({k: (weight(v) if isinstance(v, tuple) else vec(v)) for k, v in L.items()})
    return wn


def graph_forward(G, g, wn, tokens, pos, cache):
    S = len(tokens)
    idx = g.input((S,), G.FP32)
    g.set_input(idx, np.asarray(tokens, np.float32))
    h = g.embedding(wn["emb"], idx)
    k_nodes, v_nodes = [], []
    for li, L in enumerate(wn["layers"]):
        n1 = g.rms_norm(h, L["in_norm"], EPS)
        q, k, v = (g.matmul(n1, L[name], True) for name in ("q", "k", "v"))
        q = g.reshape(g.rms_norm(g.reshape(q, (S * HQ, D)), L["q_norm"], EPS), (1, S, HQ, D))
        k = g.reshape(g.rms_norm(g.reshape(k, (S * HKV, D)), L["k_norm"], EPS), (1, S, HKV, D))
        v = g.reshape(v, (1, S, HKV, D))
        q, k = g.rope(q, THETA, pos), g.rope(k, THETA, pos)
        k_nodes.append(k)
        v_nodes.append(v)
        scale = 1.0 / np.sqrt(D)
        if cache.is_empty():
            attn = g.attention(q, k, v, scale, pos)
        else:
            attn = g.attention_int8_hybrid(q, k, v, scale, pos, cache.get_keys_int8(li), cache.get_values_int8(li),
                                           cache.get_key_scales(li), cache.get_value_scales(li),
                                           cache.get_effective_seq_len(), HKV, D)
        h = g.add(h, g.matmul(g.reshape(attn, (S, HQ * D)), L["o"], True))
        n2 = g.rms_norm(h, L["post_norm"], EPS)
        act = g.multiply(g.silu(g.matmul(n2, L["gate"], True)), g.matmul(n2, L["up"], True))
        h = g.add(h, g.matmul(act, L["down"], True))
    last = g.reshape(g.index(g.rms_norm(h, wn["final_norm"], EPS), S - 1, 0), (1, HID))   # engine_model.cpp:241-246
    logits = g.matmul(last, wn.get("head", wn["emb"]), True)
    g.execute()
    out = g.get_output(logits).reshape(-1).copy()
    cache.update_from_graph(g, k_nodes, v_nodes, S)
    g.soft_reset()
    return out


@pytest.mark.parametrize("seed,prompt_len,steps,LAYERS,VOCAB,gain", [(3, 11, 14, 4, 8192, 0.05), (8, 1, 10, 4, 8192, 0.05),
                                                                      (5, 9, 64, 28, 151936, 0.05)])
def test_greedy_tokens_identical(G, seed, prompt_len, steps, LAYERS, VOCAB, gain):
    """Every op but attention reproduces the oracle bit for bit.  Decode attention runs on the matrix
    cores (fp16 products, fp32 sums in the MFMA's order, exp2): its outputs agree with the oracle's to
    within an fp16 ulp, not bit for bit, so about one value in a thousand rounds the other way and the
    following INT8 activation quantization now and then turns that into a different quantized value.
    To judge every step on its own, each step starts from the oracle's state: after a step the rows the
    device appended to its cache are compared with the oracle's and replaced by them.
    Bars per step: logits within 2e-2 + rel * max|logit|, rel = 5e-3 for 4 blocks and 1.5e-2 for 28
    (observed: 0.27 % and 0.87 %); the greedy token identical, or (at most twice in a run) different only
    where the oracle itself separates the two candidates by less than twice the difference between the
    rows.  The first step (prompt, FP16 attention in the oracle's summation order) and every step's
    first-block K/V rows (computed before any attention) must be bit-exact.  A synthetic model soon repeats
    one token: when the chosen token has been produced before, the run continues from a fresh seeded
    token instead."""
    import cactus_amd.hip as H
    rng = np.random.default_rng(seed)
    m = make_model(rng, LAYERS, VOCAB, gain)
    prompt = rng.integers(0, VOCAB, prompt_len).tolist()
    g = G.CactusGraph()
    wn = bind(G, g, m)
    cache = G.KVCache(LAYERS, 2048, HKV, D)
    ocaches = [None] * LAYERS

    tokens, near_ties = [], 0
    feed, pos = prompt, 0
    for step in range(steps):
        lg_d = graph_forward(G, g, wn, feed, pos, cache)
        lg_o = oracle_forward(m, feed, pos, ocaches).view(np.float16)
        ref = lg_o.astype(np.float64)
        err = np.abs(lg_d.astype(np.float64) - ref).max()
        exact = np.array_equal(lg_d.view(np.uint16), lg_o.view(np.uint16))
        assert err <= 2e-2 + (5e-3 if LAYERS <= 4 else 1.5e-2) * np.abs(ref).max(), (step, err, np.abs(ref).max())
        assert exact or step > 0
        td = int(H.sample_greedy(torch.from_numpy(lg_d[None].copy()).cuda()).cpu()[0])
        to = int(O.sample_greedy(lg_o[None])[0])
        tokens.append(to)
        if td != to:
            assert not exact and ref[to] - ref[td] <= 2 * err, (step, td, to, ref[to], ref[td], err)
            near_ties += 1
        # the rows this step appended: compared, then replaced by the oracle's so the next step starts equal
        n = len(feed)
        for li in range(LAYERS):
            dk, dks, dv, dvs = cache.download(li)
            ok_, oks, ov, ovs = (a[pos:pos + n] for a in ocaches[li])
            same = np.array_equal(dk[pos:], ok_) and np.array_equal(dks[pos:], oks) and \
                np.array_equal(dv[pos:], ov) and np.array_equal(dvs[pos:], ovs)
            assert same or li > 0, (step, li)
            if not same:
                cache.upload_rows(li, pos, ok_, oks, ov, ovs)
        pos += n
        feed = [to] if tokens.count(to) <= 1 else [int(rng.integers(0, VOCAB))]
    assert near_ties <= 2 and cache.get_total_seq_len() == prompt_len + steps - 1
    assert len(set(tokens)) > 1           # not a degenerate stream
    g.close()
    cache.close()


# ------------------------------------------------------------------------ Gemma-3-270m shape
# models/model_gemma.cpp:57-189: embedding scaled by sqrt(hidden), four norms per block (the
# attention and MLP outputs are normalized before their clipped residual adds), GELU, RoPE theta
# 10000 on the sliding-window blocks and 1e6 on every sixth (global) one, window 512 in the
# prefill attention.  Dimensions from the public model card (the reference reads config.txt):
# hidden 640, ffn 2048, 4 q / 1 kv heads of 256, 18 blocks, vocabulary 262144.
GEM = dict(HID=640, HQ=4, HKV=1, D=256, FF=2048, LAYERS=18, VOCAB=262144, EPS=1e-6)


def gemma_model(rng, gain, post=1.0):
    c = GEM

    def mat(n, k, g):
        B, Bs = rand_weights_int8(rng, n, k)
        return B, (Bs.astype(np.float32) * g).astype(np.float16)
    layers = []
    for _ in range(c["LAYERS"]):
        L = {name: mat(n, k, gain) for name, (n, k) in {
            "q": (c["HQ"] * c["D"], c["HID"]), "k": (c["HKV"] * c["D"], c["HID"]), "v": (c["HKV"] * c["D"], c["HID"]),
            "o": (c["HID"], c["HQ"] * c["D"]), "gate": (c["FF"], c["HID"]), "up": (c["FF"], c["HID"]),
            "down": (c["HID"], c["FF"])}.items()}
        for name, n in {"in_norm": c["HID"], "post_attn_norm": c["HID"], "pre_ffn_norm": c["HID"],
                        "post_ffn_norm": c["HID"], "q_norm": c["D"], "k_norm": c["D"]}.items():
            L[name] = ((post if name.startswith("post") else 1.0) * (1.0 + 0.1 * rng.standard_normal(n))).astype(np.float16)
        layers.append(L)
    # embedding rows of standard deviation about 0.05: scaled by sqrt(640) they are the size of one
    # normalized branch output, so the blocks, not the tied embedding, decide the next token
    return {"layers": layers, "emb": mat(c["VOCAB"], c["HID"], 0.02),
            "final_norm": (1.0 + 0.1 * rng.standard_normal(c["HID"])).astype(np.float16)}


def gemma_oracle_forward(m, tokens, pos, caches):
    c = GEM
    S, Hq, Hkv, Dh, eps = len(tokens), c["HQ"], c["HKV"], c["D"], c["EPS"]
    h = O.embedding(m["emb"][0], np.asarray(tokens, np.int32), m["emb"][1].view(np.uint16), 128)
    h = O.scalar_op("multiply", h, np.sqrt(np.float32(c["HID"])))
    for li, L in enumerate(m["layers"]):
        glob = (li + 1) % 6 == 0
        n1 = O.rms_norm(h, L["in_norm"], eps)
        q = O.rms_norm(omm(n1, L["q"]).reshape(S * Hq, Dh), L["q_norm"], eps).reshape(1, S, Hq, Dh)
        k = O.rms_norm(omm(n1, L["k"]).reshape(S * Hkv, Dh), L["k_norm"], eps).reshape(1, S, Hkv, Dh)
        v = omm(n1, L["v"]).reshape(1, S, Hkv, Dh)
        theta = 1e6 if glob else 1e4
        q, k = O.rope(q, theta, pos), O.rope(k, theta, pos)
        scale = 1.0 / np.sqrt(Dh)
        if caches[li] is None:
            attn = O.attention_f16(q, k, v, scale, position_offset=pos, window_size=0 if glob else 512)
        else:
            kq, ks, vq, vs = caches[li]
            attn = O.attention_hybrid(q, kq[None], vq[None], ks[None], vs[None], k, v, scale, position_offset=pos)
        new = O.quantize_kv(k.reshape(S, Hkv, Dh)) + O.quantize_kv(v.reshape(S, Hkv, Dh))
        caches[li] = list(new) if caches[li] is None else [np.concatenate([a, b]) for a, b in zip(caches[li], new)]
        ao = O.rms_norm(omm(attn.reshape(S, Hq * Dh), L["o"]), L["post_attn_norm"], eps)
        h = O.add_clipped(h, ao)
        n2 = O.rms_norm(h, L["pre_ffn_norm"], eps)
        act = O.binary("multiply", O.gelu(omm(n2, L["gate"])), omm(n2, L["up"]))
        h = O.add_clipped(h, O.rms_norm(omm(act, L["down"]), L["post_ffn_norm"], eps))
    return omm(O.rms_norm(h[S - 1:S], m["final_norm"], eps), m["emb"])[0]


def gemma_graph_forward(G, g, wn, tokens, pos, cache):
    c = GEM
    S, Hq, Hkv, Dh, eps = len(tokens), c["HQ"], c["HKV"], c["D"], c["EPS"]
    idx = g.input((S,), G.FP32)
    g.set_input(idx, np.asarray(tokens, np.float32))
    h = g.scalar_multiply(g.embedding(wn["emb"], idx), float(np.sqrt(np.float32(c["HID"]))))
    k_nodes, v_nodes = [], []
    for li, L in enumerate(wn["layers"]):
        glob = (li + 1) % 6 == 0
        n1 = g.rms_norm(h, L["in_norm"], eps)
        q, k, v = (g.matmul(n1, L[name], True) for name in ("q", "k", "v"))
        q = g.reshape(g.rms_norm(g.reshape(q, (S * Hq, Dh)), L["q_norm"], eps), (1, S, Hq, Dh))
        k = g.reshape(g.rms_norm(g.reshape(k, (S * Hkv, Dh)), L["k_norm"], eps), (1, S, Hkv, Dh))
        v = g.reshape(v, (1, S, Hkv, Dh))
        theta = 1e6 if glob else 1e4
        q, k = g.rope(q, theta, pos), g.rope(k, theta, pos)
        k_nodes.append(k)
        v_nodes.append(v)
        scale = 1.0 / np.sqrt(Dh)
        if cache.is_empty():
            attn = g.attention(q, k, v, scale, pos, 0 if glob else 512)
        else:
            attn = g.attention_int8_hybrid(q, k, v, scale, pos, cache.get_keys_int8(li), cache.get_values_int8(li),
                                           cache.get_key_scales(li), cache.get_value_scales(li),
                                           cache.get_effective_seq_len(), Hkv, Dh)
        ao = g.rms_norm(g.matmul(g.reshape(attn, (S, Hq * Dh)), L["o"], True), L["post_attn_norm"], eps)
        h = g.add_clipped(h, ao)
        n2 = g.rms_norm(h, L["pre_ffn_norm"], eps)
        act = g.multiply(g.gelu(g.matmul(n2, L["gate"], True)), g.matmul(n2, L["up"], True))
        h = g.add_clipped(h, g.rms_norm(g.matmul(act, L["down"], True), L["post_ffn_norm"], eps))
    last = g.reshape(g.index(g.rms_norm(h, wn["final_norm"], eps), S - 1, 0), (1, c["HID"]))
    logits = g.matmul(last, wn["emb"], True)
    g.execute()
    out = g.get_output(logits).reshape(-1).copy()
    cache.update_from_graph(g, k_nodes, v_nodes, S)
    g.soft_reset()
    return out


@pytest.mark.parametrize("post,rel,max_ties", [(1.0, 5e-2, 2), (0.15, 2.5e-2, 0)])
def test_gemma3_270m_shape_greedy_64_tokens(G, post, rel, max_ties):
    """The reference's CPU-runnable configuration (Gemma-3-270m, INT8, 64 greedy tokens) on a
    synthetic model of that shape: at every step the device's greedy token equals the oracle's and
    the logits agree.  Random weights with normalized branch outputs fall into a one-token loop, so
    whenever the chosen token was produced before, both sides continue from a fresh seeded token
    instead (the comparison of that step's choice has already been made).  At most two of the 64
    choices may differ, and only as near ties of the oracle's own logits.
    Second case: post-attention / post-MLP norm weights around 0.15 instead of 1, so that a block's output is a
    correction to the residual stream rather than its equal; one flipped INT8 activation then no longer reshapes
    the logits (observed: at most 1.7 % of the largest logit instead of 4 %) and all 64 tokens must be identical."""
    import cactus_amd.hip as H
    rng = np.random.default_rng(12)
    m = gemma_model(rng, 0.02, post)
    prompt = rng.integers(0, GEM["VOCAB"], 6).tolist()
    g = G.CactusGraph()
    wn = bind(G, g, m)
    cache = G.KVCache(GEM["LAYERS"], 2048, GEM["HKV"], GEM["D"])
    ocaches = [None] * GEM["LAYERS"]
    dev_tokens, ora_tokens, near_ties = [], [], 0
    feed, pos = prompt, 0
    for step in range(64):
        lg_d = gemma_graph_forward(G, g, wn, feed, pos, cache)
        lg_o = gemma_oracle_forward(m, feed, pos, ocaches).view(np.float16)
        ref = lg_o.astype(np.float64)
        err = np.abs(lg_d.astype(np.float64) - ref).max()
        # Differences start as single fp16 ulps in the attention output and are absorbed by the INT8
        # activation quantization of the next matmul until one flips a quantized value; from then on
        # the K/V rows in the two caches differ and Gemma's four norms per block carry the relative
        # difference through 18 blocks: observed up to 3.5 % of the largest logit, 0.6 % rms.
        assert err <= 2e-2 + rel * np.abs(ref).max(), (step, err, np.abs(ref).max())
        td = int(H.sample_greedy(torch.from_numpy(lg_d[None].copy()).cuda()).cpu()[0])
        to = int(O.sample_greedy(lg_o[None])[0])
        dev_tokens.append(td)
        ora_tokens.append(to)
        if td != to:
            # a synthetic model's logits are nearly flat (largest about 7 over 262144 entries): the two
            # sides may only part where the oracle itself separates the two candidates by less than
            # the difference between the logit rows; both then continue from the oracle's token
            assert ref[to] - ref[td] <= 2 * err, (step, td, to, ref[to], ref[td], err)
            near_ties += 1
            td = to
        pos += len(feed)
        feed = [td] if dev_tokens.count(td) <= 1 and ora_tokens.count(td) <= 1 else [int(rng.integers(0, GEM["VOCAB"]))]
    assert cache.get_total_seq_len() == 6 + 63 and near_ties <= max_ties
    g.close()
    cache.close()


# ---------------------------------------------------------------------------- LFM2-1.2B shape
# models/model_lfm2.cpp:123-331: blocks with the short convolution (in-projection to [L, 3, C], B * X,
# the ConvCache rows in front, 3-tap causal depthwise convolution, C gate, out-projection) or with
# attention (per-head q/k norms, RoPE, INT8 KV cache), each followed by the SwiGLU MLP.  Dimensions
# from the public model card (hidden 2048, ffn 8192, 32 q / 8 kv heads of 64, vocabulary 65536,
# conv_L_cache 3, eps 1e-5), all 16 blocks with attention in blocks 2, 5, 8, 10, 12, 14.
LFM = dict(HID=2048, HQ=32, HKV=8, D=64, FF=8192, VOCAB=65536, EPS=1e-5, THETA=1e6, K=3, W=2,
           TYPES="ccaccaccacacacac")


def lfm2_model(rng, gain):
    c = LFM

    def mat(n, k, g):
        B, Bs = rand_weights_int8(rng, n, k)
        return B, (Bs.astype(np.float32) * g).astype(np.float16)
    layers = []
    lo = 0.3 * gain     # the matrices that write to the residual stream (and v): small branches, see make_model
    for t in c["TYPES"]:
        L = {"gate": mat(c["FF"], c["HID"], gain), "up": mat(c["FF"], c["HID"], gain), "down": mat(c["HID"], c["FF"], lo),
             "in_norm": (1.0 + 0.1 * rng.standard_normal(c["HID"])).astype(np.float16),
             "post_norm": (1.0 + 0.1 * rng.standard_normal(c["HID"])).astype(np.float16)}
        if t == "a":
            L.update(q=mat(c["HQ"] * c["D"], c["HID"], gain), k=mat(c["HKV"] * c["D"], c["HID"], gain),
                     v=mat(c["HKV"] * c["D"], c["HID"], lo), o=mat(c["HID"], c["HQ"] * c["D"], lo),
                     q_norm=(1.0 + 0.1 * rng.standard_normal(c["D"])).astype(np.float16),
                     k_norm=(1.0 + 0.1 * rng.standard_normal(c["D"])).astype(np.float16))
        else:
            L.update(in_proj=mat(3 * c["HID"], c["HID"], gain), out_proj=mat(c["HID"], c["HID"], lo),
                     conv_w=(rng.standard_normal((c["HID"], 1, c["K"])) * 0.5).astype(np.float16))
        layers.append(L)
    return {"layers": layers, "emb": mat(c["VOCAB"], c["HID"], 1.0),
            "final_norm": (1.0 + 0.1 * rng.standard_normal(c["HID"])).astype(np.float16)}


def lfm2_oracle_forward(m, tokens, pos, kv, conv):
    c = LFM
    S, Hq, Hkv, Dh, eps = len(tokens), c["HQ"], c["HKV"], c["D"], c["EPS"]
    h = O.embedding(m["emb"][0], np.asarray(tokens, np.int32), m["emb"][1].view(np.uint16), 128)
    for li, L in enumerate(m["layers"]):
        n1 = O.rms_norm(h, L["in_norm"], eps)
        if "q" in L:
            q = O.rms_norm(omm(n1, L["q"]).reshape(S * Hq, Dh), L["q_norm"], eps).reshape(1, S, Hq, Dh)
            k = O.rms_norm(omm(n1, L["k"]).reshape(S * Hkv, Dh), L["k_norm"], eps).reshape(1, S, Hkv, Dh)
            v = omm(n1, L["v"]).reshape(1, S, Hkv, Dh)
            q, k = O.rope(q, c["THETA"], pos), O.rope(k, c["THETA"], pos)
            scale = 1.0 / np.sqrt(Dh)
            if kv[li] is None:
                attn = O.attention_f16(q, k, v, scale, position_offset=pos)
            else:
                kq, ks, vq, vs = kv[li]
                attn = O.attention_hybrid(q, kq[None], vq[None], ks[None], vs[None], k, v, scale, position_offset=pos)
            new = O.quantize_kv(k.reshape(S, Hkv, Dh)) + O.quantize_kv(v.reshape(S, Hkv, Dh))
            kv[li] = list(new) if kv[li] is None else [np.concatenate([a, b]) for a, b in zip(kv[li], new)]
            block = omm(attn.reshape(S, Hq * Dh), L["o"])
        else:
            gated = O.lfm2_conv_gate(omm(n1, L["in_proj"]), L["conv_w"].reshape(c["HID"], c["K"]), conv[li])
            block = omm(gated, L["out_proj"])
        h = O.binary("add", h, block)
        n2 = O.rms_norm(h, L["post_norm"], eps)
        act = O.binary("multiply", O.silu(omm(n2, L["gate"])), omm(n2, L["up"]))
        h = O.binary("add", h, omm(act, L["down"]))
    return omm(O.rms_norm(h[S - 1:S], m["final_norm"], eps), m["emb"])[0]


def lfm2_graph_forward(G, g, wn, m, tokens, pos, cache, windows):
    """The node sequence of LFM2Model::forward; `windows` holds each conv block's ConvCache rows on
    the host (oldest first), fed back as input nodes the way the reference feeds its circular views."""
    c = LFM
    S, Hq, Hkv, Dh, eps, C_, K_, W_ = len(tokens), c["HQ"], c["HKV"], c["D"], c["EPS"], c["HID"], c["K"], c["W"]
    idx = g.input((S,), G.FP32)
    g.set_input(idx, np.asarray(tokens, np.float32))
    h = g.embedding(wn["emb"], idx)
    k_nodes, v_nodes, bx_nodes = [], [], {}
    ai = 0
    for li, L in enumerate(wn["layers"]):
        n1 = g.rms_norm(h, L["in_norm"], eps)
        if "q" in L:
            q, k, v = (g.matmul(n1, L[name], True) for name in ("q", "k", "v"))
            q = g.reshape(g.rms_norm(g.reshape(q, (S * Hq, Dh)), L["q_norm"], eps), (1, S, Hq, Dh))
            k = g.reshape(g.rms_norm(g.reshape(k, (S * Hkv, Dh)), L["k_norm"], eps), (1, S, Hkv, Dh))
            v = g.reshape(v, (1, S, Hkv, Dh))
            q, k = g.rope(q, c["THETA"], pos), g.rope(k, c["THETA"], pos)
            k_nodes.append(k)
            v_nodes.append(v)
            scale = 1.0 / np.sqrt(Dh)
            if cache.is_empty():
                attn = g.attention(q, k, v, scale, pos)
            else:
                attn = g.attention_int8_hybrid(q, k, v, scale, pos, cache.get_keys_int8(ai), cache.get_values_int8(ai),
                                               cache.get_key_scales(ai), cache.get_value_scales(ai),
                                               cache.get_effective_seq_len(), Hkv, Dh)
            ai += 1
            block = g.matmul(g.reshape(attn, (S, Hq * Dh)), L["o"], True)
        else:
            trip = g.reshape(g.matmul(n1, L["in_proj"], True), (S, 3, C_))
            Bn, Cg, X = (g.reshape(g.slice(trip, 1, i, 1), (S, C_)) for i in range(3))
            bx = g.multiply(Bn, X)
            bx_nodes[li] = bx
            conv_in, total = bx, S
            if len(windows[li]):
                wnode = g.input(windows[li].shape, G.FP16)
                g.set_input(wnode, windows[li])
                conv_in, total = g.concat(wnode, bx, 0), S + len(windows[li])
            y = g.conv1d_causal(g.reshape(conv_in, (1, total, C_)), L["conv_w"], K_, 1)
            y = g.reshape(g.slice(y, 1, total - S, S), (S, C_))
            block = g.matmul(g.multiply(Cg, y), L["out_proj"], True)
        h = g.add(h, block)
        n2 = g.rms_norm(h, L["post_norm"], eps)
        act = g.multiply(g.silu(g.matmul(n2, L["gate"], True)), g.matmul(n2, L["up"], True))
        h = g.add(h, g.matmul(act, L["down"], True))
    last = g.reshape(g.index(g.rms_norm(h, wn["final_norm"], eps), S - 1, 0), (1, C_))
    logits = g.matmul(last, wn["emb"], True)
    g.execute()
    out = g.get_output(logits).reshape(-1).copy()
    cache.update_from_graph(g, k_nodes, v_nodes, S)
    for li, bx in bx_nodes.items():                                   # ConvCache::update
        windows[li] = np.concatenate([windows[li], g.get_output(bx)])[-W_:]
    g.soft_reset()
    return out


def test_lfm2_shape_greedy_tokens(G):
    import cactus_amd.hip as H
    c = LFM
    rng = np.random.default_rng(21)
    m = lfm2_model(rng, 0.008)
    prompt = rng.integers(0, c["VOCAB"], 7).tolist()
    g = G.CactusGraph()
    wn = bind(G, g, m)
    n_attn = c["TYPES"].count("a")
    cache = G.KVCache(n_attn, 2048, c["HKV"], c["D"])
    kv = [None] * len(c["TYPES"])
    conv = [O.Lfm2ConvState(c["W"], c["HID"]) for _ in c["TYPES"]]
    windows = [np.zeros((0, c["HID"]), np.float16) for _ in c["TYPES"]]
    toks, feed, pos = [], prompt, 0
    for step in range(12):
        lg_d = lfm2_graph_forward(G, g, wn, m, feed, pos, cache, windows)
        lg_o = lfm2_oracle_forward(m, feed, pos, kv, conv).view(np.float16)
        ref = lg_o.astype(np.float64)
        err = np.abs(lg_d.astype(np.float64) - ref).max()
        assert err <= 2e-2 + 1e-2 * np.abs(ref).max(), (step, err, np.abs(ref).max())    # observed: 0.55 %
        td = int(H.sample_greedy(torch.from_numpy(lg_d[None].copy()).cuda()).cpu()[0])
        to = int(O.sample_greedy(lg_o[None])[0])
        assert td == to, (step, toks, td, to, np.sort(ref)[-2:])
        toks.append(td)
        pos += len(feed)
        # a synthetic model soon repeats one token: continue from a fresh seeded one when it does
        feed = [td] if toks.count(td) == 1 else [int(rng.integers(0, c["VOCAB"]))]
    assert cache.get_total_seq_len() == 7 + 11
    g.close()
    cache.close()
