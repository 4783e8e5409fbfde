"""The whole-step wiring that bench.py times (cactus_amd/steps.py: batched decode steps of the Qwen3 and
LFM2 op sequences, the chunked Qwen3 prefill) against the CPU oracle, sequence by sequence: every
sequence of a batch must produce the logits (fp16 tolerance: attention sums in another order on the
device) and the greedy token that the oracle produces for that sequence run alone, step after step.
Cut-down models (a few blocks, a small vocabulary) with the real widths; synthetic weights.

Covered: batches of at most 4 (streaming launches with produced activations) and above (tile kernel,
shared launches); sequences of different lengths in one batch; the converter's MIXED INT4 / INT8
policy; the sliding window with sink held as a ring once sequences outgrow it; a three-chunk prefill."""
import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import rand_weights_int8, rand_weights_int4

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HID, HQ, HKV, D, FF = 1024, 16, 8, 128, 3072
EPS, THETA = 1e-6, 1e6


@pytest.fixture(scope="module")
def ST():
    import cactus_amd.hip as hip
    hip.init(0)
    from cactus_amd import steps
    return steps


def make_model(rng, layers, vocab, gain, mixed):
    from cactus_amd.steps import mixed_policy_int4

    def mat(n, k, g, int4=False):
        B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, n, k)
        # nibbles span [-8, 7] where bytes span [-127, 127]: larger group scales keep the weights' magnitude
        return B, (Bs.astype(np.float32) * g * (16.0 if int4 else 1.0)).astype(np.float16)
    # Branches small against the residual stream (embedding rows: standard deviation 2.6; a branch adds about a
    # tenth of that), as in a trained network.  With every matrix at one gain each block multiplies the
    # stream many times over and the model is a chaotic map: one flipped INT8 value anywhere moves the logits
    # by several per cent, and a comparison of logits says nothing.
    rel = {"q": 1.0, "k": 1.0, "v": 0.1, "o": 0.1, "gate": 0.2, "up": 0.2, "down": 0.1}
    Ls = []
    for li in range(layers):
        L = {name: mat(n, k, gain * rel[name], mixed and mixed_policy_int4(name, li, layers))
             for name, (n, k) in {"q": (HQ * D, HID), "k": (HKV * D, HID), "v": (HKV * D, HID), "o": (HID, HQ * D),
                                  "gate": (FF, HID), "up": (FF, HID), "down": (HID, FF)}.items()}
        for name, n in {"in_norm": HID, "post_norm": HID, "q_norm": D, "k_norm": D}.items():
            L[name] = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float16)
        Ls.append(L)
    B, Bs = rand_weights_int8(rng, vocab, HID)
    return {"layers": Ls, "emb": (B, Bs), "final_norm": (1.0 + 0.1 * rng.standard_normal(HID)).astype(np.float16)}


def omm(x_bits, wt):
    q, s = O.quantize_activations(x_bits)
    fn = O.matmul_int4 if wt[0].dtype == np.uint8 else O.matmul_int8
    return fn(q, s, wt[0], wt[1].view(np.uint16), 128)


def trim(cache, window, sink):
    """The reference's sliding window (engine_cache.cpp:268-329): first `sink` tokens + the most recent ones."""
    if not window or cache[0].shape[0] <= window:
        return cache
    return [np.concatenate([a[:sink], a[a.shape[0] - (window - sink):]]) for a in cache]


def oracle_forward(m, tokens, pos, caches, window=0, sink=0):
    """Logits bits of the last of `tokens` for ONE sequence; caches: per block [kq, ks, vq, vs] or None."""
    S = len(tokens)
    h = O.embedding(m["emb"][0], np.asarray(tokens, np.int32), m["emb"][1].view(np.uint16), 128)
    for li, L in enumerate(m["layers"]):
        n1 = O.rms_norm(h, L["in_norm"], EPS)
        q = O.rms_norm(omm(n1, L["q"]).reshape(S * HQ, D), L["q_norm"], EPS).reshape(1, S, HQ, D)
        k = O.rms_norm(omm(n1, L["k"]).reshape(S * HKV, D), L["k_norm"], EPS).reshape(1, S, HKV, D)
        v = omm(n1, L["v"]).reshape(1, S, HKV, D)
        q, k = O.rope(q, THETA, pos), O.rope(k, THETA, pos)
        if caches[li] is None:
            attn = O.attention_f16(q, k, v, 0.0, position_offset=pos)
        else:
            kq, ks, vq, vs = caches[li]
            attn = O.attention_hybrid(q, kq[None], vq[None], ks[None], vs[None], k, v, 0.0, position_offset=pos)
        new = O.quantize_kv(k.reshape(S, HKV, D)) + O.quantize_kv(v.reshape(S, HKV, D))
        new = [new[0], new[1], new[2], new[3]]
        caches[li] = new if caches[li] is None else [np.concatenate([a, b]) for a, b in zip(caches[li], new)]
        caches[li] = trim(caches[li], window, sink)
        h = O.binary("add", h, omm(attn.reshape(S, HQ * D), L["o"]))
        n2 = O.rms_norm(h, L["post_norm"], EPS)
        act = O.binary("multiply", O.silu(omm(n2, L["gate"])), omm(n2, L["up"]))
        h = O.binary("add", h, omm(act, L["down"]))
    last = O.rms_norm(h[S - 1:S], m["final_norm"], EPS)
    return omm(last, m.get("head", m["emb"]))[0]


def ring_slot(i, window, sink):
    return i if not window or i < window else sink + (i - window) % (window - sink)


def cache_token_indices(total, window, sink):
    """Absolute token index of every row of the oracle's (sliding-window) cache after `total` tokens."""
    if not window or total <= window:
        return list(range(total))
    return list(range(sink)) + list(range(total - (window - sink), total))


def load_caches(kvstate, index, b, caches, total, window=0, sink=0):
    """Oracle caches of one sequence -> the device caches of batch row b: the row of token i goes to the slot
    the device keeps token i in (slot i, or its ring slot once the window is full)."""
    slots = torch.tensor([ring_slot(i, window, sink) for i in cache_token_indices(total, window, sink)]).cuda()
    for li, ai in index.items():
        kq, ks, vq, vs = caches[li]
        assert kq.shape[0] == slots.numel()
        kvstate.kc[ai, b, slots] = torch.from_numpy(kq).cuda()
        kvstate.vc[ai, b, slots] = torch.from_numpy(vq).cuda()
        kvstate.ksc[ai, b, slots] = torch.from_numpy(ks).cuda()
        kvstate.vsc[ai, b, slots] = torch.from_numpy(vs).cuda()


class StepChecker:
    """Per (step, sequence) comparison of a device decode step that started from the oracle's state.
    All ops but attention reproduce the oracle bit for bit; decode attention runs on the matrix cores and
    agrees with the oracle to within an fp16 ulp per output (about one value in a thousand rounds the other
    way), which the next INT8 activation quantization now and then turns into a different quantized value.
    Bars: |diff| <= 2e-2 + 1e-2 * max|logit|; the greedy token equals the oracle's unless the oracle itself
    separates the two candidates by less than twice that difference."""

    def __init__(self):
        self.pairs = self.exact = self.near_ties = 0
        self.worst = 0.0
        self.seen = set()

    def check(self, lg_dev, tok_dev, lg_ref_bits, where):
        ref = lg_ref_bits.view(np.float16).astype(np.float64)
        self.pairs += 1
        to = int(O.sample_greedy(lg_ref_bits.view(np.float16)[None])[0])
        self.seen.add(to)
        if np.array_equal(lg_dev.view(np.uint16), lg_ref_bits):
            self.exact += 1
            assert tok_dev == to, where
            return to, True
        err = np.abs(lg_dev.astype(np.float64) - ref).max()
        self.worst = max(self.worst, err / np.abs(ref).max())
        assert err <= 2e-2 + 1e-2 * np.abs(ref).max(), (where, err, np.abs(ref).max())
        if tok_dev != to:
            assert ref[to] - ref[tok_dev] <= 2 * err, (where, tok_dev, to, ref[to], ref[tok_dev], err)
            self.near_ties += 1
        return to, False


@pytest.mark.parametrize("lens,mixed,window", [
    ((5, 17, 40), False, 0),                     # at most 4 sequences: streaming launches, produced activations
    ((9, 3, 28), True, 0),                       # MIXED policy on the streaming path (q/k INT8, v INT4 launches)
    ((3, 9, 16, 31, 33, 40), False, 0),          # above 4: tile kernel, shared q/k/v and gate/up launches
    ((4, 12, 17, 25, 36), True, 0),              # MIXED policy on the tile path
    ((10, 20, 23), False, 24),                   # sliding window 24 with sink 4: sequences outgrow it at different steps
    ((6, 14, 19, 22, 23, 23), False, 24),
])
def test_qwen3_batched_decode_matches_oracle_per_sequence(ST, lens, mixed, window):
    layers, vocab, steps, sink = 6, 8192, 8, 4
    rng = np.random.default_rng(sum(lens) + mixed + window)
    m = make_model(rng, layers, vocab, 0.05, mixed)
    cfg = dict(ST.QWEN3_0P6B, layers=layers, vocab=vocab)
    batch = len(lens)
    dec = ST.Qwen3Decode(ST.weights_to_device(m), cfg, batch, 64, window=window, sink=sink, max_position=128)
    index = {li: li for li in range(layers)}
    caches, toks = [], []
    for b, n in enumerate(lens):        # oracle prefill of each sequence alone; its cache seeds the device's
        cb = [None] * layers
        prompt = rng.integers(0, vocab, n).tolist()
        lg = oracle_forward(m, prompt, 0, cb, window, sink)
        toks.append(int(O.sample_greedy(lg.view(np.float16)[None])[0]))
        caches.append(cb)
        load_caches(dec.kv, index, b, cb, n, window, sink)
    dec.kv.set_positions(lens)
    pos = list(lens)
    ck = StepChecker()
    exact_rows = total_rows = 0
    for step in range(steps):
        dec.kv.tok.copy_(torch.tensor(toks, dtype=torch.int32))
        dec.step()
        torch.cuda.synchronize()
        lg_d = dec.logits.cpu().numpy()
        tok_d = dec.kv.tok.cpu().tolist()
        kc_d, ksc_d = dec.kv.kc.cpu().numpy(), dec.kv.ksc.cpu().numpy()
        for b in range(batch):
            lg_o = oracle_forward(m, [toks[b]], pos[b], caches[b], window, sink)
            toks[b], exact = ck.check(lg_d[b], tok_d[b], lg_o, (step, b))
            # the token this step appended: the rows the device quantized into its slot against the oracle's
            slot = ring_slot(pos[b], window, sink)
            for li in range(layers):
                same = np.array_equal(kc_d[li, b, slot], caches[b][li][0][-1]) and \
                    np.array_equal(ksc_d[li, b, slot], caches[b][li][1][-1])
                total_rows += 1
                exact_rows += same
                if exact or li == 0:     # block 0's keys do not sit behind any attention
                    assert same, (step, b, li)
            pos[b] += 1
            load_caches(dec.kv, index, b, caches[b], pos[b], window, sink)    # next step starts from the oracle's state
    assert dec.kv.positions.cpu().tolist() == pos
    assert dec.kv.cache_lens.cpu().tolist() == [min(p, window) if window else p for p in pos]
    assert len(ck.seen) > 1 and ck.near_ties <= 2


@pytest.mark.parametrize("mixed", [False, True])
def test_qwen3_chunked_prefill_matches_oracle(ST, mixed):
    """Three chunks of 64 tokens: chunk 0 against nothing (FP16 attention), the others against the INT8 cache;
    the last token's logits and greedy choice, and the quantized K cache the chunks leave behind.
    mixed: 6 blocks under the converter's MIXED policy (INT4 attn_v / attn_output / MLP in blocks 2 and 3)."""
    layers, vocab, tokens, chunk = (6 if mixed else 4), 8192, 192, 64
    rng = np.random.default_rng(77 + mixed)
    m = make_model(rng, layers, vocab, 0.05, mixed)
    cfg = dict(ST.QWEN3_0P6B, layers=layers, vocab=vocab)
    pf = ST.Qwen3Prefill(ST.weights_to_device(m), cfg, tokens, chunk)
    prompt = rng.integers(0, vocab, tokens)
    pf.toks.copy_(torch.from_numpy(prompt.reshape(-1, chunk).astype(np.int32)))
    caches = [None] * layers
    for ci in range(tokens // chunk):
        pf.run_chunk(ci)
        lg_o = oracle_forward(m, prompt[ci * chunk:(ci + 1) * chunk].tolist(), ci * chunk, caches)
    torch.cuda.synchronize()
    StepChecker().check(pf.logits.cpu().numpy()[0], int(pf.tok.cpu()[0]), lg_o, "prefill")
    # block 0's keys depend on nothing but the embedding, the input norm and the k projection: bit-exact
    assert np.array_equal(pf.kc[0, 0].cpu().numpy(), caches[0][0])
    assert np.array_equal(pf.ksc[0, 0].cpu().numpy(), caches[0][1])
    # Later blocks sit behind attention outputs, which differ in fp16 ulps: a token whose INT8 activations
    # flipped anywhere has a different row in every block after it, and over 192 tokens times 4 blocks that
    # happens to a share of the tokens.  Block 1 (one attention upstream): nearly all values agree.
    frac = [(np.abs(pf.kc[li, 0].cpu().numpy().astype(np.int32) - caches[li][0].astype(np.int32)) <= 1).mean()
            for li in range(layers)]
    assert frac[1] > 0.9 and min(frac) > 0.5, frac


def test_qwen3_single_chunk_prefill_matches_oracle(ST):
    """The whole prompt as one chunk (Model::prefill with chunk_size >= the prompt, engine_model.cpp:198-201): FP16
    attention over all 160 tokens on the matrix cores, GEMMs with M = 160; logits, greedy choice, block 0's cache."""
    layers, vocab, tokens = 4, 8192, 160
    rng = np.random.default_rng(78)
    m = make_model(rng, layers, vocab, 0.05, False)
    cfg = dict(ST.QWEN3_0P6B, layers=layers, vocab=vocab)
    pf = ST.Qwen3Prefill(ST.weights_to_device(m), cfg, tokens, tokens)
    prompt = rng.integers(0, vocab, tokens)
    pf.toks.copy_(torch.from_numpy(prompt.reshape(1, tokens).astype(np.int32)))
    caches = [None] * layers
    pf.run_chunk(0)
    lg_o = oracle_forward(m, prompt.tolist(), 0, caches)
    torch.cuda.synchronize()
    StepChecker().check(pf.logits.cpu().numpy()[0], int(pf.tok.cpu()[0]), lg_o, "prefill")
    assert np.array_equal(pf.kc[0, 0].cpu().numpy(), caches[0][0])
    assert np.array_equal(pf.ksc[0, 0].cpu().numpy(), caches[0][1])


def test_qwen3_prefill_with_a_short_last_chunk(ST):
    """200 tokens in chunks of 64: the last chunk holds 8 (its matmuls run on the streaming kernels, its attention with
    fewer than 32 queries on the vector-ALU kernel); chunk by chunk against the oracle, then the whole prompt in one call
    with three chunks in flight against the chunk loop, bit for bit."""
    layers, vocab, tokens, chunk = 4, 8192, 200, 64
    rng = np.random.default_rng(81)
    m = make_model(rng, layers, vocab, 0.05, False)
    cfg = dict(ST.QWEN3_0P6B, layers=layers, vocab=vocab)
    pf = ST.Qwen3Prefill(ST.weights_to_device(m), cfg, tokens, chunk)
    assert pf.nchunks == 4
    prompt = rng.integers(0, vocab, tokens)
    pf.prompt.copy_(torch.from_numpy(prompt.astype(np.int32)))
    caches = [None] * layers
    for ci in range(pf.nchunks):
        pf.run_chunk(ci)
        lg_o = oracle_forward(m, prompt[ci * chunk:(ci + 1) * chunk].tolist(), ci * chunk, caches)
    torch.cuda.synchronize()
    StepChecker().check(pf.logits.cpu().numpy()[0], int(pf.tok.cpu()[0]), lg_o, "prefill")
    assert np.array_equal(pf.kc[0, 0].cpu().numpy(), caches[0][0])
    assert np.array_equal(pf.ksc[0, 0].cpu().numpy(), caches[0][1])
    ref = [t.clone() for t in (pf.logits, pf.tok, pf.kc, pf.vc, pf.ksc, pf.vsc)]
    for t in (pf.logits, pf.kc, pf.vc):
        t.fill_(1)
    pf.run(lanes=3)
    torch.cuda.synchronize()
    for a, b in zip(ref, (pf.logits, pf.tok, pf.kc, pf.vc, pf.ksc, pf.vsc)):
        assert torch.equal(a, b)


def conv_rows(cc):
    """The oracle's circular ConvCache rows of one block in temporal order, zeros for rows not held yet."""
    W, Cc = cc.W, cc.rows.shape[1]
    held, h = cc.count.value, cc.head.value
    rows = cc.rows[:held] if held < W else np.concatenate([cc.rows[h:], cc.rows[:h]])
    st = np.zeros((W, Cc), np.uint16)
    st[W - held:] = rows
    return torch.from_numpy(st.view(np.float16)).cuda()


def make_lfm2(ST, rng, types, mixed, vocab=4096, ffn=2048):
    """A cut-down LFM2 (one block per letter of `types`: c = short convolution, a = attention; hidden 2048): the host
    model, its device copy and the configuration."""
    from tests import test_gpu_greedy_decode as GD
    c = GD.LFM

    def mat(n, k, g, tolerant=False):
        if mixed and tolerant:
            B, Bs = rand_weights_int4(rng, n, k)
            return B, (Bs.astype(np.float32) * g * 16.0).astype(np.float16)
        B, Bs = rand_weights_int8(rng, n, k)
        return B, (Bs.astype(np.float32) * g).astype(np.float16)
    Ls = []
    for t in types:
        L = {"gate": mat(ffn, c["HID"], 0.01, True), "up": mat(ffn, c["HID"], 0.01, True), "down": mat(c["HID"], ffn, 0.003, True),
             "in_norm": (1.0 + 0.1 * rng.standard_normal(c["HID"])).astype(np.float16),
             "post_norm": (1.0 + 0.1 * rng.standard_normal(c["HID"])).astype(np.float16)}
        if t == "a":
            L.update(q=mat(c["HQ"] * c["D"], c["HID"], 0.01), k=mat(c["HKV"] * c["D"], c["HID"], 0.01),
                     v=mat(c["HKV"] * c["D"], c["HID"], 0.003, True), o=mat(c["HID"], c["HQ"] * c["D"], 0.003, True),
                     q_norm=(1.0 + 0.1 * rng.standard_normal(c["D"])).astype(np.float16),
                     k_norm=(1.0 + 0.1 * rng.standard_normal(c["D"])).astype(np.float16))
        else:
            L.update(in_proj=mat(3 * c["HID"], c["HID"], 0.01), out_proj=mat(c["HID"], c["HID"], 0.003),
                     conv_w=(rng.standard_normal((c["HID"], 1, c["K"])) * 0.5).astype(np.float16))
        Ls.append(L)
    m = {"layers": Ls, "emb": mat(vocab, c["HID"], 1.0),
         "final_norm": (1.0 + 0.1 * rng.standard_normal(c["HID"])).astype(np.float16)}
    attn_at = tuple(i for i, t in enumerate(types) if t == "a")
    cfg = dict(ST.LFM2_1P2B, layers=len(types), vocab=vocab, ffn=ffn, attention_layers=attn_at)
    dm = ST.weights_to_device(m)
    for L in dm["layers"]:
        if "conv_w" in L:
            L["conv_w"] = L["conv_w"].reshape(c["HID"], c["K"]).contiguous()
    return m, dm, cfg


@pytest.mark.parametrize("mixed", [False, True])
def test_lfm2_batched_decode_matches_oracle_per_sequence(ST, mixed, monkeypatch):
    """A cut-down LFM2 (conv, conv, attention, conv, attention; hidden 2048, ffn 2048, vocabulary 4096) for 5
    sequences of different lengths: conv state rows and INT8 caches per sequence on the device.
    mixed: packed INT4 for attn_v / attn_output / the MLP (the converter's MIXED policy, python/src/precision_config.py:
    52-124, here in every block so that a cut-down model has some); the convolution projections, attn_q / attn_k and the
    embedding stay INT8."""
    from tests import test_gpu_greedy_decode as GD
    monkeypatch.setattr(GD, "omm", omm)       # the oracle matmul that also takes packed INT4
    c = GD.LFM
    rng = np.random.default_rng(5 + mixed)
    types, steps = "ccaca", 6
    m, dm, cfg = make_lfm2(ST, rng, types, mixed)
    lens = (4, 9, 2, 13, 7)
    batch = len(lens)
    dec = ST.Lfm2Decode(dm, cfg, batch, 64, max_position=128)
    kvs, convs, toks = [], [], []
    for b, n in enumerate(lens):
        kv = [None] * len(types)
        conv = [O.Lfm2ConvState(c["W"], c["HID"]) for _ in types]
        prompt = rng.integers(0, cfg["vocab"], n).tolist()
        lg = GD.lfm2_oracle_forward(m, prompt, 0, kv, conv)
        toks.append(int(O.sample_greedy(lg.view(np.float16)[None])[0]))
        kvs.append(kv); convs.append(conv)
        load_caches(dec.kv, dec.attn_index, b, kv, n)
        for li in dec.state:
            dec.state[li][b] = conv_rows(conv[li])
    dec.kv.set_positions(lens)
    pos = list(lens)
    ck = StepChecker()
    for step in range(steps):
        dec.kv.tok.copy_(torch.tensor(toks, dtype=torch.int32))
        dec.step()
        torch.cuda.synchronize()
        lg_d = dec.logits.cpu().numpy()
        tok_d = dec.kv.tok.cpu().tolist()
        for b in range(batch):
            lg_o = GD.lfm2_oracle_forward(m, [toks[b]], pos[b], kvs[b], convs[b])
            toks[b], _ = ck.check(lg_d[b], tok_d[b], lg_o, (step, b))
            pos[b] += 1
            load_caches(dec.kv, dec.attn_index, b, kvs[b], pos[b])
            for li in dec.state:
                dec.state[li][b] = conv_rows(convs[b][li])
    assert ck.near_ties <= 2


@pytest.mark.parametrize("mixed", [False, True])
def test_lfm2_chunked_prefill_matches_oracle(ST, mixed, monkeypatch):
    """Three chunks of 32 tokens through conv, conv, attention, conv, attention blocks: the convolution rows carried
    from chunk to chunk on the device, chunk 0 with FP16 attention and the others against the INT8 cache; the last
    token's logits and greedy choice, the rows the first convolution block is left holding (bit-exact: nothing but the
    embedding, a norm and the in-projection in front of them) and the K cache.  Then the same prompt with three chunks
    in flight, which must leave the same bits everywhere."""
    from tests import test_gpu_greedy_decode as GD
    monkeypatch.setattr(GD, "omm", omm)
    c = GD.LFM
    rng = np.random.default_rng(15 + mixed)
    types, tokens, chunk = "ccaca", 96, 32
    m, dm, cfg = make_lfm2(ST, rng, types, mixed)
    pf = ST.Lfm2Prefill(dm, cfg, tokens, chunk, capacity=tokens + 8)
    pf.conv.fill_(3.0)                       # chunk 0 must not use what an earlier prompt left
    prompt = rng.integers(0, cfg["vocab"], tokens)
    pf.toks.copy_(torch.from_numpy(prompt.reshape(-1, chunk).astype(np.int32)))
    kv = [None] * len(types)
    conv = [O.Lfm2ConvState(c["W"], c["HID"]) for _ in types]
    for ci in range(tokens // chunk):
        pf.run_chunk(ci)
        lg_o = GD.lfm2_oracle_forward(m, prompt[ci * chunk:(ci + 1) * chunk].tolist(), ci * chunk, kv, conv)
    torch.cuda.synchronize()
    StepChecker().check(pf.logits.cpu().numpy()[0], int(pf.tok.cpu()[0]), lg_o, "lfm2 prefill")
    assert torch.equal(pf.conv[0, 0], conv_rows(conv[0]))
    attn_at = cfg["attention_layers"]
    frac = [(np.abs(pf.kc[ai, 0, :tokens].cpu().numpy().astype(np.int32) - kv[li][0].astype(np.int32)) <= 1).mean()
            for ai, li in enumerate(attn_at)]
    assert min(frac) > 0.9, frac
    for ai, li in enumerate([i for i, t in enumerate(types) if t == "c"]):
        want = conv_rows(conv[li]).float()       # blocks behind an attention: fp16 ulps upstream, a few flipped INT8 values
        close = ((pf.conv[ai, 0].float() - want).abs() <= 0.05 * want.abs().max()).float().mean()
        assert close > 0.9, (li, float(close))
    ref = [t.clone() for t in (pf.logits, pf.tok, pf.kc, pf.vc, pf.ksc, pf.vsc, pf.conv)]
    for t in (pf.logits, pf.kc, pf.vc, pf.conv):
        t.fill_(1)
    pf.run(lanes=3)
    torch.cuda.synchronize()
    for a, b in zip(ref, (pf.logits, pf.tok, pf.kc, pf.vc, pf.ksc, pf.vsc, pf.conv)):
        assert torch.equal(a[..., :tokens, :, :] if a.dim() == 5 else a, b[..., :tokens, :, :] if b.dim() == 5 else b)
    # a decode step carries on from the prefill's buffers: three tokens against the oracle, which carries on from its own
    dec = ST.Lfm2Decode(dm, cfg, 1, tokens + 8, max_position=tokens + 8)
    dec.adopt(pf)
    ck = StepChecker()
    tok = int(O.sample_greedy(lg_o.view(np.float16)[None])[0])
    for step in range(3):
        dec.kv.tok.copy_(torch.tensor([tok], dtype=torch.int32))
        dec.step()
        torch.cuda.synchronize()
        lg_o = GD.lfm2_oracle_forward(m, [tok], tokens + step, kv, conv)
        tok, _ = ck.check(dec.logits.cpu().numpy()[0], int(dec.kv.tok.cpu()[0]), lg_o, ("decode after prefill", step))
    assert dec.kv.positions.cpu().tolist() == [tokens + 3]


# ------------------------------------------------------------------ per-sequence positions and the ring window
def test_qk_norm_rope_at_per_sequence_positions_bit_exact(ST):
    """cactus_hip_qk_norm_rope_seq_f16_dev: every sequence rotated for its own position (model_qwen.cpp:63-81
    applied per sequence), and with v given the rotated k / v rows quantized into the slot of that position."""
    import cactus_amd.hip as H
    rng = np.random.default_rng(3)
    B, S, Hq, Hkv, Dh, slots = 4, 3, 4, 2, 128, 48
    q = rng.standard_normal((B, S, Hq, Dh)).astype(np.float16)
    k = rng.standard_normal((B, S, Hkv, Dh)).astype(np.float16)
    v = rng.standard_normal((B, S, Hkv, Dh)).astype(np.float16)
    qw = (1.0 + 0.1 * rng.standard_normal(Dh)).astype(np.float16)
    kw = (1.0 + 0.1 * rng.standard_normal(Dh)).astype(np.float16)
    positions = np.array([0, 7, 30, 41], np.int32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    kc = torch.zeros((B, slots, Hkv, Dh), dtype=torch.int8, device="cuda"); vc = torch.zeros_like(kc)
    ks = torch.ones((B, slots, Hkv, 1), dtype=torch.float32, device="cuda"); vs = torch.ones_like(ks)
    oq, ok = H.qk_norm_rope_seq(dev(q), dev(k), dev(v), dev(qw), dev(kw), 1e-6, 1e6, dev(positions), 64, kc, vc, ks, vs)
    torch.cuda.synchronize()
    bits = lambda t: t.cpu().view(torch.int16).numpy().view(np.uint16)
    for b in range(B):
        p = int(positions[b])
        wq = O.rope(O.rms_norm(q[b].reshape(S * Hq, Dh), qw, 1e-6).reshape(1, S, Hq, Dh), 1e6, p)
        wk = O.rope(O.rms_norm(k[b].reshape(S * Hkv, Dh), kw, 1e-6).reshape(1, S, Hkv, Dh), 1e6, p)
        assert np.array_equal(bits(oq[b:b + 1]), wq) and np.array_equal(bits(ok[b:b + 1]), wk), b
        kq_o, ks_o = O.quantize_kv(wk.reshape(S, Hkv, Dh).view(np.float16))
        vq_o, vs_o = O.quantize_kv(v[b])
        assert np.array_equal(kc[b, p:p + S].cpu().numpy(), kq_o) and np.array_equal(ks[b, p:p + S].cpu().numpy(), ks_o)
        assert np.array_equal(vc[b, p:p + S].cpu().numpy(), vq_o) and np.array_equal(vs[b, p:p + S].cpu().numpy(), vs_o)
        untouched = np.ones(slots, bool); untouched[p:p + S] = False
        assert not kc[b].cpu().numpy()[untouched].any()


@pytest.mark.parametrize("new_per_call", [1, 3])
def test_windowed_batched_append_holds_the_sliding_windows_tokens(ST, new_per_call):
    """Sequences at different positions appended to together, `new_per_call` tokens each, far past a window of 16
    with sink 4 (engine_cache.cpp:268-329 per sequence): after every call the device's ring, un-rotated, holds
    bit for bit the rows of the reference's sliding cache, the device-kept positions / cache lengths are
    right, and the batched attention over the ring equals the oracle's attention over the sliding cache
    (|diff| <= 2e-3 + 4e-3 |ref|: the ring only permutes the sums)."""
    import cactus_amd.hip as H
    rng = np.random.default_rng(40 + new_per_call)
    B, Hq, Hkv, Dh, window, sink, slots = 4, 4, 2, 128, 16, 4, 20
    S = new_per_call
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    kc = torch.zeros((B, slots, Hkv, Dh), dtype=torch.int8, device="cuda"); vc = torch.zeros_like(kc)
    ks = torch.ones((B, slots, Hkv, 1), dtype=torch.float32, device="cuda"); vs = torch.ones_like(ks)
    start = [0, 5, 11, 15]
    ocache = []
    for b in range(B):                      # sequences begin with different numbers of tokens (below the window)
        k0 = rng.uniform(-2, 2, (1, start[b], Hkv, Dh)).astype(np.float16)
        v0 = rng.uniform(-2, 2, (1, start[b], Hkv, Dh)).astype(np.float16)
        if start[b]:
            H.kv_append_windowed(dev(k0), dev(v0), kc[b:b + 1], vc[b:b + 1], ks[b:b + 1], vs[b:b + 1],
                                 torch.zeros((1,), dtype=torch.int32, device="cuda"))     # below the window: slot = position
        ocache.append(list(O.quantize_kv(k0[0])) + list(O.quantize_kv(v0[0])))
    positions = dev(np.array(start, np.int32))
    lens = dev(np.array(start, np.int32))
    pos = list(start)
    for call in range(12):
        kn = rng.uniform(-2, 2, (B, S, Hkv, Dh)).astype(np.float16)
        vn = rng.uniform(-2, 2, (B, S, Hkv, Dh)).astype(np.float16)
        qh = rng.uniform(-1, 1, (B, S, Hq, Dh)).astype(np.float16)
        # the step's attention sees the cache as it is plus the new tokens, then the cache takes them
        got = H.attention_hybrid_batched(dev(qh), kc, vc, ks, vs, dev(kn), dev(vn), 0.0, lens, positions)
        H.kv_append_windowed(dev(kn), dev(vn), kc, vc, ks, vs, positions, window, sink)
        H.kv_positions_advance(positions, lens, S, window)
        torch.cuda.synchronize()
        got = got.cpu().view(torch.int16).numpy().view(np.uint16)
        for b in range(B):
            kq, ksc, vq, vsc = ocache[b]
            want = O.attention_hybrid(qh[b:b + 1], kq[None], vq[None], ksc[None], vsc[None], kn[b:b + 1], vn[b:b + 1], 0.0,
                                      position_offset=pos[b], cache_len=kq.shape[0])
            g, w = got[b:b + 1].view(np.float16).astype(np.float64), want.view(np.float16).astype(np.float64)
            assert (np.abs(g - w) <= 2e-3 + 4e-3 * np.abs(w)).all(), (call, b)
            new = list(O.quantize_kv(kn[b])) + list(O.quantize_kv(vn[b]))
            ocache[b] = trim([np.concatenate([a, n]) for a, n in zip(ocache[b], new)], window, sink)
            pos[b] += S
            idx = [ring_slot(i, window, sink) for i in cache_token_indices(pos[b], window, sink)]
            assert np.array_equal(kc[b].cpu().numpy()[idx], ocache[b][0]), (call, b)
            assert np.array_equal(ks[b].cpu().numpy()[idx], ocache[b][1]), (call, b)
            assert np.array_equal(vc[b].cpu().numpy()[idx], ocache[b][2]), (call, b)
            assert np.array_equal(vs[b].cpu().numpy()[idx], ocache[b][3]), (call, b)
        assert positions.cpu().tolist() == pos and lens.cpu().tolist() == [min(p, window) for p in pos]


def test_windowed_append_argument_errors(ST):
    import cactus_amd.hip as H
    z16 = torch.zeros((1, 1, 1, 128), dtype=torch.float16, device="cuda")
    kc = torch.zeros((1, 8, 1, 128), dtype=torch.int8, device="cuda")
    sc = torch.ones((1, 8, 1, 1), dtype=torch.float32, device="cuda")
    p = torch.zeros((1,), dtype=torch.int32, device="cuda")
    with pytest.raises(H.CactusHipError, match="sink < window"):
        H.kv_append_windowed(z16, z16, kc, kc, sc, sc, p, window=4, sink=4)          # no recent region
    with pytest.raises(H.CactusHipError, match="sink < window"):
        H.kv_append_windowed(z16, z16, kc, kc, sc, sc, p, window=16, sink=4)         # window above the slots
    five = torch.zeros((1, 5, 1, 128), dtype=torch.float16, device="cuda")
    with pytest.raises(H.CactusHipError, match="sink < window"):
        H.kv_append_windowed(five, five, kc, kc, sc, sc, p, window=8, sink=4)        # more new tokens than recent slots


@pytest.mark.parametrize("B,Hq,Hkv,D,slots,lens", [
    (1, 16, 8, 128, 600, (512,)),                     # Qwen3 heads, key ranges split over waves
    (3, 16, 8, 128, 80, (5, 40, 79)),                 # different lengths; the last sequence fills its cache
    (2, 32, 8, 64, 4200, (4100, 33)),                 # LFM2 heads, a long and a short sequence
    (70, 16, 1, 128, 40, tuple(range(1, 36)) * 2),    # 16 query heads on one kv head, many sequences
    (2, 6, 3, 128, 64, (64, 10)),                     # a full cache appends nothing and still attends
    (3, 16, 8, 128, 16, (0, 1, 9)),                   # a sequence with nothing cached yet
    (3, 4, 1, 256, 300, (256, 31, 299)),              # Gemma heads: two scale groups per cached row
    (2, 8, 4, 256, 5000, (4999, 4097)),               # ... key ranges split
])
def test_attention_decode_fused_equals_norm_rope_append_then_attention(ST, B, Hq, Hkv, D, slots, lens):
    """One launch from the projections' rows to the attention output == q/k norm + RoPE + cache append, then the
    batched attention: outputs, appended cache rows and scales bit for bit."""
    import cactus_amd.hip as H
    rng = np.random.default_rng(B + Hq + D + slots)
    gen = torch.Generator(device="cuda").manual_seed(B + D)
    half = lambda *sh: ((torch.rand(sh, device="cuda", generator=gen) * 2 - 1) * 3).half()
    q, k, v = half(B, 1, Hq, D), half(B, 1, Hkv, D), half(B, 1, Hkv, D)
    qw = (1.0 + 0.1 * torch.randn(D, device="cuda", generator=gen)).half()
    kw = (1.0 + 0.1 * torch.randn(D, device="cuda", generator=gen)).half()
    kc = torch.randint(-127, 128, (B, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vc = torch.randint(-127, 128, (B, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((B, slots, Hkv, (D + 127) // 128), device="cuda", generator=gen) * 0.02 + 1e-3
    vs = torch.rand((B, slots, Hkv, (D + 127) // 128), device="cuda", generator=gen) * 0.02 + 1e-3
    cl = torch.tensor(lens[:B], dtype=torch.int32, device="cuda")
    pos = cl + 3                                   # positions need not equal cache lengths (a slid window)
    eps, theta, maxpos = 1e-6, 1e6, slots + 8
    caches = [t.clone() for t in (kc, vc, ks, vs)]
    # reference sequence: norm + rope (+ append at slot = cache length), then attention
    slots_t = cl.clone()
    oq, ok = H.qk_norm_rope_seq(q, k, None, qw, kw, eps, theta, pos, maxpos)
    want = H.attention_hybrid_batched(oq, kc, vc, ks, vs, ok, v, 0.0, cl, pos)
    H.kv_append_batched(ok.reshape(B, Hkv, D), v.reshape(B, Hkv, D), kc, vc, ks, vs, slots_t)
    torch.cuda.synchronize()
    kc2, vc2, ks2, vs2 = caches
    got = H.attention_decode_fused(q.reshape(B, Hq, D), k.reshape(B, Hkv, D), v.reshape(B, Hkv, D), qw, kw, kc2, vc2,
                                   ks2, vs2, cl, pos, maxpos, eps, theta)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    for a, b_ in ((kc, kc2), (vc, vc2), (ks, ks2), (vs, vs2)):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("B,Hq,Hkv,D,slots,lens", [
    (3, 16, 8, 128, 80, (5, 40, 79)),                 # one key range per item: quantized inside the launch
    (130, 16, 8, 128, 600, tuple(range(1, 131))),     # Qwen3 heads, more than a thousand items
    (5, 32, 8, 64, 90, (0, 1, 33, 64, 89)),           # LFM2 heads; a sequence with nothing cached yet
    (2, 32, 8, 64, 4200, (4100, 33)),                 # key ranges split: the row kernel runs behind the attention
    (3, 4, 1, 256, 300, (256, 31, 299)),              # one kv head: likewise
])
def test_attention_decode_fused_quantized_rows(ST, B, Hq, Hkv, D, slots, lens):
    """The fused decode attention that also leaves its rows quantized for the output projection == the fused launch
    followed by the activation quantization: fp16 rows, INT8 rows, scales and appended cache rows bit for bit."""
    import cactus_amd.hip as H
    gen = torch.Generator(device="cuda").manual_seed(B + D + 1)
    half = lambda *sh: ((torch.rand(sh, device="cuda", generator=gen) * 2 - 1) * 3).half()
    q, k, v = half(B, Hq, D), half(B, Hkv, D), half(B, Hkv, D)
    qw = (1.0 + 0.1 * torch.randn(D, device="cuda", generator=gen)).half()
    kw = (1.0 + 0.1 * torch.randn(D, device="cuda", generator=gen)).half()
    kc = torch.randint(-127, 128, (B, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    vc = torch.randint(-127, 128, (B, slots, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
    ks = torch.rand((B, slots, Hkv, (D + 127) // 128), device="cuda", generator=gen) * 0.02 + 1e-3
    vs = torch.rand((B, slots, Hkv, (D + 127) // 128), device="cuda", generator=gen) * 0.02 + 1e-3
    cl = torch.tensor(lens[:B], dtype=torch.int32, device="cuda")
    pos = cl + 2
    eps, theta, maxpos = 1e-6, 1e6, slots + 8
    kc2, vc2, ks2, vs2 = (t.clone() for t in (kc, vc, ks, vs))
    want = H.attention_decode_fused(q, k, v, qw, kw, kc, vc, ks, vs, cl, pos, maxpos, eps, theta)
    want_q, want_s = H.quantize_activations(want.view(B, Hq * D))
    got_q = torch.full((B, Hq * D), 77, dtype=torch.int8, device="cuda")
    got_s = torch.full((B,), -1.0, dtype=torch.float32, device="cuda")
    got = H.attention_decode_fused(q, k, v, qw, kw, kc2, vc2, ks2, vs2, cl, pos, maxpos, eps, theta,
                                   out_q=got_q, out_scales=got_s)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(got_s.view(torch.int32), want_s.view(torch.int32))
    assert torch.equal(got_q, want_q)
    for a, b_ in ((kc, kc2), (vc, vc2), (ks, ks2), (vs, vs2)):
        assert torch.equal(a, b_)


# ------------------------------------------------------------------ two half-batches on two streams
def _run_free(ST, make, steps, split, graph):
    """`steps` free-running decode steps of a freshly built step object; returns logits per step, final tokens, caches."""
    prev = ST.decode_step_split(split)
    try:
        dec = make()
        out = []
        if graph:
            dec.step()                       # sizes the library workspaces outside the capture
            torch.cuda.synchronize()
            dec = make()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                dec.step()
            # the capture ran nothing: replay it `steps` times
            for _ in range(steps):
                g.replay()
                torch.cuda.synchronize()
                out.append(dec.logits.clone())
                dec.kv.host_max_position += 1
        else:
            for _ in range(steps):
                dec.step()
                torch.cuda.synchronize()
                out.append(dec.logits.clone())
        state = [dec.kv.tok.clone(), dec.kv.kc.clone(), dec.kv.vc.clone(), dec.kv.ksc.clone(), dec.kv.vsc.clone(),
                 dec.kv.positions.clone(), dec.kv.cache_lens.clone()]
        if getattr(dec, "_conv", None) is not None:
            state.append(dec._conv.clone())
        return out, state
    finally:
        ST.decode_step_split(prev)


@pytest.mark.parametrize("model,batch,window", [("qwen3", 150, 0), ("qwen3", 64, 0), ("qwen3", 37, 20), ("lfm2", 130, 0)])
def test_split_decode_step_equals_the_single_stream_step(ST, model, batch, window):
    """cactus_hip_decode_step_dev run as two half-batches on two streams (attention launches taking turns) gives, bit
    for bit, the logits, tokens, caches, positions and convolution rows of the single-stream step: eagerly and as a
    captured HIP graph; 4 free-running steps of cut-down models (every block kind, uneven halves, a ring window)."""
    steps, slots = 4, 48

    def make():
        gen = torch.Generator(device="cuda").manual_seed(1234 + batch)
        if model == "qwen3":
            cfg = dict(ST.QWEN3_0P6B, layers=3, vocab=4096)
            dec = ST.Qwen3Decode(ST.qwen3_synthetic_weights(gen, cfg), cfg, batch, slots, window=window, sink=4 if window else 0,
                                 max_position=slots + 16, cache_gen=gen)
        else:
            cfg = dict(ST.LFM2_1P2B, layers=4, vocab=4096, attention_layers=(1, 3))
            dec = ST.Lfm2Decode(ST.lfm2_synthetic_weights(gen, batch, cfg), cfg, batch, slots, max_position=slots + 16, cache_gen=gen)
        lens = torch.randint(1, slots - steps - 1, (batch,), generator=torch.Generator().manual_seed(batch)).tolist()
        dec.kv.set_positions(lens)
        dec.kv.tok.copy_(torch.randint(0, cfg["vocab"], (batch,), dtype=torch.int32, device="cuda", generator=gen))
        return dec

    want_lg, want_st = _run_free(ST, make, steps, 0, False)
    for graph in (False, True):
        got_lg, got_st = _run_free(ST, make, steps, 2, graph)
        for i, (a, b) in enumerate(zip(want_lg, got_lg)):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (graph, i)
        for i, (a, b) in enumerate(zip(want_st, got_st)):
            assert torch.equal(a, b), (graph, "state", i)


@pytest.mark.parametrize("lanes,mixed", [(2, False), (4, False), (3, True)])
def test_pipelined_prefill_equals_chunk_after_chunk(ST, lanes, mixed):
    """cactus_hip_prefill_dev (several chunks in flight on the library's own streams, a block's attention waiting for the
    previous chunk's rows in that block's cache) leaves, bit for bit, the caches, logits and token of the chunks run one
    after the other with cactus_hip_prefill_chunk_dev: eagerly, and as one captured HIP graph replayed twice."""
    layers, vocab, tokens, chunk = 5, 4096, 448, 64
    cfg = dict(ST.QWEN3_0P6B, layers=layers, vocab=vocab)

    def make():
        gen = torch.Generator(device="cuda").manual_seed(99 + lanes)
        pf = ST.Qwen3Prefill(ST.qwen3_synthetic_weights(gen, cfg, mixed), cfg, tokens, chunk)
        pf.toks.copy_(torch.randint(0, vocab, pf.toks.shape, dtype=torch.int32, device="cuda", generator=gen))
        return pf

    def state(pf):
        return [pf.logits.clone(), pf.tok.clone(), pf.kc.clone(), pf.vc.clone(), pf.ksc.clone(), pf.vsc.clone()]

    ref = make()
    for ci in range(ref.nchunks):
        ref.run_chunk(ci)
    torch.cuda.synchronize()
    want = state(ref)
    pf = make()
    pf.run(lanes)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(want, state(pf))):
        assert torch.equal(a, b), ("eager", i)
    pf = make()
    pf.run(lanes)                      # sizes the library workspaces outside the capture
    torch.cuda.synchronize()
    pf.kc.zero_(); pf.vc.zero_(); pf.logits.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pf.run(lanes)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(want, state(pf))):
        assert torch.equal(a, b), ("graph", i)
