"""GPU parity tests of the decode tail: the stochastic sampler (cactus_sample_f16's top-k / top-p /
repetition-penalty branch, cactus/kernel/kernel_nn.cpp:468-651) and the entropy figure of
Model::decode (cactus/engine/engine_model.cpp:255-290), called through the C ABI and compared with
the CPU oracle on the same seeded logits, draws and token histories.

Bar: sampled tokens and histories identical.  The device evaluates exp in double and rounds to
float where the oracle calls the host's expf, so a probability may differ in its last bit; the
seeded cases below hold hundreds of draws and none of them sits on such a boundary.  Entropy:
relative 1e-6 (double-precision sums combined in a different order, one float rounding).
"""
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


def logits_rows(rng, rows, vocab, kind):
    x = (rng.standard_normal((rows, vocab)) * 3).astype(np.float16)
    if kind == "ties":            # coarse values: many equal logits around the k-th place
        x = (np.round(x.astype(np.float32) * 2) / 2).astype(np.float16)
    elif kind == "peaked":
        x[:, 7 % vocab] = np.float16(30.0)
    elif kind == "flat":
        x[:] = np.float16(0.5)
    elif kind == "clustered":     # the large logits sit next to each other, the rest is one value
        hi = (np.abs(x[:, :400].astype(np.float32)) + 3).astype(np.float16)
        x[:] = np.float16(0.5)
        x[:, :400] = hi
    elif kind == "sparse":        # fewer finite logits than top_k
        keep = x[:, :30].copy()
        x[:] = np.float16(-np.inf)
        x[:, 5::(vocab // 30)][:, :30] = keep
    elif kind == "neginf":
        x[:, ::3] = np.float16(-np.inf)
        x[:, 1] = np.float16(-0.0)
        x[:, 2 % vocab] = np.float16(0.0)
    return x


CASES = [
    # rows, vocab, temperature, top_p, top_k, kind
    (4, 151936, 0.6, 0.95, 20, "normal"),
    (2, 151936, 1.0, 0.95, 64, "ties"),
    (3, 32000, 0.7, 0.8, 20, "normal"),
    (3, 32000, 0.2, 0.95, 20, "peaked"),
    (5, 1000, 0.0, 0.0, 5, "normal"),        # no scaling, no top-p
    (5, 1000, 0.0, 0.9, 0, "normal"),        # top-p over the whole (small) vocabulary
    (5, 4096, 1.3, 1.0, 0, "ties"),          # top_p = 1 disables the cut; every logit survives
    (4, 50, 0.9, 0.5, 64, "normal"),         # top_k beyond the vocabulary
    (4, 999, 0.8, 0.95, 40, "neginf"),
    (3, 1, 0.7, 0.9, 3, "normal"),
    (2, 65, 0.5, 0.3, 7, "ties"),
    # more survivors of top-k than one launch holds in LDS (4096): the second kernel's sums over the whole row
    (2, 151936, 0.8, 0.9, 0, "normal"),      # top-p over the whole vocabulary
    (2, 151936, 1.0, 0.5, 0, "ties"),        # the cut falls inside a bin of equal logits
    (3, 20000, 0.7, 1.0, 0, "normal"),       # no cut: the draw walks every logit
    (2, 20000, 0.0, 0.97, 6000, "neginf"),   # top_k above 4096, -inf entries
    (40, 9000, 1.1, 0.8, 0, "peaked"),       # more rows than the second kernel has workgroups
    (2, 10000, 1.0, 0.9, 0, "flat"),         # every logit equal
    # the top-k threshold: bound from the threads' maxima, then the exact selection over the candidates
    (3, 151936, 0.6, 0.95, 20, "peaked"),    # the sampled token repeats: the penalty lowers a thread's maximum
    (2, 151936, 0.9, 0.9, 1000, "normal"),   # k + history reaches 1024 threads after a few steps: selection over the row
    (2, 40000, 0.8, 0.9, 1024, "normal"),    # ... from the first step
    (2, 3000, 0.8, 0.9, 50, "sparse"),       # the bound falls at -inf: selection over the row
    (2, 151936, 1.0, 0.9, 300, "clustered"), # few threads hold the large logits: every element reaches the bound, more
                                             # candidates than LDS holds, selection over the row
    (2, 70000, 0.7, 0.95, 200, "ties"),      # many equal values around the k-th place
]


@pytest.mark.parametrize("rows,vocab,temperature,top_p,top_k,kind", CASES)
def test_sampler_tokens_and_history_match_oracle(H, rows, vocab, temperature, top_p, top_k, kind):
    rng = np.random.default_rng(rows * 1000003 + vocab + top_k)
    steps = 6
    smp = H.Sampler(rows, vocab)
    hist = [[] for _ in range(rows)]
    for step in range(steps):
        x = logits_rows(rng, rows, vocab, kind)
        seeds = rng.integers(1, 2**32, rows)
        draws = np.array([H.sample_uniform(int(s)) for s in seeds], np.float32)
        got = smp.sample(dev(x), temperature, top_p, top_k, dev(draws)).cpu().numpy().astype(np.uint32)
        for r in range(rows):
            want, hist[r] = O.sample(x[r], temperature, top_p, top_k, float(draws[r]), hist[r])
            assert got[r] == want, (step, r)
            assert smp.history_of(r) == hist[r], (step, r)


def test_sampler_history_window_and_penalty_over_many_steps(H):
    """150 steps on a small vocabulary: repeated tokens are penalised once per occurrence and the
    history drops its oldest entry beyond 128."""
    rng = np.random.default_rng(5)
    rows, vocab = 2, 40
    smp = H.Sampler(rows, vocab)
    hist = [[], []]
    x = (rng.standard_normal((rows, vocab)) * 2).astype(np.float16)
    for step in range(150):
        draws = rng.random(rows, dtype=np.float32)
        got = smp.sample(dev(x), 0.9, 0.95, 10, dev(draws)).cpu().numpy()
        for r in range(rows):
            want, hist[r] = O.sample(x[r], 0.9, 0.95, 10, float(draws[r]), hist[r])
            assert got[r] == want, (step, r)
    for r in range(rows):
        assert len(hist[r]) == 128 and smp.history_of(r) == hist[r]


def test_sampler_scaled_row_is_the_same_on_aligned_and_unaligned_rows(H):
    """The temperature pass keeps aligned rows packed (two binary16 products per instruction); a row whose length is
    not a multiple of 8 takes the element-wise path.  Same logits, same scaled values, bit for bit - over every finite
    binary16 pattern, both infinities and the subnormals."""
    allbits = np.arange(65536, dtype=np.uint16)
    vals = allbits.view(np.float16)
    vals = vals[~np.isnan(vals)]                       # 63490 values
    n = (len(vals) // 8) * 8
    a = vals[:n].reshape(1, n).copy()
    b = np.concatenate([a[0], np.array([1.0], np.float16)]).reshape(1, n + 1)
    draws = dev(np.array([0.5], np.float32))
    for temperature in (0.6, 1.3, 2.5):
        sa, sb = H.Sampler(1, n), H.Sampler(1, n + 1)
        sa.sample(dev(a), temperature, 0.9, 20, draws)
        sb.sample(dev(b), temperature, 0.9, 20, draws)
        wa = sa.work.cpu().view(torch.int16).numpy()[0]
        wb = sb.work.cpu().view(torch.int16).numpy()[0, :n]
        assert np.array_equal(wa, wb)


def test_sampler_edge_draws_and_no_history(H):
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((6, 300)) * 3).astype(np.float16)
    draws = np.array([0.0, 1.0, np.nextafter(np.float32(1), np.float32(0)), 1e-30, 0.5, 0.999], np.float32)
    work = torch.empty((6, 300), dtype=torch.float16, device="cuda")
    out = torch.empty((6,), dtype=torch.int32, device="cuda")
    import ctypes as C
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = H.lib.cactus_hip_sample_f16_dev(p(dev(x)), p(out), 6, 300, 0.8, 0.9, 12, p(dev(draws)), None, None, p(work),
                                         None, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for r in range(6):
        assert got[r] == O.sample(x[r], 0.8, 0.9, 12, float(draws[r]), None)[0], r
    assert got[0] == 0     # a draw of exactly 0 returns element 0 (the running sum is >= 0 there)


def test_sampler_greedy_parameters_take_the_argmax_branch(H):
    rng = np.random.default_rng(2)
    x = (rng.standard_normal((3, 5000)) * 3).astype(np.float16)
    smp = H.Sampler(3, 5000)
    got = smp.sample(dev(x), 0.0, 0.0, 0, dev(np.zeros(3, np.float32))).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, O.sample_greedy(x))
    assert smp.history_of(0) == []      # kernel_nn.cpp:451-467 returns before the history is touched


def test_sampler_ties_beyond_one_launch(H):
    """Only ties push a small top_k beyond what one launch holds: all 10000 equal logits survive top_k = 5
    (kernel_nn.cpp:526-533 removes values BELOW the k-th); the second kernel samples the row as the oracle does."""
    x = np.zeros((3, 10000), np.float16)
    x[2, ::2] = np.float16(0.5)           # 5000 ties at the top
    smp = H.Sampler(3, 10000)
    draws = np.array([0.5, 0.123, 0.77], np.float32)
    out = smp.sample(dev(x), 1.0, 0.9, 5, dev(draws)).cpu().numpy().astype(np.uint32)
    for r in range(3):
        assert out[r] == O.sample(x[r], 1.0, 0.9, 5, float(draws[r]), [])[0], r


def test_sampler_argument_errors(H):
    import ctypes as C
    x = dev(np.zeros((1, 10), np.float16))
    out = torch.empty((1,), dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert H.lib.cactus_hip_sample_f16_dev(p(x), p(out), 1, 10, 0.5, 0.9, 3, None, None, None, None, None, None) != 0
    assert b"draws" in H.lib.cactus_hip_last_error()
    assert H.lib.cactus_hip_sample_f16_dev(None, p(out), 1, 10, 0.5, 0.9, 3, None, None, None, None, None, None) != 0
    assert H.lib.cactus_hip_sample_f16_dev(None, None, 0, 10, 0.5, 0.9, 3, None, None, None, None, None, None) == 0


@pytest.mark.parametrize("rows,vocab", [(1, 151936), (4, 32000), (3, 1000), (2, 7)])
def test_logits_entropy_matches_oracle(H, rows, vocab):
    rng = np.random.default_rng(vocab)
    x = (rng.standard_normal((rows, vocab)) * 2.5).astype(np.float16)
    x[0, :] = np.float16(1.5)                      # uniform row: exactly log(vocab) / log(vocab)
    got = H.logits_entropy(dev(x)).cpu().numpy()
    for r in range(rows):
        assert got[r] == pytest.approx(O.logits_entropy(x[r]), rel=1e-6, abs=1e-7), r
    assert got[0] == pytest.approx(1.0, rel=1e-6)


def test_graph_sample_node(H):
    """CactusGraph::sample (graph_builder.cpp:396-420, graph_ops_sample.cpp:8-39): one token from the
    last logits row, the draw taken from the node's seed, the token history carried by the graph from
    step to step; against the oracle with the same draws and history."""
    import cactus_amd.graph as G
    rng = np.random.default_rng(77)
    vocab = 32000
    g = G.CactusGraph()
    history = []
    for step in range(6):
        x = logits_rows(rng, 3, vocab, "normal")
        if history:
            x[2, history[-1]] = np.float16(12.0)     # the last token again: the repetition penalty must act
        n = g.input(x.shape, G.FP16)
        tok = g.sample(n, 0.7, 0.9, 20, random_seed=1000 + step)
        g.set_input(n, x)
        g.execute()
        got = int(g.get_output(tok).view(np.uint32)[0])
        want, history = O.sample(x[2], 0.7, 0.9, 20, O.sample_uniform(1000 + step), history)
        assert got == want, step
        g.soft_reset()
    # greedy parameters: the first maximum of the last row, no history update
    x = logits_rows(rng, 2, vocab, "ties")
    n = g.input(x.shape, G.FP16)
    tok = g.sample(n, 0.0, 0.0, 0, random_seed=5)
    g.set_input(n, x)
    g.execute()
    assert int(g.get_output(tok).view(np.uint32)[0]) == int(O.sample_greedy(x[1:2])[0])
    g.soft_reset()
    # a cleared history samples like a fresh process
    g.reset_sample_history()
    x = logits_rows(rng, 1, vocab, "normal")
    n = g.input(x.shape, G.FP16)
    tok = g.sample(n, 0.7, 0.9, 20, random_seed=9)
    g.set_input(n, x)
    g.execute()
    assert int(g.get_output(tok).view(np.uint32)[0]) == O.sample(x[0], 0.7, 0.9, 20, O.sample_uniform(9), [])[0]
    bad = g.input((vocab,), G.FP16)
    g.sample(bad, 0.7, 0.9, 20, 1)
    g.set_input(bad, x[0])
    with pytest.raises(Exception, match="2D logits"):
        g.execute()
    g.close()
