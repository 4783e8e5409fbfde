"""CPU tests of the oracle's restatement of the stochastic sampler (cactus_sample_f16,
cactus/kernel/kernel_nn.cpp:468-651), of the uniform draw and of the entropy figure of
Model::decode (cactus/engine/engine_model.cpp:255-290).

The reference holds no known-answer test for its sampler: the restatement is checked here against
hand-computed cases and float64 formulas (parity unpinned beyond that).  The uniform draw IS pinned:
tests/golden/uniform_draws.txt was produced by tests/golden/make_uniform_draws.cpp with libstdc++'s
std::mt19937 / std::uniform_real_distribution<float>, the calls the reference makes."""
import os

import numpy as np
import pytest

from oracle import binding as O

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_draws():
    out = []
    for line in open(os.path.join(HERE, "golden", "uniform_draws.txt")):
        seed, bits = line.split()
        out.append((int(seed), np.array([int(bits, 16)], np.uint32).view(np.float32)[0]))
    return out


def test_uniform_draw_matches_libstdcxx():
    for seed, want in golden_draws():
        assert O.sample_uniform(seed) == want, seed
    # std::mt19937's default seed: first output 3499211612
    assert O.sample_uniform(5489) == np.float32(3499211612) / np.float32(4294967296.0)


def test_library_uniform_draw_matches_libstdcxx():
    """The host helper of the C ABI (no device work)."""
    import cactus_amd.hip as H
    for seed, want in golden_draws():
        assert H.sample_uniform(seed) == want, seed


def test_top_k_one_is_argmax_for_any_draw():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(500) * 3).astype(np.float16)
    for d in (1e-6, 0.3, 0.999999):
        tok, hist = O.sample(x, 0.7, 0.9, 1, d, [])
        assert tok == int(np.argmax(x.astype(np.float32))) and hist == [tok]


def test_inverse_cdf_on_a_known_distribution():
    """logits = log p for p = (0.1, 0.2, 0.3, 0.4): draw d picks the first index whose running sum >= d."""
    x = np.log(np.array([0.1, 0.2, 0.3, 0.4])).astype(np.float16)
    picks = [O.sample(x, 1.0, 0.0, 4, d, None)[0] for d in (0.05, 0.2, 0.45, 0.7, 0.99)]
    assert picks == [0, 1, 2, 3, 3]
    # top-p 0.65: descending order 0.4, 0.3, 0.2, 0.1 -> cumulative 0.4, 0.7 (> 0.65 at position 1): only index 3 stays
    assert [O.sample(x, 1.0, 0.65, 4, d, None)[0] for d in (0.01, 0.5, 0.99)] == [3, 3, 3]
    # top-p 0.75: cut at position 2 -> indices 3 and 2 stay, renormalised 4/7 and 3/7
    assert [O.sample(x, 1.0, 0.75, 4, d, None)[0] for d in (0.1, 0.42, 0.44, 0.9)] == [2, 2, 3, 3]


def test_top_k_keeps_ties_with_the_kth_value():
    x = np.array([1.0, 3.0, 2.0, 2.0, 2.0, -1.0], np.float16)
    seen = {O.sample(x, 1.0, 0.0, 2, d, None)[0] for d in np.linspace(0.001, 0.999, 200)}
    assert seen == {1, 2, 3, 4}


def test_repetition_penalty_and_history_window():
    # two equal logits: after index 0 was sampled once, its logit is divided by 1.1 (positive) ...
    x = np.array([2.0, 2.0], np.float16)
    p1 = 1.0 / (1.0 + np.exp(float(np.float16(np.float16(2.0) * np.float16(1 / 1.1))) - 2.0) ** -1)
    tok, hist = O.sample(x, 1.0, 0.0, 2, 0.49, [0])
    assert hist == [0, tok]
    p0 = 1.0 - 1.0 / (1.0 + np.exp(float(np.float16(np.float16(2.0) * np.float16(1 / 1.1))) - 2.0))
    assert (tok == 0) == (0.49 <= p0) and p1 > 0
    # ... negative logits are multiplied by 1.1, once per occurrence
    x = np.array([-2.0, -2.0], np.float16)
    v = np.float16(-2.0)
    for _ in range(3):
        v = np.float16(np.float32(v) * np.float32(np.float16(1.1)))
    p0 = np.exp(float(v) + 2.0) / (1.0 + np.exp(float(v) + 2.0))
    assert O.sample(x, 1.0, 0.0, 2, p0 - 1e-3, [0, 0, 0])[0] == 0
    assert O.sample(x, 1.0, 0.0, 2, p0 + 1e-3, [0, 0, 0])[0] == 1
    # the history keeps the last 128 tokens, oldest dropped first
    hist = list(range(128))
    tok, hist2 = O.sample(np.zeros(300, np.float16), 1.0, 0.0, 300, 0.5, hist)
    assert hist2 == list(range(1, 128)) + [tok]


def test_greedy_parameters_and_degenerate_rows():
    x = np.array([0.5, 4.0, 4.0, -1.0], np.float16)
    assert O.sample(x, 0.0, 0.0, 0, 0.9, [7]) == (1, [7])           # greedy branch: no history update
    assert O.sample(x, 0.0, 0.5, 0, 0.0, None)[0] == 0             # draw 0: element 0 even if it was cut
    ninf = np.full(8, -np.inf, np.float16)
    assert O.sample(ninf, 1.0, 0.9, 3, 0.5, [1]) == (0, [1])        # no finite logit: token 0, nothing recorded


def test_entropy_against_float64():
    rng = np.random.default_rng(1)
    assert abs(O.logits_entropy(np.zeros(1000, np.float16)) - 1.0) < 1e-6
    one_hot = np.full(1000, -60.0, np.float16)
    one_hot[3] = 20.0
    assert O.logits_entropy(one_hot) < 1e-6
    x = (rng.standard_normal(5000) * 2).astype(np.float16)
    z = x.astype(np.float64)
    lp = z - (z.max() + np.log(np.exp(z - z.max()).sum()))
    p = np.exp(lp)
    want = -(p[p > 1e-10] * lp[p > 1e-10]).sum() / np.log(5000.0)
    assert O.logits_entropy(x) == pytest.approx(want, rel=1e-6)
