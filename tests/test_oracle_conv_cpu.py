"""CPU tests of the oracle's causal depthwise convolution and LFM2 short-convolution block
(kernel_conv.cpp:15-110, model_lfm2.cpp:123-207, engine_cache.cpp:494-597).  The reference's tests
hold no vector for either, so the restatement is unpinned; it is checked here against float64 sums
and against itself (token-by-token calls through the circular cache equal one whole-sequence call)."""
import numpy as np
import pytest

from oracle import binding as O


def halves(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float16)


def conv_f64(x, w, dil):
    N, L, C = x.shape
    K = w.shape[1]
    y = np.zeros((N, L, C))
    for k in range(K):
        back = k * dil
        if back < L:
            y[:, back:, :] += x[:, :L - back, :].astype(np.float64) * w[:, K - 1 - k].astype(np.float64)
    return y


@pytest.mark.parametrize("N,L,C,K,dil", [(1, 1, 8, 3, 1), (2, 9, 5, 3, 1), (1, 40, 16, 8, 1), (1, 33, 7, 11, 2),
                                          (3, 17, 4, 16, 1), (1, 6, 3, 1, 1), (1, 5, 2, 9, 3)])
def test_conv_against_float64(N, L, C, K, dil):
    rng = np.random.default_rng(L * 100 + K)
    x, w = halves(rng, (N, L, C)), halves(rng, (C, K), 0.5)
    got = O.conv1d_causal_depthwise(x, w, dil).view(np.float16).astype(np.float64)
    ref = conv_f64(x, w, dil)
    assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())


def test_conv_exact_small_integers():
    x = np.arange(1, 7, dtype=np.float16).reshape(1, 6, 1)
    w = np.array([[1, 2, 3]], np.float16)          # y[t] = 3 x[t] + 2 x[t-1] + 1 x[t-2]
    got = O.conv1d_causal_depthwise(x, w).view(np.float16).ravel().tolist()
    assert got == [3, 8, 14, 20, 26, 32]


@pytest.mark.parametrize("K,W", [(3, 2), (3, 3), (4, 1), (9, 8), (2, 4)])
def test_lfm2_block_stepwise_equals_whole_sequence(K, W):
    """Rows fed in pieces through the circular cache give the rows of one call over the whole
    sequence, as long as the cache is at least K - 1 rows deep."""
    rng = np.random.default_rng(K * 10 + W)
    C, L = 12, 23
    ip, w = halves(rng, (L, 3 * C)), halves(rng, (C, K), 0.5)
    whole = O.lfm2_conv_gate(ip, w, O.Lfm2ConvState(W, C))
    st = O.Lfm2ConvState(W, C)
    pieces, t = [], 0
    for n in [1, 1, 4, 1, 2, 7, 1, 6]:
        pieces.append(O.lfm2_conv_gate(ip[t:t + n], w, st))
        t += n
    assert t == L
    got = np.concatenate(pieces)
    if W >= K - 1:
        assert np.array_equal(got, whole)
    else:
        assert not np.array_equal(got, whole)      # a cache shorter than the taps forgets rows
    assert st.count.value == min(L, W)


def test_lfm2_block_formula():
    rng = np.random.default_rng(3)
    C, L, K = 8, 10, 3
    ip, w = halves(rng, (L, 3 * C)), halves(rng, (C, K), 0.5)
    got = O.lfm2_conv_gate(ip, w, O.Lfm2ConvState(2, C)).view(np.float16).astype(np.float64)
    b, cg, x = ip[:, :C], ip[:, C:2 * C], ip[:, 2 * C:]
    bx = (b * x).astype(np.float16)
    y = conv_f64(bx[None], w, 1)[0].astype(np.float16)
    ref = (cg * y).astype(np.float64)
    assert np.abs(got - ref).max() <= 4e-3 * max(1.0, np.abs(ref).max())
