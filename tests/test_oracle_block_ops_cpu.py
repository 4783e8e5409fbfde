"""CPU tests of the oracle's restatements of the decoder-block ops against the known-answer
tests the reference holds for them (tests/test_graph.cpp, tests/test_kernel.cpp) and against
plain numpy formulas.  The reference's NEON sources do not build on x86, so beyond these loose
checks the restatements are unpinned."""
import numpy as np

from oracle import binding as O


def halves(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float16)


def test_binary_known_answers():
    """tests/test_graph.cpp:35-72 (exact small integers)."""
    f = lambda v: np.array(v, np.float16)
    assert O.binary("add", f([1, 2, 3, 4]), f([5, 6, 7, 8])).view(np.float16).tolist() == [6, 8, 10, 12]
    assert O.binary("subtract", f([10, 8, 6, 4]), f([2, 3, 1, 2])).view(np.float16).tolist() == [8, 5, 5, 2]
    assert O.binary("multiply", f([2, 3, 4, 5]), f([3, 4, 2, 2])).view(np.float16).tolist() == [6, 12, 8, 10]
    assert O.binary("divide", f([12, 15, 8, 9]), f([3, 5, 2, 3])).view(np.float16).tolist() == [4, 3, 4, 3]


def test_binary_equals_numpy_fp16():
    """numpy's float16 arithmetic rounds once, like vaddq_f16 / vmulq_f16 (kernel_blas.cpp:30-35)."""
    rng = np.random.default_rng(0)
    a, b = halves(rng, 5000, 8.0), halves(rng, 5000, 0.5)
    with np.errstate(all="ignore"):
        for op, fn in (("add", np.add), ("subtract", np.subtract), ("multiply", np.multiply), ("divide", np.divide)):
            want = fn(a, b).view(np.uint16)
            assert np.array_equal(O.binary(op, a, b), want), op


def test_rms_norm_known_answer():
    """tests/test_graph.cpp:176-200 (tolerance 0.01 there)."""
    x = np.arange(1, 9, dtype=np.float16)[None, :]
    got = O.rms_norm(x, np.ones(8, np.float16), 1e-5).view(np.float16).astype(np.float32)
    inv = 1.0 / np.sqrt(np.float32((x.astype(np.float32) ** 2).sum() / 8 + 1e-5))
    assert np.abs(got - x.astype(np.float32) * inv).max() < 0.01


def test_rms_norm_close_to_float_formula():
    rng = np.random.default_rng(1)
    x, w = halves(rng, (6, 200), 2.0), halves(rng, 200)
    got = O.rms_norm(x, w, 1e-6).view(np.float16).astype(np.float64)
    xf = x.astype(np.float64)
    want = xf / np.sqrt((xf ** 2).mean(-1, keepdims=True) + 1e-6) * w.astype(np.float64)
    assert np.abs(got - want).max() <= 0.004 * np.abs(want).max()


def test_rope_close_to_float_formula_and_changes_input():
    """tests/test_kernel.cpp:161-186 only asks that the output differs from the input."""
    rng = np.random.default_rng(2)
    q = halves(rng, (1, 3, 2, 16))
    got = O.rope(q, 10000.0, 5).view(np.float16).astype(np.float64)
    pos = np.arange(5, 8)[:, None]
    ang = pos / (10000.0 ** (2 * np.arange(8)[None, :] / 16))
    c, s = np.cos(ang)[None, :, None, :], np.sin(ang)[None, :, None, :]
    x1, x2 = q[..., :8].astype(np.float64), q[..., 8:].astype(np.float64)
    want = np.concatenate([x1 * c - x2 * s, x2 * c + x1 * s], -1)
    assert np.abs(got - want).max() < 4e-3
    assert not np.array_equal(O.rope(q, 10000.0, 5), q.view(np.uint16))


def test_rope_tables_rows_do_not_depend_on_table_length():
    a_c, a_s = O.rope_tables(0, 64, 64, 1e6)
    b_c, b_s = O.rope_tables(40, 8, 64, 1e6)
    assert np.array_equal(a_c[40:48], b_c) and np.array_equal(a_s[40:48], b_s)


def test_silu_gelu_follow_the_reference_approximations():
    """The 8-wide paths use a polynomial exp evaluated at zf*ln2 with 2^x coefficients
    (kernel_utils.h:112-135) and a rational tanh (:150-157): close to, not equal to, the exact
    functions; the scalar tails (n % 8) are the exact ones."""
    rng = np.random.default_rng(3)
    x = halves(rng, 4003, 3.0)
    xf = x.astype(np.float64)
    silu = O.silu(x).view(np.float16).astype(np.float64)
    gelu = O.gelu(x).view(np.float16).astype(np.float64)
    exact_silu = xf / (1 + np.exp(-xf))
    exact_gelu = 0.5 * xf * (1 + np.tanh(0.7978845608028654 * (xf + 0.044715 * xf ** 3)))
    assert np.abs(silu - exact_silu)[:4000].max() < 0.08
    assert np.abs(gelu - exact_gelu)[:4000].max() < 0.03
    assert np.abs(silu - exact_silu)[4000:].max() < 2e-3
    assert np.abs(gelu - exact_gelu)[4000:].max() < 2e-3


def test_unary_ranges_only_move_the_tails():
    """The reference's result depends on how many pieces its thread pool cuts a call into
    (kernel_utils.h:516-541) only through which elements fall into a piece's scalar tail."""
    rng = np.random.default_rng(3)
    x = halves(rng, 5003, 3.0)
    one, four = O.silu(x, 1), O.silu(x, 4)
    per = 5003 // 4
    tails = np.zeros(5003, bool)
    for t in range(4):
        s, e = t * per, (5003 if t == 3 else (t + 1) * per)
        tails[s + ((e - s) // 8) * 8: e] = True
    tails[(5003 // 8) * 8:] = True
    assert np.array_equal(one[~tails], four[~tails])




def test_embedding_known_answers():
    """tests/test_graph.cpp:778-816 is the ungrouped INT8 table (a plain cast); the grouped form
    multiplies by the row's group scale (graph_ops_tensor.cpp:243-254)."""
    table = np.array([[1, 5, 9, 2], [2, 6, 10, 3], [3, 7, 11, 4], [4, 8, 12, 5]], np.int8)
    ones = np.ones((4, 1), np.float16)
    got = O.embedding(table, np.array([0, 2, 3, 1], np.int32), ones, 4).view(np.float16)
    assert got.tolist() == [[1, 5, 9, 2], [3, 7, 11, 4], [4, 8, 12, 5], [2, 6, 10, 3]]
    half = np.full((4, 1), 0.5, np.float16)
    got = O.embedding(table, np.array([1], np.int32), half, 4).view(np.float16)
    assert got.tolist() == [[1, 3, 5, 1.5]]


def test_scalar_op_known_answers_and_numpy():
    """tests/test_graph.cpp:120-150 (exact small integers) and numpy's float16 arithmetic with the
    scalar rounded to half first (kernel_scalar.cpp:8)."""
    f = lambda v: np.array(v, np.float16)
    add = O.scalar_op("add", f([1, 2, 3, 4]), 5.0)
    assert O.scalar_op("multiply", add, 2.0).view(np.float16).tolist() == [12, 14, 16, 18]
    assert O.scalar_op("subtract", f([10, 8, 6, 4]), 2.0).view(np.float16).tolist() == [8, 6, 4, 2]
    assert O.scalar_op("divide", f([10, 8, 6, 4]), 2.0).view(np.float16).tolist() == [5, 4, 3, 2]
    rng = np.random.default_rng(4)
    x = halves(rng, 3001, 6.0)
    sc = np.float16(25.298221)                       # sqrt(640), Gemma's embedding scale, rounds in half
    with np.errstate(all="ignore"):
        for op, fn in (("add", np.add), ("subtract", np.subtract), ("multiply", np.multiply), ("divide", np.divide)):
            assert np.array_equal(O.scalar_op(op, x, 25.298221), fn(x, sc).view(np.uint16)), op
