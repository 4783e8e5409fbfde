"""CPU tests of the CACT header parser (MappedFile::parse_header, graph_io.cpp:497-566) against
files written by the reference's own python/src/tensor_io.py (tests/golden/*.npz hold the
complete file bytes) and against the independent Python parser in tests/helpers.py."""
import os

import numpy as np
import pytest

from tests.helpers import parse_cact

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = ["tensor_io_a.npz", "tensor_io_b.npz", "tensor_io_pad.npz"]


@pytest.fixture(scope="module")
def G():
    import cactus_amd.graph as graph   # loads the library; no GPU call is made by the parser
    return graph


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("kind,prec", [("file_int8", 0), ("file_int4", 3), ("file_fp16", 1)])
def test_header_of_reference_written_files(G, name, kind, prec):
    d = np.load(os.path.join(GOLD, name))
    raw = d[kind].tobytes()
    h = G.parse_cact_header(raw)
    ref = parse_cact(raw)
    N, K = ref["shape"]            # K is padded to whole groups by the writer (tensor_io.py)
    assert N == d["weights"].shape[0] and K >= d["weights"].shape[1]
    assert (h["version"], h["alignment"], h["ndim"], h["d0"], h["d1"], h["precision"]) == (1, 32, 2, N, K, prec)
    want_bytes = {0: N * K, 3: N * K // 2, 1: N * K * 2}[prec]
    assert h["data_bytes"] == want_bytes
    if prec in (0, 3):
        assert h["group_size"] == 128 and h["num_groups"] == (K + 127) // 128
        assert h["scales_bytes"] == N * h["num_groups"] * 2 and h["scales_offset"] == 96
    assert h["data_offset"] % 32 == 0 and h["data_offset"] + h["data_bytes"] <= len(raw)
    got = np.frombuffer(raw, np.uint8, h["data_bytes"], h["data_offset"])
    assert np.array_equal(got, np.frombuffer(ref["data"].tobytes(), np.uint8))


def test_header_errors(G):
    """graph_io.cpp:498-500 (short file), :507-509 (magic), :561-563 (payload past the end)."""
    raw = np.load(os.path.join(GOLD, FIXTURES[0]))["file_int8"].tobytes()
    with pytest.raises(G.CactusHipError, match="File too small"):
        G.parse_cact_header(raw[:40])
    with pytest.raises(G.CactusHipError, match="missing CACT magic"):
        G.parse_cact_header(b"XXXX" + raw[4:])
    with pytest.raises(G.CactusHipError, match="data extends beyond file size"):
        G.parse_cact_header(raw[:-1])
