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


# --------------------------------------------------------------------------- config.txt
def test_model_config_files_written_by_the_reference_formatter():
    """tests/golden/config_*.txt were written with python/src/tensor_io.py:format_config_value the way
    python/src/cli.py:186-189 writes a model folder's config.txt; the parser follows
    Config::from_json (cactus/engine/engine_model.cpp:403-539)."""
    import cactus_amd.graph as G
    here = os.path.dirname(os.path.abspath(__file__))
    c = G.load_model_config(os.path.join(here, "golden", "config_lfm2.txt"))
    assert (c["vocab_size"], c["hidden_dim"], c["num_layers"], c["attention_heads"], c["attention_kv_heads"],
            c["attention_head_dim"], c["ffn_intermediate_dim"]) == (65536, 2048, 16, 32, 8, 64, 8192)
    assert c["model_type"] == 5 and c["precision"] == 0 and c["conv_L_cache"] == 3 and c["tie_word_embeddings"]
    assert c["layer_types"] == ["conv", "conv", "full_attention", "conv", "conv", "full_attention"]
    assert c["layer_norm_eps"] == np.float32(1e-5) and c["rope_theta"] == 1e6
    assert (c["bos_token_id"], c["eos_token_id"]) == (1, 7)
    # LFM2 sampling defaults (engine_model.cpp:520-523)
    assert (c["default_temperature"], c["default_top_p"], c["default_top_k"]) == (np.float32(0.3), np.float32(0.95), 20)
    q = G.load_model_config(os.path.join(here, "golden", "config_qwen.txt"))
    assert q["model_type"] == 0 and q["precision"] == 1 and not q["tie_word_embeddings"] and q["layer_types"] == []
    assert (q["default_temperature"], q["default_top_k"]) == (np.float32(0.6), 20)


def test_model_config_defaults_comments_and_errors(tmp_path):
    import cactus_amd.graph as G
    p = tmp_path / "config.txt"
    p.write_text("# comment\n\n  model_type =  gemma \nunknown_key=5\nno equals sign here\n"
                 "layer_types = ['conv', \"full_attention\"]\nvision_hidden_dim=768\ntie_word_embeddings=1\n")
    c = G.load_model_config(str(p))
    # untouched fields keep the reference's defaults (engine.h:44-59)
    assert (c["vocab_size"], c["num_layers"], c["hidden_dim"], c["attention_head_dim"]) == (151936, 28, 1024, 128)
    assert c["model_type"] == 1 and (c["default_temperature"], c["default_top_k"]) == (1.0, 64)
    assert c["layer_types"] == ["conv", "full_attention"] and c["precision"] == 2
    with pytest.raises(Exception, match="Failed to open config file"):
        G.load_model_config(str(tmp_path / "missing.txt"))
    p.write_text("vocab_size=abc\n")           # std::stoul throws in the reference
    with pytest.raises(Exception):
        G.load_model_config(str(p))


def test_mmap_weights_rejects_truncated_payload_before_touching_the_graph(tmp_path):
    """A packed INT4 file whose payload is shorter than rows x K/2 (or an INT8 one shorter than rows x K)
    would make a matmul read past its device allocation: the file is refused while it is being
    validated on the host, and the graph gains no node."""
    import struct
    import cactus_amd.graph as G
    for kind in ("file_int4", "file_int8"):
        raw = bytearray(np.load(os.path.join(GOLD, FIXTURES[0]))[kind].tobytes())
        h = G.parse_cact_header(bytes(raw))
        short = h["data_bytes"] // 2
        struct.pack_into("<Q", raw, 56, short)                       # header claims half the payload
        p = tmp_path / (kind + ".weights")
        p.write_bytes(bytes(raw[: h["data_offset"] + short]))
        g = G.CactusGraph()
        with pytest.raises(Exception, match="payload smaller than the tensor"):
            g.mmap_weights(str(p))
        assert g.get_node_count() == 0
        g.close()
