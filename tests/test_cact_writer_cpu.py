"""cactus_amd/cact.py writes tensor files byte for byte as the reference's converter does: every file held in
tests/golden (written by python/src/tensor_io.py, see make_golden.py) is parsed by an independent parser and
written again from its parts."""
import os

import numpy as np
import pytest

from cactus_amd import cact
from tests.helpers import parse_cact

HERE = os.path.dirname(__file__)


@pytest.mark.parametrize("name", ["tensor_io_a.npz", "tensor_io_b.npz", "tensor_io_pad.npz"])
@pytest.mark.parametrize("kind", ["file_int8", "file_int4", "file_fp16"])
def test_rewritten_golden_files_are_identical(name, kind):
    raw = np.load(os.path.join(HERE, "golden", name))[kind].tobytes()
    p = parse_cact(raw)
    scales = p["scales"] if p["scales"].size else None
    again = cact.cact_bytes(p["shape"], p["precision"], p["data"], scales, p["group_size"], flags=p["flags"],
                            alignment=p["align"], version=p["version"])
    assert again == raw
    if scales is not None:      # the default flags are the converter's
        assert cact.cact_bytes(p["shape"], p["precision"], p["data"], scales, p["group_size"]) == raw


def test_qwen_folder_layout(tmp_path):
    rng = np.random.default_rng(1)
    mat = lambda n, k: (rng.integers(-127, 128, (n, k), dtype=np.int8), rng.random((n, k // 128)).astype(np.float16))
    vec = lambda n: rng.random(n).astype(np.float16)
    cfg = dict(hidden=128, layers=2, q_heads=2, kv_heads=1, head_dim=64, ffn=256, eps=1e-6, theta=1e6)
    L = lambda: {"q": mat(128, 128), "k": mat(64, 128), "v": mat(64, 128), "o": mat(128, 128), "gate": mat(256, 128),
                 "up": mat(256, 128), "down": mat(128, 256), "in_norm": vec(128), "post_norm": vec(128),
                 "q_norm": vec(64), "k_norm": vec(64)}
    w = {"emb": mat(512, 128), "final_norm": vec(128), "layers": [L(), L()]}
    cact.write_qwen_folder(tmp_path, w, cfg)
    names = sorted(os.listdir(tmp_path))
    assert "config.txt" in names and "token_embeddings.weights" in names and "layer_1_ffn_down.weights" in names
    assert len(names) == 3 + 2 * 11
    p = parse_cact((tmp_path / "layer_0_attn_k.weights").read_bytes())
    assert p["shape"] == (64, 128) and p["precision"] == cact.INT8 and p["group_size"] == 128 and p["num_groups"] == 1
    assert (p["data"].view(np.int8).reshape(64, 128) == w["layers"][0]["k"][0]).all()
    assert (p["scales"] == w["layers"][0]["k"][1].view(np.uint16).reshape(-1)).all()
    p = parse_cact((tmp_path / "output_norm.weights").read_bytes())
    assert p["shape"] == (128,) and p["precision"] == cact.FP16 and (p["data"].view(np.uint16) == w["final_norm"].view(np.uint16)).all()
