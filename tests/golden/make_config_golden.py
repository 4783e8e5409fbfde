#!/usr/bin/env python3
"""Writes tests/golden/config_lfm2.txt and config_qwen.txt the way the reference's converter does
(python/src/cli.py:186-189: one `key=format_config_value(value)` line per entry, with
python/src/tensor_io.py:324-330 formatting booleans and lists).  Build container only (imports
/root/reference); the values are synthetic model dimensions.

    python tests/golden/make_config_golden.py
"""
import sys
from pathlib import Path

sys.path.insert(0, "/root/reference/python")
from src.tensor_io import format_config_value  # noqa: E402

HERE = Path(__file__).resolve().parent

CONFIGS = {
    "config_lfm2.txt": dict(vocab_size=65536, hidden_dim=2048, num_layers=16, attention_heads=32,
                            attention_kv_heads=8, attention_head_dim=64, ffn_intermediate_dim=8192,
                            layer_norm_eps=1e-05, rope_theta=1000000.0, tie_word_embeddings=True,
                            model_type="lfm2", conv_L_cache=3, precision="INT8", bos_token_id=1, eos_token_id=7,
                            layer_types=["conv", "conv", "full_attention", "conv", "conv", "full_attention"]),
    "config_qwen.txt": dict(vocab_size=151936, hidden_dim=1024, num_layers=28, attention_heads=16,
                            attention_kv_heads=8, attention_head_dim=128, ffn_intermediate_dim=3072,
                            layer_norm_eps=1e-06, rope_theta=1000000, tie_word_embeddings=False,
                            model_type="qwen", precision="FP16", model_variant="default", num_experts=0),
}

for name, cfg in CONFIGS.items():
    with open(HERE / name, "w") as f:
        for key, value in cfg.items():
            f.write(f"{key}={format_config_value(value)}\n")
    print("wrote", name)
