#!/usr/bin/env python3
"""Prints the interesting figures of a bench.py JSON line (default gpurun_out/bench.json)."""
import json, sys
d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench.json").read().strip().splitlines()[-1])
e = d.get("extra", {})
print("value", d["value"], "frac", d["roofline"]["frac"], "us", d["roofline"]["us_per_launch"], "sep", d["config"].get("separate_launches_ms_per_step"))
for k in ("gemm_int8_M512", "gemm_int4_M512"):
    print(k, {s: (v["us"], v["frac_int8_mfma_peak"]) for s, v in e.get(k, {}).items()})
for k in ("gemv_int8_M1", "gemv_int8_M16", "gemv_int4_M1", "gemv_int4_M16"):
    print(k, {s: (v["us"], v["frac_hbm"]) for s, v in e.get(k, {}).items()})
for k in ("qwen3_0p6b_decode_step", "lfm2_1p2b_decode_step"):
    print(k, {s: v["ms_per_step"] for s, v in e.get(k, {}).items()})
print("prefill", {s: v["ms"] for s, v in e.get("qwen3_0p6b_prefill", {}).items()})
print("lfm2 prefill", {s: v["ms"] for s, v in e.get("lfm2_1p2b_prefill", {}).items()})
print("attn_decode", {s: (v["us"], v["frac_hbm"]) for s, v in e.get("attn_decode_int8kv", {}).items()})
print("attn_prefill", {s: (v["us"], v.get("TFLOPs")) for s, v in e.get("attn_prefill_f16_causal", {}).items()})
print("single_launch_8192", e.get("single_launch_8192", {}).get("us_per_launch"))
