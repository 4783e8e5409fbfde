"""Not collected (run by name): error of the matrix-core FP16 prefill attention against the CPU restatement, for
comparing kernel variants: python tests/wip_pf_err.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import binding as O
from tests.helpers import f16
import cactus_amd.hip as H
H.init(0)
for (B, S, KV, Hq, Hkv, D, off, causal, amp) in [(1, 1024, 1024, 2, 1, 128, 0, True, 1.0), (1, 300, 300, 4, 2, 128, 0, True, 1.0),
                                                 (2, 40, 75, 2, 1, 128, 0, False, 1.0), (1, 512, 512, 4, 2, 64, 0, True, 1.0),
                                                 (1, 512, 512, 2, 1, 128, 0, True, 4.0), (1, 2048, 2048, 1, 1, 128, 0, True, 2.0)]:
    rng = np.random.default_rng(S * 3 + KV + D)
    q = (rng.uniform(-1, 1, (B, S, Hq, D)) * amp).astype(np.float16)
    k = rng.uniform(-1, 1, (B, KV, Hkv, D)).astype(np.float16)
    v = rng.uniform(-1, 1, (B, KV, Hkv, D)).astype(np.float16)
    want = f16(O.attention_f16(q, k, v, 0.0, position_offset=off, is_causal=causal)).astype(np.float64)
    got = f16(H.attention_f16(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(), 0.0,
                              position_offset=off, is_causal=causal).cpu().view(torch.int16).numpy().view(np.uint16)).astype(np.float64)
    d = np.abs(got - want)
    tol = 2e-3 + 4e-3 * np.abs(want)
    print(f"S={S} KV={KV} D={D} amp={amp}: max |diff| {d.max():.3e}  mean {d.mean():.3e}  max diff/tolerance {(d / tol).max():.3f}  "
          f"exact-match fraction {(d == 0).mean():.3f}")
