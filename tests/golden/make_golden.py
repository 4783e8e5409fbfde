#!/usr/bin/env python3
"""Generate tests/golden/tensor_io_*.npz by running the reference's own weight
quantizer/writer (python/src/tensor_io.py:40-321) on seeded inputs.

Run in the build container only (it imports from /root/reference, which does
not exist on the GPU box):

    python tests/golden/make_golden.py

Each fixture holds the fp32 input matrix and the bytes of the file the
reference wrote (INT8, INT4 and FP16 variants).  Tests parse those bytes; they
are the pins for the INT4 nibble order/sign rule, the scale layout
[N, K/128] fp16 and the K-padding rule that the matmul kernels consume.
"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, "/root/reference/python")
from src.tensor_io import save_tensor_with_header  # noqa: E402

HERE = Path(__file__).resolve().parent

CASES = [
    # name, N, K, seed, distribution scale
    ("a", 8, 256, 1234, 0.02),
    ("b", 13, 384, 99, 1.0),
    ("pad", 5, 200, 7, 0.5),      # K not a multiple of 128 -> zero padded to 256
]


def main():
    for name, N, K, seed, sd in CASES:
        rng = np.random.default_rng(seed)
        w = (rng.standard_normal((N, K)) * sd).astype(np.float32)
        # put exact group extremes and ties in so clip/round-half-even paths run
        w[0, 0] = np.abs(w[0, :128]).max() * 1.5
        w[1, 5] = -np.abs(w[1, :128]).max() * 2.0
        out = {"weights": w}
        with tempfile.TemporaryDirectory() as td:
            for prec in ("INT8", "INT4", "FP16"):
                p = Path(td) / f"layer_ffn_{prec.lower()}.weights"
                save_tensor_with_header(w, p, precision=prec)
                out[f"file_{prec.lower()}"] = np.frombuffer(p.read_bytes(), dtype=np.uint8).copy()
        np.savez_compressed(HERE / f"tensor_io_{name}.npz", **out)
        print("wrote", f"tensor_io_{name}.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
