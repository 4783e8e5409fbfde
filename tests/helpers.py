"""Shared test helpers: CACT file parsing, synthetic inputs, error metrics."""
import struct

import numpy as np


def parse_cact(buf):
    """Parse the bytes of a CACT v1 tensor file (layout written by the reference's
    python/src/tensor_io.py:163-197 and read by cactus/graph/graph_io.cpp:497-566)."""
    b = bytes(buf)
    assert b[:4] == b"CACT"
    version, flags, align, ndim = struct.unpack_from("<IIII", b, 4)
    dims = struct.unpack_from("<4Q", b, 20)[:ndim]
    precision, = struct.unpack_from("<I", b, 52)
    data_bytes, scales_bytes = struct.unpack_from("<QQ", b, 56)
    group_size, num_groups = struct.unpack_from("<II", b, 72)
    hdr = (80 + align - 1) // align * align
    scales = np.frombuffer(b, dtype=np.uint16, count=scales_bytes // 2, offset=hdr).copy()
    data_off = (hdr + scales_bytes + align - 1) // align * align
    data = np.frombuffer(b, dtype=np.uint8, count=data_bytes, offset=data_off).copy()
    return dict(version=version, flags=flags, align=align, shape=tuple(dims), precision=precision,
                group_size=group_size, num_groups=num_groups, scales=scales, data=data)


def rand_weights_int8(rng, N, K, gs=128):
    """INT8 weights + fp16 group scales, shaped like the reference's bench inputs
    (tests/test_threading.cpp:79-87): B ~ U{-127..127}, scale = 0.01 + |U(-1,1)|*0.05."""
    B = rng.integers(-127, 128, (N, K), dtype=np.int8)
    Bs = (0.01 + np.abs(rng.uniform(-1, 1, (N, K // gs))) * 0.05).astype(np.float16)
    return B, Bs


def rand_weights_int4(rng, N, K, gs=128):
    """Packed INT4 bytes ~ U{0..255} (tests/test_performance.cpp:368-371)."""
    Bp = rng.integers(0, 256, (N, K // 2), dtype=np.uint8)
    Bs = (0.01 + np.abs(rng.uniform(-1, 1, (N, K // gs))) * 0.05).astype(np.float16)
    return Bp, Bs


def rand_act_f16(rng, M, K, lo=-1.0, hi=1.0):
    return rng.uniform(lo, hi, (M, K)).astype(np.float16)


def f16(a):
    """uint16 bit patterns -> float64 values."""
    return np.asarray(a).view(np.float16).astype(np.float64)


def max_abs_rel(got_bits, ref_bits):
    g, r = f16(got_bits), f16(ref_bits)
    ae = np.abs(g - r)
    return float(ae.max(initial=0.0)), float((ae / np.maximum(np.abs(r), 1e-6)).max(initial=0.0))


def ulp_diff_f16(a_bits, b_bits):
    """Distance in representable fp16 values between two bit-pattern arrays."""
    def key(x):
        x = np.asarray(x).astype(np.int32)
        return np.where(x & 0x8000, -(x & 0x7FFF), x & 0x7FFF)
    return np.abs(key(a_bits) - key(b_bits))
