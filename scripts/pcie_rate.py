#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point cactus_hip_matmul_int8 (the reference's own
signature: host buffers in, host buffer out, synchronous) at M = 1, K = N = 8192: the first call
uploads the weights, later calls find them resident and move only the activation row and the output."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
H.init(0)
K = N = 8192
rng = np.random.default_rng(0)
B = rng.integers(-127, 128, (N, K), dtype=np.int8)
Bs = (0.01 + np.abs(rng.uniform(-1, 1, (N, K // 128))) * 0.05).astype(np.float16)
A = rng.integers(-127, 128, (1, K), dtype=np.int8)
As = np.array([0.01], np.float32)
Cc = np.zeros((1, N), np.float16)
p = lambda a: a.ctypes.data_as(C.c_void_p)
algo = N * K + 2 * N * K // 128 + K + 4 + 2 * N
t0 = time.perf_counter(); H.lib.cactus_hip_matmul_int8(p(A), p(As), p(B), p(Bs), p(Cc), 1, K, N, 128); t1 = time.perf_counter()
print(f"first call (weights uploaded): {(t1 - t0) * 1e3:.2f} ms = {algo / (t1 - t0) / 1e9:.1f} GB/s")
n = 50
t0 = time.perf_counter()
for _ in range(n):
    H.lib.cactus_hip_matmul_int8(p(A), p(As), p(B), p(Bs), p(Cc), 1, K, N, 128)
t1 = time.perf_counter()
dt = (t1 - t0) / n
print(f"later calls (weights resident): {dt * 1e6:.1f} us = {algo / dt / 1e9:.1f} GB/s")
