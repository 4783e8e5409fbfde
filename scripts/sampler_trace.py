#!/usr/bin/env python3
"""Step-by-step shader-clock stamps of the sampler's first row (library built with -DSAMPLE_TRACE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cactus_amd.hip as H
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
lg = (torch.randn((1, 151936), device="cuda", generator=gen) * 3).half()
dr = torch.rand((1,), device="cuda", generator=gen)
smp = H.Sampler(1, 151936)
for _ in range(140):
    smp.sample(lg, 0.6, 0.95, 20, dr)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
H.lib.cactus_hip_debug_sample_trace.argtypes = [C.c_void_p]
H.lib.cactus_hip_debug_sample_trace(buf)
names = ["temperature pass", "penalty", "threshold bound", "candidates + selection", "ordered gather", "-", "top-p", "softmax + draw"]
for i, nm in enumerate(names):
    print(f"{nm:18s} {buf[i + 1] - buf[i]:8d} cycles")
print("  of candidates + selection: candidate pass", buf[9] - buf[3], "selection", buf[10] - buf[9], "placement", buf[4] - buf[10])
print("total", buf[8] - buf[0])
