#!/usr/bin/env python3
"""Where a batch-1 decode step's time goes, launch by launch: phase stamps of the fused-quantization GEMV launches
(library built with -DFQ_TRACE, scripts/build_fq_trace.sh; run with CACTUS_HIP_LIBRARY=scripts/variants/libcactus_hip_fqtrace.so).
Stamps are 100 MHz wall-clock ticks shared by all CUs and launches: per launch the spread of workgroup entries, the
phase spans, and the time from the previous stamped launch's last exit to this launch's first entry (= the kernel
boundary, or the boundary + an unstamped launch such as attention + its boundary)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cactus_amd.hip as H
from cactus_amd import steps as ST

H.init(0)
LAUNCHES, WGS, PH = 512, 64, 8
gen = torch.Generator(device="cuda").manual_seed(5)
cfg = dict(ST.QWEN3_0P6B)
w = ST.qwen3_synthetic_weights(gen, cfg)
dec = ST.Qwen3Decode(w, cfg, 1, 513, cache_gen=gen)
dec.kv.set_positions([512])
dec.kv.tok.copy_(torch.randint(0, cfg["vocab"], (1,), dtype=torch.int32, device="cuda", generator=gen))
step = lambda: dec.step(advance=False)
step(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record()
torch.cuda.synchronize()
step_us = e0.elapsed_time(e1) * 1e3 / 5
buf = (C.c_ulonglong * (LAUNCHES * WGS * PH))()
n = C.c_uint(0)
H.lib.cactus_hip_debug_fq_trace.argtypes = [C.c_void_p, C.c_void_p]
assert H.lib.cactus_hip_debug_fq_trace(buf, C.byref(n)) == 0
tr = np.frombuffer(buf, dtype=np.uint64).reshape(LAUNCHES, WGS, PH).astype(np.int64)
total = n.value
per_step = 28 * 4 + 1                      # stamped launches of one step: 4 per block + the lm_head
last = [(total - per_step + i) % LAUNCHES for i in range(per_step)]
rows = []
prev_exit = None
names = ["norm+qkv", "o-proj", "add+norm+gate/up", "silu*up+down"]
agg = {}
for i, li in enumerate(last):
    t = tr[li]
    ok = t[:, 0] > 0
    t = t[ok]
    e0, e_last = t[:, 0].min(), t[:, 0].max()
    x_last = t[:, 5].max()
    name = names[i % 4] if i < 112 else "final norm + lm_head"
    d = {"launch": name, "wg_sampled": int(ok.sum()),
         "entry_spread_us": (e_last - e0) / 100.0,
         "launch_index_load_us": float(np.median(t[:, 6] - t[:, 0])) / 100.0,
         "rows_arrived_us": float(np.median(t[:, 2] - t[:, 6])) / 100.0,
         "absmax_quantize_barriers_us": float(np.median(t[:, 1] - t[:, 2])) / 100.0,
         "prologue_us": float(np.median(t[:, 1] - t[:, 6])) / 100.0,
         "stream_us": float(np.median(t[:, 3] - t[:, 1])) / 100.0,
         "closing_barrier_us": float(np.median(t[:, 4] - t[:, 3])) / 100.0,
         "chain_store_us": float(np.median(t[:, 5] - t[:, 4])) / 100.0,
         "first_entry_to_last_exit_us": (x_last - e0) / 100.0,
         "since_previous_stamped_exit_us": None if prev_exit is None else (e0 - prev_exit) / 100.0}
    prev_exit = x_last
    rows.append(d)
    a = agg.setdefault(name, [])
    a.append(d)
summary = {}
for name, ds in agg.items():
    summary[name] = {k: round(float(np.median([d[k] for d in ds if d[k] is not None])), 2)
                     for k in ds[0] if k not in ("launch", "wg_sampled")}
    summary[name]["launches"] = len(ds)
print(json.dumps({"step_us_with_stamps": round(step_us, 1), "per_launch_medians_us": summary}, indent=1))
if len(sys.argv) > 1:
    json.dump({"per_launch_medians_us": summary, "launches": rows}, open(sys.argv[1], "w"), indent=1)
