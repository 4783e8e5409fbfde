"""Per-step difference statistics of the Gemma-3-270m-shape greedy run (device graph path vs oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import binding as O
import cactus_amd.hip as H
H.init(0)
import cactus_amd.graph as G
from tests import test_gpu_greedy_decode as GD
rng = np.random.default_rng(12)
m = GD.gemma_model(rng, 0.02, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
prompt = rng.integers(0, GD.GEM["VOCAB"], 6).tolist()
g = G.CactusGraph(); wn = GD.bind(G, g, m); cache = G.KVCache(GD.GEM["LAYERS"], 2048, GD.GEM["HKV"], GD.GEM["D"])
oc = [None] * GD.GEM["LAYERS"]
feed, pos, dt, ot = prompt, 0, [], []
for step in range(64):
    lg_d = GD.gemma_graph_forward(G, g, wn, feed, pos, cache)
    ref = GD.gemma_oracle_forward(m, feed, pos, oc).view(np.float16).astype(np.float64)
    d = lg_d.astype(np.float64) - ref
    err = np.abs(d).max(); rms = np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean())
    td, to = int(np.argmax(lg_d)), int(np.argmax(ref)); srt = np.sort(ref)
    print(f"step {step} err/max {err / np.abs(ref).max():.4f} rms {rms:.4f} max {np.abs(ref).max():.1f} tok {td == to} margin/max {(srt[-1]-srt[-2])/np.abs(ref).max():.4f}", flush=True)
    dt.append(td); ot.append(to)
    pos += len(feed)
    feed = [to] if dt.count(to) <= 1 and ot.count(to) <= 1 else [int(rng.integers(0, GD.GEM["VOCAB"]))]
