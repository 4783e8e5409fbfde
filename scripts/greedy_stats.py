"""Per-step difference statistics of the full-shape Qwen3 greedy run (device graph path vs oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import binding as O
import cactus_amd.hip as H
H.init(0)
import cactus_amd.graph as G
from tests import test_gpu_greedy_decode as GD
seed, prompt_len, steps, LAYERS, VOCAB, gain = 5, 9, 64, 28, 151936, float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
rng = np.random.default_rng(seed)
m = GD.make_model(rng, LAYERS, VOCAB, gain)
prompt = rng.integers(0, VOCAB, prompt_len).tolist()
g = G.CactusGraph(); wn = GD.bind(G, g, m); cache = G.KVCache(LAYERS, 2048, GD.HKV, GD.D); oc = [None] * LAYERS
feed, pos, toks = prompt, 0, []
for step in range(steps):
    lg_d = GD.graph_forward(G, g, wn, feed, pos, cache)
    ref = GD.oracle_forward(m, feed, pos, oc).view(np.float16).astype(np.float64)
    d = lg_d.astype(np.float64) - ref
    err = np.abs(d).max(); rms = np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean())
    td, to = int(np.argmax(lg_d)), int(np.argmax(ref)); srt = np.sort(ref)
    print(f"step {step} err/max {err / np.abs(ref).max():.4f} rms {rms:.4f} max {np.abs(ref).max():.0f} tok {td == to} margin/max {(srt[-1]-srt[-2])/np.abs(ref).max():.4f}")
    pos += len(feed); toks.append(to)
    feed = [to] if toks.count(to) <= 1 else [int(rng.integers(0, VOCAB))]
