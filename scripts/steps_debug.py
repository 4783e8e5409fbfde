"""Debug: batched decode step path vs graph-mirror path vs oracle for one scenario."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import binding as O
import cactus_amd.hip as H
H.init(0)
import cactus_amd.graph as G
from cactus_amd import steps as ST
from tests import test_gpu_steps as T
from tests import test_gpu_greedy_decode as GD

lens, mixed, window = (5, 17, 40), False, 0
layers, vocab, steps, sink = 6, 8192, 8, 4
rng = np.random.default_rng(sum(lens) + mixed + window)
m = T.make_model(rng, layers, vocab, 0.05, mixed)
cfg = dict(ST.QWEN3_0P6B, layers=layers, vocab=vocab)
batch = len(lens)
dec = ST.Qwen3Decode(ST.weights_to_device(m), cfg, batch, 64, window=window, sink=sink, max_position=128)
caches, toks, prompts = [], [], []
for b, n in enumerate(lens):
    cb = [None] * layers
    prompt = rng.integers(0, vocab, n).tolist()
    prompts.append(prompt)
    lg = T.oracle_forward(m, prompt, 0, cb, window, sink)
    toks.append(int(O.sample_greedy(lg.view(np.float16)[None])[0]))
    caches.append(cb)
    T.load_caches(dec, b, cb)
dec.kv.set_positions(lens)
dec.kv.tok.copy_(torch.tensor(toks, dtype=torch.int32))
# graph path for every sequence: prefill with graph too (its own cache), tokens forced to the oracle's
gs, wns, gcaches = [], [], []
for b in range(batch):
    g = G.CactusGraph(); wn = GD.bind(G, g, m); cache = G.KVCache(layers, 2048, 8, 128)
    lgd = GD.graph_forward(G, g, wn, prompts[b], 0, cache)
    gs.append(g); wns.append(wn); gcaches.append(cache)
pos = list(lens)
for step in range(steps):
    dec.step(); torch.cuda.synchronize()
    lg_d = dec.logits.cpu().numpy(); tok_d = dec.kv.tok.cpu().tolist()
    for b in range(batch):
        lg_o = T.oracle_forward(m, [toks[b]], pos[b], caches[b], window, sink).view(np.float16).astype(np.float64)
        lg_g = GD.graph_forward(G, gs[b], wns[b], [toks[b]], pos[b], gcaches[b])
        ea = np.abs(lg_d[b].astype(np.float64) - lg_o).max(); eg = np.abs(lg_g.astype(np.float64) - lg_o).max()
        same = np.array_equal(lg_d[b].view(np.uint16), lg_g.view(np.uint16))
        to = int(np.argmax(lg_o)); srt = np.sort(lg_o)
        print(f"step {step} seq {b} pos {pos[b]}: steps-vs-oracle {ea:.4f} graph-vs-oracle {eg:.4f} max|ref| {np.abs(lg_o).max():.1f} "
              f"steps==graph {same} tok dev {tok_d[b]} oracle {to} margin {srt[-1]-srt[-2]:.3f}")
        toks[b] = to; pos[b] += 1
    dec.kv.tok.copy_(torch.tensor(toks, dtype=torch.int32))
