"""Free-running greedy decoding against the CPU oracle (Model::decode, engine_model.cpp:221-297, argmax branch of
cactus_sample_f16, kernel_nn.cpp:451).

The device decodes on its own: it feeds itself the tokens it chose and nothing of its state (KV caches, positions) is
ever read back, compared or replaced.  The oracle shadows the device's token stream: at every step it is given the
token the device fed itself and produces its own logits from its own state.  Recorded per run: the first step at which
the two greedy choices differ (the identity length), every such step with the oracle's margin between the two
candidates, and the largest logit difference relative to the largest logit.

Used by tests/test_gpu_free_running.py (assertions) and scripts/greedy_identity.py (the table)."""
import numpy as np

from oracle import binding as O


class Shadow:
    """One sequence's record of a device stream followed by the oracle."""

    def __init__(self):
        self.steps = 0
        self.first_diff = None          # first step whose greedy choices differ
        self.diffs = []                 # (step, oracle margin between its choice and the device's, logit error there)
        self.worst_rel = 0.0            # max over steps of max|logit difference| / max|logit|
        self.worst_abs = 0.0
        self.min_margin_identical = np.inf   # smallest oracle top-2 margin among the steps with identical choices
        self.tokens = []

    def step(self, lg_dev, tok_dev, lg_ref_bits):
        ref = lg_ref_bits.view(np.float16).astype(np.float64)
        dev = lg_dev.view(np.float16).astype(np.float64)
        to = int(O.sample_greedy(lg_ref_bits.view(np.float16)[None])[0])
        err = float(np.abs(dev - ref).max())
        top = float(np.abs(ref).max())
        self.worst_abs = max(self.worst_abs, err)
        self.worst_rel = max(self.worst_rel, err / top)
        if tok_dev != to:
            if self.first_diff is None:
                self.first_diff = self.steps
            self.diffs.append((self.steps, float(ref[to] - ref[tok_dev]), err, top))
        else:
            two = np.partition(ref, -2)[-2:]
            self.min_margin_identical = min(self.min_margin_identical, float(two[1] - two[0]))
        self.tokens.append(int(tok_dev))
        self.steps += 1
        return err, top

    def summary(self):
        return {"steps": self.steps,
                "identity_length": self.steps if self.first_diff is None else self.first_diff,
                "different_choices": [{"step": s, "oracle_margin": round(m, 5), "logit_error": round(e, 5),
                                       "largest_logit": round(t, 3)} for s, m, e, t in self.diffs],
                "worst_logit_error_rel": round(self.worst_rel, 5), "worst_logit_error_abs": round(self.worst_abs, 5),
                "smallest_margin_among_identical": round(float(self.min_margin_identical), 5),
                "distinct_tokens": len(set(self.tokens))}


def make_model(GD, rng, layers, vocab, tied):
    """The residual-dominated synthetic Qwen3-shape model of tests/test_gpu_greedy_decode.py.  tied: lm_head = the
    embedding table, as Qwen3 has it - a random model then mostly predicts the token it was fed, with a large margin.
    Not tied: an lm_head of its own - logits are a random projection of the last hidden row, nearly flat, the top two
    a few per cent of the largest apart: the hard case for token identity."""
    m = GD.make_model(rng, layers, vocab, 0.05)
    if not tied:
        from tests.helpers import rand_weights_int8
        m["head"] = rand_weights_int8(rng, vocab, GD.HID)
    return m


def next_feed(history, tok, rng, vocab):
    """A synthetic model soon repeats one token: a token seen before is replaced by a fresh seeded one (the
    comparison of the step that chose it has been made).  Deterministic in the device's own history."""
    history.append(tok)
    return tok if history.count(tok) <= 1 else int(rng.integers(0, vocab))


def run_graph_path(G, GD, m, layers, vocab, prompt, steps, seed):
    """The graph mirror driven as the engine drives it (prefill of the prompt, then one token per step), free-running.
    GD: tests.test_gpu_greedy_decode (model builder, oracle forward, graph forward)."""
    import torch
    import cactus_amd.hip as H
    g = G.CactusGraph()
    wn = GD.bind(G, g, m)
    cache = G.KVCache(layers, 2048, GD.HKV, GD.D)
    ocaches = [None] * layers
    rng = np.random.default_rng(seed + 1000)
    sh, hist = Shadow(), []
    feed, pos = list(prompt), 0
    for _ in range(steps):
        lg_d = GD.graph_forward(G, g, wn, feed, pos, cache)
        td = int(H.sample_greedy(torch.from_numpy(lg_d[None].copy()).cuda()).cpu()[0])
        lg_o = GD.oracle_forward(m, feed, pos, ocaches)
        sh.step(lg_d, td, lg_o)
        pos += len(feed)
        feed = [next_feed(hist, td, rng, vocab)]
    g.close()
    cache.close()
    return sh


def run_batch_path(ST, SD, m, layers, vocab, prompts, steps, seed):
    """cactus_amd.steps.Qwen3Decode, free-running: every sequence eats its prompt one token per step, then its own
    choices; tokens stay on the device between steps except where a repeated token is replaced (next_feed).
    SD: tests.test_gpu_steps (oracle forward).  Returns one Shadow per sequence (decode steps after the prompt)."""
    import torch
    cfg = dict(ST.QWEN3_0P6B, layers=layers, vocab=vocab)
    batch = len(prompts)
    longest = max(len(p) for p in prompts)
    dec = ST.Qwen3Decode(ST.weights_to_device(m), cfg, batch, longest + steps + 2, max_position=longest + steps + 2)
    dec.kv.set_positions([0] * batch)
    caches = [[None] * layers for _ in range(batch)]
    rngs = [np.random.default_rng(seed + 1000 + b) for b in range(batch)]
    shadows, hists = [Shadow() for _ in range(batch)], [[] for _ in range(batch)]
    feed = [p[0] for p in prompts]
    pos = [0] * batch
    for it in range(longest + steps - 1):
        dec.kv.tok.copy_(torch.tensor(feed, dtype=torch.int32))
        dec.step()
        torch.cuda.synchronize()
        lg_d = dec.logits.cpu().numpy()
        tok_d = dec.kv.tok.cpu().tolist()
        for b in range(batch):
            lg_o = SD.oracle_forward(m, [feed[b]], pos[b], caches[b])
            pos[b] += 1
            if pos[b] < len(prompts[b]):                 # still inside the prompt
                feed[b] = prompts[b][pos[b]]
                continue
            if shadows[b].steps < steps:
                shadows[b].step(lg_d[b], tok_d[b], lg_o)
            feed[b] = next_feed(hists[b], tok_d[b], rngs[b], vocab)
    return shadows
