"""Free-running greedy identity table (tests/free_running.py): the device decodes on its own tokens and state, the CPU
oracle shadows its token stream.  Prints one JSON object; `python scripts/greedy_identity.py [--full] [--out FILE]`."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cactus_amd.hip as H                                      # noqa: E402
import cactus_amd.graph as G                                    # noqa: E402
from cactus_amd import steps as ST                               # noqa: E402
from tests import free_running as FR                             # noqa: E402
from tests import test_gpu_greedy_decode as GD                   # noqa: E402
from tests import test_gpu_steps as SD                           # noqa: E402

H.init(0)
full = "--full" in sys.argv
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
cases = [("cut_down_4_blocks_vocab_8192_tied", 4, 8192, (3, 8, 11), True),
         ("cut_down_4_blocks_vocab_8192_own_head", 4, 8192, (3, 8, 11), False)]
if full:
    cases.append(("qwen3_0p6b_shape_28_blocks_vocab_151936_tied", 28, 151936, (5, 6, 7), True))
    cases.append(("qwen3_0p6b_shape_28_blocks_vocab_151936_own_head", 28, 151936, (5, 6), False))
table = {}
for name, layers, vocab, seeds, tied in cases:
    rows = {}
    for seed in seeds:
        t0 = time.time()
        rng = np.random.default_rng(seed)
        m = FR.make_model(GD, rng, layers, vocab, tied)
        prompt = rng.integers(0, vocab, 9).tolist()
        sh = FR.run_graph_path(G, GD, m, layers, vocab, prompt, 64, seed)
        rows[f"graph_mirror_seed{seed}"] = sh.summary()
        prompts = [rng.integers(0, vocab, n).tolist() for n in (5, 9, 3)]
        shs = FR.run_batch_path(ST, SD, m, layers, vocab, prompts, 64, seed)
        for b, s in enumerate(shs):
            rows[f"qwen3decode_batch3_seed{seed}_seq{b}"] = s.summary()
        if not full or seed == seeds[0]:
            prompts = [rng.integers(0, vocab, n).tolist() for n in (4, 7, 2, 6, 3, 5)]
            shs = FR.run_batch_path(ST, SD, m, layers, vocab, prompts, 64, seed + 50)
            for b, s in enumerate(shs):
                rows[f"qwen3decode_batch6_seed{seed}_seq{b}"] = s.summary()
        print(name, seed, "%.0f s" % (time.time() - t0), file=sys.stderr, flush=True)
    table[name] = rows
js = json.dumps(table, indent=1)
print(js)
if out_path:
    open(out_path, "w").write(js)
