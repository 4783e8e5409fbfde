#!/usr/bin/env python3
"""Prefill through the graph surface from the C++ host program: Qwen3-0.6B shape, prompt as one chunk."""
import os, sys, json, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cactus_amd.hip as H
from cactus_amd import steps as ST, cact
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
cfg = ST.QWEN3_0P6B
w = ST.qwen3_synthetic_weights(gen, cfg)
host = {"layers": [{k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v.cpu().numpy()) for k, v in L.items()} for L in w["layers"]],
        "emb": tuple(t.cpu().numpy() for t in w["emb"]), "final_norm": w["final_norm"].cpu().numpy()}
del w
torch.cuda.empty_cache()
folder = tempfile.mkdtemp(prefix="cactus_qwen3_")
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cactus_amd", "lib", "qwen_graph_decode")
try:
    cact.write_qwen_folder(folder, host, cfg)
    for n in [int(a) for a in sys.argv[1:]] or [512, 4096]:
        prompt = [str(t) for t in np.random.default_rng(3).integers(0, cfg["vocab"], n).tolist()]
        r = subprocess.run([exe, folder, "6", "1"] + prompt, capture_output=True, text=True, timeout=300,
                           env=dict(os.environ))      # CACTUS_PREFILL_CHUNK=256: the prompt in chunks, a graph each
        print(n, r.stderr[-300:] if r.returncode else {k: v for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items() if k != "tokens"}, flush=True)
finally:
    shutil.rmtree(folder, ignore_errors=True)
