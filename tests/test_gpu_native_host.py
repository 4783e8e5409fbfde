"""A C++ host program on the C ABI (examples/qwen_graph_decode.cpp, built to cactus_amd/lib/qwen_graph_decode): it
loads a model folder in the reference's layout (config.txt + CACT tensor files, models/model_qwen.cpp:17-44), runs
prefill + greedy decode node by node through the graph surface and prints its tokens.  They must equal the
oracle's greedy tokens on the same weights (the cut-down Qwen3 model of test_gpu_greedy_decode, whose top two
logits are far apart at every step of this prompt) and those of the Python binding driving the same library."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as O
from tests import test_gpu_greedy_decode as GD

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cactus_amd", "lib", "qwen_graph_decode")


def run_host(folder, steps, fusion, prompt):
    r = subprocess.run([EXE, str(folder), str(steps), str(int(fusion))] + [str(t) for t in prompt], capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("fusion", [True, False])
def test_cpp_host_decodes_the_oracles_tokens(tmp_path, fusion):
    from cactus_amd import cact
    assert os.path.exists(EXE), "cactus_amd/lib/qwen_graph_decode is not built (python -c 'import __graft_entry__ as g; g.build()')"
    LAYERS, VOCAB, steps = 4, 8192, 12
    rng = np.random.default_rng(3)
    m = GD.make_model(rng, LAYERS, VOCAB, 0.05)
    prompt = rng.integers(0, VOCAB, 11).tolist()
    cfg = dict(hidden=GD.HID, layers=LAYERS, q_heads=GD.HQ, kv_heads=GD.HKV, head_dim=GD.D, ffn=GD.FF, eps=GD.EPS, theta=GD.THETA)
    cact.write_qwen_folder(tmp_path, m, cfg)
    got = run_host(tmp_path, steps, fusion, prompt)
    assert len(got["tokens"]) == steps + 1 and got["timed_steps"] == steps - 2
    if fusion:
        assert got["fused_launches"] > 0
    # the oracle's greedy run; a step whose two best logits are closer than 1 % of the largest is not compared
    # (decode attention rounds about one output in thirty the other way) and ends the comparison
    caches = [None] * LAYERS
    feed, pos, compared = prompt, 0, 0
    for step in range(steps + 1):
        ref = GD.oracle_forward(m, feed, pos, caches).view(np.float16).astype(np.float64)
        srt = np.sort(ref)
        if srt[-1] - srt[-2] < 1e-2 * np.abs(ref).max():
            break
        assert got["tokens"][step] == int(np.argmax(ref)), (step, got["tokens"])
        compared += 1
        pos += len(feed)
        feed = [got["tokens"][step]]
    assert compared >= 6


def test_cpp_host_and_python_binding_agree(tmp_path):
    """The same folder through cactus_amd.steps.Qwen3GraphModel (weights bound from host arrays instead of files):
    identical tokens, step for step."""
    import cactus_amd.hip as H
    H.init(0)
    from cactus_amd import cact, steps as ST
    LAYERS, VOCAB, steps = 4, 8192, 10
    rng = np.random.default_rng(8)
    m = GD.make_model(rng, LAYERS, VOCAB, 0.05)
    prompt = rng.integers(0, VOCAB, 5).tolist()
    cfg = dict(hidden=GD.HID, layers=LAYERS, q_heads=GD.HQ, kv_heads=GD.HKV, head_dim=GD.D, ffn=GD.FF, eps=GD.EPS,
               theta=GD.THETA, vocab=VOCAB)
    cact.write_qwen_folder(tmp_path, m, cfg)
    got = run_host(tmp_path, steps, True, prompt)["tokens"]
    model = ST.Qwen3GraphModel(m, cfg, max_seq=256)
    want = [model.step(prompt)]
    for _ in range(steps):
        want.append(model.step([want[-1]]))
    model.close()
    assert got == want


BATCH_EXE = os.path.join(os.path.dirname(EXE), "qwen_batched_decode")


@pytest.mark.parametrize("batch", [3, 6])
def test_cpp_host_batched_decode_equals_the_python_driver(tmp_path, batch):
    """examples/qwen_batched_decode.cpp: a C++ host that loads the folder itself, describes the model to the library's
    step driver (cactus_hip_decode_step_dev) and decodes `batch` sequences from position 0.  The same weights through
    cactus_amd.steps.Qwen3Decode (the Python view of the same driver): identical tokens, sequence by sequence; and
    sequence 0's first choices against the oracle (3 sequences: streaming launches; 6: the tile kernel)."""
    import cactus_amd.hip as H
    H.init(0)
    from cactus_amd import cact, steps as ST
    from tests import test_gpu_steps as SD
    assert os.path.exists(BATCH_EXE), "cactus_amd/lib/qwen_batched_decode is not built"
    LAYERS, VOCAB, steps = 4, 8192, 10
    rng = np.random.default_rng(40 + batch)
    m = GD.make_model(rng, LAYERS, VOCAB, 0.05)
    first = rng.integers(0, VOCAB, batch).tolist()
    cfg = dict(ST.QWEN3_0P6B, layers=LAYERS, vocab=VOCAB)
    cact.write_qwen_folder(tmp_path, m, cfg)
    r = subprocess.run([BATCH_EXE, str(tmp_path), str(batch), str(steps)] + [str(t) for t in first], capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])["tokens"]
    dec = ST.Qwen3Decode(ST.weights_to_device(m), cfg, batch, steps + 4, max_position=steps + 4)
    dec.kv.set_positions([0] * batch)
    dec.kv.tok.copy_(torch.tensor(first, dtype=torch.int32))
    want = [[] for _ in range(batch)]
    for _ in range(steps):
        dec.step()
        for b, t in enumerate(dec.kv.tok.cpu().tolist()):
            want[b].append(t)
    assert got == want
    caches, tok, pos = [None] * LAYERS, first[0], 0
    for s in range(4):          # the oracle on sequence 0 (tied head: wide margins)
        ref = SD.oracle_forward(m, [tok], pos, caches).view(np.float16)
        tok = int(O.sample_greedy(ref[None])[0])
        assert got[0][s] == tok, (s, got[0])
        pos += 1


PROMPT_EXE = os.path.join(os.path.dirname(EXE), "qwen_prompt_decode")


@pytest.mark.parametrize("chunk,lanes", [(16, 3), (64, 1)])
def test_cpp_host_prompt_then_decode(tmp_path, chunk, lanes):
    """examples/qwen_prompt_decode.cpp: a C++ host that prefills a 40-token prompt with cactus_hip_prefill_dev (chunks of 16,
    three in flight, the last one short; or one chunk) and decodes on the same cache buffers with
    cactus_hip_decode_step_dev.  The same weights through cactus_amd.steps (Prefill, adopt, Qwen3Decode): identical tokens;
    and the first choices against the oracle run on the whole prompt."""
    import cactus_amd.hip as H
    H.init(0)
    from cactus_amd import cact, steps as ST
    from tests import test_gpu_steps as SD
    assert os.path.exists(PROMPT_EXE), "cactus_amd/lib/qwen_prompt_decode is not built"
    LAYERS, VOCAB, steps, n = 4, 8192, 6, 40
    rng = np.random.default_rng(60 + chunk)
    m = GD.make_model(rng, LAYERS, VOCAB, 0.05)
    prompt = rng.integers(0, VOCAB, n).tolist()
    cfg = dict(ST.QWEN3_0P6B, layers=LAYERS, vocab=VOCAB)
    cact.write_qwen_folder(tmp_path, m, cfg)
    r = subprocess.run([PROMPT_EXE, str(tmp_path), str(steps), str(chunk), str(lanes)] + [str(t) for t in prompt],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])["tokens"]
    assert len(got) == steps + 1
    dm = ST.weights_to_device(m)
    pf = ST.Prefill(dm, cfg, n, chunk, capacity=n + steps + 1)
    pf.prompt.copy_(torch.tensor(prompt, dtype=torch.int32))
    pf.run(lanes)
    dec = ST.Qwen3Decode(dm, cfg, 1, n + steps + 1, max_position=n + steps + 1)
    dec.adopt(pf)
    want = [int(pf.tok.cpu()[0])]
    for _ in range(steps):
        dec.step()
        want.append(int(dec.kv.tok.cpu()[0]))
    assert got == want
    caches = [None] * LAYERS
    ref = SD.oracle_forward(m, prompt, 0, caches).view(np.float16)
    tok, pos = int(O.sample_greedy(ref[None])[0]), n
    assert got[0] == tok
    for s in range(2):
        ref = SD.oracle_forward(m, [tok], pos, caches).view(np.float16)
        tok = int(O.sample_greedy(ref[None])[0])
        assert got[1 + s] == tok, (s, got)
        pos += 1


HOSTPTR_EXE = os.path.join(os.path.dirname(EXE), "host_pointer_calls")


def test_cpp_host_pointer_family_links_and_is_deterministic():
    """examples/host_pointer_calls.cpp: every host-pointer entry (the reference's argument lists, kernel.h:20-24, 41-104)
    called from a plain C++ host on host buffers, the weight matmuls repeatedly over resident, rewritten and released
    weights.  Two runs give the same checksum of all outputs (what each entry computes is compared with the oracle in
    tests/test_gpu_parity.py; this is the C++ linkage and the residency bookkeeping)."""
    assert os.path.exists(HOSTPTR_EXE), "cactus_amd/lib/host_pointer_calls is not built"
    outs = []
    for _ in range(2):
        r = subprocess.run([HOSTPTR_EXE, "3"], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["host_pointer_calls"] == "ok"
    assert outs[0]["checksum"] == outs[1]["checksum"]
