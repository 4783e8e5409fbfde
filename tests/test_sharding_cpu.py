"""World-size-2 (and 3) gloo tests of the multi-GPU partition logic on CPU: the N-split matmul
with all-gather reassembles bit-identically to the unsplit result, kv-head sharded attention
concatenates to the unsplit output, sequence sharding covers every sequence once.  The local
compute stands in for the HIP kernels with the CPU oracle (test infrastructure only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import rand_weights_int8, rand_act_f16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(world, fn):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _nsplit_case(rank, world):
    from oracle import binding as O
    from cactus_amd import sharding as S
    rng = np.random.default_rng(5)            # same inputs on every rank
    M, K, N = 3, 256, 200                     # N not a multiple of 16 * world: ragged last slice
    x = rand_act_f16(rng, M, K)
    B, Bs = rand_weights_int8(rng, N, K)
    q, s = O.quantize_activations(x)
    full = O.matmul_int8(q, s, B, Bs, 128)

    def local(b, bs):
        if b.shape[0] == 0:
            return torch.zeros((M, 0), dtype=torch.float16)
        return torch.from_numpy(O.matmul_int8(q, s, np.ascontiguousarray(b), np.ascontiguousarray(bs), 128)
                                .view(np.float16))
    got = S.matmul_nsplit(local, B, Bs, N).numpy().view(np.uint16)
    return bool((got == full).all()), S.row_shard(N, world, rank)


@pytest.mark.parametrize("world", [2, 3])
def test_nsplit_matmul_all_gather_is_bit_identical(world):
    res = _run(world, _nsplit_case)
    assert all(ok for ok, _ in res)
    # slices tile [0, N) in rank order
    edges = [r for _, r in res]
    assert edges[0][0] == 0 and edges[-1][1] == 200
    assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def _kv_head_case(rank, world):
    from oracle import binding as O
    from cactus_amd import sharding as S
    rng = np.random.default_rng(9)
    Hq, Hkv, D, cache = 8, 4, 128, 70
    qd = rng.uniform(-1, 1, (1, 1, Hq, D)).astype(np.float16)
    kq, ks = O.quantize_kv(rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16))
    vq, vs = O.quantize_kv(rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16))
    kn = rng.uniform(-2, 2, (1, 1, Hkv, D)).astype(np.float16)
    vn = rng.uniform(-2, 2, (1, 1, Hkv, D)).astype(np.float16)
    full = O.attention_hybrid(qd, kq[None], vq[None], ks[None], vs[None], kn, vn, 0.0, position_offset=cache)
    k0, k1, q0, q1 = S.kv_head_shard(Hkv, Hq, world, rank)
    c = np.ascontiguousarray
    part = O.attention_hybrid(c(qd[:, :, q0:q1]), c(kq[None, :, k0:k1]), c(vq[None, :, k0:k1]),
                              c(ks[None, :, k0:k1]), c(vs[None, :, k0:k1]), c(kn[:, :, k0:k1]),
                              c(vn[:, :, k0:k1]), 0.0, position_offset=cache)
    local = torch.from_numpy(part.reshape(1, -1).view(np.float16))
    got = S.all_gather_columns(local, Hq * D, multiple=(q1 - q0) * D).numpy().view(np.uint16)
    return bool((got.reshape(full.shape) == full).all())


def test_kv_head_sharded_attention_concatenates():
    assert all(_run(2, _kv_head_case))


def test_sequence_shard_partitions():
    from cactus_amd import sharding as S
    for world in (1, 2, 8):
        seen = sorted(i for r in range(world) for i in S.sequence_shard(37, world, r))
        assert seen == list(range(37))


def test_row_shard_alignment():
    from cactus_amd import sharding as S
    for N, world in ((12288, 8), (151936, 8), (1000, 3), (8, 4)):
        edges = [S.row_shard(N, world, r) for r in range(world)]
        assert edges[0][0] == 0 and max(e[1] for e in edges) == N
        for (a, b) in edges:
            assert a <= b and (a == b or a % 16 == 0)


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it must become the launcher: two ranks start (each
    announces itself before touching a GPU) and, with no GPU in this container, fail -- which the parent
    reports with a non-zero exit code instead of measuring one GPU and calling it two."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-extra", "--no-cpu"], capture_output=True, text=True, timeout=300, env=env)
    assert "[bench] launching" in r.stderr and "--nproc-per-node=2" in r.stderr
    # (the launcher ends the other rank as soon as one has failed, which may be before that one has announced itself)
    assert "rank 0/2" in r.stderr or "rank 1/2" in r.stderr, r.stderr[-2000:]
    if not torch.cuda.is_available():
        assert r.returncode != 0
