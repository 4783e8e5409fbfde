"""More than one rank with the HIP kernels as the local compute, on the one GPU a test box has: world sizes 2 and 3,
every rank a process of its own on cuda:0 running the library's kernels on its slice (N-split INT8 / INT4 matmul, kv-head
sharded decode attention over an INT8 cache, sequence-sharded batched attention), the slices exchanged through
torch.distributed's gloo group from host copies -- RCCL does not put two ranks on one device, so the device-side
exchange itself stays with tests/test_gpu_sharding.py (one rank) and the driver's multi-GPU runs.  The assembled result
is compared with the CPU oracle's unsplit one: bit for bit for the matmuls, to the decode attention's tolerance
(2e-3 + 4e-3 |ref|) and bit for bit against the unsplit HIP launch for the attention."""
import os

import numpy as np
import pytest

from tests.helpers import rand_weights_int8, rand_weights_int4, rand_act_f16
from tests.test_sharding_cpu import _free_port

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _worker(rank, world, port, case, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cactus_amd.hip as H
        H.init(0)
        ret[rank] = case(rank, world)
    finally:
        dist.destroy_process_group()


def _run(world, case):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _bits(t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


def _nsplit_case(rank, world):
    import cactus_amd.hip as H
    from cactus_amd import sharding as S
    from oracle import binding as O
    ok = []
    for seed, (M, K, N, int4) in enumerate([(1, 1024, 4096 + 40, False), (1, 2048, 1000, True), (40, 512, 1000, False),
                                            (300, 1024, 2048 + 16, True)]):
        rng = np.random.default_rng(20 + seed)            # the same inputs on every rank
        x = rand_act_f16(rng, M, K)
        B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
        oq, os_ = O.quantize_activations(x)
        full = (O.matmul_int4 if int4 else O.matmul_int8)(oq, os_, B, Bs.view(np.uint16), 128)
        q, s = H.quantize_activations(torch.from_numpy(x).cuda())
        Bd, Bsd = torch.from_numpy(B).cuda(), torch.from_numpy(Bs).cuda()

        def local(b, bs):
            if b.shape[0] == 0:
                return torch.zeros((M, 0), dtype=torch.float16)
            fn = H.matmul_int4 if int4 else H.matmul_int8
            return fn(q, s, b.contiguous(), bs.contiguous()).cpu()
        got = S.matmul_nsplit(local, Bd, Bsd, N).view(torch.int16).numpy().view(np.uint16)
        ok.append(bool((got == full).all()))
    return ok


@pytest.mark.parametrize("world", [2, 3])
def test_nsplit_matmul_hip_kernels_on_every_rank(world):
    res = _run(world, _nsplit_case)
    assert all(all(r) for r in res), res


def _kv_head_case(rank, world):
    import cactus_amd.hip as H
    from cactus_amd import sharding as S
    from oracle import binding as O
    rng = np.random.default_rng(31)
    Hq, Hkv, D, cache = 12, 6, 128, 300
    qd = rng.uniform(-1, 1, (1, 1, Hq, D)).astype(np.float16)
    kq, ks = O.quantize_kv(rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16))
    vq, vs = O.quantize_kv(rng.uniform(-2, 2, (cache, Hkv, D)).astype(np.float16))
    kn = rng.uniform(-2, 2, (1, 1, Hkv, D)).astype(np.float16)
    vn = rng.uniform(-2, 2, (1, 1, Hkv, D)).astype(np.float16)
    full = O.attention_hybrid(qd, kq[None], vq[None], ks[None], vs[None], kn, vn, 0.0, position_offset=cache)
    full = full.view(np.float16).astype(np.float64).reshape(1, Hq * D)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    k0, k1, q0, q1 = S.kv_head_shard(Hkv, Hq, world, rank)
    part = H.attention_hybrid(dev(qd[:, :, q0:q1]), dev(kq[None, :, k0:k1]), dev(vq[None, :, k0:k1]),
                              dev(ks[None, :, k0:k1]), dev(vs[None, :, k0:k1]), dev(kn[:, :, k0:k1]),
                              dev(vn[:, :, k0:k1]), 0.0, position_offset=cache)
    whole = H.attention_hybrid(dev(qd), dev(kq[None]), dev(vq[None]), dev(ks[None]), dev(vs[None]), dev(kn), dev(vn), 0.0,
                               position_offset=cache)
    got = S.all_gather_columns(part.reshape(1, -1).cpu(), Hq * D, multiple=(q1 - q0) * D)
    same_as_unsplit = bool(torch.equal(got.view(torch.int16), whole.reshape(1, -1).cpu().view(torch.int16)))
    err = np.abs(got.numpy().astype(np.float64) - full)
    return same_as_unsplit, bool((err <= 2e-3 + 4e-3 * np.abs(full)).all())


@pytest.mark.parametrize("world", [2, 3])
def test_kv_head_sharded_decode_attention_hip_kernels(world):
    res = _run(world, _kv_head_case)
    assert all(a and b for a, b in res), res


def _sequence_case(rank, world):
    """9 sequences of different cache lengths decoded by the ranks that own them (i % world == rank), each rank one batched
    launch over its own sequences' caches; rows gathered back into sequence order."""
    import torch.distributed as dist
    import cactus_amd.hip as H
    from cactus_amd import sharding as S
    from oracle import binding as O
    rng = np.random.default_rng(44)
    nseq, Hq, Hkv, D, slots = 9, 16, 8, 128, 96
    lens = rng.integers(5, slots, nseq)
    qd = rng.uniform(-1, 1, (nseq, 1, Hq, D)).astype(np.float16)
    kn = rng.uniform(-2, 2, (nseq, 1, Hkv, D)).astype(np.float16)
    vn = rng.uniform(-2, 2, (nseq, 1, Hkv, D)).astype(np.float16)
    kq, ks = O.quantize_kv(rng.uniform(-2, 2, (nseq * slots, Hkv, D)).astype(np.float16))
    vq, vs = O.quantize_kv(rng.uniform(-2, 2, (nseq * slots, Hkv, D)).astype(np.float16))
    kq, vq = kq.reshape(nseq, slots, Hkv, D), vq.reshape(nseq, slots, Hkv, D)
    ks, vs = ks.reshape(nseq, slots, Hkv, 1), vs.reshape(nseq, slots, Hkv, 1)
    mine = S.sequence_shard(nseq, world, rank)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = H.attention_hybrid_batched(dev(qd[mine]), dev(kq[mine]), dev(vq[mine]), dev(ks[mine]), dev(vs[mine]), dev(kn[mine]),
                                     dev(vn[mine]), 0.0, dev(lens[mine].astype(np.int32)), dev(lens[mine].astype(np.int32)))
    rows = [None] * world
    dist.all_gather_object(rows, (mine, out.cpu().reshape(len(mine), Hq * D)))
    got = np.zeros((nseq, Hq * D))
    seen = []
    for idx, t in rows:
        got[idx] = t.numpy().astype(np.float64)
        seen += idx
    ok = sorted(seen) == list(range(nseq))
    for i in range(nseq):
        n = int(lens[i])
        ref = O.attention_hybrid(qd[i:i + 1], kq[i:i + 1, :n], vq[i:i + 1, :n], ks[i:i + 1, :n], vs[i:i + 1, :n], kn[i:i + 1],
                                 vn[i:i + 1], 0.0, position_offset=n).view(np.float16).astype(np.float64).reshape(-1)
        ok = ok and bool((np.abs(got[i] - ref) <= 2e-3 + 4e-3 * np.abs(ref)).all())
    return ok


@pytest.mark.parametrize("world", [2, 3])
def test_sequence_sharded_batched_attention_hip_kernels(world):
    assert all(_run(world, _sequence_case))
