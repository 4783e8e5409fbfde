"""The multi-GPU partitions with the HIP kernels as the local compute and RCCL (torch.distributed backend
"nccl") as the exchange, on a one-rank group -- what a one-GPU box can run: the N-split matmul's slicing,
kernel launch on a weight slice, all_gather_into_tensor on device tensors and reassembly, and bench.py's
config-5 leg end to end.  (More ranks: tests/test_sharding_cpu.py runs the same partition logic on gloo
with 2 and 3 ranks.)"""
import os
import socket
import sys

import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import rand_weights_int8, rand_weights_int4, rand_act_f16

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def group():
    import torch.distributed as dist
    import cactus_amd.hip as H
    H.init(0)
    torch.cuda.set_device(0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield H
    dist.destroy_process_group()


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N", [(1, 1024, 200), (3, 2048, 4096), (40, 512, 1000)])
def test_nsplit_matmul_hip_kernel_rccl_gather(group, M, K, N, int4):
    H = group
    from cactus_amd import sharding as S
    rng = np.random.default_rng(M + K + N + int4)
    B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
    q, s = O.quantize_activations(rand_act_f16(rng, M, K))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dq, ds, dB, dBs = dev(q), dev(s), dev(B), dev(Bs)
    fn = H.matmul_int4 if int4 else H.matmul_int8
    got = S.matmul_nsplit(lambda b, bs: fn(dq, ds, b, bs, 128), dB, dBs, N)
    want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
    assert np.array_equal(got.cpu().view(torch.int16).numpy().view(np.uint16), want)


@pytest.mark.parametrize("int4", [False, True])
@pytest.mark.parametrize("M,K,N", [(1, 1024, 200), (1, 2048, 4096), (3, 2048, 4096), (40, 512, 1000), (512, 1024, 2048)])
def test_nsplit_matmul_through_the_c_entry(group, M, K, N, int4):
    """cactus_hip_matmul_int{8,4}_nsplit_dev over the library's own RCCL communicator (cactus_hip_comm_init): the rank's
    weight rows in, the full [M, N] out, against the oracle bit for bit; both exchange forms for M == 1."""
    H = group
    from cactus_amd import sharding as S
    comm = S.device_comm()
    assert comm.world == 1 and comm.rank == 0
    rng = np.random.default_rng(M + K + N + int4)
    B, Bs = (rand_weights_int4 if int4 else rand_weights_int8)(rng, N, K)
    q, s = O.quantize_activations(rand_act_f16(rng, M, K))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    start, end = S.row_shard(N, comm.world, comm.rank)
    want = (O.matmul_int4 if int4 else O.matmul_int8)(q, s, B, Bs, 128)
    for algo in ((0, 1, 2) if M == 1 else (0,)):
        got = comm.matmul_nsplit(dev(q), dev(s), dev(B[start:end]), dev(Bs[start:end]), N, algo=algo)
        assert np.array_equal(got.cpu().view(torch.int16).numpy().view(np.uint16), want), algo
    comm.close()


def test_allgather_columns_and_the_slot_copy(group):
    """cactus_hip_allgather_columns_dev on a one-rank communicator leaves C as it is; the copy step of the M > 1
    exchange (cactus_hip_slots_to_columns_dev) against numpy for 2, 3 and 8 ranks' worth of slots, ragged N included."""
    H = group
    from cactus_amd import sharding as S
    comm = S.device_comm()
    rng = np.random.default_rng(9)
    for M, N in ((1, 4096), (5, 1000)):
        c = torch.from_numpy(rng.integers(0, 65535, (M, N), dtype=np.uint16).view(np.float16)).cuda()
        keep = c.clone()
        comm.allgather_columns(c, multiple=16)
        assert torch.equal(c.view(torch.int16), keep.view(torch.int16))
    comm.close()
    for world, M, N, mult in ((2, 3, 4096, 16), (3, 7, 1000, 16), (8, 512, 12288, 16), (8, 4, 520, 16), (4, 2, 24, 2),
                              (8, 1, 4096, 512), (3, 5, 1001, 1)):
        per = S.row_shard(N, world, 0, mult)[1]
        slots = rng.integers(0, 65535, (world, M, per), dtype=np.uint16)
        want = np.concatenate([slots[r] for r in range(world)], axis=1)[:, :N]
        got = H.slots_to_columns(torch.from_numpy(slots.view(np.float16)).cuda(), N, mult)
        assert np.array_equal(got.cpu().view(torch.int16).numpy().view(np.uint16), want), (world, M, N)


def test_bench_config5_leg_runs_and_checks_itself(group):
    """bench.py's `extra.nsplit_8b` (8B-shape INT4 matmuls N-split + 32k-context attention sharded by kv head),
    here with a shorter cache: every N-split result bit-identical to the unsplit kernel, the sharded
    attention within tolerance."""
    H = group
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    out = bench.nsplit_8b(H, 1, 0, cache=4096)
    assert out["all_checks_passed"], out
    assert len(out["matmul_int4"]) == 6 and all(r["bit_identical_to_unsplit"] for r in out["matmul_int4"].values())
    assert out["attn_decode_int8kv_c32768_kv_head_shard"]["max_abs_diff_vs_unsplit"] <= 2e-3
