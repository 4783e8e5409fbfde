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
