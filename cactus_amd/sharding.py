"""Multi-GPU partitioning of the hot path: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

Two partitions, neither of which exists in the reference (its engine is single-sequence, CPU):

* independent sequences -> ranks (`sequence_shard`): every rank holds a full weight replica
  and only its own sequences' KV cache; no collective on the data path.
* N-split of one large matmul (`row_shard`, `all_gather_columns`, `matmul_nsplit`): rank r owns
  rows [start, end) of the pretransposed weight B[N, K] and of its scales [N, K/gs] -- a
  contiguous slice, no repacking -- and produces C[:, start:end].  One exchange step, an
  all-gather of M x (N / world) halves per rank.  The INT32 group sums and the fp32 order of
  every output element are untouched by an N-split, so the gathered result is bit-identical to
  the single-GPU result; a K-split would need an fp32 reduction in a different order and is not
  offered.
* attention over a long INT8 KV cache shards by kv head (`kv_head_shard`): a rank keeps the
  cache of its kv heads and computes their query heads; outputs concatenate along heads with
  the same all-gather (heads * head_dim is the N dimension of the q/k/v projections).
"""
import torch
import torch.distributed as dist


def sequence_shard(num_sequences, world, rank):
    """Indices of the sequences rank `rank` decodes: i with i % world == rank."""
    return list(range(rank, num_sequences, world))


def row_shard(N, world, rank, multiple=16):
    """Rows [start, end) of an N-row weight owned by `rank`.  Slices are multiples of
    `multiple` rows (the GEMV kernel's rows per workgroup) except the last, which takes the
    remainder; ranks past the end get an empty slice.  (cactus_hip_row_shard: the library's own rule.)"""
    from . import hip as H
    return H.row_shard(N, world, rank, multiple)


def device_comm(group=None):
    """The library's RCCL communicator over the ranks of `group` (one process per GPU, torch.cuda's current device):
    rank 0 draws the unique id, torch.distributed carries its 128 bytes to the others, every rank joins."""
    from . import hip as H
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [H.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return H.Comm(box[0], world, rank)


def kv_head_shard(num_kv_heads, num_q_heads, world, rank):
    """(kv_start, kv_end, q_start, q_end): kv heads are split evenly over ranks, each with the
    query heads that read it (q head h reads kv head h // (Hq / Hkv))."""
    if num_kv_heads % world != 0:
        raise ValueError("kv heads must divide evenly over ranks; split the sequence dimension instead")
    per = num_kv_heads // world
    gq = num_q_heads // num_kv_heads
    return rank * per, (rank + 1) * per, rank * per * gq, (rank + 1) * per * gq


def all_gather_columns(local, N, group=None, multiple=16):
    """Gather C[:, start:end] slices (row_shard layout) from every rank into the full [M, N].
    `local` is this rank's [M, end - start] tensor; every rank returns the same [M, N].  The torch.distributed
    form, for host tensors (the gloo tests of the partition arithmetic); on the GPUs the exchange is the library's
    own (device_comm(): Comm.matmul_nsplit / Comm.allgather_columns, no padding, no permute)."""
    world = dist.get_world_size(group)
    M = local.shape[0]
    per = row_shard(N, world, 0, multiple)[1]
    pad = torch.zeros((M, per), dtype=local.dtype, device=local.device)
    pad[:, :local.shape[1]] = local
    out = torch.empty((world * M, per), dtype=local.dtype, device=local.device)   # rank-major rows
    dist.all_gather_into_tensor(out, pad, group=group)
    return out.view(world, M, per).permute(1, 0, 2).reshape(M, world * per)[:, :N].contiguous()


def matmul_nsplit(matmul_local, B, B_scales, N, group=None, multiple=16):
    """N-split matmul.  `B`, `B_scales` are the FULL weight and scales (or anything sliceable by
    rows); `matmul_local(B_slice, Bs_slice)` computes this rank's [M, rows] fp16 block, e.g.
    lambda b, s: cactus_amd.hip.matmul_int8(q, q_scales, b, s, 128).  Returns the gathered [M, N]."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    start, end = row_shard(N, world, rank, multiple)
    local = matmul_local(B[start:end], B_scales[start:end])
    return all_gather_columns(local, N, group, multiple)
