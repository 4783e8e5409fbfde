"""ctypes view of libcactus_hip.so for torch device tensors.

Every function enqueues on torch's current stream and returns without
synchronising (the *_dev family of include/cactus_hip.h).  Halves are
torch.float16 tensors; their bits are what the C ABI calls cactus_half_t.
"""
import ctypes as C
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CACTUS_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libcactus_hip.so")

if not os.path.exists(SO_PATH):
    raise ImportError(
        f"{SO_PATH} is missing: build it with `python -m cactus_amd.build` "
        "(there is no fallback path)")

lib = C.CDLL(SO_PATH)

_sz, _vp, _i, _f = C.c_size_t, C.c_void_p, C.c_int, C.c_float

SIGNATURES = {
    "cactus_hip_init": (_i, [_i]),
    "cactus_hip_device_count": (_i, []),
    "cactus_hip_last_error": (C.c_char_p, []),
    "cactus_hip_clear_error": (None, []),
    "cactus_hip_malloc": (_vp, [_sz]),
    "cactus_hip_free": (None, [_vp]),
    "cactus_hip_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "cactus_hip_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "cactus_hip_synchronize": (_i, [_vp]),
    "cactus_hip_release_weights": (None, [_vp]),
    "cactus_hip_workspace_reserve": (_i, [_sz]),
    "cactus_hip_weights_pack_dev": (_i, [_vp, _sz, _sz, _i, _vp]),
    "cactus_hip_weights_release_packed_dev": (_i, [_vp]),
    "cactus_hip_weights_is_packed_dev": (_i, [_vp]),
    "cactus_hip_comm_available": (_i, []),
    "cactus_hip_comm_get_unique_id": (_i, [_vp]),
    "cactus_hip_comm_init": (_i, [C.POINTER(_vp), _vp, _i, _i]),
    "cactus_hip_comm_adopt": (_i, [C.POINTER(_vp), _vp, _i, _i]),
    "cactus_hip_comm_destroy": (_i, [_vp]),
    "cactus_hip_comm_rank": (_i, [_vp]),
    "cactus_hip_comm_world": (_i, [_vp]),
    "cactus_hip_row_shard": (None, [_sz, _i, _i, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "cactus_hip_allgather_columns_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _i, _vp]),
    "cactus_hip_matmul_int8_nsplit_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _vp]),
    "cactus_hip_matmul_int4_nsplit_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _vp]),
    "cactus_hip_slots_to_columns_dev": (_i, [_vp, _vp, _sz, _sz, _i, _sz, _vp]),
    "cactus_hip_matmul_int8_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int8_f16act_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_f16act_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int8_f16act_batch_dev": (_i, [_vp, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_f16act_batch_dev": (_i, [_vp, _sz, _sz, _vp]),
    "cactus_hip_matmul_int8_shared_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_shared_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int8_group_sums_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_group_sums_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_f16_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _vp]),
    "cactus_hip_quantize_activations_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "cactus_hip_fp16_max_abs_dev": (_i, [_vp, _sz, _vp, _vp]),
    "cactus_hip_fp16_to_int8_dev": (_i, [_vp, _vp, _sz, _f, _vp]),
    "cactus_hip_quantize_kv_fp16_to_int8_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_unpack_int4_to_int8_dev": (_i, [_vp, _vp, _sz, _vp]),
    "cactus_hip_attention_f16_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _f, _vp,
                                          _sz, _sz, _i, _vp]),
    "cactus_hip_attention_hybrid_int8_fp16_dev": (_i, [_vp] * 8 + [_sz] * 7 + [_f, _sz, _i, _sz, _vp]),
    "cactus_hip_attention_hybrid_int8_fp16_quant_dev": (_i, [_vp] * 10 + [_sz] * 7 + [_f, _sz, _i, _sz, _vp]),
    "cactus_hip_attention_hybrid_int8_fp16_noalloc_dev": (_i, [_vp] * 8 + [_sz] * 7 + [_f, _sz, _i, _sz, _vp]),
    "cactus_hip_noalloc_scratch_bytes": (_sz, []),
    "cactus_hip_attention_hybrid_records_bytes": (_sz, [_sz] * 8 + [_i, _sz]),
    "cactus_hip_attention_hybrid_int8_fp16_batched_dev": (
        _i, [_vp] * 8 + [_sz, _sz, _vp, _sz, _sz, _sz, _sz, _f, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_attention_decode_fused_f16_dev": (
        _i, [_vp] * 10 + [_sz, _vp, _vp, _sz, _sz, _sz, _sz, _f, _f, _f, _sz, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_attention_decode_fused_quant_f16_dev": (
        _i, [_vp] * 12 + [_sz, _vp, _vp, _sz, _sz, _sz, _sz, _f, _f, _f, _sz, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_attention_decode_fused_single_f16_dev": (
        _i, [_vp] * 11 + [_sz, _sz, _sz, _sz, _sz, _f, _f, _f, _sz, _vp]),
    "cactus_hip_matmul_int8": (None, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz]),
    "cactus_hip_matmul_int4": (None, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz]),
    "cactus_hip_matmul_f16": (None, [_vp, _vp, _vp, _sz, _sz, _sz]),
    "cactus_hip_attention_f16": (None, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _f, _vp, _sz,
                                        _sz, _i]),
    "cactus_hip_attention_hybrid_int8_fp16": (None, [_vp] * 8 + [_sz] * 7 + [_f, _sz, _i, _sz]),
    "cactus_hip_quantize_kv_fp16_to_int8": (None, [_vp, _vp, _vp, _sz, _sz, _sz, _sz]),
    "cactus_hip_fp16_max_abs": (_f, [_vp, _sz]),
    "cactus_hip_fp16_to_int8": (None, [_vp, _vp, _sz, _f]),
    "cactus_hip_unpack_int4_to_int8": (None, [_vp, _vp, _sz]),
    "cactus_hip_rms_norm_f16_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _f, _vp]),
    "cactus_hip_rope_f16_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _f, _vp]),
    "cactus_hip_rope_reserve": (_i, [_sz, _f, _sz]),
    "cactus_hip_silu_f16_dev": (_i, [_vp, _vp, _sz, _vp]),
    "cactus_hip_gelu_f16_dev": (_i, [_vp, _vp, _sz, _vp]),
    "cactus_hip_add_f16_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cactus_hip_add_f16_clipped_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cactus_hip_subtract_f16_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cactus_hip_multiply_f16_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cactus_hip_divide_f16_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cactus_hip_add_rms_norm_f16_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _f, _vp]),
    "cactus_hip_qk_norm_rope_f16_dev": (_i, [_vp] * 6 + [_sz] * 5 + [_f, _sz, _f, _vp]),
    "cactus_hip_silu_multiply_f16_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cactus_hip_scalar_op_f16_dev": (_i, [_vp, _vp, _sz, _f, _i, _vp]),
    "cactus_hip_conv1d_causal_depthwise_f16_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_lfm2_conv_gate_f16_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_lfm2_conv_gate_quant_f16_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_add_rms_norm_quant_f16_dev": (_i, [_vp] * 7 + [_sz, _sz, _f, _vp]),
    "cactus_hip_silu_multiply_quant_f16_dev": (_i, [_vp] * 5 + [_sz, _sz, _vp]),
    "cactus_hip_gelu_multiply_quant_f16_dev": (_i, [_vp] * 5 + [_sz, _sz, _vp]),
    "cactus_hip_gelu_multiply_f16_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cactus_hip_quantize_kv_pair_fp16_to_int8_dev": (_i, [_vp] * 6 + [_sz] * 4 + [_vp]),
    "cactus_hip_sample_greedy_f16_dev": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "cactus_hip_qk_norm_rope_kv_append_f16_dev": (_i, [_vp] * 12 + [_sz] * 5 + [_f, _sz, _f] + [_sz] * 4 + [_vp]),
    "cactus_hip_matmul_int8_norm_f16act_dev": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_norm_f16act_dev": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int8_silu_f16act_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_silu_f16act_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int8_gelu_f16act_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_matmul_int4_gelu_f16act_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "cactus_hip_sample_f16_dev": (_i, [_vp, _vp, _sz, _sz, _f, _f, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cactus_hip_sample_max_survivors": (_sz, []),
    "cactus_hip_sample_uniform": (_f, [C.c_uint32]),
    "cactus_hip_logits_entropy_f16_dev": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "cactus_hip_qk_norm_rope_seq_f16_dev": (_i, [_vp] * 12 + [_sz] * 6 + [_f, _f] + [_sz] * 6 + [_vp]),
    "cactus_hip_kv_append_windowed_dev": (_i, [_vp] * 7 + [_sz] * 10 + [_vp]),
    "cactus_hip_kv_positions_advance_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _vp]),
    "cactus_hip_kv_append_batched_dev": (_i, [_vp] * 6 + [_sz] * 4 + [_vp, _sz, _sz, _sz, _vp]),
    "cactus_hip_indices_to_int32_dev": (_i, [_vp, _i, _vp, _sz, _vp]),
    "cactus_hip_embedding_int8_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp]),
    "cactus_hip_embedding_f16_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "cactus_hip_rms_norm_f16": (None, [_vp, _vp, _vp, _sz, _sz, _f]),
    "cactus_hip_rope_f16": (None, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _f]),
    "cactus_hip_silu_f16": (None, [_vp, _vp, _sz]),
    "cactus_hip_gelu_f16": (None, [_vp, _vp, _sz]),
    "cactus_hip_add_f16": (None, [_vp, _vp, _vp, _sz]),
    "cactus_hip_add_f16_clipped": (None, [_vp, _vp, _vp, _sz]),
    "cactus_hip_subtract_f16": (None, [_vp, _vp, _vp, _sz]),
    "cactus_hip_multiply_f16": (None, [_vp, _vp, _vp, _sz]),
    "cactus_hip_divide_f16": (None, [_vp, _vp, _vp, _sz]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export the header's symbol
    _fn.restype = _res
    _fn.argtypes = _args


class CactusHipError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise CactusHipError(f"cactus_hip error {rc}: {lib.cactus_hip_last_error().decode()}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.dtype, dtype)
    return t


def init(device=0):
    _check(lib.cactus_hip_init(device))


def pack_weights(B):
    """Give the device-resident weight matrix B (int8 [N, K] or packed uint8 [N, K/2]) a tile-major twin: every
    later matmul with M <= 16 that names B streams the twin (cactus_hip_weights_pack_dev).  B must stay alive and
    unchanged until release_packed(B); the twin is released when the tensor object goes away."""
    kind = {torch.int8: 0, torch.uint8: 1, torch.float16: 2}[B.dtype]     # INT8, packed INT4, FP16
    _dev(B, B.dtype)
    N = B.shape[0]
    K = B.shape[1] * (2 if kind == 1 else 1)
    fresh = not is_packed(B)
    _check(lib.cactus_hip_weights_pack_dev(_p(B), N, K, kind, _stream()))
    if fresh:   # the twin is keyed by B's address: it must not outlive the tensor
        weakref.finalize(B, lib.cactus_hip_weights_release_packed_dev, C.c_void_p(B.data_ptr()))
    return B


def release_packed(B=None):
    """Drop B's twin (all twins when B is None)."""
    _check(lib.cactus_hip_weights_release_packed_dev(_p(B)))


def is_packed(B):
    return bool(lib.cactus_hip_weights_is_packed_dev(_p(B)))


def row_shard(N, world, rank, multiple=16):
    """Rows [start, end) of an N-row weight owned by `rank` (cactus_hip_row_shard; host arithmetic only)."""
    a, b = _sz(0), _sz(0)
    lib.cactus_hip_row_shard(N, world, rank, multiple, C.byref(a), C.byref(b))
    return a.value, b.value


class Comm:
    """An RCCL communicator of the library (cactus_hip_comm_*): one per process and GPU.  unique_id() on rank 0, the
    128 bytes carried to the other ranks by the host program, then Comm(id, world, rank) on every rank."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _check(lib.cactus_hip_comm_get_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, world, rank):
        h = _vp()
        _check(lib.cactus_hip_comm_init(C.byref(h), unique_id, world, rank))
        self._h, self.world, self.rank = h, world, rank

    def close(self):
        if self._h:
            lib.cactus_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def allgather_columns(self, Cfull, multiple=16, algo=0):
        """Cfull fp16 [M, N], this rank's columns (row_shard) written: afterwards all columns, on every rank."""
        _dev(Cfull, torch.float16)
        M, N = Cfull.shape
        _check(lib.cactus_hip_allgather_columns_dev(self._h, _p(Cfull), M, N, multiple, algo, _stream()))
        return Cfull

    def matmul_nsplit(self, A, A_scales, B_rows, B_scales_rows, N, group_size=128, multiple=16, algo=0, out=None):
        """N-split W8A8 / W4A8 matmul: this rank's weight rows in, the full [M, N] out on every rank."""
        int4 = B_rows.dtype == torch.uint8
        _dev(A, torch.int8); _dev(A_scales, torch.float32); _dev(B_scales_rows, torch.float16)
        M, K = A.shape
        Cc = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=A.device)
        fn = lib.cactus_hip_matmul_int4_nsplit_dev if int4 else lib.cactus_hip_matmul_int8_nsplit_dev
        _check(fn(self._h, _p(A), _p(A_scales), _p(B_rows), _p(B_scales_rows), _p(Cc), M, K, N, group_size, multiple,
                  algo, _stream()))
        return Cc


def slots_to_columns(slots, N, multiple=16, out=None):
    """Rank-major blocks fp16 [world, M, per] -> fp16 [M, N] (the copy step of the M > 1 exchange)."""
    _dev(slots, torch.float16)
    world, M, per = slots.shape
    Cc = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=slots.device)
    _check(lib.cactus_hip_slots_to_columns_dev(_p(slots), _p(Cc), M, N, world, multiple, _stream()))
    return Cc


def quantize_activations(x, out_q=None, out_scales=None):
    """fp16 [M,K] -> (int8 [M,K], fp32 scales [M])."""
    _dev(x, torch.float16)
    M, K = x.shape
    q = out_q if out_q is not None else torch.empty((M, K), dtype=torch.int8, device=x.device)
    s = out_scales if out_scales is not None else torch.empty((M,), dtype=torch.float32, device=x.device)
    _check(lib.cactus_hip_quantize_activations_dev(_p(x), _p(q), _p(s), M, K, _stream()))
    return q, s


def matmul_int8(A, A_scales, B, B_scales, group_size=128, out=None):
    _dev(A, torch.int8); _dev(B, torch.int8); _dev(A_scales, torch.float32); _dev(B_scales, torch.float16)
    M, K = A.shape
    N = B.shape[0]
    Cc = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=A.device)
    _check(lib.cactus_hip_matmul_int8_dev(_p(A), _p(A_scales), _p(B), _p(B_scales), _p(Cc), M, K, N,
                                          group_size, _stream()))
    return Cc


def matmul_int4(A, A_scales, B_packed, B_scales, group_size=128, out=None):
    _dev(A, torch.int8); _dev(B_packed, torch.uint8); _dev(A_scales, torch.float32)
    _dev(B_scales, torch.float16)
    M, K = A.shape
    N = B_packed.shape[0]
    Cc = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=A.device)
    _check(lib.cactus_hip_matmul_int4_dev(_p(A), _p(A_scales), _p(B_packed), _p(B_scales), _p(Cc), M, K, N,
                                          group_size, _stream()))
    return Cc


def matmul_f16act(X, B, B_scales, group_size=128, int4=False, out=None):
    """FP16 activations [M, K] x INT8 / packed-INT4 weights: row quantization + grouped matmul
    (one fused launch for M <= 4)."""
    _dev(X, torch.float16); _dev(B, torch.uint8 if int4 else torch.int8); _dev(B_scales, torch.float16)
    M, K = X.shape
    N = B.shape[0]
    Cc = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=X.device)
    fn = lib.cactus_hip_matmul_int4_f16act_dev if int4 else lib.cactus_hip_matmul_int8_f16act_dev
    _check(fn(_p(X), _p(B), _p(B_scales), _p(Cc), M, K, N, group_size, _stream()))
    return Cc


class MatmulProblem(C.Structure):
    _fields_ = [("X", C.c_void_p), ("B", C.c_void_p), ("B_scales", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_size_t), ("K", C.c_size_t), ("N", C.c_size_t)]


class MatmulBatch:
    """A fixed list of FP16-activation matmuls (X, B, B_scales, out) prepared once and enqueued
    with one call per step (cactus_hip_matmul_int{8,4}_f16act_batch_dev)."""

    def __init__(self, entries, group_size=128, int4=False):
        self._keep = list(entries)
        self._arr = (MatmulProblem * len(self._keep))()
        for i, (X, B, Bs, out) in enumerate(self._keep):
            _dev(X, torch.float16); _dev(B, torch.uint8 if int4 else torch.int8)
            _dev(Bs, torch.float16); _dev(out, torch.float16)
            M, K = X.shape
            N = B.shape[0]
            assert tuple(out.shape) == (M, N)
            self._arr[i] = MatmulProblem(X.data_ptr(), B.data_ptr(), Bs.data_ptr(), out.data_ptr(), M, K, N)
        self._fn = lib.cactus_hip_matmul_int4_f16act_batch_dev if int4 else lib.cactus_hip_matmul_int8_f16act_batch_dev
        self._gs = group_size

    def run(self):
        _check(self._fn(C.cast(self._arr, C.c_void_p), len(self._keep), self._gs, _stream()))
        return [e[3] for e in self._keep]


class MatmulWeights(C.Structure):
    _fields_ = [("B", C.c_void_p), ("B_scales", C.c_void_p), ("C", C.c_void_p), ("N", C.c_size_t)]


class MatmulShared:
    """Several weights against one quantized activation matrix: entries = [(B, B_scales, out)];
    run(A, A_scales) enqueues them (one launch when M > 16)."""

    def __init__(self, entries, group_size=128, int4=False):
        self._keep = entries
        self._arr = (MatmulWeights * len(entries))()
        for i, (B, Bs, out) in enumerate(entries):
            _dev(Bs, torch.float16); _dev(out, torch.float16)
            self._arr[i] = MatmulWeights(B.data_ptr(), Bs.data_ptr(), out.data_ptr(), B.shape[0])
        self._fn = lib.cactus_hip_matmul_int4_shared_dev if int4 else lib.cactus_hip_matmul_int8_shared_dev
        self._gs = group_size

    def run(self, A, A_scales):
        _dev(A, torch.int8); _dev(A_scales, torch.float32)
        M, K = A.shape
        _check(self._fn(_p(A), _p(A_scales), self._arr, len(self._keep), M, K, self._gs, _stream()))


class MatmulProduced:
    """Weights multiplied with FP16 rows that the same launch derives from the ops in front of them:
    entries = [(B, B_scales, out)];
      run_norm(x, addend, norm_weight, eps, out_sum)  rows = rms_norm(x [+ addend]) * norm_weight
      run_act(gate, up, act="silu" | "gelu")          rows = act(gate) * up
    One launch for M <= 4 (cactus_hip_matmul_int{8,4}_{norm,silu,gelu}_f16act_dev)."""

    def __init__(self, entries, group_size=128, int4=False):
        self._keep = entries
        self._arr = (MatmulWeights * len(entries))()
        for i, (B, Bs, out) in enumerate(entries):
            _dev(Bs, torch.float16); _dev(out, torch.float16)
            self._arr[i] = MatmulWeights(B.data_ptr(), Bs.data_ptr(), out.data_ptr(), B.shape[0])
        self._tag = "int4" if int4 else "int8"
        self._gs = group_size

    def run_norm(self, x, addend, norm_weight, eps, out_sum=None):
        _dev(x, torch.float16); _dev(norm_weight, torch.float16)
        M, K = x.shape
        fn = getattr(lib, "cactus_hip_matmul_%s_norm_f16act_dev" % self._tag)
        _check(fn(_p(x), _p(addend), _p(norm_weight), _p(out_sum), float(eps), self._arr, len(self._keep), M, K,
                  self._gs, _stream()))

    def run_act(self, gate, up, act="silu"):
        _dev(gate, torch.float16); _dev(up, torch.float16)
        M, K = gate.shape
        fn = getattr(lib, "cactus_hip_matmul_%s_%s_f16act_dev" % (self._tag, act))
        _check(fn(_p(gate), _p(up), self._arr, len(self._keep), M, K, self._gs, _stream()))


def matmul_group_sums(A, B, group_size=128, int4=False):
    M, K = A.shape
    N = B.shape[0]
    out = torch.empty((M, N, K // group_size), dtype=torch.int32, device=A.device)
    fn = lib.cactus_hip_matmul_int4_group_sums_dev if int4 else lib.cactus_hip_matmul_int8_group_sums_dev
    _check(fn(_p(A), _p(B), _p(out), M, K, N, group_size, _stream()))
    return out


def matmul_f16(a, bT, out=None):
    _dev(a, torch.float16); _dev(bT, torch.float16)
    M, K = a.shape
    N = bT.shape[0]
    c = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=a.device)
    _check(lib.cactus_hip_matmul_f16_dev(_p(a), _p(bT), _p(c), M, K, N, _stream()))
    return c


def fp16_max_abs(x):
    _dev(x, torch.float16)
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    _check(lib.cactus_hip_fp16_max_abs_dev(_p(x), x.numel(), _p(out), _stream()))
    return out


def fp16_to_int8(x, scale):
    _dev(x, torch.float16)
    q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    _check(lib.cactus_hip_fp16_to_int8_dev(_p(x), _p(q), x.numel(), float(scale), _stream()))
    return q


def quantize_kv(src, group_size=128, out=None, out_scales=None):
    """fp16 [seq, kv_heads, head_dim] -> (int8 same shape, fp32 [seq, kv_heads, ceil(D/gs)])."""
    _dev(src, torch.float16)
    S, H, D = src.shape
    G = (D + group_size - 1) // group_size
    q = out if out is not None else torch.empty((S, H, D), dtype=torch.int8, device=src.device)
    sc = out_scales if out_scales is not None else torch.empty((S, H, G), dtype=torch.float32,
                                                                device=src.device)
    _check(lib.cactus_hip_quantize_kv_fp16_to_int8_dev(_p(src), _p(q), _p(sc), S, H, D, group_size,
                                                       _stream()))
    return q, sc


def unpack_int4(packed, count, out=None):
    _dev(packed, torch.uint8)
    o = out if out is not None else torch.zeros((count,), dtype=torch.int8, device=packed.device)
    _check(lib.cactus_hip_unpack_int4_to_int8_dev(_p(packed), _p(o), count, _stream()))
    return o


def rms_norm(x, weight, eps=1e-5, out=None):
    """fp16 [rows, dims] normalised row-wise and multiplied by weight [dims]."""
    _dev(x, torch.float16); _dev(weight, torch.float16)
    rows, dims = x.shape
    o = out if out is not None else torch.empty_like(x)
    _check(lib.cactus_hip_rms_norm_f16_dev(_p(x), _p(weight), _p(o), rows, dims, float(eps), _stream()))
    return o


def add_rms_norm(a, b, weight, eps=1e-5, out_sum=None, out_norm=None):
    """(a + b, rms_norm(a + b, weight)) in one launch."""
    _dev(a, torch.float16); _dev(b, torch.float16); _dev(weight, torch.float16)
    rows, dims = a.shape
    s = out_sum if out_sum is not None else torch.empty_like(a)
    n = out_norm if out_norm is not None else torch.empty_like(a)
    _check(lib.cactus_hip_add_rms_norm_f16_dev(_p(a), _p(b), _p(weight), _p(s), _p(n), rows, dims, float(eps),
                                               _stream()))
    return s, n


def add_rms_norm_quant(a, b, weight, eps=1e-5, out_sum=None, out_norm=None, out_q=None, out_scales=None,
                       want_norm=False):
    """rms_norm(a + b, weight) (b may be None) and its row-wise INT8 quantization, one launch.
    Returns (sum or None, norm or None, q int8 [rows, dims], scales fp32 [rows])."""
    _dev(a, torch.float16); _dev(weight, torch.float16)
    rows, dims = a.shape
    s = None
    if b is not None:
        _dev(b, torch.float16)
        s = out_sum if out_sum is not None else torch.empty_like(a)
    n = out_norm if out_norm is not None else (torch.empty_like(a) if want_norm else None)
    q = out_q if out_q is not None else torch.empty((rows, dims), dtype=torch.int8, device=a.device)
    sc = out_scales if out_scales is not None else torch.empty((rows,), dtype=torch.float32, device=a.device)
    _check(lib.cactus_hip_add_rms_norm_quant_f16_dev(_p(a), _p(b), _p(weight), _p(s), _p(n), _p(q), _p(sc), rows, dims,
                                                     float(eps), _stream()))
    return s, n, q, sc


def _act_multiply_quant(fn, gate, up, out_act, out_q, out_scales, want_act):
    _dev(gate, torch.float16); _dev(up, torch.float16)
    rows, dims = gate.shape
    act = out_act if out_act is not None else (torch.empty_like(gate) if want_act else None)
    q = out_q if out_q is not None else torch.empty((rows, dims), dtype=torch.int8, device=gate.device)
    sc = out_scales if out_scales is not None else torch.empty((rows,), dtype=torch.float32, device=gate.device)
    _check(fn(_p(gate), _p(up), _p(act), _p(q), _p(sc), rows, dims, _stream()))
    return act, q, sc


def silu_multiply_quant(gate, up, out_act=None, out_q=None, out_scales=None, want_act=False):
    """silu(gate) * up of [rows, dims] and its row-wise INT8 quantization, one launch."""
    return _act_multiply_quant(lib.cactus_hip_silu_multiply_quant_f16_dev, gate, up, out_act, out_q, out_scales, want_act)


def gelu_multiply_quant(gate, up, out_act=None, out_q=None, out_scales=None, want_act=False):
    return _act_multiply_quant(lib.cactus_hip_gelu_multiply_quant_f16_dev, gate, up, out_act, out_q, out_scales, want_act)


def qk_norm_rope(q, k, q_weight, k_weight, eps, theta, start_pos=0, out_q=None, out_k=None):
    """Per-head RMS norm then RoPE of q [batch, seq, Hq, D] and k [batch, seq, Hkv, D], one launch."""
    _dev(q, torch.float16); _dev(k, torch.float16)
    B, S, Hq, D = q.shape
    Hkv = k.shape[2]
    oq = out_q if out_q is not None else torch.empty_like(q)
    ok = out_k if out_k is not None else torch.empty_like(k)
    _check(lib.cactus_hip_qk_norm_rope_f16_dev(_p(q), _p(k), _p(q_weight), _p(k_weight), _p(oq), _p(ok), B, S, Hq,
                                               Hkv, D, float(eps), start_pos, float(theta), _stream()))
    return oq, ok


def qk_norm_rope_kv_append(q, k, v, q_weight, k_weight, eps, theta, start_pos, k_cache, v_cache, k_scales, v_scales,
                           slots=None, group_size=128, out_q=None, out_k=None):
    """qk_norm_rope plus the INT8 cache append of the step's K (rotated) and V, one launch.
    q [B, S, Hq, D], k / v [B, S, Hkv, D]; caches [B, max_slots, Hkv, D] with slots int32 [B]
    (sequence b's tokens go to slots[b] .. slots[b] + S - 1), or slots None and caches [B * S, Hkv, D]."""
    _dev(q, torch.float16); _dev(k, torch.float16); _dev(v, torch.float16)
    B, S, Hq, D = q.shape
    Hkv = k.shape[2]
    oq = out_q if out_q is not None else torch.empty_like(q)
    ok = out_k if out_k is not None else torch.empty_like(k)
    if slots is not None:
        _dev(slots, torch.int32)
        cbs, sbs, ms = k_cache.stride(0), k_scales.stride(0), k_cache.shape[1]
    else:
        cbs = sbs = ms = 0
    _check(lib.cactus_hip_qk_norm_rope_kv_append_f16_dev(
        _p(q), _p(k), _p(v), _p(q_weight), _p(k_weight), _p(oq), _p(ok), _p(k_cache), _p(v_cache), _p(k_scales),
        _p(v_scales), _p(slots), B, S, Hq, Hkv, D, float(eps), start_pos, float(theta), group_size, cbs, sbs, ms,
        _stream()))
    return oq, ok


def conv1d_causal_depthwise(x, weight, dilation=1, out=None):
    """x fp16 [N, L, C], weight fp16 [C, K] (or [C, 1, K]) -> fp16 [N, L, C]."""
    _dev(x, torch.float16); _dev(weight, torch.float16)
    N, L, Cc = x.shape
    K = weight.shape[-1]
    o = out if out is not None else torch.empty_like(x)
    _check(lib.cactus_hip_conv1d_causal_depthwise_f16_dev(_p(x), _p(weight), _p(o), N, L, Cc, K, dilation, _stream()))
    return o


def lfm2_conv_gate(in_proj, weight, state, out=None):
    """in_proj fp16 [num_seqs, L, 3 * C] (B, C-gate, X), weight fp16 [C, K], state fp16 [num_seqs, W, C]
    (updated in place; None for no state) -> gated fp16 [num_seqs, L, C]."""
    _dev(in_proj, torch.float16); _dev(weight, torch.float16)
    nseq, L, C3 = in_proj.shape
    Cc = C3 // 3
    K = weight.shape[-1]
    W = 0 if state is None else state.shape[1]
    if state is not None:
        _dev(state, torch.float16)
    o = out if out is not None else torch.empty((nseq, L, Cc), dtype=torch.float16, device=in_proj.device)
    _check(lib.cactus_hip_lfm2_conv_gate_f16_dev(_p(in_proj), _p(weight), _p(state), _p(o), nseq, L, Cc, K, W,
                                                 _stream()))
    return o


def lfm2_conv_gate_quant(in_proj, weight, state, out_q=None, out_scales=None, out=None, want_gated=False):
    """lfm2_conv_gate whose gated rows leave quantized for the out-projection: returns (q int8 [num_seqs * L, C],
    scales float32 [num_seqs * L]) and, with want_gated or out, the fp16 rows as well."""
    _dev(in_proj, torch.float16); _dev(weight, torch.float16)
    nseq, L, C3 = in_proj.shape
    Cc = C3 // 3
    W = 0 if state is None else state.shape[1]
    q = out_q if out_q is not None else torch.empty((nseq * L, Cc), dtype=torch.int8, device=in_proj.device)
    sc = out_scales if out_scales is not None else torch.empty((nseq * L,), dtype=torch.float32, device=in_proj.device)
    o = out
    if o is None and want_gated:
        o = torch.empty((nseq, L, Cc), dtype=torch.float16, device=in_proj.device)
    _check(lib.cactus_hip_lfm2_conv_gate_quant_f16_dev(_p(in_proj), _p(weight), _p(state), _p(o), _p(q), _p(sc), nseq, L,
                                                       Cc, weight.shape[-1], W, _stream()))
    return (q, sc, o) if o is not None else (q, sc)


def silu_multiply(gate, up, out=None):
    return _binary(lib.cactus_hip_silu_multiply_f16_dev, gate, up, out)


def gelu_multiply(gate, up, out=None):
    return _binary(lib.cactus_hip_gelu_multiply_f16_dev, gate, up, out)


def quantize_kv_pair(k, v, out_k, out_v, out_k_scales, out_v_scales, group_size=128):
    """K and V [seq, kv_heads, head_dim] quantized in one launch into the given int8 / fp32 outputs."""
    _dev(k, torch.float16); _dev(v, torch.float16)
    S, Hh, D = k.shape
    _check(lib.cactus_hip_quantize_kv_pair_fp16_to_int8_dev(_p(k), _p(v), _p(out_k), _p(out_v), _p(out_k_scales),
                                                            _p(out_v_scales), S, Hh, D, group_size, _stream()))


def kv_append_batched(k_new, v_new, k_cache, v_cache, k_scales, v_scales, slots, group_size=128):
    """k_new / v_new fp16 [B, Hkv, D] quantized into slot slots[b] of caches int8 [B, slots, Hkv, D] and
    scales fp32 [B, slots, Hkv, G]."""
    _dev(k_new, torch.float16); _dev(v_new, torch.float16); _dev(slots, torch.int32)
    B, Hh, D = k_new.shape
    _check(lib.cactus_hip_kv_append_batched_dev(_p(k_new), _p(v_new), _p(k_cache), _p(v_cache), _p(k_scales),
                                                _p(v_scales), B, Hh, D, group_size, _p(slots), k_cache.stride(0),
                                                k_scales.stride(0), k_cache.shape[1], _stream()))


def qk_norm_rope_seq(q, k, v, q_weight, k_weight, eps, theta, positions, max_position, k_cache=None, v_cache=None,
                     k_scales=None, v_scales=None, window=0, sink=0, group_size=128, out_q=None, out_k=None):
    """qk_norm_rope for independent sequences at positions int32 [B] (tokens seen so far); with v (and the
    caches [B, slots, Hkv, D] / scales [B, slots, Hkv, G]) the rotated k and v rows are also quantized into the
    ring slot of their position.  v None: no append (append after the attention with kv_append_windowed)."""
    _dev(q, torch.float16); _dev(k, torch.float16); _dev(positions, torch.int32)
    B, S, Hq, D = q.shape
    Hkv = k.shape[2]
    oq = out_q if out_q is not None else torch.empty_like(q)
    ok = out_k if out_k is not None else torch.empty_like(k)
    cbs = sbs = ms = 0
    if v is not None:
        _dev(v, torch.float16)
        cbs, sbs, ms = k_cache.stride(0), k_scales.stride(0), k_cache.shape[1]
    _check(lib.cactus_hip_qk_norm_rope_seq_f16_dev(
        _p(q), _p(k), _p(v), _p(q_weight), _p(k_weight), _p(oq), _p(ok), _p(k_cache if v is not None else None),
        _p(v_cache if v is not None else None), _p(k_scales if v is not None else None),
        _p(v_scales if v is not None else None), _p(positions), max_position, B, S, Hq, Hkv, D, float(eps), float(theta),
        group_size, cbs, sbs, ms, window, sink, _stream()))
    return oq, ok


def kv_append_windowed(k_new, v_new, k_cache, v_cache, k_scales, v_scales, positions, window=0, sink=0, group_size=128):
    """k_new / v_new fp16 [B, S, Hkv, D] quantized into the ring slots of positions[b] + s."""
    _dev(k_new, torch.float16); _dev(v_new, torch.float16); _dev(positions, torch.int32)
    B, S, Hh, D = k_new.shape
    _check(lib.cactus_hip_kv_append_windowed_dev(_p(k_new), _p(v_new), _p(k_cache), _p(v_cache), _p(k_scales),
                                                 _p(v_scales), _p(positions), B, S, Hh, D, group_size, k_cache.stride(0),
                                                 k_scales.stride(0), k_cache.shape[1], window, sink, _stream()))


def kv_positions_advance(positions, cache_lens, advance, window=0):
    _dev(positions, torch.int32)
    _check(lib.cactus_hip_kv_positions_advance_dev(_p(positions), _p(cache_lens), positions.numel(), advance, window,
                                                   _stream()))


def rope(x, theta, start_pos=0, out=None):
    """Half-split rotary embedding on fp16 [batch, seq, heads, head_dim]."""
    _dev(x, torch.float16)
    B, S, H, D = x.shape
    o = out if out is not None else torch.empty_like(x)
    _check(lib.cactus_hip_rope_f16_dev(_p(x), _p(o), B, S, H, D, start_pos, float(theta), _stream()))
    return o


def rope_reserve(head_dim, theta, max_positions):
    _check(lib.cactus_hip_rope_reserve(head_dim, float(theta), max_positions))


def _unary(fn, x, out):
    _dev(x, torch.float16)
    o = out if out is not None else torch.empty_like(x)
    _check(fn(_p(x), _p(o), x.numel(), _stream()))
    return o


def silu(x, out=None):
    return _unary(lib.cactus_hip_silu_f16_dev, x, out)


def gelu(x, out=None):
    return _unary(lib.cactus_hip_gelu_f16_dev, x, out)


def _binary(fn, a, b, out):
    _dev(a, torch.float16); _dev(b, torch.float16)
    assert a.shape == b.shape
    o = out if out is not None else torch.empty_like(a)
    _check(fn(_p(a), _p(b), _p(o), a.numel(), _stream()))
    return o


SCALAR_OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "exp": 4, "sqrt": 5, "cos": 6, "sin": 7}


def scalar_op(op, x, value, out=None):
    """x (op) half(value), element-wise on fp16."""
    _dev(x, torch.float16)
    o = out if out is not None else torch.empty_like(x)
    _check(lib.cactus_hip_scalar_op_f16_dev(_p(x), _p(o), x.numel(), float(value), SCALAR_OPS[op], _stream()))
    return o


def add(a, b, out=None):
    return _binary(lib.cactus_hip_add_f16_dev, a, b, out)


def add_clipped(a, b, out=None):
    return _binary(lib.cactus_hip_add_f16_clipped_dev, a, b, out)


def subtract(a, b, out=None):
    return _binary(lib.cactus_hip_subtract_f16_dev, a, b, out)


def multiply(a, b, out=None):
    return _binary(lib.cactus_hip_multiply_f16_dev, a, b, out)


def divide(a, b, out=None):
    return _binary(lib.cactus_hip_divide_f16_dev, a, b, out)


def sample_greedy(logits, out=None):
    """First index of the row maximum of fp16 [rows, vocab] -> int32 [rows] (the greedy sampler)."""
    _dev(logits, torch.float16)
    rows, vocab = logits.shape
    o = out if out is not None else torch.empty((rows,), dtype=torch.int32, device=logits.device)
    _check(lib.cactus_hip_sample_greedy_f16_dev(_p(logits), _p(o), rows, vocab, _stream()))
    return o


def sample_uniform(seed):
    """The uniform draw the reference takes from std::mt19937(seed) (host computation)."""
    return float(lib.cactus_hip_sample_uniform(seed & 0xFFFFFFFF))


class Sampler:
    """cactus_sample_f16 for `rows` independent sequences: keeps each row's token history (the
    reference's process-wide history of the last 128 sampled tokens), the workspace and the status
    flag on the device.  draws: float32 [rows] uniform numbers (sample_uniform(seed) per row)."""

    def __init__(self, rows, vocab, device="cuda"):
        self.rows, self.vocab = rows, vocab
        self.history = torch.zeros((rows, 128), dtype=torch.int32, device=device)
        self.counts = torch.zeros((rows,), dtype=torch.int32, device=device)
        self.work = torch.empty((rows, vocab), dtype=torch.float16, device=device)
        self.status = torch.zeros((1,), dtype=torch.int32, device=device)

    def sample(self, logits, temperature, top_p, top_k, draws, out=None, check=True):
        _dev(logits, torch.float16)
        rows, vocab = logits.shape
        if (rows, vocab) != (self.rows, self.vocab):
            raise ValueError("Sampler: logits shape does not match")
        _dev(draws, torch.float32)
        o = out if out is not None else torch.empty((rows,), dtype=torch.int32, device=logits.device)
        _check(lib.cactus_hip_sample_f16_dev(_p(logits), _p(o), rows, vocab, temperature, top_p, top_k, _p(draws),
                                             _p(self.history), _p(self.counts), _p(self.work), _p(self.status),
                                             _stream()))
        if check and int(self.status.item()) != 0:
            self.status.zero_()
            raise RuntimeError("sample: more than %d logits survive top-k in a row"
                               % lib.cactus_hip_sample_max_survivors())
        return o

    def history_of(self, row):
        return self.history[row, : int(self.counts[row].item())].cpu().tolist()


def logits_entropy(logits, out=None):
    """Entropy of softmax(row) / log(vocab) per row of fp16 [rows, vocab] -> float32 [rows]."""
    _dev(logits, torch.float16)
    rows, vocab = logits.shape
    o = out if out is not None else torch.empty((rows,), dtype=torch.float32, device=logits.device)
    _check(lib.cactus_hip_logits_entropy_f16_dev(_p(logits), _p(o), rows, vocab, _stream()))
    return o


def embedding(table, indices, scales=None, group_size=128, out=None, check=True):
    """Rows of an INT8 (per-group fp16 scales) or FP16 table; indices int32 [n].  Raises on an
    index outside the table (after synchronising, like the reference's exception); check=False
    skips the flag and the synchronisation (e.g. under stream capture)."""
    _dev(indices, torch.int32)
    vocab, hidden = table.shape
    n = indices.numel()
    o = out if out is not None else torch.empty((n, hidden), dtype=torch.float16, device=table.device)
    bad = torch.zeros((1,), dtype=torch.int32, device=table.device) if check else None
    if table.dtype == torch.int8:
        _dev(table, torch.int8); _dev(scales, torch.float16)
        _check(lib.cactus_hip_embedding_int8_dev(_p(table), _p(scales), _p(indices), _p(o), n, vocab, hidden,
                                                 group_size, _p(bad), _stream()))
    else:
        _dev(table, torch.float16)
        _check(lib.cactus_hip_embedding_f16_dev(_p(table), _p(indices), _p(o), n, vocab, hidden, _p(bad),
                                                _stream()))
    if check and int(bad.item()):
        raise CactusHipError("Embedding index out of bounds")
    return o


def attention_f16(q, k, v, scale, mask=None, position_offset=0, window_size=0, is_causal=True, out=None):
    _dev(q, torch.float16); _dev(k, torch.float16); _dev(v, torch.float16)
    B, S, Hq, D = q.shape
    KV, Hkv = k.shape[1], k.shape[2]
    out = out if out is not None else torch.empty_like(q)
    _check(lib.cactus_hip_attention_f16_dev(_p(q), _p(k), _p(v), _p(out), B, S, KV, Hq, Hkv, D,
                                            float(scale), _p(mask), position_offset, window_size,
                                            int(bool(is_causal)), _stream()))
    return out


def attention_hybrid(q, k_cached, v_cached, k_scales, v_scales, k_new, v_new, scale, position_offset=0,
                     is_causal=True, group_size=128, cache_len=None, out=None, noalloc=False, out_q=None,
                     out_scales=None):
    """noalloc: the entry point that keeps its split-KV records in the library's fixed scratch
    (pass `out`, so that nothing is allocated on the way either).
    out_q int8 [B * S, Hq * D] / out_scales fp32 [B * S]: the output rows are also left quantized for the
    output projection."""
    _dev(q, torch.float16)
    B, S, Hq, D = q.shape
    new_len, Hkv = k_new.shape[1], k_new.shape[2]
    if cache_len is None:
        cache_len = k_cached.shape[-3]
    out = out if out is not None else torch.empty_like(q)
    if out_q is not None:
        _check(lib.cactus_hip_attention_hybrid_int8_fp16_quant_dev(
            _p(q), _p(k_cached), _p(v_cached), _p(k_scales), _p(v_scales), _p(k_new), _p(v_new), _p(out), _p(out_q),
            _p(out_scales), B, S, cache_len, new_len, Hq, Hkv, D, float(scale), position_offset, int(bool(is_causal)),
            group_size, _stream()))
        return out
    fn = lib.cactus_hip_attention_hybrid_int8_fp16_noalloc_dev if noalloc else lib.cactus_hip_attention_hybrid_int8_fp16_dev
    _check(fn(
        _p(q), _p(k_cached), _p(v_cached), _p(k_scales), _p(v_scales), _p(k_new), _p(v_new), _p(out),
        B, S, cache_len, new_len, Hq, Hkv, D, float(scale), position_offset, int(bool(is_causal)),
        group_size, _stream()))
    return out


def attention_decode_fused(q, k, v, q_weight, k_weight, k_cache, v_cache, k_scales, v_scales, cache_lens, positions,
                           max_position, eps, theta, scale=0.0, group_size=128, out=None, out_q=None, out_scales=None):
    """One new token per sequence, from the projections' rows to the attention output in one launch: q [B, Hq, D],
    k / v [B, Hkv, D] fp16; per-head norm + RoPE at positions[b], K / V appended at slot cache_lens[b] of the caches
    int8 [B, slots, Hkv, D] / fp32 [B, slots, Hkv, G], attention over the cached tokens and the new one.
    out_q int8 [B, Hq * D] / out_scales fp32 [B]: the output rows are also left quantized for the output projection."""
    B, Hq, D = q.shape[0], q.shape[-2], q.shape[-1]
    Hkv, slots, G = k.shape[-2], k_cache.shape[1], k_scales.shape[-1]
    out = out if out is not None else torch.empty((B, 1, Hq, D), dtype=torch.float16, device=q.device)
    if out_q is not None:
        _check(lib.cactus_hip_attention_decode_fused_quant_f16_dev(
            _p(q), _p(k), _p(v), _p(q_weight), _p(k_weight), _p(k_cache), _p(v_cache), _p(k_scales), _p(v_scales),
            _p(out), _p(out_q), _p(out_scales), B, _p(cache_lens), _p(positions), int(max_position), Hq, Hkv, D,
            float(eps), float(theta), float(scale), group_size, slots * Hkv * D, slots * Hkv * G, slots, slots,
            _stream()))
        return out
    _check(lib.cactus_hip_attention_decode_fused_f16_dev(
        _p(q), _p(k), _p(v), _p(q_weight), _p(k_weight), _p(k_cache), _p(v_cache), _p(k_scales), _p(v_scales), _p(out),
        B, _p(cache_lens), _p(positions), int(max_position), Hq, Hkv, D, float(eps), float(theta), float(scale),
        group_size, slots * Hkv * D, slots * Hkv * G, slots, slots, _stream()))
    return out


def attention_decode_fused_single(q, k, v, q_weight, k_weight, k_cache, v_cache, k_scales, v_scales, cache_len, position,
                                  eps, theta, scale=0.0, group_size=128):
    """One sequence, cache length and position given by the host, nothing appended: q [Hq, D], k / v [Hkv, D] fp16
    (the projections' rows), caches int8 [>= cache_len, Hkv, D] / fp32 [.., Hkv, G].  Returns (attention output
    [1, 1, Hq, D], the normalized and rotated key rows [1, 1, Hkv, D])."""
    Hq, D, Hkv = q.shape[-2], q.shape[-1], k.shape[-2]
    out = torch.empty((1, 1, Hq, D), dtype=torch.float16, device=q.device)
    k_out = torch.empty((1, 1, Hkv, D), dtype=torch.float16, device=q.device)
    _check(lib.cactus_hip_attention_decode_fused_single_f16_dev(
        _p(q), _p(k), _p(v), _p(q_weight), _p(k_weight), _p(k_cache), _p(v_cache), _p(k_scales), _p(v_scales), _p(out),
        _p(k_out), int(cache_len), int(position), Hq, Hkv, D, float(eps), float(theta), float(scale), group_size,
        _stream()))
    return out, k_out


def attention_hybrid_batched(q, k_cached, v_cached, k_scales, v_scales, k_new, v_new, scale, cache_lens,
                             position_offsets, group_size=128, out=None):
    """k_cached/v_cached int8 [B, max_cache, Hkv, D]; scales fp32 [B, max_cache, Hkv, G];
    cache_lens / position_offsets int32 [B]."""
    B, S, Hq, D = q.shape
    new_len, Hkv = k_new.shape[1], k_new.shape[2]
    max_cache = k_cached.shape[1]
    G = k_scales.shape[-1]
    out = out if out is not None else torch.empty_like(q)
    _check(lib.cactus_hip_attention_hybrid_int8_fp16_batched_dev(
        _p(q), _p(k_cached), _p(v_cached), _p(k_scales), _p(v_scales), _p(k_new), _p(v_new), _p(out),
        B, S, _p(cache_lens), new_len, Hq, Hkv, D, float(scale), _p(position_offsets), group_size,
        max_cache * Hkv * D, max_cache * Hkv * G, max_cache, _stream()))
    return out
