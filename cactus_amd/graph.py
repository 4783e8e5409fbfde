"""Python view of the C ABI in include/cactus_hip_graph.h: the device-resident mirror of the
reference's CactusGraph / KVCache for the hot path (cactus_amd/csrc/graph.h).  Host arrays are
numpy; node ids are the graph's own.  Method names and argument meaning follow
cactus/graph/graph.h:386-500 and cactus/engine/engine.h:324-381."""
import ctypes as C
import os

import numpy as np

from .hip import lib, CactusHipError

INT8, FP16, FP32, INT4 = 0, 1, 2, 3

_sz, _vp, _i, _f, _l = C.c_size_t, C.c_void_p, C.c_int, C.c_float, C.c_long
_szp = C.POINTER(C.c_size_t)

GRAPH_SIGNATURES = {
    "cactus_hip_graph_create": (_vp, []),
    "cactus_hip_graph_destroy": (None, [_vp]),
    "cactus_hip_graph_input": (_l, [_vp, _szp, _sz, _i]),
    "cactus_hip_graph_matmul": (_l, [_vp, _sz, _sz, _i]),
    "cactus_hip_graph_quantize_activations": (_l, [_vp, _sz]),
    "cactus_hip_graph_attention": (_l, [_vp, _sz, _sz, _sz, _f, _sz, _sz, _i]),
    "cactus_hip_graph_attention_int8_hybrid": (_l, [_vp, _sz, _sz, _sz, _f, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _sz]),
    "cactus_hip_graph_binary": (_l, [_vp, _i, _sz, _sz]),
    "cactus_hip_graph_scalar_op": (_l, [_vp, _i, _sz, _f]),
    "cactus_hip_graph_silu": (_l, [_vp, _sz]),
    "cactus_hip_graph_gelu": (_l, [_vp, _sz]),
    "cactus_hip_graph_reshape": (_l, [_vp, _sz, _szp, _sz]),
    "cactus_hip_graph_slice": (_l, [_vp, _sz, _i, _sz, _sz]),
    "cactus_hip_graph_concat": (_l, [_vp, _sz, _sz, _i]),
    "cactus_hip_graph_index": (_l, [_vp, _sz, _sz, _i]),
    "cactus_hip_graph_sample": (_l, [_vp, _sz, _f, _f, _sz, C.c_uint32]),
    "cactus_hip_graph_reset_sample_history": (_i, [_vp]),
    "cactus_hip_graph_embedding": (_l, [_vp, _sz, _sz]),
    "cactus_hip_graph_rms_norm": (_l, [_vp, _sz, _sz, _f]),
    "cactus_hip_graph_rope": (_l, [_vp, _sz, _f, _sz]),
    "cactus_hip_graph_conv1d_causal": (_l, [_vp, _sz, _sz, _sz, _sz]),
    "cactus_hip_graph_mmap_weights": (_l, [_vp, C.c_char_p]),
    "cactus_hip_graph_node_info": (_i, [_vp, _sz, _szp, _szp, C.POINTER(C.c_int), _szp, _szp, C.POINTER(C.c_int)]),
    "cactus_hip_cact_parse_header": (_i, [_vp, _sz, C.POINTER(C.c_uint64)]),
    "cactus_hip_model_config_load": (_i, [C.c_char_p, _vp]),
    "cactus_hip_graph_set_input": (_i, [_vp, _sz, _vp, _i]),
    "cactus_hip_graph_set_external_input": (_i, [_vp, _sz, _vp, _i]),
    "cactus_hip_graph_set_grouped_scales": (_i, [_vp, _sz, _sz, _sz, _vp]),
    "cactus_hip_graph_set_packed_int4": (_i, [_vp, _sz, _vp, _sz]),
    "cactus_hip_graph_execute": (_i, [_vp]),
    "cactus_hip_graph_get_output": (_i, [_vp, _sz, _vp, _sz]),
    "cactus_hip_graph_get_device_output": (_vp, [_vp, _sz]),
    "cactus_hip_graph_soft_reset": (_i, [_vp]),
    "cactus_hip_graph_hard_reset": (_i, [_vp]),
    "cactus_hip_graph_node_count": (_sz, [_vp]),
    "cactus_hip_graph_pool_allocations": (_sz, [_vp]),
    "cactus_hip_graph_pool_idle_bytes": (_sz, [_vp]),
    "cactus_hip_graph_set_pool_max_idle_bytes": (None, [_vp, _sz]),
    "cactus_hip_graph_set_streaming_twins": (None, [_vp, C.c_int]),
    "cactus_hip_graph_set_fusion": (None, [_vp, _i]),
    "cactus_hip_graph_fused_launches": (_sz, [_vp]),
    "cactus_hip_kvcache_create": (_vp, [_sz, _sz, _sz, _sz]),
    "cactus_hip_kvcache_destroy": (None, [_vp]),
    "cactus_hip_kvcache_set_window": (_i, [_vp, _sz, _sz]),
    "cactus_hip_kvcache_set_ring": (_i, [_vp, _i]),
    "cactus_hip_kvcache_ring_head": (_sz, [_vp]),
    "cactus_hip_kvcache_reset": (None, [_vp]),
    "cactus_hip_kvcache_update": (_i, [_vp, _vp, _szp, _szp, _sz]),
    "cactus_hip_kvcache_len": (_sz, [_vp]),
    "cactus_hip_kvcache_total_len": (_sz, [_vp]),
    "cactus_hip_kvcache_keys": (_vp, [_vp, _sz]),
    "cactus_hip_kvcache_values": (_vp, [_vp, _sz]),
    "cactus_hip_kvcache_key_scales": (_vp, [_vp, _sz]),
    "cactus_hip_kvcache_value_scales": (_vp, [_vp, _sz]),
}
for _name, (_res, _args) in GRAPH_SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

_ITEMSIZE = {INT8: 1, FP16: 2, FP32: 4, INT4: 1}
_NPTYPE = {INT8: np.int8, FP16: np.float16, FP32: np.float32}


def _err():
    return CactusHipError(lib.cactus_hip_last_error().decode())


def _node(rc):
    if rc < 0:
        raise _err()
    return int(rc)


def _ok(rc):
    if rc != 0:
        raise _err()


class CactusGraph:
    def __init__(self):
        self._g = lib.cactus_hip_graph_create()
        self._keep = []      # host arrays whose addresses key resident device copies
        self._shapes = {}

    def close(self):
        if self._g:
            lib.cactus_hip_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        self.close()

    def input(self, shape, precision=INT8):
        arr = (C.c_size_t * len(shape))(*shape)
        n = _node(lib.cactus_hip_graph_input(self._g, arr, len(shape), precision))
        self._shapes[n] = (tuple(shape), precision)
        return n

    def matmul(self, a, b, pretransposed_rhs=False):
        n = _node(lib.cactus_hip_graph_matmul(self._g, a, b, int(pretransposed_rhs)))
        M = self._shapes[a][0][0]
        N = self._shapes[b][0][0] if pretransposed_rhs else self._shapes[b][0][1]
        self._shapes[n] = ((M, N), FP16)
        return n

    def quantize_activations(self, x):
        n = _node(lib.cactus_hip_graph_quantize_activations(self._g, x))
        self._shapes[n] = (self._shapes[x][0], INT8)
        return n

    def _same(self, n, src, precision=None):
        self._shapes[n] = (self._shapes[src][0], self._shapes[src][1] if precision is None else precision)
        return n

    def _binary(self, op, a, b):
        return self._same(_node(lib.cactus_hip_graph_binary(self._g, op, a, b)), a)

    def scalar_add(self, x, value):
        return self._same(_node(lib.cactus_hip_graph_scalar_op(self._g, 0, x, value)), x)

    def scalar_subtract(self, x, value):
        return self._same(_node(lib.cactus_hip_graph_scalar_op(self._g, 1, x, value)), x)

    def scalar_multiply(self, x, value):
        return self._same(_node(lib.cactus_hip_graph_scalar_op(self._g, 2, x, value)), x)

    def scalar_divide(self, x, value):
        return self._same(_node(lib.cactus_hip_graph_scalar_op(self._g, 3, x, value)), x)

    def add(self, a, b):
        return self._binary(0, a, b)

    def subtract(self, a, b):
        return self._binary(1, a, b)

    def multiply(self, a, b):
        return self._binary(2, a, b)

    def divide(self, a, b):
        return self._binary(3, a, b)

    def add_clipped(self, a, b):
        return self._binary(4, a, b)

    def silu(self, x):
        return self._same(_node(lib.cactus_hip_graph_silu(self._g, x)), x)

    def gelu(self, x):
        return self._same(_node(lib.cactus_hip_graph_gelu(self._g, x)), x)

    def reshape(self, x, shape):
        arr = (C.c_size_t * len(shape))(*shape)
        n = _node(lib.cactus_hip_graph_reshape(self._g, x, arr, len(shape)))
        self._shapes[n] = (tuple(shape), self._shapes[x][1])
        return n

    def slice(self, x, axis, start, length):
        n = _node(lib.cactus_hip_graph_slice(self._g, x, axis, start, length))
        shp = list(self._shapes[x][0])
        shp[axis] = length
        self._shapes[n] = (tuple(shp), self._shapes[x][1])
        return n

    def concat(self, a, b, axis=0):
        n = _node(lib.cactus_hip_graph_concat(self._g, a, b, axis))
        shp = list(self._shapes[a][0])
        ax = axis + len(shp) if axis < 0 else axis
        shp[ax] += self._shapes[b][0][ax]
        self._shapes[n] = (tuple(shp), self._shapes[a][1])
        return n

    def index(self, x, index_value, dim):
        n = _node(lib.cactus_hip_graph_index(self._g, x, index_value, dim))
        shp = list(self._shapes[x][0])
        d = dim + len(shp) if dim < 0 else dim
        del shp[d]
        self._shapes[n] = (tuple(shp) or (1,), self._shapes[x][1])
        return n

    def sample(self, logits, temperature, top_p, top_k, random_seed=0):
        """One token from the last logits row; get_output gives a float32-typed [1] array whose bits are
        the uint32 token (use .view(np.uint32))."""
        n = _node(lib.cactus_hip_graph_sample(self._g, logits, temperature, top_p, top_k, random_seed))
        self._shapes[n] = ((1,), FP32)
        return n

    def reset_sample_history(self):
        _ok(lib.cactus_hip_graph_reset_sample_history(self._g))

    def embedding(self, table, indices):
        n = _node(lib.cactus_hip_graph_embedding(self._g, table, indices))
        self._shapes[n] = (tuple(self._shapes[indices][0]) + (self._shapes[table][0][1],), FP16)
        return n

    def rms_norm(self, x, weight, epsilon=1e-5):
        return self._same(_node(lib.cactus_hip_graph_rms_norm(self._g, x, weight, epsilon)), x)

    def conv1d_causal(self, x, weight, kernel_size, dilation=1):
        return self._same(_node(lib.cactus_hip_graph_conv1d_causal(self._g, x, weight, kernel_size, dilation)), x)

    def rope(self, x, theta, position_offset=0):
        return self._same(_node(lib.cactus_hip_graph_rope(self._g, x, theta, position_offset)), x)

    def mmap_weights(self, filename):
        """A CACT tensor file as a weight node; returns the node id (shape / precision via node_info)."""
        n = _node(lib.cactus_hip_graph_mmap_weights(self._g, str(filename).encode()))
        info = self.node_info(n)
        self._shapes[n] = (info["shape"], info["precision"])
        return n

    def node_info(self, n):
        shape = (C.c_size_t * 4)()
        ndim, gs, ng = C.c_size_t(), C.c_size_t(), C.c_size_t()
        prec, packed = C.c_int(), C.c_int()
        _ok(lib.cactus_hip_graph_node_info(self._g, n, shape, C.byref(ndim), C.byref(prec), C.byref(gs), C.byref(ng),
                                           C.byref(packed)))
        return {"shape": tuple(shape[i] for i in range(ndim.value)), "precision": prec.value,
                "group_size": gs.value, "num_groups": ng.value, "packed_int4": bool(packed.value)}

    def attention(self, q, k, v, scale, position_offset=0, window_size=0, is_causal=True):
        n = _node(lib.cactus_hip_graph_attention(self._g, q, k, v, scale, position_offset, window_size,
                                                 int(is_causal)))
        self._shapes[n] = (self._shapes[q][0], FP16)
        return n

    def attention_int8_hybrid(self, q, key_new, value_new, scale, position_offset, cached_keys, cached_values,
                              k_scales, v_scales, cache_len, num_kv_heads, head_dim):
        n = _node(lib.cactus_hip_graph_attention_int8_hybrid(
            self._g, q, key_new, value_new, scale, position_offset, cached_keys, cached_values, k_scales,
            v_scales, cache_len, num_kv_heads, head_dim))
        self._shapes[n] = (self._shapes[q][0], FP16)
        return n

    def set_input(self, node, data, precision=None):
        a = np.ascontiguousarray(data)
        _ok(lib.cactus_hip_graph_set_input(self._g, node, a.ctypes.data_as(C.c_void_p),
                                           self._shapes[node][1] if precision is None else precision))

    def set_external_input(self, node, data, precision=None):
        a = np.ascontiguousarray(data)
        self._keep.append(a)
        _ok(lib.cactus_hip_graph_set_external_input(self._g, node, a.ctypes.data_as(C.c_void_p),
                                                    self._shapes[node][1] if precision is None else precision))

    def set_grouped_scales(self, node, group_size, num_groups, scales):
        a = np.ascontiguousarray(scales)
        self._keep.append(a)
        _ok(lib.cactus_hip_graph_set_grouped_scales(self._g, node, group_size, num_groups,
                                                    a.ctypes.data_as(C.c_void_p)))

    def set_packed_int4(self, node, packed):
        a = np.ascontiguousarray(packed)
        self._keep.append(a)
        _ok(lib.cactus_hip_graph_set_packed_int4(self._g, node, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def execute(self):
        _ok(lib.cactus_hip_graph_execute(self._g))

    def get_output(self, node):
        shape, prec = self._shapes[node]
        out = np.empty(shape, _NPTYPE[prec])
        _ok(lib.cactus_hip_graph_get_output(self._g, node, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def soft_reset(self):
        _ok(lib.cactus_hip_graph_soft_reset(self._g))

    def hard_reset(self):
        _ok(lib.cactus_hip_graph_hard_reset(self._g))
        self._keep.clear()

    def get_node_count(self):
        return lib.cactus_hip_graph_node_count(self._g)

    def pool_allocations(self):
        return lib.cactus_hip_graph_pool_allocations(self._g)

    def pool_idle_bytes(self):
        return lib.cactus_hip_graph_pool_idle_bytes(self._g)

    def set_pool_max_idle_bytes(self, nbytes):
        """Idle node buffers kept across soft resets (default 1 GiB); the excess is freed, largest first."""
        lib.cactus_hip_graph_set_pool_max_idle_bytes(self._g, int(nbytes))

    def set_streaming_twins(self, on):
        """Whether weights bound from now on get a tile-major twin for the decode matmuls (default on)."""
        lib.cactus_hip_graph_set_streaming_twins(self._g, 1 if on else 0)

    def set_fusion(self, on):
        """Whether execute() hands [add ->] rms_norm -> matmuls and silu/gelu -> multiply -> matmuls runs to the
        fused launches (default on; same results bit for bit)."""
        lib.cactus_hip_graph_set_fusion(self._g, 1 if on else 0)

    def fused_launches(self):
        return lib.cactus_hip_graph_fused_launches(self._g)


def parse_cact_header(data):
    """MappedFile::parse_header on a bytes-like object (no GPU needed)."""
    buf = bytes(data)
    out = (C.c_uint64 * 14)()
    _ok(lib.cactus_hip_cact_parse_header(buf, len(buf), out))
    keys = ("version", "alignment", "ndim", "d0", "d1", "d2", "d3", "precision", "data_bytes", "scales_bytes",
            "group_size", "num_groups", "scales_offset", "data_offset")
    return dict(zip(keys, [int(v) for v in out]))


class _ModelConfigStruct(C.Structure):
    _fields_ = ([(n, C.c_uint32) for n in ("vocab_size", "bos_token_id", "eos_token_id", "num_layers", "hidden_dim",
                                           "ffn_intermediate_dim", "attention_heads", "attention_kv_heads",
                                           "attention_head_dim")] +
                [("layer_norm_eps", C.c_float), ("rope_theta", C.c_float)] +
                [(n, C.c_uint32) for n in ("num_experts", "num_shared_experts", "num_top_experts",
                                           "moe_every_n_layers")] +
                [(n, C.c_int) for n in ("tie_word_embeddings", "model_type", "model_variant", "precision")] +
                [("default_temperature", C.c_float), ("default_top_p", C.c_float)] +
                [(n, C.c_uint32) for n in ("default_top_k", "conv_L_cache", "num_layer_types")] +
                [("layer_types", C.c_char * 2048)])


def load_model_config(path):
    """config.txt of a converted model folder -> dict (Config::from_json, engine_model.cpp:403-539)."""
    st = _ModelConfigStruct()
    _ok(lib.cactus_hip_model_config_load(os.fsencode(path), C.byref(st)))
    out = {n: getattr(st, n) for n, _ in _ModelConfigStruct._fields_}
    out["tie_word_embeddings"] = bool(out["tie_word_embeddings"])
    lt = out.pop("layer_types").decode()
    out["layer_types"] = lt.split(",") if lt else []
    out.pop("num_layer_types")
    return out


class KVCache:
    def __init__(self, num_layers, max_seq, num_kv_heads, head_dim):
        self._c = lib.cactus_hip_kvcache_create(num_layers, max_seq, num_kv_heads, head_dim)
        self.num_layers, self.num_kv_heads, self.head_dim = num_layers, num_kv_heads, head_dim

    def close(self):
        if self._c:
            lib.cactus_hip_kvcache_destroy(self._c)
            self._c = None

    def __del__(self):
        self.close()

    def set_window_size(self, window, sink=4):
        _ok(lib.cactus_hip_kvcache_set_window(self._c, window, sink))

    def set_ring(self, on=True):
        _ok(lib.cactus_hip_kvcache_set_ring(self._c, int(on)))

    def ring_head(self):
        return lib.cactus_hip_kvcache_ring_head(self._c)

    def reset(self):
        lib.cactus_hip_kvcache_reset(self._c)

    def update_from_graph(self, graph, k_nodes, v_nodes, seq_len):
        k = (C.c_size_t * len(k_nodes))(*k_nodes)
        v = (C.c_size_t * len(v_nodes))(*v_nodes)
        _ok(lib.cactus_hip_kvcache_update(self._c, graph._g, k, v, seq_len))

    def is_empty(self):
        return self.get_effective_seq_len() == 0

    def get_effective_seq_len(self):
        return lib.cactus_hip_kvcache_len(self._c)

    def get_total_seq_len(self):
        return lib.cactus_hip_kvcache_total_len(self._c)

    def get_keys_int8(self, layer):
        return lib.cactus_hip_kvcache_keys(self._c, layer)

    def get_values_int8(self, layer):
        return lib.cactus_hip_kvcache_values(self._c, layer)

    def get_key_scales(self, layer):
        return lib.cactus_hip_kvcache_key_scales(self._c, layer)

    def get_value_scales(self, layer):
        return lib.cactus_hip_kvcache_value_scales(self._c, layer)

    def upload_rows(self, layer, first, keys, key_scales, values, value_scales):
        """Overwrite slots [first, first + n) of one layer with host rows (keys / values int8 [n, Hkv, D], scales
        fp32 [n, Hkv, G]): lets a test continue a device cache from a state computed elsewhere."""
        G = (self.head_dim + 127) // 128
        n = keys.shape[0]
        assert first + n <= self.get_effective_seq_len()
        for ptr, a, row_bytes in ((self.get_keys_int8(layer), keys, self.num_kv_heads * self.head_dim),
                                  (self.get_key_scales(layer), key_scales, self.num_kv_heads * G * 4),
                                  (self.get_values_int8(layer), values, self.num_kv_heads * self.head_dim),
                                  (self.get_value_scales(layer), value_scales, self.num_kv_heads * G * 4)):
            a = np.ascontiguousarray(a)
            assert a.nbytes == n * row_bytes
            _ok(lib.cactus_hip_memcpy_h2d(C.c_void_p(ptr + first * row_bytes), a.ctypes.data_as(C.c_void_p), a.nbytes, None))
        _ok(lib.cactus_hip_synchronize(None))

    def download(self, layer):
        """(keys int8 [len,Hkv,D], key scales, values, value scales) of one layer as numpy."""
        L = self.get_effective_seq_len()
        G = (self.head_dim + 127) // 128
        out = []
        for ptr, shape, dt in ((self.get_keys_int8(layer), (L, self.num_kv_heads, self.head_dim), np.int8),
                               (self.get_key_scales(layer), (L, self.num_kv_heads, G), np.float32),
                               (self.get_values_int8(layer), (L, self.num_kv_heads, self.head_dim), np.int8),
                               (self.get_value_scales(layer), (L, self.num_kv_heads, G), np.float32)):
            a = np.empty(shape, dt)
            if L:
                rc = lib.cactus_hip_memcpy_d2h(a.ctypes.data_as(C.c_void_p), ptr, a.nbytes, None)
                if rc != 0:
                    raise _err()
            out.append(a)
        return tuple(out)
