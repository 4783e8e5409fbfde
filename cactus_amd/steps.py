"""Whole decode / prefill steps wired from the library's device entry points.

The reference builds one graph per token for ONE sequence (models/model_qwen.cpp:51-172,
models/model_lfm2.cpp:123-331; Model::decode / Model::prefill, engine_model.cpp:170-297).  These
classes enqueue the same op sequence for `batch` independent sequences on the current stream with
the back-to-back small ops going out as the library's one-launch sequences, and keep every piece
of per-sequence state on the device (token ids, positions, INT8 KV caches, LFM2 conv rows), so a
step can be captured into a HIP graph and replayed.  They are what bench.py times and what
tests/test_gpu_steps.py checks sequence by sequence against a CPU restatement of the reference.

Weights: dicts of device tensors.  A quantized matrix is a pair (B, B_scales): int8 [N, K] or packed
uint8 [N, K/2] (INT4) with fp16 scales [N, K/128]; a norm weight is an fp16 vector.
"""
import ctypes as C

import torch

from . import hip as H

QWEN3_0P6B = dict(hidden=1024, q_heads=16, kv_heads=8, head_dim=128, ffn=3072, vocab=151936, layers=28,
                  eps=1e-6, theta=1e6)                     # cactus/engine/engine.h:44-59
LFM2_1P2B = dict(hidden=2048, q_heads=32, kv_heads=8, head_dim=64, ffn=8192, vocab=65536, layers=16,
                 eps=1e-5, theta=1e6, conv_taps=3, conv_rows=2,
                 attention_layers=(2, 5, 8, 10, 12, 14))    # public model card; the reference reads config.txt


def synthetic_matrix(gen, N, K, int4=False):
    """Random weights shaped like the reference's bench inputs (tests/test_threading.cpp:79-87;
    packed bytes as tests/test_performance.cpp:368-371)."""
    if int4:
        B = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device="cuda", generator=gen)
    else:
        B = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=gen)
    Bs = (0.01 + torch.rand((N, K // 128), device="cuda", generator=gen) * 0.05).half()
    return B, Bs


def mixed_policy_int4(name, layer, layers):
    """The converter's MIXED policy for a model of Qwen3-0.6B's size (python/src/precision_config.py:52-68,
    110-124): INT8 for the embedding / lm_head, attn_q, attn_k and the first and last two blocks, packed
    INT4 for attn_v, attn_output and the MLP elsewhere."""
    return name in ("v", "o", "gate", "up", "down") and 2 <= layer < layers - 2


def _is4(w):
    return w[0].dtype == torch.uint8


def _half(*shape):
    return torch.empty(shape, dtype=torch.float16, device="cuda")


def _i8(*shape):
    return torch.empty(shape, dtype=torch.int8, device="cuda")


def _f32(*shape):
    return torch.empty(shape, dtype=torch.float32, device="cuda")


def pack_model_weights(weights):
    """Tile-major twins for every quantized matrix of a model dict (cactus_hip_weights_pack_dev): the decode
    matmuls of batches up to 16 rows stream them; the row-major tensors stay as they are for everything else."""
    for L in weights["layers"]:
        for v in L.values():
            if isinstance(v, tuple):
                H.pack_weights(v[0])
    H.pack_weights(weights["emb"][0])
    if "head" in weights:
        H.pack_weights(weights["head"][0])
    return weights


def _norm_weight(gen, n):
    return (1.0 + 0.1 * torch.randn((n,), device="cuda", generator=gen)).half()


def qwen3_synthetic_weights(gen, cfg=QWEN3_0P6B, mixed=False):
    c = cfg
    Hd, Hq, Hkv, D, F, V, NL = c["hidden"], c["q_heads"], c["kv_heads"], c["head_dim"], c["ffn"], c["vocab"], c["layers"]
    shapes = {"q": (Hq * D, Hd), "k": (Hkv * D, Hd), "v": (Hkv * D, Hd), "o": (Hd, Hq * D), "gate": (F, Hd),
              "up": (F, Hd), "down": (Hd, F)}
    layers = []
    for li in range(NL):
        L = {n: synthetic_matrix(gen, N, K, mixed and mixed_policy_int4(n, li, NL)) for n, (N, K) in shapes.items()}
        L.update(in_norm=_norm_weight(gen, Hd), post_norm=_norm_weight(gen, Hd), q_norm=_norm_weight(gen, D),
                 k_norm=_norm_weight(gen, D))
        layers.append(L)
    return {"layers": layers, "emb": synthetic_matrix(gen, V, Hd), "final_norm": _norm_weight(gen, Hd)}


def weights_to_device(m):
    """A host model (numpy arrays in the same dict layout) uploaded as it is."""
    def up(x):
        if isinstance(x, tuple):
            return tuple(up(y) for y in x)
        import numpy as np
        return torch.from_numpy(np.ascontiguousarray(x)).cuda()
    out = {"layers": [{k: up(v) for k, v in L.items()} for L in m["layers"]], "emb": up(m["emb"]),
           "final_norm": up(m["final_norm"])}
    if "head" in m:
        out["head"] = up(m["head"])
    return out


class _KvState:
    """Per-sequence decode state on the device: token ids, positions (tokens seen), cache lengths, and
    the INT8 caches int8 [n_attn, batch, slots, Hkv, D] with fp32 scales [n_attn, batch, slots, Hkv, G].
    window > 0: the reference's sliding window with `sink` leading tokens, held as a ring."""

    def __init__(self, n_attn, batch, slots, kv_heads, head_dim, window=0, sink=4, gen=None):
        if window:
            assert sink < window <= slots
        self.batch, self.slots, self.window, self.sink = batch, slots, window, sink if window else 0
        G = (head_dim + 127) // 128
        if gen is None:
            self.kc, self.vc = _i8(n_attn, batch, slots, kv_heads, head_dim), _i8(n_attn, batch, slots, kv_heads, head_dim)
            self.ksc = torch.ones((n_attn, batch, slots, kv_heads, G), dtype=torch.float32, device="cuda")
            self.vsc = torch.ones((n_attn, batch, slots, kv_heads, G), dtype=torch.float32, device="cuda")
            self.kc.zero_(); self.vc.zero_()
        else:   # benchmark contents
            shp = (n_attn, batch, slots, kv_heads, head_dim)
            self.kc = torch.randint(-127, 128, shp, dtype=torch.int8, device="cuda", generator=gen)
            self.vc = torch.randint(-127, 128, shp, dtype=torch.int8, device="cuda", generator=gen)
            self.ksc = torch.rand((n_attn, batch, slots, kv_heads, G), device="cuda", generator=gen) * 0.02
            self.vsc = torch.rand((n_attn, batch, slots, kv_heads, G), device="cuda", generator=gen) * 0.02
        self.positions = torch.zeros((batch,), dtype=torch.int32, device="cuda")
        self.cache_lens = torch.zeros((batch,), dtype=torch.int32, device="cuda")
        self.tok = torch.zeros((batch,), dtype=torch.int32, device="cuda")
        self.host_max_position = 0      # host mirror of max(positions): a step past the last slot is refused, not dropped

    def set_positions(self, positions):
        """positions: tokens each sequence has seen (host list); cache lengths follow."""
        p = torch.tensor(list(positions), dtype=torch.int32)
        self.host_max_position = int(p.max()) if p.numel() else 0
        self.positions.copy_(p)
        self.cache_lens.copy_(torch.clamp(p, max=self.window) if self.window else p)

    def advance(self, n=1):
        H.kv_positions_advance(self.positions, self.cache_lens, n, self.window)
        self.host_max_position += n

    def check_room(self):
        """Without a window every token needs a slot of its own: the kernels clamp a sequence that has run out of
        slots (its new token is not appended) rather than write past its slice, so running out is an error here."""
        if not self.window and self.host_max_position >= self.slots:
            raise RuntimeError(f"KV cache full: a sequence is at position {self.host_max_position} of {self.slots} slots")


# ------------------------------------------------------------------------------------------------
# The step drivers themselves live in the library (include/cactus_hip_steps.h, cactus_amd/csrc/steps.cpp): a C or
# C++ engine calls them directly.  The classes below describe a model and its state to them and own the tensors.
class _QMatrix(C.Structure):
    _fields_ = [("B", C.c_void_p), ("scales", C.c_void_p), ("N", C.c_size_t), ("K", C.c_size_t), ("int4", C.c_int)]


class _Block(C.Structure):
    _fields_ = [("kind", C.c_int), ("q", _QMatrix), ("k", _QMatrix), ("v", _QMatrix), ("o", _QMatrix),
                ("in_proj", _QMatrix), ("out_proj", _QMatrix), ("conv_weight", C.c_void_p),
                ("gate", _QMatrix), ("up", _QMatrix), ("down", _QMatrix),
                ("in_norm", C.c_void_p), ("post_norm", C.c_void_p), ("q_norm", C.c_void_p), ("k_norm", C.c_void_p)]


class _Weights(C.Structure):
    _fields_ = [("hidden", C.c_size_t), ("q_heads", C.c_size_t), ("kv_heads", C.c_size_t), ("head_dim", C.c_size_t),
                ("ffn", C.c_size_t), ("vocab", C.c_size_t), ("layers", C.c_size_t), ("conv_taps", C.c_size_t),
                ("conv_rows", C.c_size_t), ("eps", C.c_float), ("theta", C.c_float), ("blocks", C.POINTER(_Block)),
                ("embedding", _QMatrix), ("lm_head", _QMatrix), ("final_norm", C.c_void_p)]


class _DecodeState(C.Structure):
    _fields_ = [("batch", C.c_size_t), ("slots", C.c_size_t), ("window", C.c_size_t), ("sink", C.c_size_t),
                ("max_position", C.c_size_t), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p), ("k_scales", C.c_void_p),
                ("v_scales", C.c_void_p), ("positions", C.c_void_p), ("cache_lens", C.c_void_p), ("tokens", C.c_void_p),
                ("conv_state", C.c_void_p)]


class _PrefillState(C.Structure):
    _fields_ = [("capacity", C.c_size_t), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p), ("k_scales", C.c_void_p),
                ("v_scales", C.c_void_p), ("conv_state", C.c_void_p)]


_lib = H.lib
_lib.cactus_hip_prefill_workspace_bytes.restype = C.c_size_t
_lib.cactus_hip_prefill_workspace_bytes.argtypes = [C.POINTER(_Weights), C.c_size_t, C.c_size_t]
_lib.cactus_hip_prefill_dev.restype = C.c_int
_lib.cactus_hip_prefill_dev.argtypes = [C.POINTER(_Weights), C.POINTER(_PrefillState), C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.cactus_hip_decode_step_split.restype = C.c_size_t
_lib.cactus_hip_decode_step_split.argtypes = [C.c_size_t]
_lib.cactus_hip_decode_step_workspace_bytes.restype = C.c_size_t
_lib.cactus_hip_decode_step_workspace_bytes.argtypes = [C.POINTER(_Weights), C.c_size_t]
_lib.cactus_hip_decode_step_dev.restype = C.c_int
_lib.cactus_hip_decode_step_dev.argtypes = [C.POINTER(_Weights), C.POINTER(_DecodeState), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
_lib.cactus_hip_prefill_chunk_workspace_bytes.restype = C.c_size_t
_lib.cactus_hip_prefill_chunk_workspace_bytes.argtypes = [C.POINTER(_Weights), C.c_size_t]
_lib.cactus_hip_prefill_chunk_dev.restype = C.c_int
_lib.cactus_hip_prefill_chunk_dev.argtypes = [C.POINTER(_Weights), C.POINTER(_PrefillState), C.c_void_p, C.c_size_t, C.c_size_t,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


def _qm(pair):
    if pair is None:
        return _QMatrix(None, None, 0, 0, 0)
    B, Bs = pair
    int4 = B.dtype == torch.uint8
    H._dev(B, torch.uint8 if int4 else torch.int8); H._dev(Bs, torch.float16)
    return _QMatrix(B.data_ptr(), Bs.data_ptr(), B.shape[0], B.shape[1] * (2 if int4 else 1), int(int4))


def _vp(t):
    return None if t is None else t.data_ptr()


def describe_model(weights, cfg):
    """The model dict of this module as the library's cactus_hip_decoder_weights (the dict must outlive it)."""
    Ls = weights["layers"]
    blocks = (_Block * len(Ls))()
    for i, L in enumerate(Ls):
        conv = "q" not in L
        blocks[i] = _Block(
            int(conv), _qm(L.get("q")), _qm(L.get("k")), _qm(L.get("v")), _qm(L.get("o")), _qm(L.get("in_proj")),
            _qm(L.get("out_proj")), _vp(L.get("conv_w")), _qm(L["gate"]), _qm(L["up"]), _qm(L["down"]),
            _vp(L["in_norm"]), _vp(L["post_norm"]), _vp(L.get("q_norm")), _vp(L.get("k_norm")))
    w = _Weights(cfg["hidden"], cfg["q_heads"], cfg["kv_heads"], cfg["head_dim"], cfg["ffn"], cfg["vocab"], len(Ls),
                 cfg.get("conv_taps", 0), cfg.get("conv_rows", 0), cfg["eps"], cfg["theta"], blocks, _qm(weights["emb"]),
                 _qm(weights.get("head")), _vp(weights["final_norm"]))
    w._keep = (blocks, weights)
    return w


def decode_step_split(min_batch):
    """Smallest batch a decode step runs as two half-batches on two streams (0: never); returns the previous value."""
    return int(_lib.cactus_hip_decode_step_split(int(min_batch)))


class _DecodeStep:
    """One decode step of a decoder model for `batch` independent sequences through cactus_hip_decode_step_dev:
    embedding lookup of each sequence's last token, every block (attention over its INT8 cache plus the new FP16 token,
    the new token's K/V quantized into the cache; or LFM2's short convolution over its carried rows), the SwiGLU MLP,
    final norm, lm_head, greedy sampler; token ids, positions, caches and convolution rows stay on the device."""

    def __init__(self, weights, cfg, batch, slots, window=0, sink=4, max_position=None, cache_gen=None):
        c = self.cfg = cfg
        self.w, self.batch = weights, batch
        Ls = weights["layers"]
        if batch <= 16:        # the streaming kernels take these batches: give them the tile-major layout
            pack_model_weights(weights)
        self.attn_index = {}
        for li, L in enumerate(Ls):
            if "q" in L:
                self.attn_index[li] = len(self.attn_index)
        self.kv = _KvState(len(self.attn_index), batch, slots, c["kv_heads"], c["head_dim"], window, sink, cache_gen)
        self.max_position = max_position if max_position is not None else slots
        H.rope_reserve(c["head_dim"], c["theta"], self.max_position + 1)
        conv_layers = [li for li, L in enumerate(Ls) if "q" not in L]
        self._conv = None
        self.state = {}
        if conv_layers:
            self._conv = torch.zeros((len(conv_layers), batch, c["conv_rows"], c["hidden"]), dtype=torch.float16, device="cuda")
            self.state = {li: self._conv[i] for i, li in enumerate(conv_layers)}
        self.logits = _half(batch, c["vocab"])
        self._desc = describe_model(weights, cfg)
        kv = self.kv
        self._state = _DecodeState(batch, slots, window, kv.sink, self.max_position, _vp(kv.kc), _vp(kv.vc), _vp(kv.ksc),
                                   _vp(kv.vsc), _vp(kv.positions), _vp(kv.cache_lens), _vp(kv.tok), _vp(self._conv))
        nbytes = _lib.cactus_hip_decode_step_workspace_bytes(C.byref(self._desc), batch)
        self._ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")

    def adopt(self, prefill):
        """Carry on from a prefill of one sequence: the decode state points at the prefill's caches and convolution rows
        (same layout at batch 1 with slots = the prefill's capacity), position = the prompt length, token = its choice."""
        kv = self.kv
        assert self.batch == 1 and not kv.window and kv.slots == prefill.capacity
        kv.kc, kv.vc, kv.ksc, kv.vsc = prefill.kc, prefill.vc, prefill.ksc, prefill.vsc
        if self._conv is not None:
            self._conv = prefill.conv
            self.state = {li: self._conv[i] for i, li in enumerate(sorted(self.state))}
        st = self._state
        st.k_cache, st.v_cache, st.k_scales, st.v_scales = _vp(kv.kc), _vp(kv.vc), _vp(kv.ksc), _vp(kv.vsc)
        st.conv_state = _vp(self._conv)
        kv.set_positions([prefill.tokens])
        kv.tok.copy_(prefill.tok)

    def step(self, advance=True):
        self.kv.check_room()
        H._check(_lib.cactus_hip_decode_step_dev(C.byref(self._desc), C.byref(self._state), self._ws.data_ptr(),
                                                 self.logits.data_ptr(), int(bool(advance)), H._stream()))
        if advance:
            self.kv.host_max_position += 1

    def weight_bytes(self):
        """Algorithmic bytes of one step's weight matmuls (weights + scales + activations + outputs)."""
        def one(w, M):
            N = w[0].shape[0]
            K = w[0].shape[1] * (2 if _is4(w) else 1)
            return w[0].numel() + 2 * N * (K // 128) + 2 * M * K + 2 * M * N
        total = one(self.w.get("head", self.w["emb"]), self.batch)
        for L in self.w["layers"]:
            total += sum(one(v, self.batch) for v in L.values() if isinstance(v, tuple))
        return total


class Qwen3Decode(_DecodeStep):
    """The Qwen3 op sequence (model_qwen.cpp:51-129, 143-172; the decode tail of engine_model.cpp:221-253).  Batches of
    at most 4 run every matmul as a streaming launch that also produces its own input rows (norm / activation in the
    prologue); larger ones quantize rows in the producing launch and use the tile kernel, q/k/v and gate/up in shared
    launches."""


class Prefill:
    """A chunked prefill of one sequence, the way the engine walks a long prompt (Model::prefill, engine_model.cpp:
    170-219: chunks of `chunk` tokens, the first against nothing, the others against the INT8 cache), through
    cactus_hip_prefill_chunk_dev: per block norm + activation quantization, then either q/k/v GEMM, q/k norm + RoPE +
    cache append, attention (FP16 for chunk 0, INT8-cache hybrid after), o GEMM, or LFM2's in-projection, short
    convolution over the rows the previous chunk left, out-projection; add + norm, gate/up GEMM, SiLU x up, down GEMM,
    add; after the last chunk the final norm, the lm_head on the last token and the greedy sampler.  No window: the
    cache holds the whole prompt.  kc / vc / ksc / vsc: [attention_blocks, 1, tokens, kv_heads, ...]; conv: half
    [conv_blocks, 1, conv_rows, hidden] or None."""

    def __init__(self, weights, cfg, tokens, chunk, capacity=None):
        c = self.cfg = cfg
        self.w, self.tokens, self.chunk, self.nchunks = weights, tokens, chunk, (tokens + chunk - 1) // chunk   # the last may be short
        Hkv, D, V = c["kv_heads"], c["head_dim"], c["vocab"]
        NA = sum("q" in L for L in weights["layers"])
        NC = len(weights["layers"]) - NA
        G = (D + 127) // 128
        cap = self.capacity = capacity or tokens        # cache slots: room for the tokens a decode adds afterwards
        assert cap >= tokens
        H.rope_reserve(D, c["theta"], cap + 1)
        self._toks_all = torch.zeros((self.nchunks * chunk,), dtype=torch.int32, device="cuda")
        self.toks = self._toks_all.view(self.nchunks, chunk)       # row ci: chunk ci's ids (a short last chunk: its head)
        self.prompt = self._toks_all[:tokens]
        self.kc, self.vc = _i8(NA, 1, cap, Hkv, D), _i8(NA, 1, cap, Hkv, D)
        self.ksc, self.vsc = _f32(NA, 1, cap, Hkv, G), _f32(NA, 1, cap, Hkv, G)
        self.conv = torch.zeros((NC, 1, c["conv_rows"], c["hidden"]), dtype=torch.float16, device="cuda") if NC else None
        self.logits, self.tok = _half(1, V), torch.empty((1,), dtype=torch.int32, device="cuda")
        self.slots = [torch.full((1,), ci * chunk, dtype=torch.int32, device="cuda") for ci in range(self.nchunks)]
        self._desc = describe_model(weights, cfg)
        self._state = _PrefillState(cap, _vp(self.kc), _vp(self.vc), _vp(self.ksc), _vp(self.vsc), _vp(self.conv))
        nbytes = _lib.cactus_hip_prefill_chunk_workspace_bytes(C.byref(self._desc), chunk)
        self._ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")

    def run_chunk(self, ci):
        last = ci + 1 == self.nchunks
        H._check(_lib.cactus_hip_prefill_chunk_dev(C.byref(self._desc), C.byref(self._state), self.toks[ci].data_ptr(),
                                                   min(self.chunk, self.tokens - ci * self.chunk), ci * self.chunk, self.slots[ci].data_ptr(), self._ws.data_ptr(),
                                                   int(last), self.logits.data_ptr(), self.tok.data_ptr(), H._stream()))

    def run(self, lanes=4):
        """The whole prompt in one call (cactus_hip_prefill_dev): up to `lanes` chunks in flight on streams of the
        library's own, each chunk's attention in a block waiting for the previous chunk's rows in that block's cache."""
        if getattr(self, "_lanes", 0) < lanes:
            nbytes = _lib.cactus_hip_prefill_workspace_bytes(C.byref(self._desc), self.chunk, lanes)
            self._ws_all = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
            self._starts = torch.arange(self.nchunks, dtype=torch.int32, device="cuda") * self.chunk
            self._lanes = lanes
        H._check(_lib.cactus_hip_prefill_dev(C.byref(self._desc), C.byref(self._state), self.toks.data_ptr(), self.tokens,
                                             self.chunk, self._starts.data_ptr(), self._ws_all.data_ptr(), lanes,
                                             self.logits.data_ptr(), self.tok.data_ptr(), H._stream()))


Qwen3Prefill = Prefill
Lfm2Prefill = Prefill


def lfm2_synthetic_weights(gen, batch, cfg=LFM2_1P2B, mixed=False):
    """mixed: the converter's MIXED policy (python/src/precision_config.py:52-124): packed INT4 for attn_v, attn_output
    and the MLP outside the first and last two blocks; the convolution blocks' projections match no tolerant pattern
    and stay INT8, as do attn_q / attn_k and the embedding."""
    c = cfg
    Hd, Hq, Hkv, D, F, V = c["hidden"], c["q_heads"], c["kv_heads"], c["head_dim"], c["ffn"], c["vocab"]
    NL = c["layers"]
    layers = []
    for li in range(NL):
        i4 = lambda name: mixed and mixed_policy_int4(name, li, NL)
        L = dict(gate=synthetic_matrix(gen, F, Hd, i4("gate")), up=synthetic_matrix(gen, F, Hd, i4("up")),
                 down=synthetic_matrix(gen, Hd, F, i4("down")), in_norm=_norm_weight(gen, Hd), post_norm=_norm_weight(gen, Hd))
        if li in c["attention_layers"]:
            L.update(q=synthetic_matrix(gen, Hq * D, Hd), k=synthetic_matrix(gen, Hkv * D, Hd),
                     v=synthetic_matrix(gen, Hkv * D, Hd, i4("v")), o=synthetic_matrix(gen, Hd, Hq * D, i4("o")),
                     q_norm=_norm_weight(gen, D), k_norm=_norm_weight(gen, D))
        else:
            L.update(in_proj=synthetic_matrix(gen, 3 * Hd, Hd), out_proj=synthetic_matrix(gen, Hd, Hd),
                     conv_w=(torch.rand((Hd, c["conv_taps"]), device="cuda", generator=gen) - 0.5).half())
        layers.append(L)
    return {"layers": layers, "emb": synthetic_matrix(gen, V, Hd), "final_norm": _norm_weight(gen, Hd)}


class Lfm2Decode(_DecodeStep):
    """The LFM2 op sequence (model_lfm2.cpp:123-331): blocks with the short convolution (in-projection to 3 x hidden;
    B*X, causal conv over the carried rows, C-gate, state update and the out-projection's row quantization in one
    launch; out-projection) or with attention (q/k/v, per-head q/k norms, RoPE, INT8 cache append, attention over the
    cache, o), each followed by residual add, norm, gate/up, SiLU x up, down, residual add; final norm, lm_head, greedy
    sampler; embedding lookup of the previous token in front.  INT8 weights.  self.state[li]: the carried rows
    half [batch, conv_rows, hidden] of conv block li."""


class Qwen3GraphModel:
    """One sequence driven node by node through the graph surface, as the reference's engine does:
    the Qwen forward pass (models/model_qwen.cpp:51-172) rebuilt on a CactusGraph every step, the INT8
    KVCache fed from the k / v nodes (engine_cache.cpp:186-266), greedy sampling as a SAMPLE node, soft
    reset between steps (Model::prefill / Model::decode, engine_model.cpp:170-253).

    weights: the host model (numpy arrays, the dict layout of this module); they are bound once as
    external inputs and stay resident on the device across soft resets."""

    def __init__(self, weights, cfg, max_seq=2048, fusion=True):
        from . import graph as G
        import numpy as np
        self.G, self.np, self.cfg = G, np, cfg
        self.g = G.CactusGraph()
        self.g.set_fusion(fusion)
        self.cache = G.KVCache(cfg["layers"], max_seq, cfg["kv_heads"], cfg["head_dim"])
        self.pos = 0

        def weight(pair):
            B, Bs = pair
            K = B.shape[1] * (2 if B.dtype == np.uint8 else 1)
            n = self.g.input((B.shape[0], K), G.INT8)
            if B.dtype == np.uint8:
                self.g.set_packed_int4(n, B)
            else:
                self.g.set_external_input(n, B)
            self.g.set_grouped_scales(n, 128, K // 128, Bs)
            return n

        def vec(a):
            n = self.g.input(a.shape, G.FP16)
            self.g.set_external_input(n, a)
            return n
        self.wn = {"emb": weight(weights["emb"]), "final_norm": vec(weights["final_norm"]),
                   "layers": [{k: (weight(v) if isinstance(v, tuple) else vec(v)) for k, v in L.items()}
                              for L in weights["layers"]]}

    def forward(self, tokens):
        """Adds the nodes of one forward pass over `tokens` at the current position; returns (logits node,
        k nodes, v nodes).  Nothing runs until execute()."""
        g, G, c, np = self.g, self.G, self.cfg, self.np
        S, Hq, Hkv, D, Hd = len(tokens), c["q_heads"], c["kv_heads"], c["head_dim"], c["hidden"]
        idx = g.input((S,), G.FP32)
        g.set_input(idx, np.asarray(tokens, np.float32))
        h = g.embedding(self.wn["emb"], idx)
        ks, vs = [], []
        scale = 1.0 / float(np.sqrt(D))
        for li, L in enumerate(self.wn["layers"]):
            n1 = g.rms_norm(h, L["in_norm"], c["eps"])
            q, k, v = (g.matmul(n1, L[name], True) for name in ("q", "k", "v"))
            q = g.reshape(g.rms_norm(g.reshape(q, (S * Hq, D)), L["q_norm"], c["eps"]), (1, S, Hq, D))
            k = g.reshape(g.rms_norm(g.reshape(k, (S * Hkv, D)), L["k_norm"], c["eps"]), (1, S, Hkv, D))
            v = g.reshape(v, (1, S, Hkv, D))
            q, k = g.rope(q, c["theta"], self.pos), g.rope(k, c["theta"], self.pos)
            ks.append(k)
            vs.append(v)
            if self.cache.is_empty():
                attn = g.attention(q, k, v, scale, self.pos)
            else:
                attn = g.attention_int8_hybrid(q, k, v, scale, self.pos, self.cache.get_keys_int8(li),
                                               self.cache.get_values_int8(li), self.cache.get_key_scales(li),
                                               self.cache.get_value_scales(li), self.cache.get_effective_seq_len(),
                                               Hkv, D)
            h = g.add(h, g.matmul(g.reshape(attn, (S, Hq * D)), L["o"], True))
            n2 = g.rms_norm(h, L["post_norm"], c["eps"])
            act = g.multiply(g.silu(g.matmul(n2, L["gate"], True)), g.matmul(n2, L["up"], True))
            h = g.add(h, g.matmul(act, L["down"], True))
        last = g.reshape(g.index(g.rms_norm(h, self.wn["final_norm"], c["eps"]), S - 1, 0), (1, Hd))   # engine_model.cpp:241-246
        return g.matmul(last, self.wn["emb"], True), ks, vs

    def step(self, tokens):
        """Prefill (several tokens) or decode (one): runs the pass, appends to the cache, returns the greedy token."""
        logits, ks, vs = self.forward(tokens)
        tok = self.g.sample(logits, 0.0, 0.0, 0)          # temperature 0, no top-p / top-k: greedy (kernel_nn.cpp:451)
        self.g.execute()
        out = int(self.g.get_output(tok).view(self.np.uint32)[0])
        self.cache.update_from_graph(self.g, ks, vs, len(tokens))
        self.g.soft_reset()
        self.pos += len(tokens)
        return out

    def close(self):
        self.g.close()
        self.cache.close()
