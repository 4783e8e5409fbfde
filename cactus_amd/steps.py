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


class Qwen3Decode:
    """One decode step of the Qwen3 op sequence (model_qwen.cpp:51-129, 143-172; the decode tail of
    engine_model.cpp:221-253) for `batch` independent sequences: embedding lookup of each sequence's last
    token (the table is the tied lm_head weight), per block input norm, q/k/v, per-head q/k norms + RoPE at
    each sequence's own position, attention over its INT8 cache plus the new FP16 token, the new token's
    K/V quantized into the cache, o projection, residual add, post-attention norm, gate/up, SiLU x up, down,
    residual add; final norm, lm_head, greedy sampler.  Batches of at most 4 run every matmul as a
    streaming launch that also produces its own input rows (norm / activation in the prologue); larger
    ones quantize rows in the producing launch and use the tile kernel, q/k/v and gate/up in shared
    launches."""

    def __init__(self, weights, cfg, batch, slots, window=0, sink=4, max_position=None, cache_gen=None):
        c = self.cfg = cfg
        self.w, self.batch = weights, batch
        Hd, Hq, Hkv, D, F, V = c["hidden"], c["q_heads"], c["kv_heads"], c["head_dim"], c["ffn"], c["vocab"]
        NL = len(weights["layers"])
        self.kv = _KvState(NL, batch, slots, Hkv, D, window, sink, cache_gen)
        self.max_position = max_position if max_position is not None else slots
        H.rope_reserve(D, c["theta"], self.max_position + 1)
        b = batch
        self.h0, self.hid, self.x, self.h2 = _half(b, Hd), _half(b, Hd), _half(b, Hd), _half(b, Hd)
        self.qo, self.ko, self.vo = _half(b, Hq * D), _half(b, Hkv * D), _half(b, Hkv * D)
        self.qr, self.kr, self.ao = _half(b, Hq * D), _half(b, Hkv * D), _half(b, 1, Hq, D)
        self.oo, self.go, self.uo, self.do = _half(b, Hd), _half(b, F), _half(b, F), _half(b, Hd)
        self.logits = _half(b, V)
        self.xq, self.xs, self.xq2, self.xs2 = _i8(b, Hd), _f32(b), _i8(b, Hd), _f32(b)
        self.aq, self.asc = _i8(b, F), _f32(b)
        self.oq, self.osc = _i8(b, Hq * D), _f32(b)      # attention output rows, quantized for the output projection
        self.small = batch <= 4
        if batch <= 16:        # the streaming kernels take these batches: give them the tile-major layout
            pack_model_weights(weights)
        Ls = weights["layers"]
        head = weights.get("head", weights["emb"])     # tied embeddings unless the model brings an lm_head of its own

        def groups(entries, cls):
            """entries [(weight pair, out)] split by precision: one launch object per precision."""
            out = []
            for want4 in (False, True):
                sel = [(w[0], w[1], o) for w, o in entries if _is4(w) == want4]
                if sel:
                    out.append(cls(sel, 128, int4=want4))
            return out
        self.qkv, self.gu, self.down = [], [], []
        cls = H.MatmulProduced if self.small else H.MatmulShared
        for L in Ls:
            self.qkv.append(groups([(L["q"], self.qo), (L["k"], self.ko), (L["v"], self.vo)], cls))
            self.gu.append(groups([(L["gate"], self.go), (L["up"], self.uo)], cls))
            if self.small:
                self.down.append(groups([(L["down"], self.do)], cls))
        if self.small:
            self.head_p = H.MatmulProduced([(head[0], head[1], self.logits)], 128, int4=_is4(head))

    def step(self, advance=True):
        c, kv, b = self.cfg, self.kv, self.batch
        kv.check_room()
        Hq, Hkv, D, eps, theta = c["q_heads"], c["kv_heads"], c["head_dim"], c["eps"], c["theta"]
        Ls, emb = self.w["layers"], self.w["emb"]
        head = self.w.get("head", emb)
        NL = len(Ls)
        H.embedding(emb[0], kv.tok, emb[1], 128, out=self.h0, check=False)
        if not self.small:
            H.add_rms_norm_quant(self.h0, None, Ls[0]["in_norm"], eps, out_q=self.xq, out_scales=self.xs)
        res = self.h0
        q4, k4, v4 = self.qo.view(b, 1, Hq, D), self.ko.view(b, 1, Hkv, D), self.vo.view(b, 1, Hkv, D)
        qr4, kr4 = self.qr.view(b, 1, Hq, D), self.kr.view(b, 1, Hkv, D)
        for li, L in enumerate(Ls):
            if self.small:
                # residual add of the previous block + this block's input norm + q/k/v, one launch per precision
                for mp in self.qkv[li]:
                    if li == 0:
                        mp.run_norm(self.h0, None, L["in_norm"], eps)
                    else:
                        mp.run_norm(self.h2, self.do, L["in_norm"], eps, self.hid)
                if li:
                    res = self.hid
            else:
                for sh in self.qkv[li]:
                    sh.run(self.xq, self.xs)       # rows quantized by the norm launch that produced them
            # q/k norm + RoPE at each sequence's position; without a window the new token's K and V join the
            # cache in the same launch (the slot beyond the cached length: this step's attention reads them
            # as FP16), with a window they are appended after the attention, which must still see the token
            # a full window is about to drop
            # Without a window all of it - norm, RoPE, cache append, attention - is one launch (head_dim 64 / 128 / 256).
            fused_append = kv.window == 0
            # batches above 4 rows: the attention launch also leaves its rows quantized for the output projection
            o_quant = fused_append and D == 128 and not self.small   # (head_dim 64: the wider workgroups cost more than the launch)
            if fused_append and D in (64, 128, 256):
                H.attention_decode_fused(self.qo.view(b, Hq, D), self.ko.view(b, Hkv, D), self.vo.view(b, Hkv, D),
                                         L["q_norm"], L["k_norm"], kv.kc[li], kv.vc[li], kv.ksc[li], kv.vsc[li],
                                         kv.cache_lens, kv.positions, self.max_position, eps, theta, out=self.ao,
                                         out_q=self.oq if o_quant else None, out_scales=self.osc if o_quant else None)
            else:
                H.qk_norm_rope_seq(q4, k4, v4 if fused_append else None, L["q_norm"], L["k_norm"], eps, theta,
                                   kv.positions, self.max_position, kv.kc[li], kv.vc[li], kv.ksc[li], kv.vsc[li], 0, 0,
                                   128, out_q=qr4, out_k=kr4)
                H.attention_hybrid_batched(qr4, kv.kc[li], kv.vc[li], kv.ksc[li], kv.vsc[li], kr4, v4, 0.0,
                                           kv.cache_lens, kv.positions, out=self.ao)
            if not fused_append:
                H.kv_append_windowed(kr4, v4, kv.kc[li], kv.vc[li], kv.ksc[li], kv.vsc[li], kv.positions, kv.window,
                                     kv.sink)
            if o_quant:
                (H.matmul_int4 if _is4(L["o"]) else H.matmul_int8)(self.oq, self.osc, L["o"][0], L["o"][1], 128, out=self.oo)
            else:
                H.matmul_f16act(self.ao.view(b, Hq * D), L["o"][0], L["o"][1], 128, int4=_is4(L["o"]), out=self.oo)
            nxt = Ls[li + 1]["in_norm"] if li + 1 < NL else self.w["final_norm"]
            if self.small:
                for mp in self.gu[li]:
                    mp.run_norm(res, self.oo, L["post_norm"], eps, self.h2)     # add + norm + gate/up
                self.down[li][0].run_act(self.go, self.uo, "silu")              # silu x up + down
            else:   # norm / activation launches that also quantize the next matmul's input rows
                H.add_rms_norm_quant(res, self.oo, L["post_norm"], eps, out_sum=self.h2, out_q=self.xq2,
                                     out_scales=self.xs2)
                for sh in self.gu[li]:
                    sh.run(self.xq2, self.xs2)
                H.silu_multiply_quant(self.go, self.uo, out_q=self.aq, out_scales=self.asc)
                (H.matmul_int4 if _is4(L["down"]) else H.matmul_int8)(self.aq, self.asc, L["down"][0], L["down"][1], 128,
                                                                      out=self.do)
                last = li + 1 == NL   # the lm_head below takes the FP16 row
                H.add_rms_norm_quant(self.h2, self.do, nxt, eps, out_sum=self.hid, out_norm=self.x if last else None,
                                     out_q=self.xq, out_scales=self.xs)
                res = self.hid
        if self.small:   # last residual add + final norm + lm_head
            self.head_p.run_norm(self.h2, self.do, self.w["final_norm"], eps, self.hid)
        else:
            H.matmul_f16act(self.x, head[0], head[1], 128, int4=_is4(head), out=self.logits)
        H.sample_greedy(self.logits, out=kv.tok)
        if advance:
            kv.advance(1)

    def weight_bytes(self):
        """Algorithmic bytes of one step's weight matmuls (weights + scales + activations + outputs)."""
        def one(w, M):
            N = w[0].shape[0]
            K = w[0].shape[1] * (2 if _is4(w) else 1)
            return w[0].numel() + 2 * N * (K // 128) + 2 * M * K + 2 * M * N
        total = one(self.w["emb"], self.batch)
        for L in self.w["layers"]:
            total += sum(one(L[n], self.batch) for n in ("q", "k", "v", "o", "gate", "up", "down"))
        return total


class Qwen3Prefill:
    """A chunked prefill of one sequence through the Qwen3 op sequence, the way the engine walks a long
    prompt (Model::prefill, engine_model.cpp:170-219: chunks of `chunk` tokens, the first against nothing,
    the others against the INT8 cache): per block norm + activation quantization, q/k/v GEMM, q/k norm +
    RoPE + cache append, attention (FP16 for chunk 0, INT8-cache hybrid after), o GEMM, add + norm, gate/up
    GEMM, SiLU x up, down GEMM, add; after the last chunk the final norm, the lm_head on the last token and
    the greedy sampler.  No window: the cache holds the whole prompt."""

    def __init__(self, weights, cfg, tokens, chunk):
        c = self.cfg = cfg
        assert tokens % chunk == 0
        self.w, self.tokens, self.chunk, self.nchunks = weights, tokens, chunk, tokens // chunk
        Hd, Hq, Hkv, D, F, V = c["hidden"], c["q_heads"], c["kv_heads"], c["head_dim"], c["ffn"], c["vocab"]
        NL = len(weights["layers"])
        M = chunk
        H.rope_reserve(D, c["theta"], tokens + 1)
        self.toks = torch.zeros((self.nchunks, M), dtype=torch.int32, device="cuda")
        self.kc, self.vc = _i8(NL, 1, tokens, Hkv, D), _i8(NL, 1, tokens, Hkv, D)
        self.ksc, self.vsc = _f32(NL, 1, tokens, Hkv, 1), _f32(NL, 1, tokens, Hkv, 1)
        self.h0, self.hid, self.h2, self.x = _half(M, Hd), _half(M, Hd), _half(M, Hd), _half(M, Hd)
        self.qo, self.ko, self.vo = _half(M, Hq * D), _half(M, Hkv * D), _half(M, Hkv * D)
        self.qr, self.kr = _half(M, Hq * D), _half(M, Hkv * D)
        self.ao, self.oo, self.go, self.uo, self.do = _half(1, M, Hq, D), _half(M, Hd), _half(M, F), _half(M, F), _half(M, Hd)
        self.logits, self.tok = _half(1, V), torch.empty((1,), dtype=torch.int32, device="cuda")
        self.xq, self.xs, self.xq2, self.xs2 = _i8(M, Hd), _f32(M), _i8(M, Hd), _f32(M)
        self.aq, self.asc, self.oq, self.os_ = _i8(M, F), _f32(M), _i8(M, Hq * D), _f32(M)

        def groups(entries):
            out = []
            for want4 in (False, True):
                sel = [(w[0], w[1], o) for w, o in entries if _is4(w) == want4]
                if sel:
                    out.append(H.MatmulShared(sel, 128, int4=want4))
            return out
        self.qkv = [groups([(L["q"], self.qo), (L["k"], self.ko), (L["v"], self.vo)]) for L in weights["layers"]]
        self.gu = [groups([(L["gate"], self.go), (L["up"], self.uo)]) for L in weights["layers"]]
        self.slots = [torch.full((1,), ci * chunk, dtype=torch.int32, device="cuda") for ci in range(self.nchunks)]

    def run_chunk(self, ci):
        c, M = self.cfg, self.chunk
        Hq, Hkv, D, eps, theta = c["q_heads"], c["kv_heads"], c["head_dim"], c["eps"], c["theta"]
        Ls, head = self.w["layers"], self.w["emb"]
        NL = len(Ls)
        cache = ci * M
        H.embedding(head[0], self.toks[ci], head[1], 128, out=self.h0, check=False)
        H.add_rms_norm_quant(self.h0, None, Ls[0]["in_norm"], eps, out_q=self.xq, out_scales=self.xs)
        res = self.h0
        q4, k4, v4 = self.qo.view(1, M, Hq, D), self.ko.view(1, M, Hkv, D), self.vo.view(1, M, Hkv, D)
        qr4, kr4 = self.qr.view(1, M, Hq, D), self.kr.view(1, M, Hkv, D)
        for li, L in enumerate(Ls):
            for sh in self.qkv[li]:
                sh.run(self.xq, self.xs)
            H.qk_norm_rope_kv_append(q4, k4, v4, L["q_norm"], L["k_norm"], eps, theta, cache, self.kc[li], self.vc[li],
                                     self.ksc[li], self.vsc[li], self.slots[ci], 128, out_q=qr4, out_k=kr4)
            if cache == 0:
                a = H.attention_f16(qr4, kr4, v4, 0.0, out=self.ao)
            else:
                a = H.attention_hybrid(qr4, self.kc[li, 0, :cache], self.vc[li, 0, :cache], self.ksc[li, 0, :cache],
                                       self.vsc[li, 0, :cache], kr4, v4, 0.0, position_offset=cache, out=self.ao)
            H.quantize_activations(a.view(M, Hq * D), out_q=self.oq, out_scales=self.os_)
            (H.matmul_int4 if _is4(L["o"]) else H.matmul_int8)(self.oq, self.os_, L["o"][0], L["o"][1], 128, out=self.oo)
            H.add_rms_norm_quant(res, self.oo, L["post_norm"], eps, out_sum=self.h2, out_q=self.xq2, out_scales=self.xs2)
            for sh in self.gu[li]:
                sh.run(self.xq2, self.xs2)
            H.silu_multiply_quant(self.go, self.uo, out_q=self.aq, out_scales=self.asc)
            (H.matmul_int4 if _is4(L["down"]) else H.matmul_int8)(self.aq, self.asc, L["down"][0], L["down"][1], 128,
                                                                  out=self.do)
            nxt = Ls[li + 1]["in_norm"] if li + 1 < NL else self.w["final_norm"]
            last = li + 1 == NL and ci + 1 == self.nchunks
            H.add_rms_norm_quant(self.h2, self.do, nxt, eps, out_sum=self.hid, out_norm=self.x if last else None,
                                 out_q=self.xq, out_scales=self.xs)
            res = self.hid
        if ci + 1 == self.nchunks:
            H.matmul_f16act(self.x[M - 1:M], head[0], head[1], 128, int4=_is4(head), out=self.logits)
            H.sample_greedy(self.logits, out=self.tok)


def lfm2_synthetic_weights(gen, batch, cfg=LFM2_1P2B):
    c = cfg
    Hd, Hq, Hkv, D, F, V = c["hidden"], c["q_heads"], c["kv_heads"], c["head_dim"], c["ffn"], c["vocab"]
    layers = []
    for li in range(c["layers"]):
        L = dict(gate=synthetic_matrix(gen, F, Hd), up=synthetic_matrix(gen, F, Hd), down=synthetic_matrix(gen, Hd, F),
                 in_norm=_norm_weight(gen, Hd), post_norm=_norm_weight(gen, Hd))
        if li in c["attention_layers"]:
            L.update(q=synthetic_matrix(gen, Hq * D, Hd), k=synthetic_matrix(gen, Hkv * D, Hd),
                     v=synthetic_matrix(gen, Hkv * D, Hd), o=synthetic_matrix(gen, Hd, Hq * D),
                     q_norm=_norm_weight(gen, D), k_norm=_norm_weight(gen, D))
        else:
            L.update(in_proj=synthetic_matrix(gen, 3 * Hd, Hd), out_proj=synthetic_matrix(gen, Hd, Hd),
                     conv_w=(torch.rand((Hd, c["conv_taps"]), device="cuda", generator=gen) - 0.5).half())
        layers.append(L)
    return {"layers": layers, "emb": synthetic_matrix(gen, V, Hd), "final_norm": _norm_weight(gen, Hd)}


class Lfm2Decode:
    """One decode step of the LFM2 op sequence (model_lfm2.cpp:123-331) for `batch` independent sequences:
    blocks with the short convolution (in-projection to 3 x hidden; B*X, causal conv over the carried rows,
    C-gate, state update and the out-projection's row quantization in one launch; out-projection) or with
    attention (q/k/v, per-head q/k norms, RoPE, INT8 cache append, attention over the cache, o), each
    followed by residual add, norm, gate/up, SiLU x up, down, residual add; final norm, lm_head, greedy
    sampler; embedding lookup of the previous token in front.  INT8 weights."""

    def __init__(self, weights, cfg, batch, slots, window=0, sink=4, max_position=None, cache_gen=None):
        c = self.cfg = cfg
        self.w, self.batch = weights, batch
        Hd, Hq, Hkv, D, F, V = c["hidden"], c["q_heads"], c["kv_heads"], c["head_dim"], c["ffn"], c["vocab"]
        Ls = weights["layers"]
        if batch <= 16:
            pack_model_weights(weights)
        self.attn_index = {}
        for li, L in enumerate(Ls):
            if "q" in L:
                self.attn_index[li] = len(self.attn_index)
        self.kv = _KvState(len(self.attn_index), batch, slots, Hkv, D, window, sink, cache_gen)
        self.max_position = max_position if max_position is not None else slots
        H.rope_reserve(D, c["theta"], self.max_position + 1)
        b = batch
        self.state = {li: torch.zeros((b, c["conv_rows"], Hd), dtype=torch.float16, device="cuda")
                      for li, L in enumerate(Ls) if "q" not in L}
        self.h0, self.hid, self.x, self.h2 = _half(b, Hd), _half(b, Hd), _half(b, Hd), _half(b, Hd)
        self.qo, self.ko, self.vo = _half(b, Hq * D), _half(b, Hkv * D), _half(b, Hkv * D)
        self.qr, self.kr, self.ao = _half(b, Hq * D), _half(b, Hkv * D), _half(b, 1, Hq, D)
        self.ipj, self.oo, self.go, self.uo, self.do = _half(b, 3 * Hd), _half(b, Hd), _half(b, F), _half(b, F), _half(b, Hd)
        self.logits = _half(b, V)
        self.xq, self.xs, self.xq2, self.xs2 = _i8(b, Hd), _f32(b), _i8(b, Hd), _f32(b)
        self.aq, self.asc, self.cq, self.cs = _i8(b, F), _f32(b), _i8(b, Hd), _f32(b)
        self.qkv = {li: H.MatmulShared([(L["q"][0], L["q"][1], self.qo), (L["k"][0], L["k"][1], self.ko),
                                        (L["v"][0], L["v"][1], self.vo)], 128) for li, L in enumerate(Ls) if "q" in L}
        self.gu = [H.MatmulShared([(L["gate"][0], L["gate"][1], self.go), (L["up"][0], L["up"][1], self.uo)], 128)
                   for L in Ls]

    def step(self, advance=True):
        c, kv, b = self.cfg, self.kv, self.batch
        kv.check_room()
        Hd, Hq, Hkv, D, eps, theta = c["hidden"], c["q_heads"], c["kv_heads"], c["head_dim"], c["eps"], c["theta"]
        Ls, head = self.w["layers"], self.w["emb"]
        NL = len(Ls)
        H.embedding(head[0], kv.tok, head[1], 128, out=self.h0, check=False)
        H.add_rms_norm_quant(self.h0, None, Ls[0]["in_norm"], eps, out_q=self.xq, out_scales=self.xs)
        res = self.h0
        q4, k4, v4 = self.qo.view(b, 1, Hq, D), self.ko.view(b, 1, Hkv, D), self.vo.view(b, 1, Hkv, D)
        qr4, kr4 = self.qr.view(b, 1, Hq, D), self.kr.view(b, 1, Hkv, D)
        for li, L in enumerate(Ls):
            if "q" in L:
                ai = self.attn_index[li]
                self.qkv[li].run(self.xq, self.xs)
                fused_append = kv.window == 0
                if fused_append and D in (64, 128, 256):     # norm, RoPE, cache append and attention in one launch
                    H.attention_decode_fused(self.qo.view(b, Hq, D), self.ko.view(b, Hkv, D), self.vo.view(b, Hkv, D),
                                             L["q_norm"], L["k_norm"], kv.kc[ai], kv.vc[ai], kv.ksc[ai], kv.vsc[ai],
                                             kv.cache_lens, kv.positions, self.max_position, eps, theta, out=self.ao)
                else:
                    H.qk_norm_rope_seq(q4, k4, v4 if fused_append else None, L["q_norm"], L["k_norm"], eps, theta,
                                       kv.positions, self.max_position, kv.kc[ai], kv.vc[ai], kv.ksc[ai], kv.vsc[ai], 0,
                                       0, 128, out_q=qr4, out_k=kr4)
                    H.attention_hybrid_batched(qr4, kv.kc[ai], kv.vc[ai], kv.ksc[ai], kv.vsc[ai], kr4, v4, 0.0,
                                               kv.cache_lens, kv.positions, out=self.ao)
                if not fused_append:
                    H.kv_append_windowed(kr4, v4, kv.kc[ai], kv.vc[ai], kv.ksc[ai], kv.vsc[ai], kv.positions, kv.window,
                                         kv.sink)
                H.matmul_f16act(self.ao.view(b, Hq * D), L["o"][0], L["o"][1], 128, out=self.oo)
            else:
                H.matmul_int8(self.xq, self.xs, L["in_proj"][0], L["in_proj"][1], 128, out=self.ipj)
                H.lfm2_conv_gate_quant(self.ipj.view(b, 1, 3 * Hd), L["conv_w"], self.state[li], out_q=self.cq,
                                       out_scales=self.cs)
                H.matmul_int8(self.cq, self.cs, L["out_proj"][0], L["out_proj"][1], 128, out=self.oo)
            H.add_rms_norm_quant(res, self.oo, L["post_norm"], eps, out_sum=self.h2, out_q=self.xq2, out_scales=self.xs2)
            self.gu[li].run(self.xq2, self.xs2)
            H.silu_multiply_quant(self.go, self.uo, out_q=self.aq, out_scales=self.asc)
            H.matmul_int8(self.aq, self.asc, L["down"][0], L["down"][1], 128, out=self.do)
            last = li + 1 == NL
            nxt = self.w["final_norm"] if last else Ls[li + 1]["in_norm"]
            H.add_rms_norm_quant(self.h2, self.do, nxt, eps, out_sum=self.hid, out_norm=self.x if last else None,
                                 out_q=self.xq, out_scales=self.xs)
            res = self.hid
        H.matmul_f16act(self.x, head[0], head[1], 128, out=self.logits)
        H.sample_greedy(self.logits, out=kv.tok)
        if advance:
            kv.advance(1)


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
