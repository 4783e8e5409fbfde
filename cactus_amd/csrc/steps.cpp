// Whole decode / prefill steps (include/cactus_hip_steps.h): host code that enqueues the launches of cactus_hip.h in
// the order the reference's models build their graphs (models/model_qwen.cpp:51-172, models/model_lfm2.cpp:123-331;
// Model::decode / Model::prefill, engine_model.cpp:170-297), for batches of independent sequences.
#include "../../include/cactus_hip_steps.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace cactus_hip {
void set_last_error(const char* msg);
int workspace_lane(int lane);
}

namespace {

using half_t = cactus_half_t;

int fail(const char* what) {
    cactus_hip::set_last_error(what);
    return CACTUS_HIP_EINVAL;
}

#define STEP_TRY(expr)                      \
    do {                                    \
        const int rc_ = (expr);             \
        if (rc_ != CACTUS_HIP_OK) return rc_; \
    } while (0)

// carve a workspace into 256-byte aligned pieces; with base == nullptr it only measures
struct Carver {
    unsigned char* base;
    size_t used = 0;
    explicit Carver(void* b) : base(static_cast<unsigned char*>(b)) {}
    template <typename T>
    T* take(size_t count) {
        const size_t at = used;
        used += (count * sizeof(T) + 255) & ~size_t(255);
        return base ? reinterpret_cast<T*>(base + at) : nullptr;
    }
};

struct Out {
    const cactus_hip_qmatrix* w;
    half_t* c;
};

// Several weights against one quantized activation matrix (q/k/v, gate/up): one launch per precision present
int shared_matmuls(const int8_t* A, const float* As, const Out* outs, size_t count, size_t M, size_t K, void* stream) {
    for (int want4 = 0; want4 < 2; want4++) {
        cactus_hip_matmul_weights ws[8];
        size_t n = 0;
        for (size_t i = 0; i < count; i++)
            if ((outs[i].w->int4 != 0) == (want4 != 0)) ws[n++] = cactus_hip_matmul_weights{outs[i].w->B, outs[i].w->scales, outs[i].c, outs[i].w->N};
        if (n == 0) continue;
        STEP_TRY(want4 ? cactus_hip_matmul_int4_shared_dev(A, As, ws, n, M, K, 128, stream)
                       : cactus_hip_matmul_int8_shared_dev(A, As, ws, n, M, K, 128, stream));
    }
    return CACTUS_HIP_OK;
}

// The same with the rows produced in the launch: rms_norm(x [+ addend]) * w (kind 1) or silu(x) * addend (kind 2)
int produced_matmuls(int kind, const half_t* x, const half_t* addend, const half_t* norm_w, half_t* sum_out, float eps,
                     const Out* outs, size_t count, size_t M, size_t K, void* stream) {
    for (int want4 = 0; want4 < 2; want4++) {
        cactus_hip_matmul_weights ws[8];
        size_t n = 0;
        for (size_t i = 0; i < count; i++)
            if ((outs[i].w->int4 != 0) == (want4 != 0)) ws[n++] = cactus_hip_matmul_weights{outs[i].w->B, outs[i].w->scales, outs[i].c, outs[i].w->N};
        if (n == 0) continue;
        if (kind == 1)
            STEP_TRY(want4 ? cactus_hip_matmul_int4_norm_f16act_dev(x, addend, norm_w, sum_out, eps, ws, n, M, K, 128, stream)
                           : cactus_hip_matmul_int8_norm_f16act_dev(x, addend, norm_w, sum_out, eps, ws, n, M, K, 128, stream));
        else
            STEP_TRY(want4 ? cactus_hip_matmul_int4_silu_f16act_dev(x, addend, ws, n, M, K, 128, stream)
                           : cactus_hip_matmul_int8_silu_f16act_dev(x, addend, ws, n, M, K, 128, stream));
    }
    return CACTUS_HIP_OK;
}

int matmul_q(const int8_t* A, const float* As, const cactus_hip_qmatrix& w, half_t* C, size_t M, void* stream) {
    return w.int4 ? cactus_hip_matmul_int4_dev(A, As, static_cast<const uint8_t*>(w.B), w.scales, C, M, w.K, w.N, 128, stream)
                  : cactus_hip_matmul_int8_dev(A, As, static_cast<const int8_t*>(w.B), w.scales, C, M, w.K, w.N, 128, stream);
}

int matmul_f16act(const half_t* X, const cactus_hip_qmatrix& w, half_t* C, size_t M, void* stream) {
    return w.int4 ? cactus_hip_matmul_int4_f16act_dev(X, static_cast<const uint8_t*>(w.B), w.scales, C, M, w.K, w.N, 128, stream)
                  : cactus_hip_matmul_int8_f16act_dev(X, static_cast<const int8_t*>(w.B), w.scales, C, M, w.K, w.N, 128, stream);
}

bool check_matrix(const cactus_hip_qmatrix& m, size_t N, size_t K) {
    return m.B != nullptr && m.scales != nullptr && m.N == N && m.K == K;
}

const char* check_weights(const cactus_hip_decoder_weights* w) {
    if (!w || !w->blocks || w->layers == 0) return "decoder step: null model description";
    if (w->hidden == 0 || w->q_heads == 0 || w->kv_heads == 0 || w->head_dim == 0 || w->ffn == 0 || w->vocab == 0)
        return "decoder step: zero model dimension";
    if (w->q_heads % w->kv_heads != 0) return "decoder step: q heads must be a multiple of kv heads";
    if (w->hidden % 128 != 0 || w->ffn % 128 != 0 || (w->q_heads * w->head_dim) % 128 != 0)
        return "decoder step: hidden, ffn and heads * head_dim must be multiples of the weight group (128)";
    if (!check_matrix(w->embedding, w->vocab, w->hidden) || w->embedding.int4) return "decoder step: embedding must be INT8 [vocab, hidden]";
    if (w->lm_head.B && !check_matrix(w->lm_head, w->vocab, w->hidden)) return "decoder step: lm_head must be [vocab, hidden]";
    if (!w->final_norm) return "decoder step: null final norm";
    const size_t Hd = w->hidden, QD = w->q_heads * w->head_dim, KD = w->kv_heads * w->head_dim;
    for (size_t i = 0; i < w->layers; i++) {
        const cactus_hip_block_weights& b = w->blocks[i];
        if (!b.in_norm || !b.post_norm) return "decoder step: null block norm";
        if (!check_matrix(b.gate, w->ffn, Hd) || !check_matrix(b.up, w->ffn, Hd) || !check_matrix(b.down, Hd, w->ffn))
            return "decoder step: MLP matrices must be gate / up [ffn, hidden], down [hidden, ffn]";
        if (b.kind == 0) {
            if (!check_matrix(b.q, QD, Hd) || !check_matrix(b.k, KD, Hd) || !check_matrix(b.v, KD, Hd) || !check_matrix(b.o, Hd, QD))
                return "decoder step: attention matrices must be q [Hq D, hidden], k / v [Hkv D, hidden], o [hidden, Hq D]";
            if (!b.q_norm || !b.k_norm) return "decoder step: null q / k norm";
        } else if (b.kind == 1) {
            if (!check_matrix(b.in_proj, 3 * Hd, Hd) || !check_matrix(b.out_proj, Hd, Hd) || !b.conv_weight)
                return "decoder step: conv block needs in_proj [3 hidden, hidden], out_proj [hidden, hidden], conv_weight";
            if (b.in_proj.int4 || b.out_proj.int4) return "decoder step: conv block projections must be INT8";
            if (w->conv_taps < 2 || w->conv_rows == 0) return "decoder step: conv_taps / conv_rows missing";
        } else {
            return "decoder step: unknown block kind";
        }
    }
    return nullptr;
}

struct DecodeBuffers {
    half_t *h0, *hid, *x, *h2, *qo, *ko, *vo, *qr, *kr, *ao, *oo, *go, *uo, *dn, *ipj;
    int8_t *xq, *xq2, *aq, *oq, *cq;
    float *xs, *xs2, *as, *os, *cs;
};

size_t carve_decode(const cactus_hip_decoder_weights* w, size_t b, void* base, DecodeBuffers* out) {
    Carver c(base);
    const size_t Hd = w->hidden, QD = w->q_heads * w->head_dim, KD = w->kv_heads * w->head_dim, F = w->ffn;
    DecodeBuffers d;
    d.h0 = c.take<half_t>(b * Hd); d.hid = c.take<half_t>(b * Hd); d.x = c.take<half_t>(b * Hd); d.h2 = c.take<half_t>(b * Hd);
    d.qo = c.take<half_t>(b * QD); d.ko = c.take<half_t>(b * KD); d.vo = c.take<half_t>(b * KD);
    d.qr = c.take<half_t>(b * QD); d.kr = c.take<half_t>(b * KD); d.ao = c.take<half_t>(b * QD);
    d.oo = c.take<half_t>(b * Hd); d.go = c.take<half_t>(b * F); d.uo = c.take<half_t>(b * F); d.dn = c.take<half_t>(b * Hd);
    d.ipj = c.take<half_t>(b * 3 * Hd);
    d.xq = c.take<int8_t>(b * Hd); d.xq2 = c.take<int8_t>(b * Hd); d.aq = c.take<int8_t>(b * F);
    d.oq = c.take<int8_t>(b * QD); d.cq = c.take<int8_t>(b * Hd);
    d.xs = c.take<float>(b); d.xs2 = c.take<float>(b); d.as = c.take<float>(b); d.os = c.take<float>(b); d.cs = c.take<float>(b);
    if (out) *out = d;
    return c.used;
}

// the rows [r0, ...) of every buffer
DecodeBuffers rows_from(const cactus_hip_decoder_weights* w, const DecodeBuffers& d, size_t r0) {
    const size_t Hd = w->hidden, QD = w->q_heads * w->head_dim, KD = w->kv_heads * w->head_dim, F = w->ffn;
    DecodeBuffers v = d;
    v.h0 += r0 * Hd; v.hid += r0 * Hd; v.x += r0 * Hd; v.h2 += r0 * Hd;
    v.qo += r0 * QD; v.ko += r0 * KD; v.vo += r0 * KD; v.qr += r0 * QD; v.kr += r0 * KD; v.ao += r0 * QD;
    v.oo += r0 * Hd; v.go += r0 * F; v.uo += r0 * F; v.dn += r0 * Hd; v.ipj += r0 * 3 * Hd;
    v.xq += r0 * Hd; v.xq2 += r0 * Hd; v.aq += r0 * F; v.oq += r0 * QD; v.cq += r0 * Hd;
    v.xs += r0; v.xs2 += r0; v.as += r0; v.os += r0; v.cs += r0;
    return v;
}

// Two half-batches in step (see cactus_hip_decode_step_dev): the second half's stream and the events that order the
// halves' attention launches, one set per device, made on first use and kept.
constexpr int PF_MAX_LANES = 8;       // chunks of a prompt in flight at a time
struct StepPipeline {
    hipStream_t side = nullptr;                 // the decode step's second half; lane 1 of a prefill
    hipStream_t more[PF_MAX_LANES - 2] = {};    // lanes 2.. of a prefill
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t lane_join[PF_MAX_LANES - 2] = {};
    std::vector<hipEvent_t> a_done, b_done;     // per attention layer
    std::vector<hipEvent_t> appended[PF_MAX_LANES];   // prefill: per lane and layer, "this chunk's K / V rows are in the cache"
};
std::mutex g_pipe_mu;
StepPipeline g_pipe[64];

StepPipeline* step_pipeline(size_t attention_layers) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_pipe_mu);
    StepPipeline& p = g_pipe[dev];
    if (!p.side) {
        if (hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking) != hipSuccess) { p.side = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p.join, hipEventDisableTiming) != hipSuccess)
            return nullptr;
    }
    if (!p.fork || !p.join) return nullptr;
    while (p.a_done.size() < attention_layers) {
        hipEvent_t ea = nullptr, eb = nullptr;
        if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess)
            return nullptr;
        p.a_done.push_back(ea);
        p.b_done.push_back(eb);
    }
    return &p;
}

// the prefill's further streams and its per-lane, per-layer events (on top of step_pipeline's)
StepPipeline* prefill_pipeline(size_t layers, int lanes) {
    StepPipeline* p = step_pipeline(0);
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(g_pipe_mu);
    for (int l = 2; l < lanes; l++)
        if (!p->more[l - 2]) {
            if (hipStreamCreateWithFlags(&p->more[l - 2], hipStreamNonBlocking) != hipSuccess) { p->more[l - 2] = nullptr; return nullptr; }
            if (hipEventCreateWithFlags(&p->lane_join[l - 2], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
    for (int l = 0; l < lanes; l++)
        while (p->appended[l].size() < layers) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            p->appended[l].push_back(e);
        }
    return p;
}

// Batches from this size on are run as two half-batches (0: never, the default: on the shapes measured a half-batch's
// attention, 128 workgroups of 8 waves, streams its cache at 3.2 - 3.7 TB/s against 4.5 - 5.4 TB/s for the whole batch,
// which costs more than hiding the other half's matmuls returns).  CACTUS_HIP_STEP_SPLIT sets the initial value.
size_t& step_split_value() {
    static size_t t = [] {
        const char* e = std::getenv("CACTUS_HIP_STEP_SPLIT");
        return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)0;
    }();
    return t;
}
size_t step_split_threshold() { return step_split_value(); }
// experiment switch: the halves' attention launches take turns (1, the default) or the halves run freely (0)
bool step_split_alternate() {
    static const bool a = [] {
        const char* e = std::getenv("CACTUS_HIP_STEP_ALTERNATE");
        return e ? std::atoi(e) != 0 : true;
    }();
    return a;
}

#define HIP_STEP_TRY(expr)                                        \
    do {                                                          \
        if ((expr) != hipSuccess) return fail("decode step: stream / event call failed"); \
    } while (0)

// One chunk.  appended / previous: per-layer events of this chunk's lane and of the lane of the chunk before it (null:
// chunks follow one another on one stream): this chunk's rows are announced once they are in a block's cache, and its
// attention over the cache waits for the previous chunk's announcement of that block.
int prefill_chunk(const cactus_hip_decoder_weights* w, const cactus_hip_prefill_state* st, const int32_t* tokens,
                  size_t chunk, size_t cache_start, const int32_t* cache_start_dev, void* workspace, int last,
                  cactus_half_t* logits, int32_t* token_out, void* stream, const hipEvent_t* appended,
                  const hipEvent_t* previous) {
    if (const char* bad = check_weights(w)) return fail(bad);
    if (!st || !st->k_cache || !st->v_cache || !st->k_scales || !st->v_scales) return fail("prefill chunk: null cache");
    size_t n_conv = 0;
    for (size_t i = 0; i < w->layers; i++) n_conv += w->blocks[i].kind == 1;
    if (n_conv && !st->conv_state) return fail("prefill chunk: null convolution state");
    if (!tokens || !cache_start_dev || !workspace || chunk == 0) return fail("prefill chunk: null operand");
    if (cache_start + chunk > st->capacity) return fail("prefill chunk: the cache cannot hold cache_start + chunk tokens");
    if (last && (!logits || !token_out)) return fail("prefill chunk: the last chunk needs logits and token_out");
    const size_t M = chunk, Hd = w->hidden, Hq = w->q_heads, Hkv = w->kv_heads, D = w->head_dim, F = w->ffn, V = w->vocab;
    const size_t G = (D + 127) / 128, NL = w->layers, QD = Hq * D;
    const size_t cbs = st->capacity * Hkv * D, sbs = st->capacity * Hkv * G;
    DecodeBuffers d;
    carve_decode(w, M, workspace, &d);
    const cactus_hip_qmatrix& head = w->lm_head.B ? w->lm_head : w->embedding;
    STEP_TRY(cactus_hip_embedding_int8_dev(static_cast<const int8_t*>(w->embedding.B), w->embedding.scales, tokens, d.h0, M, V, Hd,
                                           128, nullptr, stream));
    STEP_TRY(cactus_hip_add_rms_norm_quant_f16_dev(d.h0, nullptr, w->blocks[0].in_norm, nullptr, nullptr, d.xq, d.xs, M, Hd, w->eps,
                                                   stream));
    const half_t* res = d.h0;
    size_t ai = 0, ci = 0;
    for (size_t li = 0; li < NL; li++) {
        const cactus_hip_block_weights& L = w->blocks[li];
        if (L.kind == 1) {
            // LFM2 short convolution over the rows the chunk before left in this block's state (the first chunk
            // starts from none): the in-projection needs nothing of the chunk before, the convolution launch does
            half_t* state = st->conv_state + ci++ * w->conv_rows * Hd;
            STEP_TRY(matmul_q(d.xq, d.xs, L.in_proj, d.ipj, M, stream));
            if (cache_start == 0)
                HIP_STEP_TRY(hipMemsetAsync(state, 0, w->conv_rows * Hd * sizeof(half_t), static_cast<hipStream_t>(stream)));
            else if (previous)
                HIP_STEP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), previous[li], 0));
            STEP_TRY(cactus_hip_lfm2_conv_gate_quant_f16_dev(d.ipj, L.conv_weight, state, nullptr, d.cq, d.cs, 1, M, Hd, w->conv_taps,
                                                            w->conv_rows, stream));
            if (appended) HIP_STEP_TRY(hipEventRecord(appended[li], static_cast<hipStream_t>(stream)));
            STEP_TRY(matmul_q(d.cq, d.cs, L.out_proj, d.oo, M, stream));
        } else {
            const Out qkv[3] = {{&L.q, d.qo}, {&L.k, d.ko}, {&L.v, d.vo}};
            STEP_TRY(shared_matmuls(d.xq, d.xs, qkv, 3, M, Hd, stream));
            int8_t* kc = st->k_cache + ai * cbs;
            int8_t* vc = st->v_cache + ai * cbs;
            float* ks = st->k_scales + ai * sbs;
            float* vs = st->v_scales + ai * sbs;
            ai++;
            STEP_TRY(cactus_hip_qk_norm_rope_kv_append_f16_dev(d.qo, d.ko, d.vo, L.q_norm, L.k_norm, d.qr, d.kr, kc, vc, ks, vs,
                                                              cache_start_dev, 1, M, Hq, Hkv, D, w->eps, cache_start, w->theta, 128, cbs,
                                                              sbs, st->capacity, stream));
            if (appended) HIP_STEP_TRY(hipEventRecord(appended[li], static_cast<hipStream_t>(stream)));
            if (previous && cache_start != 0) HIP_STEP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), previous[li], 0));
            if (cache_start == 0) {
                STEP_TRY(cactus_hip_attention_f16_dev(d.qr, d.kr, d.vo, d.ao, 1, M, M, Hq, Hkv, D, 0.0f, nullptr, 0, 0, 1, stream));
                STEP_TRY(cactus_hip_quantize_activations_dev(d.ao, d.oq, d.os, M, QD, stream));
            } else {
                STEP_TRY(cactus_hip_attention_hybrid_int8_fp16_quant_dev(d.qr, kc, vc, ks, vs, d.kr, d.vo, d.ao, d.oq, d.os, 1, M, cache_start,
                                                                        M, Hq, Hkv, D, 0.0f, cache_start, 1, 128, stream));
            }
            STEP_TRY(matmul_q(d.oq, d.os, L.o, d.oo, M, stream));
        }
        STEP_TRY(cactus_hip_add_rms_norm_quant_f16_dev(res, d.oo, L.post_norm, d.h2, nullptr, d.xq2, d.xs2, M, Hd, w->eps, stream));
        const Out gu[2] = {{&L.gate, d.go}, {&L.up, d.uo}};
        STEP_TRY(shared_matmuls(d.xq2, d.xs2, gu, 2, M, Hd, stream));
        STEP_TRY(cactus_hip_silu_multiply_quant_f16_dev(d.go, d.uo, nullptr, d.aq, d.as, M, F, stream));
        STEP_TRY(matmul_q(d.aq, d.as, L.down, d.dn, M, stream));
        const half_t* nxt = li + 1 < NL ? w->blocks[li + 1].in_norm : w->final_norm;
        const bool fin = li + 1 == NL && last;
        STEP_TRY(cactus_hip_add_rms_norm_quant_f16_dev(d.h2, d.dn, nxt, d.hid, fin ? d.x : nullptr, d.xq, d.xs, M, Hd, w->eps, stream));
        res = d.hid;
    }
    if (last) {
        STEP_TRY(matmul_f16act(d.x + (M - 1) * Hd, head, logits, 1, stream));
        STEP_TRY(cactus_hip_sample_greedy_f16_dev(logits, reinterpret_cast<uint32_t*>(token_out), 1, V, stream));
    }
    return CACTUS_HIP_OK;
}


}  // namespace

extern "C" {

size_t cactus_hip_decode_step_split(size_t min_batch) {
    const size_t prev = step_split_value();
    step_split_value() = min_batch;
    return prev;
}

size_t cactus_hip_decode_step_workspace_bytes(const cactus_hip_decoder_weights* w, size_t batch) {
    if (!w || batch == 0) return 0;
    if (w->blocks && step_split_threshold() && batch >= step_split_threshold()) {   // outside any stream capture
        size_t n_attn = 0;
        for (size_t i = 0; i < w->layers; i++) n_attn += w->blocks[i].kind == 0;
        step_pipeline(n_attn);
    }
    return carve_decode(w, batch, nullptr, nullptr);
}

int cactus_hip_decode_step_dev(const cactus_hip_decoder_weights* w, const cactus_hip_decode_state* kv, void* workspace,
                               cactus_half_t* logits, int advance, void* stream) {
    if (const char* bad = check_weights(w)) return fail(bad);
    if (!kv || kv->batch == 0) return fail("decode step: null or empty state");
    if (!workspace || !logits) return fail("decode step: null workspace or logits");
    if (!kv->positions || !kv->cache_lens || !kv->tokens) return fail("decode step: null positions / cache_lens / tokens");
    if (kv->window && (kv->sink >= kv->window || kv->window > kv->slots)) return fail("decode step: need sink < window <= slots");
    const size_t b = kv->batch, Hd = w->hidden, Hq = w->q_heads, Hkv = w->kv_heads, D = w->head_dim, F = w->ffn, V = w->vocab;
    const size_t G = (D + 127) / 128, NL = w->layers;
    size_t n_attn = 0, n_conv = 0;
    for (size_t i = 0; i < NL; i++) (w->blocks[i].kind == 0 ? n_attn : n_conv)++;
    if (n_attn && (!kv->k_cache || !kv->v_cache || !kv->k_scales || !kv->v_scales || kv->slots == 0))
        return fail("decode step: null KV cache");
    if (n_conv && !kv->conv_state) return fail("decode step: null convolution state");
    DecodeBuffers d;
    carve_decode(w, b, workspace, &d);
    const cactus_hip_qmatrix& head = w->lm_head.B ? w->lm_head : w->embedding;
    const bool small = b <= 4 && n_conv == 0;      // streaming launches that produce their own input rows
    const size_t cbs = kv->slots * Hkv * D, sbs = kv->slots * Hkv * G;
    const bool fused_append = kv->window == 0;
    const bool fused_attention = fused_append && (D == 64 || D == 128 || D == 256);
    const bool o_quant = fused_append && D == 128 && !small && n_conv == 0;

    STEP_TRY(cactus_hip_embedding_int8_dev(static_cast<const int8_t*>(w->embedding.B), w->embedding.scales, kv->tokens, d.h0, b,
                                           V, Hd, 128, nullptr, stream));
    if (!small) {
        STEP_TRY(cactus_hip_add_rms_norm_quant_f16_dev(d.h0, nullptr, w->blocks[0].in_norm, nullptr, nullptr, d.xq, d.xs, b, Hd,
                                                       w->eps, stream));
        // Large batches run as two half-batches on two streams, their attention launches taking turns (A0 B0 A1 B1 ...):
        // a half's attention streams the KV cache (HBM-bound) while the other half runs its matmuls, norms and
        // activations (short launches, latency-bound, few bytes), instead of the two kinds of work alternating on an
        // otherwise idle chip.  Rows are independent, so every result is what the single stream gives.  The second
        // stream forks from and joins the caller's, so a stream capture of the caller's takes in both.
        StepPipeline* pipe = nullptr;
        if (step_split_threshold() && b >= step_split_threshold() && b >= 2) pipe = step_pipeline(n_attn);
        const int nh = pipe ? 2 : 1;
        struct Half { size_t r0, n; DecodeBuffers d; void* stream; const half_t* res; };
        Half hv[2];
        const size_t b0 = nh == 2 ? ((b / 2 + 15) / 16) * 16 : b;     // whole 16-row tiles in the first half
        hv[0] = Half{0, nh == 2 ? (b0 < b ? b0 : b / 2) : b, d, stream, d.h0};
        if (nh == 2) {
            hv[1] = Half{hv[0].n, b - hv[0].n, rows_from(w, d, hv[0].n), pipe->side, d.h0 + hv[0].n * Hd};
            HIP_STEP_TRY(hipEventRecord(pipe->fork, static_cast<hipStream_t>(stream)));
            HIP_STEP_TRY(hipStreamWaitEvent(pipe->side, pipe->fork, 0));
        }
        struct LaneGuard {     // the second half's calls use the library workspace's second lane
            int prev;
            explicit LaneGuard(int lane) : prev(cactus_hip::workspace_lane(lane)) {}
            ~LaneGuard() { cactus_hip::workspace_lane(prev); }
        };
        size_t ai = 0, ci = 0;
        for (size_t li = 0; li < NL; li++) {
            const cactus_hip_block_weights& L = w->blocks[li];
            const half_t* nxt = li + 1 < NL ? w->blocks[li + 1].in_norm : w->final_norm;
            const bool last = li + 1 == NL;     // the lm_head below takes the FP16 row
            if (L.kind == 0) {
                for (int h = 0; h < nh; h++) {
                    LaneGuard lg(h);
                    Half& H = hv[h];
                    const Out qkv[3] = {{&L.q, H.d.qo}, {&L.k, H.d.ko}, {&L.v, H.d.vo}};
                    STEP_TRY(shared_matmuls(H.d.xq, H.d.xs, qkv, 3, H.n, Hd, H.stream));   // rows quantized by the launch that produced them
                }
                for (int h = 0; h < nh; h++) {
                    LaneGuard lg(h);
                    Half& H = hv[h];
                    int8_t* kc = kv->k_cache + (ai * b + H.r0) * cbs;
                    int8_t* vc = kv->v_cache + (ai * b + H.r0) * cbs;
                    float* ks = kv->k_scales + (ai * b + H.r0) * sbs;
                    float* vs = kv->v_scales + (ai * b + H.r0) * sbs;
                    const int32_t* lens = kv->cache_lens + H.r0;
                    const int32_t* pos = kv->positions + H.r0;
                    hipStream_t hs = static_cast<hipStream_t>(H.stream);
                    if (nh == 2 && step_split_alternate()) {       // A(i) after B(i - 1), B(i) after A(i)
                        if (h == 0 && ai > 0) HIP_STEP_TRY(hipStreamWaitEvent(hs, pipe->b_done[ai - 1], 0));
                        if (h == 1) HIP_STEP_TRY(hipStreamWaitEvent(hs, pipe->a_done[ai], 0));
                    }
                    // q/k norm + RoPE at each sequence's position.  Without a window the new token's K and V join the
                    // cache in the same launch as the attention; with a window they are appended after the attention,
                    // which must still see the token a full window is about to drop.
                    if (fused_attention) {
                        if (o_quant)
                            STEP_TRY(cactus_hip_attention_decode_fused_quant_f16_dev(H.d.qo, H.d.ko, H.d.vo, L.q_norm, L.k_norm, kc, vc, ks, vs,
                                                                                    H.d.ao, H.d.oq, H.d.os, H.n, lens, pos, kv->max_position, Hq,
                                                                                    Hkv, D, w->eps, w->theta, 0.0f, 128, cbs, sbs, kv->slots,
                                                                                    kv->slots, H.stream));
                        else
                            STEP_TRY(cactus_hip_attention_decode_fused_f16_dev(H.d.qo, H.d.ko, H.d.vo, L.q_norm, L.k_norm, kc, vc, ks, vs, H.d.ao,
                                                                              H.n, lens, pos, kv->max_position, Hq, Hkv, D, w->eps, w->theta,
                                                                              0.0f, 128, cbs, sbs, kv->slots, kv->slots, H.stream));
                    } else {
                        const bool app = fused_append;
                        STEP_TRY(cactus_hip_qk_norm_rope_seq_f16_dev(H.d.qo, H.d.ko, app ? H.d.vo : nullptr, L.q_norm, L.k_norm, H.d.qr, H.d.kr,
                                                                    app ? kc : nullptr, app ? vc : nullptr, app ? ks : nullptr,
                                                                    app ? vs : nullptr, pos, kv->max_position, H.n, 1, Hq, Hkv, D, w->eps,
                                                                    w->theta, 128, app ? cbs : 0, app ? sbs : 0, app ? kv->slots : 0, 0, 0,
                                                                    H.stream));
                        STEP_TRY(cactus_hip_attention_hybrid_int8_fp16_batched_dev(H.d.qr, kc, vc, ks, vs, H.d.kr, H.d.vo, H.d.ao, H.n, 1, lens, 1,
                                                                                  Hq, Hkv, D, 0.0f, pos, 128, cbs, sbs, kv->slots, H.stream));
                    }
                    if (nh == 2) HIP_STEP_TRY(hipEventRecord(h == 0 ? pipe->a_done[ai] : pipe->b_done[ai], hs));
                    if (!fused_append)
                        STEP_TRY(cactus_hip_kv_append_windowed_dev(H.d.kr, H.d.vo, kc, vc, ks, vs, pos, H.n, 1, Hkv, D, 128, cbs, sbs,
                                                                  kv->slots, kv->window, kv->sink, H.stream));
                    if (o_quant) STEP_TRY(matmul_q(H.d.oq, H.d.os, L.o, H.d.oo, H.n, H.stream));
                    else STEP_TRY(matmul_f16act(H.d.ao, L.o, H.d.oo, H.n, H.stream));
                }
                ai++;
            } else {
                // LFM2 short convolution: in-projection; B * X, convolution over the carried rows, C gate, state update
                // and the out-projection's row quantization in one launch; out-projection
                for (int h = 0; h < nh; h++) {
                    LaneGuard lg(h);
                    Half& H = hv[h];
                    half_t* state = kv->conv_state + (ci * b + H.r0) * w->conv_rows * Hd;
                    STEP_TRY(matmul_q(H.d.xq, H.d.xs, L.in_proj, H.d.ipj, H.n, H.stream));
                    STEP_TRY(cactus_hip_lfm2_conv_gate_quant_f16_dev(H.d.ipj, L.conv_weight, state, nullptr, H.d.cq, H.d.cs, H.n, 1, Hd,
                                                                    w->conv_taps, w->conv_rows, H.stream));
                    STEP_TRY(matmul_q(H.d.cq, H.d.cs, L.out_proj, H.d.oo, H.n, H.stream));
                }
                ci++;
            }
            for (int h = 0; h < nh; h++) {   // norm / activation launches that also quantize the next matmul's input rows
                LaneGuard lg(h);
                Half& H = hv[h];
                const Out gu[2] = {{&L.gate, H.d.go}, {&L.up, H.d.uo}};
                STEP_TRY(cactus_hip_add_rms_norm_quant_f16_dev(H.res, H.d.oo, L.post_norm, H.d.h2, nullptr, H.d.xq2, H.d.xs2, H.n, Hd, w->eps,
                                                               H.stream));
                STEP_TRY(shared_matmuls(H.d.xq2, H.d.xs2, gu, 2, H.n, Hd, H.stream));
                STEP_TRY(cactus_hip_silu_multiply_quant_f16_dev(H.d.go, H.d.uo, nullptr, H.d.aq, H.d.as, H.n, F, H.stream));
                STEP_TRY(matmul_q(H.d.aq, H.d.as, L.down, H.d.dn, H.n, H.stream));
                STEP_TRY(cactus_hip_add_rms_norm_quant_f16_dev(H.d.h2, H.d.dn, nxt, H.d.hid, last ? H.d.x : nullptr, H.d.xq, H.d.xs, H.n, Hd,
                                                               w->eps, H.stream));
                H.res = H.d.hid;
            }
        }
        if (nh == 2) {
            HIP_STEP_TRY(hipEventRecord(pipe->join, pipe->side));
            HIP_STEP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), pipe->join, 0));
        }
    }
    const half_t* res = d.h0;
    size_t ai = 0;
    for (size_t li = 0; small && li < NL; li++) {
        const cactus_hip_block_weights& L = w->blocks[li];
        const Out qkv[3] = {{&L.q, d.qo}, {&L.k, d.ko}, {&L.v, d.vo}};
        // residual add of the previous block + this block's input norm + q/k/v, one launch per precision
        if (li == 0) STEP_TRY(produced_matmuls(1, d.h0, nullptr, L.in_norm, nullptr, w->eps, qkv, 3, b, Hd, stream));
        else STEP_TRY(produced_matmuls(1, d.h2, d.dn, L.in_norm, d.hid, w->eps, qkv, 3, b, Hd, stream));
        if (li) res = d.hid;
        int8_t* kc = kv->k_cache + ai * b * cbs;
        int8_t* vc = kv->v_cache + ai * b * cbs;
        float* ks = kv->k_scales + ai * b * sbs;
        float* vs = kv->v_scales + ai * b * sbs;
        ai++;
        if (fused_attention) {
            STEP_TRY(cactus_hip_attention_decode_fused_f16_dev(d.qo, d.ko, d.vo, L.q_norm, L.k_norm, kc, vc, ks, vs, d.ao, b,
                                                              kv->cache_lens, kv->positions, kv->max_position, Hq, Hkv, D,
                                                              w->eps, w->theta, 0.0f, 128, cbs, sbs, kv->slots, kv->slots,
                                                              stream));
        } else {
            const bool app = fused_append;
            STEP_TRY(cactus_hip_qk_norm_rope_seq_f16_dev(d.qo, d.ko, app ? d.vo : nullptr, L.q_norm, L.k_norm, d.qr, d.kr,
                                                        app ? kc : nullptr, app ? vc : nullptr, app ? ks : nullptr,
                                                        app ? vs : nullptr, kv->positions, kv->max_position, b, 1, Hq, Hkv, D,
                                                        w->eps, w->theta, 128, app ? cbs : 0, app ? sbs : 0,
                                                        app ? kv->slots : 0, 0, 0, stream));
            STEP_TRY(cactus_hip_attention_hybrid_int8_fp16_batched_dev(d.qr, kc, vc, ks, vs, d.kr, d.vo, d.ao, b, 1, kv->cache_lens, 1,
                                                                      Hq, Hkv, D, 0.0f, kv->positions, 128, cbs, sbs, kv->slots,
                                                                      stream));
        }
        if (!fused_append)
            STEP_TRY(cactus_hip_kv_append_windowed_dev(d.kr, d.vo, kc, vc, ks, vs, kv->positions, b, 1, Hkv, D, 128, cbs, sbs,
                                                      kv->slots, kv->window, kv->sink, stream));
        STEP_TRY(matmul_f16act(d.ao, L.o, d.oo, b, stream));
        const Out gu[2] = {{&L.gate, d.go}, {&L.up, d.uo}};
        STEP_TRY(produced_matmuls(1, res, d.oo, L.post_norm, d.h2, w->eps, gu, 2, b, Hd, stream));   // add + norm + gate/up
        const Out dn[1] = {{&L.down, d.dn}};
        STEP_TRY(produced_matmuls(2, d.go, d.uo, nullptr, nullptr, 0.0f, dn, 1, b, F, stream));        // silu x up + down
    }
    if (small) {   // last residual add + final norm + lm_head
        const Out hd[1] = {{&head, logits}};
        STEP_TRY(produced_matmuls(1, d.h2, d.dn, w->final_norm, d.hid, w->eps, hd, 1, b, Hd, stream));
    } else {
        STEP_TRY(matmul_f16act(d.x, head, logits, b, stream));
    }
    STEP_TRY(cactus_hip_sample_greedy_f16_dev(logits, reinterpret_cast<uint32_t*>(kv->tokens), b, V, stream));
    if (advance) STEP_TRY(cactus_hip_kv_positions_advance_dev(kv->positions, kv->cache_lens, b, 1, kv->window, stream));
    return CACTUS_HIP_OK;
}

size_t cactus_hip_prefill_chunk_workspace_bytes(const cactus_hip_decoder_weights* w, size_t chunk) {
    if (!w || chunk == 0) return 0;
    return carve_decode(w, chunk, nullptr, nullptr);
}

int cactus_hip_prefill_chunk_dev(const cactus_hip_decoder_weights* w, const cactus_hip_prefill_state* st, const int32_t* tokens,
                                 size_t chunk, size_t cache_start, const int32_t* cache_start_dev, void* workspace, int last,
                                 cactus_half_t* logits, int32_t* token_out, void* stream) {
    return prefill_chunk(w, st, tokens, chunk, cache_start, cache_start_dev, workspace, last, logits, token_out, stream, nullptr,
                         nullptr);
}

size_t cactus_hip_prefill_workspace_bytes(const cactus_hip_decoder_weights* w, size_t chunk, size_t lanes) {
    if (!w || chunk == 0 || lanes == 0) return 0;
    if (lanes > (size_t)PF_MAX_LANES) lanes = PF_MAX_LANES;
    if (lanes > 1 && w->blocks) prefill_pipeline(w->layers, (int)lanes);     // outside any stream capture
    return lanes * carve_decode(w, chunk, nullptr, nullptr);
}

int cactus_hip_prefill_dev(const cactus_hip_decoder_weights* w, const cactus_hip_prefill_state* st, const int32_t* tokens,
                           size_t n_tokens, size_t chunk, const int32_t* chunk_starts_dev, void* workspace, size_t lanes,
                           cactus_half_t* logits, int32_t* token_out, void* stream) {
    if (const char* bad = check_weights(w)) return fail(bad);
    if (!tokens || !chunk_starts_dev || !workspace || chunk == 0 || n_tokens == 0) return fail("prefill: null or empty operand");
    if (!logits || !token_out) return fail("prefill: null logits / token_out");
    if (lanes == 0) lanes = 1;
    if (lanes > (size_t)PF_MAX_LANES) lanes = PF_MAX_LANES;
    const size_t nchunks = (n_tokens + chunk - 1) / chunk;
    if (lanes > nchunks) lanes = nchunks;
    StepPipeline* pipe = lanes > 1 ? prefill_pipeline(w->layers, (int)lanes) : nullptr;
    if (lanes > 1 && !pipe) lanes = 1;           // no second stream to be had: one chunk after the other
    const size_t per_lane = carve_decode(w, chunk, nullptr, nullptr);
    hipStream_t main_stream = static_cast<hipStream_t>(stream);
    auto lane_stream = [&](size_t l) { return l == 0 ? main_stream : (l == 1 ? pipe->side : pipe->more[l - 2]); };
    if (lanes > 1) {
        HIP_STEP_TRY(hipEventRecord(pipe->fork, main_stream));
        for (size_t l = 1; l < lanes; l++) HIP_STEP_TRY(hipStreamWaitEvent(lane_stream(l), pipe->fork, 0));
    }
    int rc = CACTUS_HIP_OK;
    for (size_t c = 0; c < nchunks && rc == CACTUS_HIP_OK; c++) {
        const size_t l = c % lanes, start = c * chunk, len = start + chunk <= n_tokens ? chunk : n_tokens - start;
        const int prev_lane = cactus_hip::workspace_lane((int)l);
        rc = prefill_chunk(w, st, tokens + start, len, start, chunk_starts_dev + c, static_cast<unsigned char*>(workspace) + l * per_lane,
                           c + 1 == nchunks, logits, token_out, lane_stream(l), lanes > 1 ? pipe->appended[l].data() : nullptr,
                           lanes > 1 && c > 0 ? pipe->appended[(c - 1) % lanes].data() : nullptr);
        cactus_hip::workspace_lane(prev_lane);
    }
    if (lanes > 1) {      // joined even after a failure: a capture of `stream` must not be left with open branches
        for (size_t l = 1; l < lanes; l++) {
            hipEvent_t j = l == 1 ? pipe->join : pipe->lane_join[l - 2];
            if (hipEventRecord(j, lane_stream(l)) != hipSuccess || hipStreamWaitEvent(main_stream, j, 0) != hipSuccess)
                return fail("prefill: stream / event call failed");
        }
    }
    return rc;
}

}  // extern "C"
