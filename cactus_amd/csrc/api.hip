// C ABI of the gfx950 backend (include/cactus_hip.h): argument checks, kernel selection,
// the split-KV workspace, and the host-pointer drop-in family (staging + weight cache).
#include "../../include/cactus_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

using namespace cactus_hip;

namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, #expr, e_); \
    } while (0)

}  // namespace

namespace cactus_hip {
void set_last_error(const char* msg) { g_err = msg ? msg : ""; }
}  // namespace cactus_hip

namespace {

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Grow-only device scratch (split-KV partials).  One per device; calls that overlap on a
// device and both need it must be on one stream (they share the buffer).
struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
};
std::mutex g_ws_mu;
std::unordered_map<int, Workspace> g_ws;

int workspace(size_t bytes, void** out) {
    *out = nullptr;
    if (bytes == 0) return CACTUS_HIP_OK;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ws_mu);
    Workspace& w = g_ws[dev];
    if (w.bytes < bytes) {
        if (w.ptr) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(w.ptr));
            w.ptr = nullptr;
            w.bytes = 0;
        }
        size_t want = bytes < (size_t(8) << 20) ? (size_t(8) << 20) : bytes + bytes / 2;
        if (hipMalloc(&w.ptr, want) != hipSuccess) return fail(CACTUS_HIP_ENOMEM, "workspace hipMalloc");
        w.bytes = want;
    }
    *out = w.ptr;
    return CACTUS_HIP_OK;
}

int matmul_q(bool int4, const int8_t* A, const float* As, const void* B, const cactus_half_t* Bs,
             cactus_half_t* C, int32_t* gsums, size_t M, size_t K, size_t N, size_t gs, void* stream) {
    if (M == 0 || K == 0 || N == 0) return CACTUS_HIP_OK;  // kernel_gemm.cpp:137, :338
    if (gs == 0) return fail(CACTUS_HIP_EINVAL, "matmul: group_size must be > 0");
    if (!A || !B || (!C && !gsums) || (C && (!As || !Bs)))
        return fail(CACTUS_HIP_EINVAL, "matmul: null operand");
    if (int4 && (K & 1)) return fail(CACTUS_HIP_EINVAL, "matmul_int4: K must be even");
    if (K / gs == 0) {
        // no whole group: the reference's group loop does not run and C = fp16(0 * a_scale)
        if (C) {
            hipError_t e = hipMemsetAsync(C, 0, M * N * sizeof(cactus_half_t), S(stream));
            if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "hipMemsetAsync", e);
        }
        return CACTUS_HIP_OK;
    }
    const bool tiled_ok = gs == 128 && (K % 128) == 0 && aligned(A, 16) && aligned(B, 16) &&
                          K < (size_t(1) << 30) && N < (size_t(1) << 30);
    hipError_t e;
    if (tiled_ok && M <= 16 && (!int4 || K % 256 == 0))
        e = launch_matmul_skinny(int4, A, As, B, Bs, C, gsums, M, K, N, S(stream));
    else if (tiled_ok)
        e = launch_matmul_gemm(int4, A, As, B, Bs, C, gsums, M, K, N, S(stream));
    else
        e = launch_matmul_generic(int4, A, As, B, Bs, C, gsums, M, K, N, gs, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "matmul launch", e);
    return CACTUS_HIP_OK;
}

// FP16 activations: one fused launch when the skinny kernel can quantize in place, otherwise
// row quantization into the device workspace followed by the INT8-activation path.
int matmul_q_f16act(bool int4, const cactus_half_t* X, const void* B, const cactus_half_t* Bs,
                    cactus_half_t* C, size_t M, size_t K, size_t N, size_t gs, void* stream) {
    if (M == 0 || K == 0 || N == 0) return CACTUS_HIP_OK;
    if (!X || !B || !Bs || !C) return fail(CACTUS_HIP_EINVAL, "matmul_f16act: null operand");
    if (gs == 128 && K % 128 == 0 && aligned(X, 16) && aligned(B, 16) && aligned(Bs, 2) &&
        K < (size_t(1) << 30) && N < (size_t(1) << 30) && skinny_fused_quant_ok(int4, M, K)) {
        hipError_t e = launch_matmul_skinny_fq(int4, X, nullptr, B, Bs, C, M, K, N, S(stream));
        if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "matmul_f16act launch", e);
        return CACTUS_HIP_OK;
    }
    void* ws = nullptr;
    const size_t qbytes = (M * K + 255) & ~size_t(255);
    int rc = workspace(qbytes + M * sizeof(float), &ws);
    if (rc != CACTUS_HIP_OK) return rc;
    int8_t* q = static_cast<int8_t*>(ws);
    float* qs = reinterpret_cast<float*>(static_cast<char*>(ws) + qbytes);
    hipError_t e = launch_act_quant(X, q, qs, M, K, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "act_quant launch", e);
    return matmul_q(int4, q, qs, B, Bs, C, nullptr, M, K, N, gs, stream);
}

int matmul_q_f16act_batch(bool int4, const cactus_hip_matmul_problem* pr, size_t count, size_t gs, void* stream) {
    if (count == 0) return CACTUS_HIP_OK;
    if (!pr) return fail(CACTUS_HIP_EINVAL, "matmul_f16act_batch: null problem list");
    std::vector<SkinnyProblem> fused;
    fused.reserve(count);
    for (size_t i = 0; i < count; i++) {
        const cactus_hip_matmul_problem& q = pr[i];
        if (q.M == 0 || q.K == 0 || q.N == 0) continue;
        if (!q.X || !q.B || !q.B_scales || !q.C) return fail(CACTUS_HIP_EINVAL, "matmul_f16act_batch: null operand");
        const bool ok = gs == 128 && q.K % 128 == 0 && aligned(q.X, 16) && aligned(q.B, 16) &&
                        q.K < (size_t(1) << 30) && q.N < (size_t(1) << 30) && skinny_fused_quant_ok(int4, q.M, q.K);
        if (ok) {
            fused.push_back(SkinnyProblem{q.X, q.B, q.B_scales, q.C, q.M, q.K, q.N});
        } else {
            int rc = matmul_q_f16act(int4, q.X, q.B, q.B_scales, q.C, q.M, q.K, q.N, gs, stream);
            if (rc != CACTUS_HIP_OK) return rc;
        }
    }
    for (size_t i = 0; i < fused.size(); i += 16) {
        const size_t n = fused.size() - i < 16 ? fused.size() - i : 16;
        hipError_t e = launch_matmul_skinny_fq_batch(int4, fused.data() + i, n, S(stream));
        if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "matmul_f16act_batch launch", e);
    }
    return CACTUS_HIP_OK;
}

int attention(const AttnArgs& a, void* stream) {
    if (a.B == 0 || a.S == 0 || a.Hq == 0 || a.D == 0) return CACTUS_HIP_OK;
    if (a.Hkv == 0 || a.Hq % a.Hkv != 0) return fail(CACTUS_HIP_EINVAL, "attention: Hq must be a multiple of Hkv");
    if (a.D > 8000) return fail(CACTUS_HIP_EINVAL, "attention: head_dim above 8000");
    if (!a.f16_mode && a.qgs == 0) return fail(CACTUS_HIP_EINVAL, "attention: KV group_size must be nonzero");
    if (!a.q || !a.out) return fail(CACTUS_HIP_EINVAL, "attention: null operand");
    void* ws = nullptr;
    int rc = workspace(attn_workspace_bytes(a), &ws);
    if (rc != CACTUS_HIP_OK) return rc;
    hipError_t e = launch_attention(a, ws, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "attention launch", e);
    return CACTUS_HIP_OK;
}

}  // namespace

extern "C" {

int cactus_hip_init(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));
    return CACTUS_HIP_OK;
}

int cactus_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* cactus_hip_last_error(void) { return g_err.c_str(); }
void cactus_hip_clear_error(void) { g_err.clear(); }

void* cactus_hip_malloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        fail(CACTUS_HIP_ENOMEM, "hipMalloc");
        return nullptr;
    }
    return p;
}
void cactus_hip_free(void* dptr) {
    if (dptr) (void)hipFree(dptr);
}
int cactus_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
    return CACTUS_HIP_OK;
}
int cactus_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    return CACTUS_HIP_OK;
}
int cactus_hip_synchronize(void* stream) {
    HIP_TRY(hipStreamSynchronize(S(stream)));
    return CACTUS_HIP_OK;
}

int cactus_hip_matmul_int8_dev(const int8_t* A, const float* A_scales, const int8_t* B,
                               const cactus_half_t* B_scales, cactus_half_t* C, size_t M, size_t K,
                               size_t N, size_t group_size, void* stream) {
    if (!C && M && K && N) return fail(CACTUS_HIP_EINVAL, "matmul_int8: null C");
    return matmul_q(false, A, A_scales, B, B_scales, C, nullptr, M, K, N, group_size, stream);
}
int cactus_hip_matmul_int4_dev(const int8_t* A, const float* A_scales, const uint8_t* B_packed,
                               const cactus_half_t* B_scales, cactus_half_t* C, size_t M, size_t K,
                               size_t N, size_t group_size, void* stream) {
    if (!C && M && K && N) return fail(CACTUS_HIP_EINVAL, "matmul_int4: null C");
    return matmul_q(true, A, A_scales, B_packed, B_scales, C, nullptr, M, K, N, group_size, stream);
}
int cactus_hip_matmul_int8_group_sums_dev(const int8_t* A, const int8_t* B, int32_t* sums, size_t M,
                                          size_t K, size_t N, size_t group_size, void* stream) {
    if (!sums && M && K && N) return fail(CACTUS_HIP_EINVAL, "group_sums: null output");
    return matmul_q(false, A, nullptr, B, nullptr, nullptr, sums, M, K, N, group_size, stream);
}
int cactus_hip_matmul_int4_group_sums_dev(const int8_t* A, const uint8_t* B_packed, int32_t* sums,
                                          size_t M, size_t K, size_t N, size_t group_size, void* stream) {
    if (!sums && M && K && N) return fail(CACTUS_HIP_EINVAL, "group_sums: null output");
    return matmul_q(true, A, nullptr, B_packed, nullptr, nullptr, sums, M, K, N, group_size, stream);
}

int cactus_hip_matmul_int8_f16act_dev(const cactus_half_t* X, const int8_t* B, const cactus_half_t* B_scales,
                                      cactus_half_t* C, size_t M, size_t K, size_t N, size_t group_size,
                                      void* stream) {
    return matmul_q_f16act(false, X, B, B_scales, C, M, K, N, group_size, stream);
}
int cactus_hip_matmul_int4_f16act_dev(const cactus_half_t* X, const uint8_t* B_packed,
                                      const cactus_half_t* B_scales, cactus_half_t* C, size_t M, size_t K,
                                      size_t N, size_t group_size, void* stream) {
    return matmul_q_f16act(true, X, B_packed, B_scales, C, M, K, N, group_size, stream);
}

int cactus_hip_matmul_int8_f16act_batch_dev(const cactus_hip_matmul_problem* problems, size_t count,
                                            size_t group_size, void* stream) {
    return matmul_q_f16act_batch(false, problems, count, group_size, stream);
}
int cactus_hip_matmul_int4_f16act_batch_dev(const cactus_hip_matmul_problem* problems, size_t count,
                                            size_t group_size, void* stream) {
    return matmul_q_f16act_batch(true, problems, count, group_size, stream);
}

int cactus_hip_matmul_f16_dev(const cactus_half_t* a, const cactus_half_t* bT, cactus_half_t* c,
                              size_t M, size_t K, size_t N, void* stream) {
    if (M == 0 || N == 0) return CACTUS_HIP_OK;
    if (!a || !bT || !c) return fail(CACTUS_HIP_EINVAL, "matmul_f16: null operand");
    hipError_t e = launch_matmul_f16(a, bT, c, M, K, N, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "matmul_f16 launch", e);
    return CACTUS_HIP_OK;
}

int cactus_hip_quantize_activations_dev(const cactus_half_t* src, int8_t* dst, float* scales, size_t M,
                                        size_t K, void* stream) {
    if (M == 0 || K == 0) return CACTUS_HIP_OK;
    if (!src || !dst || !scales) return fail(CACTUS_HIP_EINVAL, "quantize_activations: null operand");
    hipError_t e = launch_act_quant(src, dst, scales, M, K, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "act_quant launch", e);
    return CACTUS_HIP_OK;
}

int cactus_hip_fp16_max_abs_dev(const cactus_half_t* src, size_t count, float* out_dev, void* stream) {
    if (!out_dev || (!src && count)) return fail(CACTUS_HIP_EINVAL, "fp16_max_abs: null operand");
    hipError_t e = launch_max_abs(src, count, out_dev, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "max_abs launch", e);
    return CACTUS_HIP_OK;
}

int cactus_hip_fp16_to_int8_dev(const cactus_half_t* src, int8_t* dst, size_t count, float scale,
                                void* stream) {
    if (count == 0) return CACTUS_HIP_OK;
    if (!src || !dst) return fail(CACTUS_HIP_EINVAL, "fp16_to_int8: null operand");
    hipError_t e = launch_fp16_to_int8(src, dst, count, scale, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "fp16_to_int8 launch", e);
    return CACTUS_HIP_OK;
}

int cactus_hip_quantize_kv_fp16_to_int8_dev(const cactus_half_t* src, int8_t* dst, float* scales,
                                            size_t seq_len, size_t kv_heads, size_t head_dim,
                                            size_t group_size, void* stream) {
    if (seq_len * kv_heads * head_dim == 0) return CACTUS_HIP_OK;
    if (!src || !dst || !scales || group_size == 0)
        return fail(CACTUS_HIP_EINVAL, "quantize_kv: bad operand");
    hipError_t e = launch_kv_quant(src, dst, scales, seq_len, kv_heads, head_dim, group_size, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "kv_quant launch", e);
    return CACTUS_HIP_OK;
}

int cactus_hip_unpack_int4_to_int8_dev(const uint8_t* packed, int8_t* unpacked, size_t unpacked_count,
                                       void* stream) {
    if (unpacked_count == 0) return CACTUS_HIP_OK;
    if (!packed || !unpacked) return fail(CACTUS_HIP_EINVAL, "unpack_int4: null operand");
    hipError_t e = launch_unpack_int4(packed, unpacked, unpacked_count, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, "unpack launch", e);
    return CACTUS_HIP_OK;
}

int cactus_hip_attention_f16_dev(const cactus_half_t* queries, const cactus_half_t* keys,
                                 const cactus_half_t* values, cactus_half_t* output, size_t batch_size,
                                 size_t seq_len, size_t kv_seq_len, size_t num_q_heads,
                                 size_t num_kv_heads, size_t head_dim, float scale,
                                 const cactus_half_t* mask, size_t position_offset, size_t window_size,
                                 int is_causal, void* stream) {
    AttnArgs a{};
    a.q = queries; a.k_new = keys; a.v_new = values; a.out = output; a.mask = mask;
    a.B = batch_size; a.S = seq_len; a.cache_len = 0; a.new_len = kv_seq_len;
    a.Hq = num_q_heads; a.Hkv = num_kv_heads; a.D = head_dim; a.scale = scale;
    a.position_offset = position_offset; a.window_size = window_size; a.is_causal = is_causal;
    a.qgs = CACTUS_KV_QUANT_GROUP_SIZE; a.f16_mode = 1;
    if (kv_seq_len && (!keys || !values)) return fail(CACTUS_HIP_EINVAL, "attention_f16: null K/V");
    return attention(a, stream);
}

int cactus_hip_attention_hybrid_int8_fp16_dev(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales, const cactus_half_t* keys_new,
    const cactus_half_t* values_new, cactus_half_t* output, size_t batch_size, size_t seq_len,
    size_t cache_len, size_t new_len, size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
    float scale, size_t position_offset, int is_causal, size_t group_size, void* stream) {
    AttnArgs a{};
    a.q = queries; a.k_cached = keys_cached; a.v_cached = values_cached;
    a.k_scales = k_scales; a.v_scales = v_scales; a.k_new = keys_new; a.v_new = values_new;
    a.out = output; a.B = batch_size; a.S = seq_len; a.cache_len = cache_len; a.new_len = new_len;
    a.Hq = num_q_heads; a.Hkv = num_kv_heads; a.D = head_dim; a.scale = scale;
    a.position_offset = position_offset; a.is_causal = is_causal; a.qgs = group_size;
    a.cache_batch_stride = cache_len * num_kv_heads * head_dim;  // kernel_attention.cpp:292
    a.scale_batch_stride = 0;                                    // kernel_attention.cpp:353
    a.f16_mode = 0;
    if (cache_len && (!keys_cached || !values_cached || !k_scales || !v_scales))
        return fail(CACTUS_HIP_EINVAL, "attention_hybrid: null cache");
    if (new_len && (!keys_new || !values_new)) return fail(CACTUS_HIP_EINVAL, "attention_hybrid: null new K/V");
    return attention(a, stream);
}

int cactus_hip_attention_hybrid_int8_fp16_batched_dev(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales, const cactus_half_t* keys_new,
    const cactus_half_t* values_new, cactus_half_t* output, size_t batch_size, size_t seq_len,
    const int32_t* cache_lens, size_t new_len, size_t num_q_heads, size_t num_kv_heads,
    size_t head_dim, float scale, const int32_t* position_offsets, size_t group_size,
    size_t cache_batch_stride, size_t scale_batch_stride, size_t max_cache_len, void* stream) {
    if (!cache_lens || !position_offsets)
        return fail(CACTUS_HIP_EINVAL, "attention_hybrid_batched: null length arrays");
    AttnArgs a{};
    a.q = queries; a.k_cached = keys_cached; a.v_cached = values_cached;
    a.k_scales = k_scales; a.v_scales = v_scales; a.k_new = keys_new; a.v_new = values_new;
    a.out = output; a.B = batch_size; a.S = seq_len; a.cache_len = max_cache_len; a.new_len = new_len;
    a.Hq = num_q_heads; a.Hkv = num_kv_heads; a.D = head_dim; a.scale = scale;
    a.is_causal = 1; a.qgs = group_size; a.cache_lens = cache_lens; a.pos_offsets = position_offsets;
    a.cache_batch_stride = cache_batch_stride; a.scale_batch_stride = scale_batch_stride;
    a.f16_mode = 0;
    return attention(a, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Host-pointer drop-in family
// ---------------------------------------------------------------------------------------
namespace {

struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
};
std::mutex g_host_mu;
std::unordered_map<const void*, DevBuf> g_weights;  // host address -> resident device copy
std::vector<DevBuf> g_stage;                        // per-call staging slots, grow-only

void* stage(size_t slot, size_t bytes) {
    if (g_stage.size() <= slot) g_stage.resize(slot + 1);
    DevBuf& b = g_stage[slot];
    if (b.bytes < bytes) {
        if (b.ptr) (void)hipFree(b.ptr);
        b.ptr = nullptr;
        b.bytes = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&b.ptr, want) != hipSuccess) {
            fail(CACTUS_HIP_ENOMEM, "staging hipMalloc");
            return nullptr;
        }
        b.bytes = want;
    }
    return b.ptr;
}

void* upload(size_t slot, const void* host, size_t bytes) {
    if (bytes == 0) return stage(slot, 1);
    void* d = stage(slot, bytes);
    if (!d) return nullptr;
    if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        fail(CACTUS_HIP_ERUNTIME, "hipMemcpy H2D");
        return nullptr;
    }
    return d;
}

// Weights are immutable mmap'd regions: upload once, keep resident, key by address.
void* resident(const void* host, size_t bytes) {
    auto it = g_weights.find(host);
    if (it != g_weights.end() && it->second.bytes == bytes) return it->second.ptr;
    if (it != g_weights.end()) {
        (void)hipFree(it->second.ptr);
        g_weights.erase(it);
    }
    DevBuf b;
    if (hipMalloc(&b.ptr, bytes ? bytes : 1) != hipSuccess) {
        fail(CACTUS_HIP_ENOMEM, "weight hipMalloc");
        return nullptr;
    }
    b.bytes = bytes;
    if (bytes && hipMemcpy(b.ptr, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(b.ptr);
        fail(CACTUS_HIP_ERUNTIME, "weight hipMemcpy");
        return nullptr;
    }
    g_weights[host] = b;
    return b.ptr;
}

bool download(void* host, const void* dev, size_t bytes) {
    if (bytes == 0) return true;
    if (hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        fail(CACTUS_HIP_ERUNTIME, "hipMemcpy D2H");
        return false;
    }
    return true;
}

void host_matmul_q(bool int4, const int8_t* A, const float* As, const void* B, const cactus_half_t* Bs,
                   cactus_half_t* C, size_t M, size_t K, size_t N, size_t gs) {
    if (M == 0 || K == 0 || N == 0) return;
    if (gs == 0) { fail(CACTUS_HIP_EINVAL, "matmul: group_size must be > 0"); return; }
    std::lock_guard<std::mutex> lk(g_host_mu);
    const size_t G = K / gs;
    void* dB = resident(B, int4 ? N * (K / 2) : N * K);
    void* dBs = resident(Bs, N * G * sizeof(cactus_half_t));
    void* dA = upload(0, A, M * K);
    void* dAs = upload(1, As, M * sizeof(float));
    void* dC = stage(2, M * N * sizeof(cactus_half_t));
    if (!dB || !dBs || !dA || !dAs || !dC) return;
    int rc = int4 ? cactus_hip_matmul_int4_dev((const int8_t*)dA, (const float*)dAs, (const uint8_t*)dB,
                                               (const cactus_half_t*)dBs, (cactus_half_t*)dC, M, K, N, gs, nullptr)
                  : cactus_hip_matmul_int8_dev((const int8_t*)dA, (const float*)dAs, (const int8_t*)dB,
                                               (const cactus_half_t*)dBs, (cactus_half_t*)dC, M, K, N, gs, nullptr);
    if (rc != CACTUS_HIP_OK) return;
    download(C, dC, M * N * sizeof(cactus_half_t));
}

}  // namespace

extern "C" {

void cactus_hip_release_weights(const void* host_ptr) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (host_ptr) {
        auto it = g_weights.find(host_ptr);
        if (it != g_weights.end()) {
            (void)hipFree(it->second.ptr);
            g_weights.erase(it);
        }
        return;
    }
    for (auto& kv : g_weights) (void)hipFree(kv.second.ptr);
    g_weights.clear();
}

void cactus_hip_matmul_int8(const int8_t* A, const float* A_scales, const int8_t* B,
                            const cactus_half_t* B_scales, cactus_half_t* C, size_t M, size_t K, size_t N,
                            size_t group_size) {
    host_matmul_q(false, A, A_scales, B, B_scales, C, M, K, N, group_size);
}

void cactus_hip_matmul_int4(const int8_t* A, const float* A_scales, const uint8_t* B_packed,
                            const cactus_half_t* B_scales, cactus_half_t* C, size_t M, size_t K, size_t N,
                            size_t group_size) {
    host_matmul_q(true, A, A_scales, B_packed, B_scales, C, M, K, N, group_size);
}

void cactus_hip_matmul_f16(const cactus_half_t* a, const cactus_half_t* bT, cactus_half_t* c, size_t M,
                           size_t K, size_t N) {
    if (M == 0 || N == 0) return;
    std::lock_guard<std::mutex> lk(g_host_mu);
    void* dB = resident(bT, N * K * 2);
    void* dA = upload(0, a, M * K * 2);
    void* dC = stage(2, M * N * 2);
    if (!dB || !dA || !dC) return;
    if (cactus_hip_matmul_f16_dev((const cactus_half_t*)dA, (const cactus_half_t*)dB, (cactus_half_t*)dC, M, K,
                                  N, nullptr) != CACTUS_HIP_OK)
        return;
    download(c, dC, M * N * 2);
}

void cactus_hip_attention_f16(const cactus_half_t* queries, const cactus_half_t* keys,
                              const cactus_half_t* values, cactus_half_t* output, size_t B, size_t Sq,
                              size_t KV, size_t Hq, size_t Hkv, size_t D, float scale,
                              const cactus_half_t* mask, size_t position_offset, size_t window_size,
                              int is_causal) {
    if (B * Sq * Hq * D == 0) return;
    std::lock_guard<std::mutex> lk(g_host_mu);
    const size_t qb = B * Sq * Hq * D * 2, kb = B * KV * Hkv * D * 2;
    void* dq = upload(0, queries, qb);
    void* dk = upload(1, keys, kb);
    void* dv = upload(2, values, kb);
    void* dm = mask ? upload(3, mask, B * Sq * KV * 2) : nullptr;
    void* dout = stage(4, qb);
    if (!dq || !dk || !dv || !dout || (mask && !dm)) return;
    if (cactus_hip_attention_f16_dev((const cactus_half_t*)dq, (const cactus_half_t*)dk,
                                     (const cactus_half_t*)dv, (cactus_half_t*)dout, B, Sq, KV, Hq, Hkv, D,
                                     scale, (const cactus_half_t*)dm, position_offset, window_size,
                                     is_causal, nullptr) != CACTUS_HIP_OK)
        return;
    download(output, dout, qb);
}

void cactus_hip_attention_hybrid_int8_fp16(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales, const cactus_half_t* keys_new,
    const cactus_half_t* values_new, cactus_half_t* output, size_t B, size_t Sq, size_t cache_len,
    size_t new_len, size_t Hq, size_t Hkv, size_t D, float scale, size_t position_offset, int is_causal,
    size_t group_size) {
    if (B * Sq * Hq * D == 0) return;
    if (group_size == 0) { fail(CACTUS_HIP_EINVAL, "attention_hybrid: group_size must be > 0"); return; }
    std::lock_guard<std::mutex> lk(g_host_mu);
    const size_t G = (D + group_size - 1) / group_size;
    const size_t qb = B * Sq * Hq * D * 2;
    void* dq = upload(0, queries, qb);
    void* dkc = upload(1, keys_cached, B * cache_len * Hkv * D);
    void* dvc = upload(2, values_cached, B * cache_len * Hkv * D);
    void* dks = upload(3, k_scales, cache_len * Hkv * G * sizeof(float));
    void* dvs = upload(4, v_scales, cache_len * Hkv * G * sizeof(float));
    void* dkn = upload(5, keys_new, B * new_len * Hkv * D * 2);
    void* dvn = upload(6, values_new, B * new_len * Hkv * D * 2);
    void* dout = stage(7, qb);
    if (!dq || !dkc || !dvc || !dks || !dvs || !dkn || !dvn || !dout) return;
    if (cactus_hip_attention_hybrid_int8_fp16_dev(
            (const cactus_half_t*)dq, (const int8_t*)dkc, (const int8_t*)dvc, (const float*)dks,
            (const float*)dvs, (const cactus_half_t*)dkn, (const cactus_half_t*)dvn, (cactus_half_t*)dout, B,
            Sq, cache_len, new_len, Hq, Hkv, D, scale, position_offset, is_causal, group_size,
            nullptr) != CACTUS_HIP_OK)
        return;
    download(output, dout, qb);
}

void cactus_hip_quantize_kv_fp16_to_int8(const cactus_half_t* src, int8_t* dst, float* scales, size_t seq_len,
                                         size_t kv_heads, size_t head_dim, size_t group_size) {
    const size_t n = seq_len * kv_heads * head_dim;
    if (n == 0) return;
    if (group_size == 0) { fail(CACTUS_HIP_EINVAL, "quantize_kv: group_size must be > 0"); return; }
    std::lock_guard<std::mutex> lk(g_host_mu);
    const size_t G = (head_dim + group_size - 1) / group_size;
    void* ds = upload(0, src, n * 2);
    void* dd = stage(1, n);
    void* dsc = stage(2, seq_len * kv_heads * G * sizeof(float));
    if (!ds || !dd || !dsc) return;
    if (cactus_hip_quantize_kv_fp16_to_int8_dev((const cactus_half_t*)ds, (int8_t*)dd, (float*)dsc, seq_len,
                                                kv_heads, head_dim, group_size, nullptr) != CACTUS_HIP_OK)
        return;
    if (download(dst, dd, n)) download(scales, dsc, seq_len * kv_heads * G * sizeof(float));
}

float cactus_hip_fp16_max_abs(const cactus_half_t* src, size_t count) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    void* ds = upload(0, src, count * 2);
    void* dout = stage(1, sizeof(float));
    float r = 0.0f;
    if (!ds || !dout) return r;
    if (cactus_hip_fp16_max_abs_dev((const cactus_half_t*)ds, count, (float*)dout, nullptr) != CACTUS_HIP_OK)
        return r;
    download(&r, dout, sizeof(float));
    return r;
}

void cactus_hip_fp16_to_int8(const cactus_half_t* src, int8_t* dst, size_t count, float scale) {
    if (count == 0) return;
    std::lock_guard<std::mutex> lk(g_host_mu);
    void* ds = upload(0, src, count * 2);
    void* dd = stage(1, count);
    if (!ds || !dd) return;
    if (cactus_hip_fp16_to_int8_dev((const cactus_half_t*)ds, (int8_t*)dd, count, scale, nullptr) !=
        CACTUS_HIP_OK)
        return;
    download(dst, dd, count);
}

void cactus_hip_unpack_int4_to_int8(const uint8_t* packed, int8_t* unpacked, size_t unpacked_count) {
    if (unpacked_count == 0) return;
    std::lock_guard<std::mutex> lk(g_host_mu);
    void* dp = upload(0, packed, (unpacked_count + 1) / 2);
    void* du = upload(1, unpacked, unpacked_count);  // elements the reference leaves untouched stay
    if (!dp || !du) return;
    if (cactus_hip_unpack_int4_to_int8_dev((const uint8_t*)dp, (int8_t*)du, unpacked_count, nullptr) !=
        CACTUS_HIP_OK)
        return;
    download(unpacked, du, unpacked_count);
}

}  // extern "C"
