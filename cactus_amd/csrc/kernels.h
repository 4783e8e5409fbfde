// Internal launcher declarations shared by the kernel translation units and the C ABI.
// Every launcher validates nothing: operand shapes are checked by the C ABI (api.hip)
// before a kernel is enqueued.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cactus_hip {

// quant_kernels.hip
hipError_t launch_act_quant(const uint16_t* src, int8_t* dst, float* scales, size_t M, size_t K,
                            hipStream_t st);
hipError_t launch_fp16_to_int8(const uint16_t* src, int8_t* dst, size_t count, float scale,
                               hipStream_t st);
hipError_t launch_max_abs(const uint16_t* src, size_t count, float* out, hipStream_t st);
hipError_t launch_kv_quant(const uint16_t* src, int8_t* dst, float* scales, size_t seq, size_t heads,
                           size_t D, size_t gs, hipStream_t st);
// equally shaped quantization jobs ([seq, heads, D] rows each) in launches of up to KV_QUANT_MAX_JOBS
constexpr int KV_QUANT_MAX_JOBS = 64;
struct KvQuantJob {
    const uint16_t* src;
    int8_t* dst;
    float* scales;
};
hipError_t launch_kv_quant_jobs(const KvQuantJob* jobs, size_t count, size_t seq, size_t heads, size_t D, size_t gs,
                                hipStream_t st);
hipError_t launch_kv_quant_pair(const uint16_t* k_src, int8_t* k_dst, float* k_scales, const uint16_t* v_src,
                                int8_t* v_dst, float* v_scales, size_t seq, size_t heads, size_t D, size_t gs,
                                hipStream_t st);
// one new token per sequence, K and V, quantized straight into slot slots[b] of each sequence's cache
hipError_t launch_kv_append(const uint16_t* k_src, int8_t* k_cache, float* k_scales, const uint16_t* v_src,
                            int8_t* v_cache, float* v_scales, size_t batch, size_t heads, size_t D, size_t gs,
                            const int32_t* slots, size_t cache_batch_stride, size_t scale_batch_stride,
                            size_t max_slots, hipStream_t st);
// seq_len new tokens per sequence at absolute positions positions[b] + s, each stored at
// ring_slot(position, window, sink) of its sequence's cache (window 0: slot = position)
hipError_t launch_kv_append_windowed(const uint16_t* k_src, int8_t* k_cache, float* k_scales, const uint16_t* v_src,
                                     int8_t* v_cache, float* v_scales, const int32_t* positions, size_t batch,
                                     size_t seq_len, size_t heads, size_t D, size_t gs, size_t cache_batch_stride,
                                     size_t scale_batch_stride, size_t max_slots, size_t window, size_t sink,
                                     hipStream_t st);
hipError_t launch_kv_positions_advance(int32_t* positions, int32_t* cache_lens, size_t batch, size_t advance,
                                       size_t window, hipStream_t st);
hipError_t launch_unpack_int4(const uint8_t* packed, int8_t* unpacked, size_t count, hipStream_t st);

// matmul_skinny.hip: M <= 16, group_size == 128.  B is int8[N,K] or packed uint8[N,K/2].
// If gsums != nullptr the kernel also writes the INT32 group sums [M,N,G]; C may then be null.
// packed: B is the tile-major twin that launch_pack_weights made of the row-major matrix (matmul_skinny.hip).
hipError_t launch_matmul_skinny(bool int4, const int8_t* A, const float* A_scales, const void* B,
                                const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                                size_t K, size_t N, hipStream_t st, bool packed = false);
// packed must hold ((N + 15) & ~15) * row_bytes bytes; row_bytes (K for INT8, K / 2 for INT4, 2 K for FP16) a multiple of 128
hipError_t launch_pack_weights(size_t row_bytes, const void* B, void* packed, size_t N, hipStream_t st);

// FP16 activations quantized inside the GEMV kernel (one launch).  Only for shapes
// skinny_fused_quant_ok accepts (M <= 4, activations + group sums fit LDS); scales_out may be null.
bool skinny_fused_quant_ok(bool int4, size_t M, size_t K);
hipError_t launch_matmul_skinny_fq(bool int4, const uint16_t* X, float* scales_out, const void* B,
                                   const uint16_t* B_scales, uint16_t* C, size_t M, size_t K, size_t N,
                                   hipStream_t st, bool packed = false);

// One launch for up to 16 independent FP16-activation matmuls (each skinny_fused_quant_ok).
struct SkinnyProblem {
    const uint16_t* X;
    const void* B;
    const uint16_t* Bs;
    uint16_t* C;
    size_t M, K, N;
    bool packed = false;   // B is the tile-major twin
};
hipError_t launch_matmul_skinny_fq_batch(bool int4, const SkinnyProblem* problems, size_t count, hipStream_t st);

// The same launch with the activation rows produced in the prologue (all problems read them):
// kind 1 rms_norm(X [+ x2]) * w with the sum written to sum_out, 2 silu(X) * x2, 3 gelu(X) * x2.
bool skinny_fq_pro_ok(bool int4, int kind, size_t M, size_t K);
hipError_t launch_matmul_skinny_fq_batch_pro(bool int4, int kind, const SkinnyProblem* problems, size_t count,
                                             const uint16_t* x2, const uint16_t* w, uint16_t* sum_out, float eps,
                                             hipStream_t st);

extern int g_skinny_ablate;
hipError_t skinny_read_trace(unsigned long long* out512);
// Benchmark hook: variant = waves_per_workgroup*100 + ring_depth (0 = the default choice).
hipError_t launch_matmul_skinny_variant(int variant, bool int4, const int8_t* A, const float* A_scales,
                                        const void* B, const uint16_t* B_scales, uint16_t* C, int32_t* gsums,
                                        size_t M, size_t K, size_t N, hipStream_t st);

// matmul_gemm.hip: any M, group_size == 128, MFMA tiles staged through LDS.
bool launch_matmul_f16_tiles(const uint16_t* a, const uint16_t* bT, uint16_t* c, size_t M, size_t K, size_t N,
                             hipStream_t st, hipError_t* err);
hipError_t launch_matmul_gemm(bool int4, const int8_t* A, const float* A_scales, const void* B,
                              const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                              size_t K, size_t N, hipStream_t st);

// Several weight matrices against one quantized activation matrix [M, K] (group 128), one launch.
struct GemmProblem {
    const void* B;
    const uint16_t* Bs;
    uint16_t* C;
    size_t N;
};
bool gemm_batch_ok(size_t count, size_t K);
hipError_t launch_matmul_gemm_batch(bool int4, const int8_t* A, const float* A_scales, const GemmProblem* problems,
                                    size_t count, size_t M, size_t K, hipStream_t st);

// matmul_generic.hip: any group_size (whole groups of K only), one thread per output.
hipError_t launch_matmul_generic(bool int4, const int8_t* A, const float* A_scales, const void* B,
                                 const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                                 size_t K, size_t N, size_t gs, hipStream_t st);

// matmul_f16.hip
// streaming GEMV for up to 16 activation rows (K a multiple of 64, rows staged in LDS); packed: bT is the tile-major twin
bool gemv_f16_ok(const uint16_t* a, const uint16_t* bT, size_t M, size_t K, size_t N);
hipError_t launch_gemv_f16(const uint16_t* a, const uint16_t* bT, uint16_t* c, size_t M, size_t K, size_t N,
                           hipStream_t st, bool packed);
hipError_t launch_matmul_f16(const uint16_t* a, const uint16_t* bT, uint16_t* c, size_t M, size_t K,
                             size_t N, hipStream_t st);

hipError_t launch_transpose_f16(const uint16_t* src, uint16_t* dst, size_t R, size_t C, hipStream_t st);

// attention.hip
struct AttnArgs {
    const uint16_t* q;
    const int8_t* k_cached;
    const int8_t* v_cached;
    const float* k_scales;
    const float* v_scales;
    const uint16_t* k_new;   // hybrid: new keys; f16 mode: all keys
    const uint16_t* v_new;
    uint16_t* out;
    const uint16_t* mask;        // f16 mode only
    const int32_t* cache_lens;   // optional per-sequence cache length (batched decode)
    const int32_t* pos_offsets;  // optional per-sequence position offset
    size_t B, S, cache_len, new_len, Hq, Hkv, D;
    float scale;
    size_t position_offset, window_size;
    int is_causal;
    size_t qgs;                  // KV quant group size
    size_t cache_batch_stride;   // elements between sequences in k_cached/v_cached
    size_t scale_batch_stride;   // elements between sequences in k_scales/v_scales
    int f16_mode;                // 1: cactus_attention_f16 semantics, 0: hybrid
};
// workspace: float scratch for split-KV partials; needs attn_workspace_bytes(args) bytes.
size_t attn_workspace_bytes(const AttnArgs& a);
hipError_t launch_attention(const AttnArgs& a, void* workspace, hipStream_t st);
// the same, and the output rows [B * S, Hq * D] quantized as launch_act_quant would (q_out int8, q_scales [B * S])
hipError_t launch_attention_quant(const AttnArgs& a, void* workspace, int8_t* q_out, float* q_scales, hipStream_t st);

// attention_decode.hip: cactus_attention_hybrid_int8_fp16 for one query position per sequence on the matrix
// cores (head_dim 64 / 128 with one scale group per row or 256 with two, up to 16 query heads per kv head); key ranges split
// over workgroups leave records in ws as attn_kernel does
bool attn_decode_mfma_ok(const AttnArgs& a);
// The producer of a decode step's q / k rows and the cache append, folded into the attention launch
// (attn_decode_mfma_kernel<.., FUSED>): norm weights, RoPE tables from position 0 up to max_position, the cache
// to append to (the buffers the attention reads, writable) and its slot count.
struct AttnFusedIn {
    const uint16_t* q_weight = nullptr;
    const uint16_t* k_weight = nullptr;
    const uint16_t* cos_t = nullptr;
    const uint16_t* sin_t = nullptr;
    int8_t* kc = nullptr;
    int8_t* vc = nullptr;
    float* ks = nullptr;
    float* vs = nullptr;
    float eps = 0.0f;
    int max_position = 0;
    size_t max_slots = 0;      // 0: nothing is appended
    uint16_t* k_out = nullptr; // when given: the normalized, rotated key rows [B, Hkv, D] are also written here
    int8_t* q_out = nullptr;   // when given (attn_decode_quant_ok): the output rows [B, Hq * D] quantized for the
    float* q_scales_out = nullptr;   // output projection, and their scales [B]
};
hipError_t launch_attention_decode_mfma(const AttnArgs& a, float* ws, int nsplit, hipStream_t st,
                                        const AttnFusedIn* fused = nullptr);
// nonzero: launch_attention_decode_mfma merges the key ranges inside its workgroups (that many ranges per item)
// and no merge launch follows; 0: it leaves `planned_splits` records per row in ws for attn_merge_kernel
int attn_decode_merge_waves(const AttnArgs& a, int planned_splits);
// true: launch_attention_decode_mfma with fused->q_out set also writes the quantized output rows and their scales
bool attn_decode_quant_ok(const AttnArgs& a, int planned_splits);
// the whole fused launch (decode kernel + split merge); a must satisfy attn_decode_mfma_ok with new_len 1 and
// per-sequence cache_lens / pos_offsets (one sequence: a.cache_len / a.position_offset will do); workspace as
// attn_workspace_bytes(a)
hipError_t launch_attention_decode_fused(const AttnArgs& a, const AttnFusedIn& f, void* workspace, hipStream_t st);

// attention_prefill.hip: cactus_attention_f16 for many queries on the FP16 matrix cores (head_dim 64 /
// 128, no window, no mask tensor); launch_attention routes eligible calls there
bool attn_prefill_ok(const AttnArgs& a);
hipError_t launch_attention_prefill(const AttnArgs& a, hipStream_t st);
// causal FP16 attention whose sliding window cuts visible keys (every query's 32-key blocks start at its own first
// visible key, kernel_attention.cpp:78-92): same shapes as above
bool attn_prefill_win_ok(const AttnArgs& a);
hipError_t launch_attention_prefill_win(const AttnArgs& a, hipStream_t st);
// the same for cactus_attention_hybrid_int8_fp16 with many queries (a prefill chunk against the INT8
// cache): key ranges split over workgroups, records merged by attn_merge_kernel (ws as in
// attn_workspace_bytes)
bool attn_prefill_hybrid_ok(const AttnArgs& a);
int attn_prefill_hybrid_splits(const AttnArgs& a);
hipError_t launch_attention_prefill_hybrid(const AttnArgs& a, float* ws, hipStream_t st);

// block_ops.hip: the ops between the hot kernels of a decoder block
hipError_t launch_rms_norm(const uint16_t* x, const uint16_t* w, uint16_t* y, size_t rows, size_t dims, float eps,
                           hipStream_t st);
// sum_out = a + b, y = rms_norm(sum_out, w): the residual add and the norm after it, one launch
hipError_t launch_add_rms_norm(const uint16_t* a, const uint16_t* b, const uint16_t* w, uint16_t* sum_out,
                               uint16_t* y, size_t rows, size_t dims, float eps, hipStream_t st);
// per-head RMS norm of q [tokens, Hq, D] and k [tokens, Hkv, D], then RoPE of both, one launch
hipError_t launch_qk_norm_rope(const uint16_t* q, const uint16_t* k, const uint16_t* qw, const uint16_t* kw,
                               uint16_t* q_out, uint16_t* k_out, const uint16_t* cos_t, const uint16_t* sin_t,
                               size_t tokens, size_t Hq, size_t Hkv, size_t D, size_t seq_len, float eps,
                               hipStream_t st);
hipError_t launch_unary_mul(int op, const uint16_t* in, const uint16_t* mul, uint16_t* out, size_t n,
                            hipStream_t st);   // act(in) * mul
// cos_t / sin_t: [seq][head_dim / 2] table rows for positions start_pos .. start_pos + seq - 1
hipError_t launch_rope(const uint16_t* in, uint16_t* out, const uint16_t* cos_t, const uint16_t* sin_t,
                       size_t batch, size_t seq, size_t heads, size_t head_dim, hipStream_t st);
hipError_t launch_unary(int op, const uint16_t* in, uint16_t* out, size_t n, hipStream_t st);     // 0 silu, 1 gelu
hipError_t launch_binary(int op, const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n,
                         hipStream_t st);   // 0 add, 1 subtract, 2 multiply, 3 divide, 4 clipped add
// cactus_scalar_op_f16's arithmetic ops: 0 add, 1 subtract, 2 multiply, 3 divide by half(scalar)
hipError_t launch_scalar_op(int op, const uint16_t* a, float scalar, uint16_t* out, size_t n, hipStream_t st);
hipError_t launch_embedding_int8(const int8_t* emb, const uint16_t* scales, const int32_t* indices, uint16_t* out,
                                 size_t num_indices, size_t vocab, size_t hidden, size_t group_size, int* bad,
                                 hipStream_t st);
hipError_t launch_argmax(const uint16_t* logits, uint32_t* out, size_t rows, size_t vocab,
                         unsigned long long* scratch, hipStream_t st);
// the same launch also quantizes the rotated k rows and the v rows into the INT8 cache
hipError_t launch_qk_norm_rope_append(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* qw,
                                      const uint16_t* kw, uint16_t* q_out, uint16_t* k_out, const uint16_t* cos_t,
                                      const uint16_t* sin_t, int8_t* kc, int8_t* vc, float* ks, float* vs,
                                      const int32_t* slots, size_t batch, size_t seq_len, size_t Hq, size_t Hkv,
                                      size_t D, float eps, size_t gs, size_t cache_batch_stride,
                                      size_t scale_batch_stride, size_t max_slots, hipStream_t st);
// per-sequence positions (device int32 [batch]); cos_t / sin_t start at position 0 and hold rows up to
// max_position; v == nullptr: no cache append
hipError_t launch_qk_norm_rope_seq(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* qw,
                                   const uint16_t* kw, uint16_t* q_out, uint16_t* k_out, const uint16_t* cos_t,
                                   const uint16_t* sin_t, int8_t* kc, int8_t* vc, float* ks, float* vs,
                                   const int32_t* positions, size_t max_position, size_t batch, size_t seq_len,
                                   size_t Hq, size_t Hkv, size_t D, float eps, size_t gs, size_t cache_batch_stride,
                                   size_t scale_batch_stride, size_t max_slots, size_t window, size_t sink,
                                   hipStream_t st);
hipError_t launch_row_quant(int mode, const uint16_t* a, const uint16_t* b, const uint16_t* w, uint16_t* sum_out,
                            uint16_t* y_out, int8_t* q_out, float* scales_out, size_t rows, size_t dims, float eps,
                            hipStream_t st);
// sampler.hip: cactus_sample_f16's stochastic branch (one workgroup per row; `work` = fp16
// [rows, vocab] scratch) and the normalised entropy of a logits row
size_t sample_wide_scratch_bytes(size_t rows);
hipError_t launch_sample(const uint16_t* logits, uint16_t* work, uint32_t* out, int* status, size_t rows,
                         size_t vocab, float temperature, float top_p, size_t top_k, const float* draws,
                         uint32_t* history, uint32_t* hist_counts, void* wide_scratch, hipStream_t st);
size_t entropy_scratch_bytes(size_t rows, size_t vocab);
hipError_t launch_entropy(const uint16_t* logits, float* out, size_t rows, size_t vocab, void* scratch, hipStream_t st);
hipError_t launch_indices_to_i32(const void* src, bool is_f32, int32_t* dst, size_t n, hipStream_t st);
hipError_t launch_embedding_f16(const uint16_t* emb, const int32_t* indices, uint16_t* out, size_t num_indices,
                                size_t vocab, size_t hidden, int* bad, hipStream_t st);

// conv_ops.hip: causal depthwise convolution [N, L, C] and the LFM2 short-convolution block
// (B * X, convolution over the carried state rows and the new rows, C-gate product, state update)
hipError_t launch_conv1d_causal_depthwise(const uint16_t* x, const uint16_t* w, uint16_t* y, size_t N, size_t L,
                                          size_t C, size_t K, size_t dilation, hipStream_t st);
// the same launch also quantizes every gated row for the out-projection (C a multiple of 512 up to 2048, 2..4 taps)
bool lfm2_conv_gate_quant_ok(size_t C, size_t K);
hipError_t launch_lfm2_conv_gate_quant(const uint16_t* in_proj, const uint16_t* w, uint16_t* state, uint16_t* gated,
                                       int8_t* q_out, float* scales_out, size_t nseq, size_t L, size_t C, size_t K,
                                       size_t W, hipStream_t st);
hipError_t launch_lfm2_conv_gate(const uint16_t* in_proj, const uint16_t* w, uint16_t* state, uint16_t* gated,
                                 size_t nseq, size_t L, size_t C, size_t K, size_t W, hipStream_t st);

}  // namespace cactus_hip
