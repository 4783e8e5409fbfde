/*
 * cactus_oracle.h -- CPU restatement of the Cactus quantized hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker for the HIP kernels:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under cactus_amd/ links, imports or calls it.
 *
 * Every function restates, in plain scalar C (no NEON), the arithmetic of one
 * reference kernel; the reference file:line each one follows is cited at its
 * definition in cactus_oracle.c.  Halves are carried as uint16_t (IEEE
 * binary16 bit patterns) so the library builds with gcc on x86.
 *
 * Pinning status:
 *   - oracle_unpack_int4_to_int8, the INT8/INT4 weight layout consumed by
 *     oracle_matmul_int8/int4: pinned by fixtures generated from the
 *     reference's own python/src/tensor_io.py (tests/golden/).
 *   - oracle_matmul_int8 / oracle_matmul_int4: pinned by the reference's own
 *     known-answer tests tests/test_kernel.cpp:214-405 (scalar formula,
 *     tolerance 0.1) re-run against this library.
 *   - oracle_matmul_f16: pinned by tests/test_graph.cpp:75-90 (exact ints).
 *   - oracle_quantize_kv_fp16_to_int8: pinned only by
 *     tests/test_kv_cache.cpp:108-124 (+-2 LSB on a constant fill).
 *   - oracle_fp16_max_abs / oracle_fp16_to_int8 / oracle_attention_f16 /
 *     oracle_attention_hybrid_int8_fp16: PARITY UNPINNED -- the reference holds
 *     no golden vector for them and its NEON sources do not build on x86.
 *   - oracle_rms_norm_f16 / oracle_rope_f16 / oracle_silu_f16 / oracle_gelu_f16 /
 *     oracle_binary_f16 / oracle_add_f16_clipped / oracle_embedding_*: PARITY
 *     UNPINNED beyond the reference's loose tests (tests/test_kernel.cpp,
 *     tolerance-based on unseeded inputs), for the same reason.
 */
#ifndef CACTUS_ORACLE_H
#define CACTUS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* IEEE binary16 <-> binary32, round-to-nearest-even, subnormals kept. */
float    oracle_h2f(uint16_t h);
uint16_t oracle_f2h(float f);

/* 1 (default): statements the ARM compilers contract are evaluated with one
 * rounding (fmaf); 0: multiply and add round separately. */
void oracle_set_fma(int fused);
/* Worker threads used by the matmul/attention loops (OpenMP); default 1. */
void oracle_set_threads(int n);

void oracle_matmul_int8(const int8_t* A, const float* A_scales,
                        const int8_t* B, const uint16_t* B_scales,
                        uint16_t* C, size_t M, size_t K, size_t N,
                        size_t group_size, int32_t* group_sums_out);

void oracle_matmul_int4(const int8_t* A, const float* A_scales,
                        const uint8_t* B_packed, const uint16_t* B_scales,
                        uint16_t* C, size_t M, size_t K, size_t N,
                        size_t group_size, int32_t* group_sums_out);

void oracle_matmul_f16(const uint16_t* a, const uint16_t* b_transposed,
                       uint16_t* c, size_t M, size_t K, size_t N);

float oracle_fp16_max_abs(const uint16_t* src, size_t count);
void  oracle_fp16_to_int8(const uint16_t* src, int8_t* dst, size_t count, float scale);
void  oracle_quantize_activations(const uint16_t* src, int8_t* dst, float* scales,
                                  size_t M, size_t K);

void oracle_quantize_kv_fp16_to_int8(const uint16_t* src, int8_t* dst, float* scales,
                                     size_t seq_len, size_t kv_heads, size_t head_dim,
                                     size_t group_size);

void oracle_unpack_int4_to_int8(const uint8_t* packed, int8_t* unpacked, size_t unpacked_count);

void oracle_attention_f16(const uint16_t* queries, const uint16_t* keys, const uint16_t* values,
                          uint16_t* output, size_t batch_size, size_t seq_len, size_t kv_seq_len,
                          size_t num_q_heads, size_t num_kv_heads, size_t head_dim, float scale,
                          const uint16_t* mask, size_t position_offset, size_t window_size,
                          int is_causal);

void oracle_attention_hybrid_int8_fp16(const uint16_t* queries,
                                       const int8_t* keys_cached, const int8_t* values_cached,
                                       const float* k_scales, const float* v_scales,
                                       const uint16_t* keys_new, const uint16_t* values_new,
                                       uint16_t* output,
                                       size_t batch_size, size_t seq_len, size_t cache_len,
                                       size_t new_len, size_t num_q_heads, size_t num_kv_heads,
                                       size_t head_dim, float scale, size_t position_offset,
                                       int is_causal, size_t group_size);

/* Ops between the hot kernels of a decoder block.  `ranges` is the number of
 * contiguous pieces the reference's thread pool would cut the call into (1 =
 * one piece: groups of 8 from the start take the vector formula, the last
 * n % 8 elements the scalar libm one). */
void oracle_rms_norm_f16(const uint16_t* input, const uint16_t* weight, uint16_t* output,
                         size_t batch_size, size_t dims, float eps);
void oracle_rope_tables(uint16_t* cos_t, uint16_t* sin_t, size_t pos0, size_t npos,
                        size_t head_dim, float theta);
void oracle_rope_f16(const uint16_t* input, uint16_t* output, size_t batch_size, size_t seq_len,
                     size_t num_heads, size_t head_dim, size_t start_pos, float theta);
void oracle_silu_f16(const uint16_t* in, uint16_t* out, size_t n, size_t ranges);
void oracle_gelu_f16(const uint16_t* in, uint16_t* out, size_t n, size_t ranges);
void oracle_binary_f16(int op, const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n);
void oracle_add_f16_clipped(const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n);
int  oracle_embedding_int8(const int8_t* emb, const uint16_t* scales, const int32_t* indices,
                           uint16_t* out, size_t num_indices, size_t vocab_size, size_t hidden_dim,
                           size_t group_size);
uint32_t oracle_sample_greedy_f16(const uint16_t* logits, size_t vocab_size);
uint32_t oracle_sample_f16(const uint16_t* logits, size_t vocab_size, float temperature, float top_p,
                           size_t top_k, float sample, uint32_t* history, size_t* hist_count);
float oracle_sample_uniform(uint32_t seed);
float oracle_logits_entropy_f16(const uint16_t* logits, size_t vocab_size);
int  oracle_embedding_f16(const uint16_t* emb, const int32_t* indices, uint16_t* out,
                          size_t num_indices, size_t vocab_size, size_t hidden_dim);

void oracle_scalar_op_f16(const uint16_t* in, uint16_t* out, size_t n, float scalar, int op);

/* LFM2's short convolution (PARITY UNPINNED: the reference's tests hold nothing
 * for the causal convolution or the conv cache; checked against float64 sums). */
void oracle_conv1d_causal_depthwise_f16(const uint16_t* input, const uint16_t* weight, uint16_t* output,
                                        size_t N, size_t L, size_t C, size_t K, size_t dilation);
void oracle_lfm2_conv_gate_f16(const uint16_t* in_proj, const uint16_t* weight, uint16_t* state,
                               size_t* head, size_t* count, uint16_t* gated,
                               size_t L, size_t C, size_t K, size_t W);

#ifdef __cplusplus
}
#endif
#endif
