/*
 * cactus_hip.h -- C ABI of the MI355X (gfx950) backend for Cactus's quantized
 * hot path: grouped INT8/INT4 weight matmul, FP16 matmul, activation / KV
 * quantization and attention over the INT8 KV cache.
 *
 * Each entry point replaces one free function of the reference's kernel ABI
 * (cactus/kernel/kernel.h); the line it replaces is cited beside it.  Halves
 * are IEEE binary16 bit patterns (uint16_t) so the header is plain C.
 *
 * Two families:
 *   cactus_hip_<op>_dev(...)  operands are DEVICE pointers already resident in
 *                             HBM; work is enqueued on `stream` (a hipStream_t,
 *                             or NULL for HIP's null stream) and the call
 *                             returns without synchronising.  This is the hot
 *                             path.
 *   cactus_hip_<op>(...)      operands are HOST pointers with exactly the
 *                             reference's signature; results are complete on
 *                             return (the reference's contract).  The weights and
 *                             weight scales of the quantized matmuls are uploaded
 *                             once and cached by (device, address) (they are mmap'd
 *                             file regions that live as long as the graph:
 *                             graph_io.cpp:43-45) and re-validated per call by size
 *                             and a sampled content signature; every other operand,
 *                             cactus_hip_matmul_f16's b_transposed included, is
 *                             staged per call.
 *
 * Errors: the reference kernels return void and cannot fail.  The *_dev family
 * returns 0 or a negative CACTUS_HIP_E* code; the host family records the
 * failure, readable with cactus_hip_last_error(), so that the graph layer can
 * throw (graph_ops_nn.cpp's std::runtime_error convention).
 */
#ifndef CACTUS_HIP_H
#define CACTUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t cactus_half_t;

#define CACTUS_HIP_OK            0
#define CACTUS_HIP_EINVAL       -1   /* shape/argument the kernels do not accept */
#define CACTUS_HIP_ERUNTIME     -2   /* a HIP runtime call failed */
#define CACTUS_HIP_ENOMEM       -3   /* device allocation failed */

/* KV quantization group (cactus/kernel/kernel.h:18). */
#define CACTUS_KV_QUANT_GROUP_SIZE 128

/* ---------------------------------------------------------------- runtime */

/* Select the device this thread's calls run on and create the library stream. */
int  cactus_hip_init(int device);
int  cactus_hip_device_count(void);
const char* cactus_hip_last_error(void);
void cactus_hip_clear_error(void);

void* cactus_hip_malloc(size_t bytes);
void  cactus_hip_free(void* dptr);
int   cactus_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int   cactus_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int   cactus_hip_synchronize(void* stream);

/* The library keeps one grow-only device workspace per device (split-KV records of the attention
 * kernels, quantized activation rows of the FP16-activation matmuls with more than 4 rows, the
 * slice results of cactus_hip_logits_entropy_f16_dev: 24 bytes per row and slice, at most 64 slices).  It
 * grows by allocating, which HIP stream capture forbids: reserve the largest size a captured
 * sequence needs beforehand (or run the sequence once uncaptured).  An outgrown workspace stays
 * allocated, so graphs captured earlier remain valid.  No counterpart in the reference. */
int   cactus_hip_workspace_reserve(size_t bytes);

/* Host-pointer family: drop cached device copies of weights (all, or the one
 * registered under host_ptr on the current device).  A weight passed by host pointer is uploaded on
 * first use and must not change while cached: a later call re-checks the whole buffer up to 1 MiB but
 * only 64 samples of a larger one, so a partial rewrite in place can go unnoticed - release it first. */
void cactus_hip_release_weights(const void* host_ptr);

/* Tile-major twins of device-resident weights.  No counterpart in the reference, whose weights stay in the
 * row-major [N, K] layout of its files (graph_io.cpp:49-73); the matmul entries below keep taking that layout.
 * Call this once when a weight matrix (kind 0: B int8[N, K]; 1: packed INT4 uint8[N, K/2]; 2: half[N, K], the
 * b_transposed operand of cactus_hip_matmul_f16_dev) has been made resident: the library allocates a second copy in which every 16 rows x 128 bytes unit of the streaming
 * kernels is one contiguous 2 KB run (a whole 16-row tile is one contiguous stretch), and every later call with
 * M <= 16 rows that names B with the same N and K (the _dev matmuls, their _f16act, _batch, _norm / _silu /
 * _gelu forms) reads the twin instead: same integer sums, same fp32 chain, same bits, about a fifth less
 * time.  Launches with more rows read B itself, which must stay as it is.  B must not change while it has a
 * twin: call again to refresh the twin after rewriting B, or release it.  The call allocates (not under stream
 * capture); K a multiple of 128 (INT4: 256, FP16: 64), B 16-byte aligned and below 2 GiB.  release(NULL) drops every twin;
 * release synchronizes the device before it frees. */
int cactus_hip_weights_pack_dev(const void* B, size_t N, size_t K, int kind, void* stream);
int cactus_hip_weights_release_packed_dev(const void* B);
int cactus_hip_weights_is_packed_dev(const void* B);

/* ------------------------------------------------- device-resident hot path */

/* cactus_matmul_int8 (kernel.h:41-43; kernel_gemm.cpp:128-327).
 * C[m,n] = fp16( A_scales[m] * sum_g float(sum_{k in g} A[m,k]*B[n,k]) * B_scales[n,g] )
 * A int8[M,K], B int8[N,K] (K contiguous), B_scales half[N, K/group_size], C half[M,N].
 * K/group_size whole groups are used, a K tail is ignored as in the reference. */
int cactus_hip_matmul_int8_dev(const int8_t* A, const float* A_scales,
                               const int8_t* B, const cactus_half_t* B_scales,
                               cactus_half_t* C, size_t M, size_t K, size_t N,
                               size_t group_size, void* stream);

/* cactus_matmul_int4 (kernel.h:45-47; kernel_gemm.cpp:329-529).  B_packed
 * uint8[N, K/2]: byte j of a row holds element 2j in its low nibble and 2j+1 in
 * its high nibble, two's-complement 4-bit (kernel_utils.h:165-185). */
int cactus_hip_matmul_int4_dev(const int8_t* A, const float* A_scales,
                               const uint8_t* B_packed, const cactus_half_t* B_scales,
                               cactus_half_t* C, size_t M, size_t K, size_t N,
                               size_t group_size, void* stream);

/* The exact INT32 group sums [M, N, K/group_size] of the two kernels above
 * (kernel_gemm.cpp:173, 404: group_acc), for bit-exact parity checks. */
int cactus_hip_matmul_int8_group_sums_dev(const int8_t* A, const int8_t* B, int32_t* sums,
                                          size_t M, size_t K, size_t N, size_t group_size,
                                          void* stream);
int cactus_hip_matmul_int4_group_sums_dev(const int8_t* A, const uint8_t* B_packed, int32_t* sums,
                                          size_t M, size_t K, size_t N, size_t group_size,
                                          void* stream);

/* The FP16-activation branch of compute_matmul_node (graph_ops_nn.cpp:166-177 for
 * INT8 weights, :196-207 for INT4): quantize_activations_fp16_to_int8 on X
 * half[M,K] followed by cactus_matmul_int8 / cactus_matmul_int4, same results bit
 * for bit.  For M <= 4 (decode) the row quantization happens inside the GEMV kernel
 * and the call is a single launch; otherwise the quantized rows go through the
 * library's device workspace (calls on one device that need it must share a stream). */
int cactus_hip_matmul_int8_f16act_dev(const cactus_half_t* X, const int8_t* B,
                                      const cactus_half_t* B_scales, cactus_half_t* C,
                                      size_t M, size_t K, size_t N, size_t group_size, void* stream);
int cactus_hip_matmul_int4_f16act_dev(const cactus_half_t* X, const uint8_t* B_packed,
                                      const cactus_half_t* B_scales, cactus_half_t* C,
                                      size_t M, size_t K, size_t N, size_t group_size, void* stream);

/* Several independent FP16-activation matmuls enqueued together: what a decoder
 * block issues back to back on one activation row (q/k/v: model_qwen.cpp:63-65,
 * gate/up: :139-141), or any list of small decode matmuls.  Results are those of
 * one cactus_hip_matmul_int{8,4}_f16act_dev call per entry; entries with M <= 4 are
 * merged into single launches (up to 16 per launch), which removes the per-launch
 * latency that bounds small GEMVs.  Outputs must not alias other entries' inputs. */
typedef struct cactus_hip_matmul_problem {
    const cactus_half_t* X;        /* half[M, K] activations                          */
    const void* B;                 /* int8[N, K] or packed uint8[N, K/2] weights      */
    const cactus_half_t* B_scales; /* half[N, K/group_size]                           */
    cactus_half_t* C;              /* half[M, N] output                               */
    size_t M, K, N;
} cactus_hip_matmul_problem;
int cactus_hip_matmul_int8_f16act_batch_dev(const cactus_hip_matmul_problem* problems, size_t count,
                                            size_t group_size, void* stream);
int cactus_hip_matmul_int4_f16act_batch_dev(const cactus_hip_matmul_problem* problems, size_t count,
                                            size_t group_size, void* stream);

/* Several weight matrices against ONE quantized activation matrix: the q/k/v (model_qwen.cpp:
 * 55-57) or gate/up (:112-113) projections of a prefill or batched-decode step.  Results are those
 * of one cactus_hip_matmul_int{8,4}_dev call per entry; for M > 16 and group_size 128 up to 8
 * entries share a launch (small N leaves most of the chip idle when launched alone). */
typedef struct cactus_hip_matmul_weights {
    const void* B;                 /* int8[N, K] or packed uint8[N, K/2]  */
    const cactus_half_t* B_scales; /* half[N, K/group_size]               */
    cactus_half_t* C;              /* half[M, N] output                   */
    size_t N;
} cactus_hip_matmul_weights;
int cactus_hip_matmul_int8_shared_dev(const int8_t* A, const float* A_scales,
                                      const cactus_hip_matmul_weights* weights, size_t count, size_t M,
                                      size_t K, size_t group_size, void* stream);
int cactus_hip_matmul_int4_shared_dev(const int8_t* A, const float* A_scales,
                                      const cactus_hip_matmul_weights* weights, size_t count, size_t M,
                                      size_t K, size_t group_size, void* stream);

/* The matmuls of a decoder block together with the op that produces their FP16 input, one launch
 * for M <= 4 (a decode step; otherwise a producer + quantization launch followed by the matmuls):
 *   _norm_: rows = rms_norm(input [+ addend]) * norm_weight -- the residual add and norm in front
 *           of q/k/v or gate/up (model_qwen.cpp:51-57, 109-113); input + addend is also written
 *           to sum_output (required when addend is given; must not alias input or addend).
 *   _silu_ / _gelu_: rows = act(gate) * up -- in front of the down projection (:114-116).
 * Each row is then quantized as compute_matmul_node quantizes FP16 activations
 * (graph_ops_nn.cpp:35-66, 163-165) and multiplied with every weight entry.  Results equal
 * cactus_hip_add_rms_norm_f16_dev / cactus_hip_{silu,gelu}_multiply_f16_dev followed by
 * cactus_hip_matmul_int{8,4}_f16act_dev per entry, bit for bit.  K <= 16384. */
int cactus_hip_matmul_int8_norm_f16act_dev(const cactus_half_t* input, const cactus_half_t* addend,
                                           const cactus_half_t* norm_weight, cactus_half_t* sum_output,
                                           float eps, const cactus_hip_matmul_weights* weights,
                                           size_t count, size_t M, size_t K, size_t group_size,
                                           void* stream);
int cactus_hip_matmul_int4_norm_f16act_dev(const cactus_half_t* input, const cactus_half_t* addend,
                                           const cactus_half_t* norm_weight, cactus_half_t* sum_output,
                                           float eps, const cactus_hip_matmul_weights* weights,
                                           size_t count, size_t M, size_t K, size_t group_size,
                                           void* stream);
int cactus_hip_matmul_int8_silu_f16act_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                           const cactus_hip_matmul_weights* weights, size_t count,
                                           size_t M, size_t K, size_t group_size, void* stream);
int cactus_hip_matmul_int4_silu_f16act_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                           const cactus_hip_matmul_weights* weights, size_t count,
                                           size_t M, size_t K, size_t group_size, void* stream);
int cactus_hip_matmul_int8_gelu_f16act_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                           const cactus_hip_matmul_weights* weights, size_t count,
                                           size_t M, size_t K, size_t group_size, void* stream);
int cactus_hip_matmul_int4_gelu_f16act_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                           const cactus_hip_matmul_weights* weights, size_t count,
                                           size_t M, size_t K, size_t group_size, void* stream);

/* cactus_matmul_f16 (kernel.h:49-50; kernel_gemm.cpp:102-126): C = A * B^T,
 * a half[M,K], b_transposed half[N,K].  Accumulates in fp32 (the reference
 * accumulates in fp16 lanes), so it is compared by tolerance.  Any shape is taken; K a multiple of 64
 * with 16-byte aligned operands runs on the pipelined tile kernel (about three times the rate). */
int cactus_hip_matmul_f16_dev(const cactus_half_t* a, const cactus_half_t* b_transposed,
                              cactus_half_t* c, size_t M, size_t K, size_t N, void* stream);

/* quantize_activations_fp16_to_int8 (graph_ops_nn.cpp:35-66): per row
 * scale = max(absmax/127, 1e-10), q = sat8(rne(clamp(x * (1/scale)))). */
int cactus_hip_quantize_activations_dev(const cactus_half_t* src, int8_t* dst, float* scales,
                                        size_t M, size_t K, void* stream);

/* cactus_fp16_max_abs (kernel.h:149); result written to *out_dev (device float). */
int cactus_hip_fp16_max_abs_dev(const cactus_half_t* src, size_t count, float* out_dev,
                                void* stream);
/* cactus_fp16_to_int8 (kernel.h:148; kernel_quants.cpp:158-194). */
int cactus_hip_fp16_to_int8_dev(const cactus_half_t* src, int8_t* dst, size_t count, float scale,
                                void* stream);

/* cactus_quantize_kv_fp16_to_int8 (kernel.h:151-156; kernel_quants.cpp:266-293).
 * src half[seq,kv_heads,head_dim] -> dst int8 same shape, scales float[seq,kv_heads,ceil(D/gs)]. */
int cactus_hip_quantize_kv_fp16_to_int8_dev(const cactus_half_t* src, int8_t* dst, float* scales,
                                            size_t seq_len, size_t kv_heads, size_t head_dim,
                                            size_t group_size, void* stream);

/* K and V of one step in a single launch: two cactus_quantize_kv_fp16_to_int8 calls
 * (engine_cache.cpp:253-263 makes them back to back); dst / scales may point into cache slots. */
int cactus_hip_quantize_kv_pair_fp16_to_int8_dev(const cactus_half_t* k_src, const cactus_half_t* v_src,
                                                 int8_t* k_dst, int8_t* v_dst, float* k_scales,
                                                 float* v_scales, size_t seq_len, size_t kv_heads,
                                                 size_t head_dim, size_t group_size, void* stream);

/* cactus_unpack_int4_to_int8 (kernel.h:163; kernel_quants.cpp:295-351). */
int cactus_hip_unpack_int4_to_int8_dev(const uint8_t* packed, int8_t* unpacked,
                                       size_t unpacked_count, void* stream);

/* ---- ops between the hot kernels of a decoder block (device pointers) ---- */

/* cactus_rms_norm_f16 (kernel.h:72-73; kernel_attention.cpp:484-563): input half[batch,dims],
 * weight half[dims]; fp32 sum of squares in the reference's lane order, fp16 products. */
int cactus_hip_rms_norm_f16_dev(const cactus_half_t* input, const cactus_half_t* weight,
                                cactus_half_t* output, size_t batch_size, size_t dims, float eps,
                                void* stream);

/* cactus_rope_f16 (kernel.h:75-76; kernel_attention.cpp:614-676): half-split rotation on
 * half[batch,seq,heads,head_dim], token s at position start_pos + s.  The cos/sin tables are
 * built on the host as the reference builds them (:594-604) and kept per device; a call that
 * needs more positions than the table holds grows it (a blocking allocation and copy), which
 * cactus_hip_rope_reserve does ahead of time (e.g. before stream capture). */
int cactus_hip_rope_f16_dev(const cactus_half_t* input, cactus_half_t* output, size_t batch_size,
                            size_t seq_len, size_t num_heads, size_t head_dim, size_t start_pos,
                            float theta, void* stream);
int cactus_hip_rope_reserve(size_t head_dim, float theta, size_t max_positions);

/* cactus_silu_f16 / cactus_gelu_f16 (kernel.h:81, 83; kernel_nn.cpp:12-91): the call is one
 * piece: whole groups of 8 from the start take the reference's polynomial / rational formulas,
 * the last n % 8 elements its scalar expf / tanhf ones. */
int cactus_hip_silu_f16_dev(const cactus_half_t* input, cactus_half_t* output, size_t num_elements, void* stream);
int cactus_hip_gelu_f16_dev(const cactus_half_t* input, cactus_half_t* output, size_t num_elements, void* stream);

/* cactus_add_f16, cactus_add_f16_clipped, cactus_subtract_f16, cactus_multiply_f16,
 * cactus_divide_f16 (kernel.h:20-24; kernel_blas.cpp:187-343): one fp16 operation per element;
 * the clipped add sums in fp32 and clamps to +-65500. */
int cactus_hip_add_f16_dev(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements, void* stream);
int cactus_hip_add_f16_clipped_dev(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements, void* stream);
int cactus_hip_subtract_f16_dev(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements, void* stream);
int cactus_hip_multiply_f16_dev(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements, void* stream);
int cactus_hip_divide_f16_dev(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements, void* stream);

/* cactus_scalar_op_f16 (kernel.h:7-16, 39; kernel_scalar.cpp:7-154) for op_type ADD (0), SUBTRACT (1),
 * MULTIPLY (2), DIVIDE (3): the scalar is rounded to binary16, then one fp16 operation per element.
 * EXP / SQRT / COS / SIN (4..7) are not on the device: CACTUS_HIP_EINVAL. */
int cactus_hip_scalar_op_f16_dev(const cactus_half_t* input, cactus_half_t* output, size_t num_elements,
                                 float scalar_value, int op_type, void* stream);

/* Sequences of the calls above that a decoder block makes back to back, in one launch each, with
 * the arithmetic of the separate calls (results are bit-identical to making them one by one):
 *   add_rms_norm    sum = cactus_add_f16(a, b); norm = cactus_rms_norm_f16(sum, weight)
 *                   (model_qwen.cpp:125-126; :128 with the next block's :123)
 *   qk_norm_rope    per-head cactus_rms_norm_f16 of q [batch*seq*Hq, D] and k [batch*seq*Hkv, D],
 *                   then cactus_rope_f16 of both (model_qwen.cpp:63-81)
 *   silu_multiply / gelu_multiply   cactus_multiply_f16(cactus_silu_f16(gate), up)
 *                   (model_qwen.cpp:114-115; model_gemma.cpp uses gelu) */
int cactus_hip_add_rms_norm_f16_dev(const cactus_half_t* a, const cactus_half_t* b,
                                    const cactus_half_t* weight, cactus_half_t* sum_output,
                                    cactus_half_t* norm_output, size_t batch_size, size_t dims, float eps,
                                    void* stream);
int cactus_hip_qk_norm_rope_f16_dev(const cactus_half_t* q, const cactus_half_t* k,
                                    const cactus_half_t* q_weight, const cactus_half_t* k_weight,
                                    cactus_half_t* q_output, cactus_half_t* k_output, size_t batch_size,
                                    size_t seq_len, size_t num_q_heads, size_t num_kv_heads,
                                    size_t head_dim, float eps, size_t start_pos, float theta,
                                    void* stream);
/* cactus_hip_qk_norm_rope_f16_dev plus the cache append of the step (KVCache::update_from_graph,
 * engine_cache.cpp:234-267), one launch: the rotated k rows and the v rows half[batch, seq, Hkv, D]
 * are quantized as cactus_quantize_kv_fp16_to_int8 does and stored at slot slots[b] + s of sequence
 * b (k_cache / v_cache: sequence b at b * cache_batch_stride elements, k_scales / v_scales at
 * b * scale_batch_stride; a slot outside [0, max_slots) stores nothing); slots == NULL stores row
 * for row into int8[batch * seq, Hkv, D] / float[batch * seq, Hkv, G].  k_output still receives the
 * FP16 rows the attention of this step reads. */
int cactus_hip_qk_norm_rope_kv_append_f16_dev(const cactus_half_t* q, const cactus_half_t* k,
                                              const cactus_half_t* v, const cactus_half_t* q_weight,
                                              const cactus_half_t* k_weight, cactus_half_t* q_output,
                                              cactus_half_t* k_output, int8_t* k_cache, int8_t* v_cache,
                                              float* k_scales, float* v_scales, const int32_t* slots,
                                              size_t batch_size, size_t seq_len, size_t num_q_heads,
                                              size_t num_kv_heads, size_t head_dim, float eps,
                                              size_t start_pos, float theta, size_t group_size,
                                              size_t cache_batch_stride, size_t scale_batch_stride,
                                              size_t max_slots, void* stream);
int cactus_hip_silu_multiply_f16_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                     cactus_half_t* output, size_t num_elements, void* stream);
int cactus_hip_gelu_multiply_f16_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                     cactus_half_t* output, size_t num_elements, void* stream);

/* cactus_conv1d_causal_depthwise_f16 (kernel.h:106-111; kernel_conv.cpp:15-110): input
 * half[N, L, C], weight half[C, K], output[n, t, c] = sum_k weight[c, K-1-k] * input[n, t - k*dilation, c]
 * over the taps that lie inside the sequence, fp32 sums in the reference's order, one rounding. */
int cactus_hip_conv1d_causal_depthwise_f16_dev(const cactus_half_t* input, const cactus_half_t* weight,
                                               cactus_half_t* output, size_t N, size_t L, size_t C,
                                               size_t K, size_t dilation, void* stream);
/* The LFM2 short-convolution block between its two projections (LFM2Model::build_conv1d,
 * model_lfm2.cpp:136-204, with ConvCache, engine_cache.cpp:494-597), one launch:
 * in_proj half[num_seqs, L, 3, C] holds B, C-gate, X of each row; Bx = B * X;
 * y = causal depthwise convolution (dilation 1) over the sequence's last W rows of Bx from earlier
 * calls followed by this call's L rows; gated[num_seqs, L, C] = C-gate * y.  state
 * half[num_seqs, W, C] carries those last W rows in temporal order between calls (W = conv_L_cache - 1
 * in the reference) and must start as zeros, which stands for "no rows held yet": results equal the
 * reference's, whose circular buffer convolves over only the rows it holds.  The call leaves the new
 * last W rows in state.  W == 0: no state (state may be NULL). */
int cactus_hip_lfm2_conv_gate_f16_dev(const cactus_half_t* in_proj, const cactus_half_t* weight,
                                      cactus_half_t* state, cactus_half_t* gated, size_t num_seqs, size_t L,
                                      size_t C, size_t K, size_t W, void* stream);

/* cactus_hip_lfm2_conv_gate_f16_dev plus the activation quantization of the out-projection that consumes
 * the gated rows (quantize_activations_fp16_to_int8, graph_ops_nn.cpp:35-66), one launch: q_output
 * int8[num_seqs * L, C], scales_output float[num_seqs * L], bit for bit what
 * cactus_hip_quantize_activations_dev gives on the gated rows; gated may be NULL.  C a multiple of 512 up
 * to 2048 and K in 2..4 (LFM2: 2048, 3), 16-byte aligned buffers; otherwise CACTUS_HIP_EINVAL and the two
 * separate calls apply. */
int cactus_hip_lfm2_conv_gate_quant_f16_dev(const cactus_half_t* in_proj, const cactus_half_t* weight,
                                            cactus_half_t* state, cactus_half_t* gated, int8_t* q_output,
                                            float* scales_output, size_t num_seqs, size_t L, size_t C, size_t K,
                                            size_t W, void* stream);

/* A row producer followed by the activation quantization of the matmul that consumes it
 * (quantize_activations_fp16_to_int8, graph_ops_nn.cpp:35-66), one launch, one workgroup per row:
 * results equal cactus_hip_add_rms_norm_f16_dev / cactus_hip_silu_multiply_f16_dev /
 * cactus_hip_gelu_multiply_f16_dev followed by cactus_hip_quantize_activations_dev bit for bit.
 * addend may be NULL (plain norm; sum_output unused); norm_output / act_output may be NULL when
 * only the quantized row is needed.  dims <= 16384. */
int cactus_hip_add_rms_norm_quant_f16_dev(const cactus_half_t* input, const cactus_half_t* addend,
                                          const cactus_half_t* weight, cactus_half_t* sum_output,
                                          cactus_half_t* norm_output, int8_t* q_output,
                                          float* scales_output, size_t rows, size_t dims, float eps,
                                          void* stream);
int cactus_hip_silu_multiply_quant_f16_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                           cactus_half_t* act_output, int8_t* q_output,
                                           float* scales_output, size_t rows, size_t dims, void* stream);
int cactus_hip_gelu_multiply_quant_f16_dev(const cactus_half_t* gate, const cactus_half_t* up,
                                           cactus_half_t* act_output, int8_t* q_output,
                                           float* scales_output, size_t rows, size_t dims, void* stream);

/* compute_embedding_node (graph_ops_tensor.cpp:226-320): output row i = row indices[i] of the
 * table; INT8 rows with per-group fp16 scales dequantize as half(float(q) * float(scale))
 * (:243-254), FP16 rows are copied (:301-316).  indices: device int32[num_indices].  An index
 * outside [0, vocab_size) writes nothing for that row and stores 1 to *bad_index_flag (device
 * int, may be NULL; the caller clears it) -- the reference throws there (:261-263). */
/* The greedy branch of cactus_sample_f16 (kernel.h; kernel_nn.cpp:446-466: temperature 0, no
 * top-p / top-k): output[r] = first index of the maximum of logits[r, :] (NaNs never win; a NaN
 * in element 0 gives 0; vocab_size 0 gives 0).  The stochastic branches (:468-651: host RNG,
 * process-wide token history) are cactus_hip_sample_f16_dev below. */
int cactus_hip_sample_greedy_f16_dev(const cactus_half_t* logits, uint32_t* output, size_t rows,
                                     size_t vocab_size, void* stream);

/* cactus_sample_f16 (kernel.h:130-133; kernel_nn.cpp:446-651) for `rows` independent logits rows
 * (the reference samples one row per call, compute_sample_node graph_ops_sample.cpp:24-32):
 * temperature scaling in fp16, repetition penalty 1.1 per occurrence in the row's token history,
 * top-k, top-p, softmax and the inverse-CDF draw, in the reference's order of fp32 operations.
 *   - temperature == 0 && top_p <= 0 && top_k == 0 is the greedy branch (draws, history and
 *     workspace may then be NULL).
 *   - draws: device float[rows], the uniform number of each row (the reference draws it from
 *     std::mt19937(random_seed), :617-620: cactus_hip_sample_uniform gives that number for a seed).
 *   - history: device uint32[rows, 128] and history_counts: device uint32[rows] stand for the
 *     reference's process-wide `token_history` (:498-500), one per row; the sampled token is
 *     appended (oldest dropped beyond 128).  Both NULL: no penalty, nothing recorded.
 *   - the logit bias arguments of the reference are not taken: it overwrites the biased values
 *     before use (:473-496), so they never change a result.
 *   - workspace: device half[rows, vocab_size] scratch.
 *   - a row of which at most cactus_hip_sample_max_survivors() logits survive top-k is sampled by one
 *     launch.  Rows with more survivors (top_k == 0: every finite logit survives; a top_k above that
 *     number; ties with the k-th value, which all survive, :526-533) are finished by a second, slower
 *     launch (about a millisecond for 150k survivors: the reference's sequential fp32 sums in
 *     descending and in index order are kept) that works in the library workspace
 *     (cactus_hip_workspace_reserve before stream capture: 25 MB); it is enqueued whenever vocab_size
 *     exceeds that number and returns at once for rows that are done.  *status (device int, may be
 *     NULL) is not written by this entry.
 *   - equal logits keep ascending index order in the top-p pass (the reference leaves it to
 *     std::sort); exp is computed in double and rounded (the reference uses its libm's expf). */
int cactus_hip_sample_f16_dev(const cactus_half_t* logits, uint32_t* output, size_t rows,
                              size_t vocab_size, float temperature, float top_p, size_t top_k,
                              const float* draws, uint32_t* history, uint32_t* history_counts,
                              cactus_half_t* workspace, int* status, void* stream);
size_t cactus_hip_sample_max_survivors(void);
float cactus_hip_sample_uniform(uint32_t seed);   /* host computation, no device work */

/* The confidence figure of Model::decode (engine_model.cpp:255-290): entropy of softmax(row) in
 * double precision divided by log(vocab_size), one float per row.  The reference adds in element
 * order; partial sums are combined as a tree here (the float result may differ in the last bit).
 * A row is spread over up to 64 workgroups whose slice results pass through the library workspace
 * (calls on one device share a stream, as for the attention records). */
int cactus_hip_logits_entropy_f16_dev(const cactus_half_t* logits, float* output, size_t rows,
                                      size_t vocab_size, void* stream);

/* The index conversion of compute_embedding_node (graph_ops_tensor.cpp:257-275: indices held as
 * FP32 or INT8 values, `static_cast<size_t>(indices[i])`): precision 2 = float, 0 = int8; values
 * that are not a valid index (negative, NaN) become -1. */
int cactus_hip_indices_to_int32_dev(const void* indices, int precision, int32_t* output, size_t count,
                                    void* stream);
int cactus_hip_embedding_int8_dev(const int8_t* embeddings, const cactus_half_t* scales,
                                  const int32_t* indices, cactus_half_t* output, size_t num_indices,
                                  size_t vocab_size, size_t hidden_dim, size_t group_size,
                                  int* bad_index_flag, void* stream);
int cactus_hip_embedding_f16_dev(const cactus_half_t* embeddings, const int32_t* indices,
                                 cactus_half_t* output, size_t num_indices, size_t vocab_size,
                                 size_t hidden_dim, int* bad_index_flag, void* stream);

/* cactus_attention_f16 (kernel.h:87-90; kernel_attention.cpp:10-255).
 * Q half[B,S,Hq,D], K/V half[B,KV,Hkv,D], O half[B,S,Hq,D]; mask half[B,S,KV]
 * (0 = masked) or NULL.  scale 0 means 1/sqrt(D). */
int cactus_hip_attention_f16_dev(const cactus_half_t* queries, const cactus_half_t* keys,
                                 const cactus_half_t* values, cactus_half_t* output,
                                 size_t batch_size, size_t seq_len, size_t kv_seq_len,
                                 size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
                                 float scale, const cactus_half_t* mask, size_t position_offset,
                                 size_t window_size, int is_causal, void* stream);

/* cactus_attention_hybrid_int8_fp16 (kernel.h:92-104; kernel_attention.cpp:257-482).
 * Cached K/V int8[B, cache_len, Hkv, D] with scales float[cache_len, Hkv, ceil(D/gs)]
 * (the reference indexes scales WITHOUT a batch stride, kernel_attention.cpp:353,425;
 * so does this entry point), new K/V half[B,new_len,Hkv,D]. */
int cactus_hip_attention_hybrid_int8_fp16_dev(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales,
    const cactus_half_t* keys_new, const cactus_half_t* values_new, cactus_half_t* output,
    size_t batch_size, size_t seq_len, size_t cache_len, size_t new_len,
    size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
    float scale, size_t position_offset, int is_causal, size_t group_size, void* stream);

/* cactus_attention_hybrid_int8_fp16 (kernel.h:92-104) followed by the activation quantization that the
 * output projection applies to the attention node's output (graph_ops_nn.cpp:35-66; model_qwen.cpp:103-106):
 * output as above, and output_q int8[B * S, Hq * D] (8-byte aligned) with output_scales float[B * S] equal to
 * cactus_hip_quantize_activations_dev(output, ...) bit for bit.  For a prefill chunk whose key range is
 * split over workgroups the merge of the splits and the quantization are one launch (one workgroup per
 * token); other shapes run the quantization kernel behind the attention. */
int cactus_hip_attention_hybrid_int8_fp16_quant_dev(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales,
    const cactus_half_t* keys_new, const cactus_half_t* values_new, cactus_half_t* output,
    int8_t* output_q, float* output_scales,
    size_t batch_size, size_t seq_len, size_t cache_len, size_t new_len,
    size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
    float scale, size_t position_offset, int is_causal, size_t group_size, void* stream);

/* No counterpart in the reference (its kernels borrow nothing but their arguments and the host
 * heap).  For callers that cannot allocate device memory: code under HIP stream capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Same arguments and the same results, bit for
 * bit, as cactus_hip_attention_hybrid_int8_fp16_dev, but the split-KV records (running maximum,
 * sum and unnormalised output row per query head and key range) go to a fixed 16 MiB array that
 * is part of the library's code object instead of the grow-only workspace, so the call makes no
 * hipMalloc / hipFree and calls no allocating routine.  A shape whose records need more than
 * cactus_hip_noalloc_scratch_bytes() is rejected with CACTUS_HIP_EINVAL (decode steps need
 * Hq * splits * (D + 2) * 4 bytes per sequence with at most 256 splits: 2.1 MiB for one Qwen3
 * sequence at any context length; nothing when the batch alone fills the chip).  Each device
 * holds one array for the life of the process; cactus_hip_init(device) resolves its address
 * (call it before capturing).  Calls on one device share the array: two that overlap (different
 * streams, or threads) race.  Nothing in it has to be cleared between calls. */
size_t cactus_hip_noalloc_scratch_bytes(void);
/* Bytes of split-KV records the hybrid attention entries keep for this shape with 16-byte-aligned operands (0: none): what the
 * allocation-free entry compares with cactus_hip_noalloc_scratch_bytes(), and what
 * cactus_hip_workspace_reserve needs at least before the allocating entry is captured. */
size_t cactus_hip_attention_hybrid_records_bytes(size_t batch_size, size_t seq_len, size_t cache_len, size_t new_len,
                                                 size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
                                                 size_t position_offset, int is_causal, size_t group_size);
int cactus_hip_attention_hybrid_int8_fp16_noalloc_dev(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales,
    const cactus_half_t* keys_new, const cactus_half_t* values_new, cactus_half_t* output,
    size_t batch_size, size_t seq_len, size_t cache_len, size_t new_len,
    size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
    float scale, size_t position_offset, int is_causal, size_t group_size, void* stream);

/* Batched-decode form of the same kernel for independent sequences (one per
 * batch row): per-sequence cache length and position offset, and cache/scales
 * addressed with explicit per-sequence strides (in elements).  Has no
 * counterpart in the reference, whose engine is batch-1 (model_qwen.cpp:74-76);
 * with one sequence it computes exactly what the entry point above computes. */
int cactus_hip_attention_hybrid_int8_fp16_batched_dev(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales,
    const cactus_half_t* keys_new, const cactus_half_t* values_new, cactus_half_t* output,
    size_t batch_size, size_t seq_len, const int32_t* cache_lens, size_t new_len,
    size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
    float scale, const int32_t* position_offsets, size_t group_size,
    size_t cache_batch_stride, size_t scale_batch_stride, size_t max_cache_len, void* stream);

/* A decode step's attention in one launch, from the q / k / v projections' rows to the attention output
 * (model_qwen.cpp:63-99 for one new token per sequence): per-head RMS norm of q and k (q_weight, k_weight),
 * RoPE at positions[b], the rotated key and the value quantized into slot cache_lens[b] of sequence b's cache
 * (KVCache::update_from_graph, engine_cache.cpp:234-267), and cactus_attention_hybrid_int8_fp16 over the
 * cache_lens[b] cached tokens plus the new one.  Results equal cactus_hip_qk_norm_rope_kv_append_f16_dev (or
 * _seq_ with window 0) followed by cactus_hip_attention_hybrid_int8_fp16_batched_dev, bit for bit, one launch
 * (plus the split merge) instead of two.  q half[B, Hq, D], k / v half[B, Hkv, D], output half[B, Hq, D];
 * caches addressed as in the batched attention (max_cache_len: an upper bound of cache_lens[b], at most
 * max_slots; a sequence whose cache is full appends nothing); RoPE tables as
 * cactus_hip_qk_norm_rope_seq_f16_dev (max_position).  For
 * head_dim 64 / 128 with group_size >= head_dim or head_dim 256 with group_size 128, up to 16 query heads per kv head, 16-byte aligned rows and at
 * least one cached token per call; other shapes take the two separate calls. */
int cactus_hip_attention_decode_fused_f16_dev(
    const cactus_half_t* q, const cactus_half_t* k, const cactus_half_t* v, const cactus_half_t* q_weight,
    const cactus_half_t* k_weight, int8_t* k_cache, int8_t* v_cache, float* k_scales, float* v_scales,
    cactus_half_t* output, size_t batch_size, const int32_t* cache_lens, const int32_t* positions,
    size_t max_position, size_t num_q_heads, size_t num_kv_heads, size_t head_dim, float eps, float theta,
    float scale, size_t group_size, size_t cache_batch_stride, size_t scale_batch_stride, size_t max_slots,
    size_t max_cache_len, void* stream);

/* The launch above that also leaves the output rows quantized for the output projection, as
 * cactus_hip_quantize_activations_dev would (graph_ops_nn.cpp:35-66 on the attention node's output, bit for bit):
 * output_q int8[B, Hq * D] (8-byte aligned), output_scales float[B]; output is written as well.  With 8 kv heads of
 * head_dim 128 and one key range per sequence a workgroup holds a whole row and no extra launch is made;
 * other shapes run the row quantization kernel behind the attention. */
int cactus_hip_attention_decode_fused_quant_f16_dev(
    const cactus_half_t* q, const cactus_half_t* k, const cactus_half_t* v, const cactus_half_t* q_weight,
    const cactus_half_t* k_weight, int8_t* k_cache, int8_t* v_cache, float* k_scales, float* v_scales,
    cactus_half_t* output, int8_t* output_q, float* output_scales, size_t batch_size, const int32_t* cache_lens,
    const int32_t* positions, size_t max_position, size_t num_q_heads, size_t num_kv_heads, size_t head_dim, float eps,
    float theta, float scale, size_t group_size, size_t cache_batch_stride, size_t scale_batch_stride, size_t max_slots,
    size_t max_cache_len, void* stream);

/* The same launch for ONE sequence whose cache length and position the host knows, with nothing appended to the
 * cache: CactusGraph::execute hands it the run rms_norm -> rope (on the q and k projections) ->
 * attention_int8_hybrid of a decode step (model_qwen.cpp:63-99).  The normalized, rotated key rows are also
 * written to k_out half[Hkv, D], from which KVCache::update_from_graph feeds the cache afterwards
 * (engine_cache.cpp:186-266).  output and k_out equal those of cactus_hip_rms_norm_f16_dev + cactus_hip_rope_f16_dev
 * + cactus_hip_attention_hybrid_int8_fp16_dev bit for bit.  Same shape limits as above; cache_len >= 1. */
int cactus_hip_attention_decode_fused_single_f16_dev(
    const cactus_half_t* q, const cactus_half_t* k, const cactus_half_t* v, const cactus_half_t* q_weight,
    const cactus_half_t* k_weight, const int8_t* k_cache, const int8_t* v_cache, const float* k_scales,
    const float* v_scales, cactus_half_t* output, cactus_half_t* k_out, size_t cache_len, size_t position,
    size_t num_q_heads, size_t num_kv_heads, size_t head_dim, float eps, float theta, float scale, size_t group_size,
    void* stream);

/* Batched-decode append: one new token per independent sequence, K and V [B,Hkv,D] quantized
 * (cactus_quantize_kv_fp16_to_int8 arithmetic) straight into slot slots[b] of sequence b's cache,
 * caches addressed as in the batched attention above (per-sequence strides in elements; a slot is
 * Hkv*D int8 values / Hkv*ceil(D/gs) scales).  A slot outside [0, max_slots) stores nothing.  No
 * counterpart in the reference, whose cache is batch-1; with one sequence it is the append of
 * engine_cache.cpp:234-267. */
int cactus_hip_kv_append_batched_dev(const cactus_half_t* k_new, const cactus_half_t* v_new,
                                     int8_t* k_cache, int8_t* v_cache, float* k_scales, float* v_scales,
                                     size_t batch_size, size_t kv_heads, size_t head_dim,
                                     size_t group_size, const int32_t* slots,
                                     size_t cache_batch_stride, size_t scale_batch_stride,
                                     size_t max_slots, void* stream);

/* Independent sequences at different positions, with the reference's sliding window (engine_cache.cpp:
 * 268-329: the first `sink` tokens and the most recent `window - sink` are kept; window = min(context, 512),
 * sink 4 in engine_model.cpp:136-146) held as a ring: the token with absolute index p of a sequence lives in
 * slot p while p < window and in slot sink + (p - window) % (window - sink) afterwards, i.e. a new token
 * overwrites the oldest recent one instead of every layer's window being moved down by one per token.  The
 * cache then holds exactly the tokens the reference's holds, rotated; the hybrid attention sees every cached
 * key, so only the order of its sums changes.  No counterpart in the reference, whose cache is batch-1.
 *
 * positions: device int32 [batch], tokens each sequence has seen so far (its RoPE position offset and
 * attention position_offset; KVCache::total_seq_len).  window == 0: no window (slot = position).
 *
 *   cactus_hip_qk_norm_rope_seq_f16_dev   cactus_hip_qk_norm_rope_kv_append_f16_dev with per-sequence
 *       positions: token s of sequence b is rotated for position positions[b] + s (clamped to max_position, which
 *       bounds the cos/sin table the call builds) and, when v != NULL, k (rotated) and v are quantized into
 *       that token's slot.  Appending in this launch is only equivalent to the reference while no
 *       sequence's window is full (the attention of the step must still see the token a full window is
 *       about to drop): pass v == NULL then and append after the attention with
 *   cactus_hip_kv_append_windowed_dev     k_new / v_new half[batch, seq_len, Hkv, D] (k already rotated)
 *       quantized into the slots of their positions.
 *   cactus_hip_kv_positions_advance_dev   positions[b] += advance and, when cache_lens != NULL,
 *       cache_lens[b] = min(positions[b], window) (window 0: positions[b]): the two arrays the batched
 *       attention takes, kept on the device between steps.
 * At most window - sink new tokens per sequence and call. */
int cactus_hip_qk_norm_rope_seq_f16_dev(const cactus_half_t* q, const cactus_half_t* k, const cactus_half_t* v,
                                        const cactus_half_t* q_weight, const cactus_half_t* k_weight,
                                        cactus_half_t* q_output, cactus_half_t* k_output, int8_t* k_cache,
                                        int8_t* v_cache, float* k_scales, float* v_scales, const int32_t* positions,
                                        size_t max_position, size_t batch_size, size_t seq_len, size_t num_q_heads,
                                        size_t num_kv_heads, size_t head_dim, float eps, float theta, size_t group_size,
                                        size_t cache_batch_stride, size_t scale_batch_stride, size_t max_slots,
                                        size_t window, size_t sink, void* stream);
int cactus_hip_kv_append_windowed_dev(const cactus_half_t* k_new, const cactus_half_t* v_new, int8_t* k_cache,
                                      int8_t* v_cache, float* k_scales, float* v_scales, const int32_t* positions,
                                      size_t batch_size, size_t seq_len, size_t kv_heads, size_t head_dim,
                                      size_t group_size, size_t cache_batch_stride, size_t scale_batch_stride,
                                      size_t max_slots, size_t window, size_t sink, void* stream);
int cactus_hip_kv_positions_advance_dev(int32_t* positions, int32_t* cache_lens, size_t batch_size, size_t advance,
                                        size_t window, void* stream);

/* ----------------------------------------------------------- multi-GPU (RCCL over xGMI)
 * No counterpart in the reference (a single-process CPU engine with no multi-device code).  One process per GPU;
 * a matmul splits by weight rows: rank r owns rows cactus_hip_row_shard(N, world, r, multiple) of B[N, K] and of
 * its scales [N, K/group_size] - a contiguous slice of the reference's layout, no repacking - and computes
 * C[:, start:end]; ONE exchange leaves C[M, N] on every rank.  No output element's sum is split, so the result is
 * bit-identical to the one-GPU call.  (A K-split would reorder the fp32 chain and is not offered.)  Attention over
 * a long INT8 cache shards by kv head and gathers its output rows with the same call (N = heads * head_dim,
 * multiple = a rank's heads * head_dim).
 *
 * The communicator is RCCL's: rank 0 calls cactus_hip_comm_get_unique_id, the host program hands the 128 bytes to
 * the other ranks by its own means, and every rank calls cactus_hip_comm_init on its device (collective:
 * ncclCommInitRank); cactus_hip_comm_adopt wraps an ncclComm_t the host already owns.  RCCL is loaded on first
 * use; cactus_hip_comm_available() == 0 when it cannot be.  Calls enqueue on `stream`, allocate nothing beyond the
 * library workspace (M > 1: world * M * per halves; reserve it before stream capture) and can be captured.
 *
 * algo: 0 the library's choice; 1 ncclAllGather where every rank owns the same count; 2 grouped ncclSend/ncclRecv
 * to and from every peer (the direct all-pairs schedule: all xGMI links of a GPU at once; M == 1 only). */
#define CACTUS_HIP_COMM_ID_BYTES 128
int  cactus_hip_comm_available(void);
int  cactus_hip_comm_get_unique_id(void* id128);
int  cactus_hip_comm_init(void** comm_out, const void* id128, int world, int rank);
int  cactus_hip_comm_adopt(void** comm_out, void* nccl_comm, int world, int rank);
int  cactus_hip_comm_destroy(void* comm);
int  cactus_hip_comm_rank(void* comm);
int  cactus_hip_comm_world(void* comm);
/* Rows [*start, *end) of an N-row weight that `rank` owns: ceil(N / world) rounded up to `multiple` rows per rank
 * (16 = the streaming kernel's rows per workgroup), the last ranks take what is left (possibly nothing).  Host only. */
void cactus_hip_row_shard(size_t N, int world, int rank, size_t multiple, size_t* start, size_t* end);
/* C half[M, N] on every rank, each having written its own columns (row_shard of N): afterwards every rank holds
 * all columns.  M == 1: exchanged in place.  M > 1: through rank-major slots in the workspace and one copy kernel. */
int  cactus_hip_allgather_columns_dev(void* comm, cactus_half_t* C, size_t M, size_t N, size_t multiple, int algo,
                                      void* stream);
/* cactus_matmul_int8 / _int4 (kernel.h:41-47) N-split: B_rows / B_scales_rows are THIS rank's rows of the weight
 * and of its scales, A / A_scales the full quantized activations (replicated), C half[M, N] the full output on
 * every rank.  M > 1: the local launch writes its block straight into its slot (no padding copy). */
int  cactus_hip_matmul_int8_nsplit_dev(void* comm, const int8_t* A, const float* A_scales, const int8_t* B_rows,
                                       const cactus_half_t* B_scales_rows, cactus_half_t* C, size_t M, size_t K,
                                       size_t N, size_t group_size, size_t multiple, int algo, void* stream);
int  cactus_hip_matmul_int4_nsplit_dev(void* comm, const int8_t* A, const float* A_scales,
                                       const uint8_t* B_rows_packed, const cactus_half_t* B_scales_rows,
                                       cactus_half_t* C, size_t M, size_t K, size_t N, size_t group_size,
                                       size_t multiple, int algo, void* stream);
/* The copy kernel of the M > 1 exchange on its own: slots half[world][M][per] (per = a rank's column count as
 * row_shard gives it) -> C half[M, N]. */
int  cactus_hip_slots_to_columns_dev(const cactus_half_t* slots, cactus_half_t* C, size_t M, size_t N, int world,
                                     size_t multiple, void* stream);

/* ------------------------------------------------ host-pointer drop-in family
 * Same argument lists as cactus/kernel/kernel.h, host memory in and out.      */

void  cactus_hip_matmul_int8(const int8_t* A, const float* A_scales,
                             const int8_t* B, const cactus_half_t* B_scales,
                             cactus_half_t* C, size_t M, size_t K, size_t N, size_t group_size);
void  cactus_hip_matmul_int4(const int8_t* A, const float* A_scales,
                             const uint8_t* B_packed, const cactus_half_t* B_scales,
                             cactus_half_t* C, size_t M, size_t K, size_t N, size_t group_size);
void  cactus_hip_matmul_f16(const cactus_half_t* a, const cactus_half_t* b_transposed,
                            cactus_half_t* c, size_t M, size_t K, size_t N);
void  cactus_hip_attention_f16(const cactus_half_t* queries, const cactus_half_t* keys,
                               const cactus_half_t* values, cactus_half_t* output,
                               size_t batch_size, size_t seq_len, size_t kv_seq_len,
                               size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
                               float scale, const cactus_half_t* mask, size_t position_offset,
                               size_t window_size, int is_causal);
void  cactus_hip_attention_hybrid_int8_fp16(
    const cactus_half_t* queries, const int8_t* keys_cached, const int8_t* values_cached,
    const float* k_scales, const float* v_scales,
    const cactus_half_t* keys_new, const cactus_half_t* values_new, cactus_half_t* output,
    size_t batch_size, size_t seq_len, size_t cache_len, size_t new_len,
    size_t num_q_heads, size_t num_kv_heads, size_t head_dim,
    float scale, size_t position_offset, int is_causal, size_t group_size);
void  cactus_hip_quantize_kv_fp16_to_int8(const cactus_half_t* src, int8_t* dst, float* scales,
                                          size_t seq_len, size_t kv_heads, size_t head_dim,
                                          size_t group_size);
float cactus_hip_fp16_max_abs(const cactus_half_t* src, size_t count);
void  cactus_hip_fp16_to_int8(const cactus_half_t* src, int8_t* dst, size_t count, float scale);
void  cactus_hip_unpack_int4_to_int8(const uint8_t* packed, int8_t* unpacked, size_t unpacked_count);

/* Host-pointer forms of the block ops, same argument meaning as kernel.h:20-24, 72-83. */
void  cactus_hip_rms_norm_f16(const cactus_half_t* input, const cactus_half_t* weight, cactus_half_t* output,
                              size_t batch_size, size_t dims, float eps);
void  cactus_hip_rope_f16(const cactus_half_t* input, cactus_half_t* output, size_t batch_size, size_t seq_len,
                          size_t num_heads, size_t head_dim, size_t start_pos, float theta);
void  cactus_hip_silu_f16(const cactus_half_t* input, cactus_half_t* output, size_t num_elements);
void  cactus_hip_gelu_f16(const cactus_half_t* input, cactus_half_t* output, size_t num_elements);
void  cactus_hip_add_f16(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements);
void  cactus_hip_add_f16_clipped(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements);
void  cactus_hip_subtract_f16(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements);
void  cactus_hip_multiply_f16(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements);
void  cactus_hip_divide_f16(const cactus_half_t* a, const cactus_half_t* b, cactus_half_t* output, size_t num_elements);

#ifdef __cplusplus
}
#endif
#endif /* CACTUS_HIP_H */
