/*
 * cactus_hip_graph.h -- C ABI over the device-resident mirror of the reference's graph layer
 * for the hot path (cactus_amd/csrc/graph.h).  One call per CactusGraph builder / executor
 * method a model or test uses on this path; node ids are the graph's own ids.  Calls that the
 * reference signals with a C++ exception return -1 (or NULL) here and leave the exception's
 * message in cactus_hip_last_error().
 *
 *   CactusGraph::input                 graph.h:396   graph_builder.cpp:7-9
 *   CactusGraph::matmul                graph.h:420   graph_builder.cpp:61-81
 *   CactusGraph::quantize_activations  graph.h:398   graph_builder.cpp:560-570
 *   CactusGraph::attention (x3)        graph.h:451-453  graph_builder.cpp:286-299
 *   CactusGraph::attention_int8_hybrid graph.h:455-458  graph_builder.cpp:301-316
 *   CactusGraph::set_input / set_external_input / get_output / execute
 *                                      graph.h:469-473  graph_execute.cpp:197-253
 *   CactusGraph::set_grouped_scales    graph.h:437   graph_io.cpp:106-112
 *   CactusGraph::soft_reset / hard_reset  graph.h:474-475  graph_execute.cpp:629-675
 *   KVCache::init / set_window_size / reset / update_from_graph / get_*
 *                                      engine.h:324-381  engine_cache.cpp:11-51, 101-337
 *
 * Precision values: 0 INT8, 1 FP16, 2 FP32, 3 INT4 (graph.h:99-104).
 */
#ifndef CACTUS_HIP_GRAPH_H
#define CACTUS_HIP_GRAPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void*  cactus_hip_graph_create(void);
void   cactus_hip_graph_destroy(void* graph);
long   cactus_hip_graph_input(void* graph, const size_t* shape, size_t ndim, int precision);
long   cactus_hip_graph_matmul(void* graph, size_t a, size_t b, int pretransposed_rhs);
long   cactus_hip_graph_quantize_activations(void* graph, size_t input);
long   cactus_hip_graph_attention(void* graph, size_t q, size_t k, size_t v, float scale,
                                  size_t position_offset, size_t window_size, int is_causal);
/* cached_* are DEVICE pointers (e.g. from cactus_hip_kvcache_keys). */
long   cactus_hip_graph_attention_int8_hybrid(void* graph, size_t q, size_t key_new, size_t value_new,
                                              float scale, size_t position_offset,
                                              const int8_t* cached_keys, const int8_t* cached_values,
                                              const float* k_scales, const float* v_scales,
                                              size_t cache_len, size_t num_kv_heads, size_t head_dim);
/* Decoder-block ops (cactus/graph/graph.h:400-404, 416-417, 423, 443, 448-449).  binary op:
 * 0 add, 1 subtract, 2 multiply, 3 divide, 4 add_clipped; equal shapes only.  reshape is a view.
 * embedding: table node INT8 (with grouped scales) or FP16, indices node FP32 or INT8 values. */
long   cactus_hip_graph_binary(void* graph, int op, size_t a, size_t b);
/* CactusGraph::scalar_add / scalar_subtract / scalar_multiply / scalar_divide (graph.h:407-410): op 0..3. */
long   cactus_hip_graph_scalar_op(void* graph, int op, size_t input, float value);
long   cactus_hip_graph_silu(void* graph, size_t input);
long   cactus_hip_graph_gelu(void* graph, size_t input);
long   cactus_hip_graph_reshape(void* graph, size_t input, const size_t* shape, size_t ndim);
/* CactusGraph::slice / concat / index (graph.h:426-428, 444): along axis 0 slice and index are views. */
long   cactus_hip_graph_slice(void* graph, size_t input, int axis, size_t start, size_t length);
long   cactus_hip_graph_concat(void* graph, size_t input1, size_t input2, int axis);
long   cactus_hip_graph_index(void* graph, size_t input, size_t index_value, int dim);
/* CactusGraph::sample (graph.h:446; graph_ops_sample.cpp:8-39): one token from the last row of FP16 logits
 * [seq_len, vocab]; read it with cactus_hip_graph_get_output (4 bytes, uint32).  random_seed 0: the clock, as
 * the reference seeds every node.  The history of sampled tokens (repetition penalty) is kept per graph. */
long   cactus_hip_graph_sample(void* graph, size_t logits, float temperature, float top_p, size_t top_k,
                               uint32_t random_seed);
int    cactus_hip_graph_reset_sample_history(void* graph);
long   cactus_hip_graph_embedding(void* graph, size_t table, size_t indices);
long   cactus_hip_graph_rms_norm(void* graph, size_t input, size_t weight, float epsilon);
long   cactus_hip_graph_rope(void* graph, size_t input, float theta, size_t position_offset);
/* CactusGraph::conv1d_causal (graph.h:460): input [N, L, C], weight [C, 1, K], FP16. */
long   cactus_hip_graph_conv1d_causal(void* graph, size_t input, size_t weight, size_t kernel_size, size_t dilation);
/* CactusGraph::mmap_weights (graph.h:435; graph_io.cpp:48-72): a CACT tensor file becomes a weight
 * node (payload and group scales copied to HBM once, INT4 kept packed); the same file name gives
 * the same node.  node_info reports what was loaded (shape4: room for 4 dims). */
long   cactus_hip_graph_mmap_weights(void* graph, const char* filename);
int    cactus_hip_graph_node_info(void* graph, size_t node, size_t* shape4, size_t* ndim, int* precision,
                                  size_t* group_size, size_t* num_groups, int* packed_int4);
/* MappedFile::parse_header (graph_io.cpp:497-566) on a byte buffer, no GPU involved.  out14 =
 * version, alignment, ndim, dims[4], precision, data_bytes, scales_bytes, group_size, num_groups,
 * scales_offset, data_offset.  -1 + cactus_hip_last_error() on the reference's error conditions. */
int    cactus_hip_cact_parse_header(const void* bytes, size_t size, uint64_t* out14);
/* Config::from_json (cactus/engine/engine_model.cpp:403-539; fields and defaults of
 * cactus/engine/engine.h:43-116): the `key = value` lines of a model folder's config.txt.  Only the
 * fields a decoder needs are returned; the vision / image keys are accepted and ignored.
 * layer_types holds the items joined by ','.  -1 when the file cannot be opened or a number does
 * not parse (the reference returns false / throws from std::stoul). */
typedef struct cactus_hip_model_config {
    uint32_t vocab_size, bos_token_id, eos_token_id, num_layers, hidden_dim, ffn_intermediate_dim;
    uint32_t attention_heads, attention_kv_heads, attention_head_dim;
    float layer_norm_eps, rope_theta;
    uint32_t num_experts, num_shared_experts, num_top_experts, moe_every_n_layers;
    int tie_word_embeddings;
    int model_type;      /* QWEN 0, GEMMA 1, SMOL 2, NOMIC 3, LFM2 5, WHISPER 7 */
    int model_variant;   /* DEFAULT 0, VLM 1, EXTRACT 2, RAG 3 */
    int precision;       /* INT8 0, FP16 1, FP32 2 */
    float default_temperature, default_top_p;
    uint32_t default_top_k, conv_L_cache, num_layer_types;
    char layer_types[2048];
} cactus_hip_model_config;
int    cactus_hip_model_config_load(const char* path, cactus_hip_model_config* out);
int    cactus_hip_graph_set_input(void* graph, size_t node, const void* host_data, int precision);
int    cactus_hip_graph_set_external_input(void* graph, size_t node, void* host_data, int precision);
int    cactus_hip_graph_set_grouped_scales(void* graph, size_t node, size_t group_size, size_t num_groups,
                                           void* host_scales);
/* Packed INT4 payload uint8[N, K/2] of a weight node (what MappedFile::load_into_graph sets,
 * graph_io.cpp:470-495); it stays packed in HBM. */
int    cactus_hip_graph_set_packed_int4(void* graph, size_t node, const void* host_packed, size_t bytes);
int    cactus_hip_graph_execute(void* graph);
int    cactus_hip_graph_get_output(void* graph, size_t node, void* host_dst, size_t bytes);
void*  cactus_hip_graph_get_device_output(void* graph, size_t node);
int    cactus_hip_graph_soft_reset(void* graph);
int    cactus_hip_graph_hard_reset(void* graph);
size_t cactus_hip_graph_node_count(void* graph);
/* Device allocations made so far for node outputs and per-step scratch.  The graph keeps those
 * buffers in a pool across execute() and soft_reset() (the reference's BufferPool,
 * graph_execute.cpp:640-675), so the count stops growing once a step has run. */
size_t cactus_hip_graph_pool_allocations(void* graph);
/* The pool hands out buffers in power-of-two sizes from 1 KiB (as the reference's BufferPool buckets them,
 * graph_core.cpp:5-19), so passes of different sequence lengths share buffers.  soft_reset() keeps at most
 * max_idle_bytes of idle buffers (default: an eighth of the device's memory, at least 1 GiB) and frees the rest, largest
 * first; a device allocation that fails frees every idle buffer and is tried once more. */
size_t cactus_hip_graph_pool_idle_bytes(void* graph);
void   cactus_hip_graph_set_pool_max_idle_bytes(void* graph, size_t bytes);
/* Grouped weights bound from now on (set_external_input / set_packed_int4 + set_grouped_scales, mmap_weights)
 * get a tile-major twin for the decode matmuls (cactus_hip_weights_pack_dev; default on, costs a second copy of
 * the weight in HBM); hard_reset() releases them. */
void   cactus_hip_graph_set_streaming_twins(void* graph, int on);
/* execute() recognizes the runs a decoder block is made of and hands each to one fused launch of
 * cactus_hip.h, with the separate nodes' results bit for bit (on by default; the reference executes node
 * by node, graph_execute.cpp:480-560, and fuses nothing):
 *   [add ->] rms_norm -> the matmuls reading the normalized rows   (model_qwen.cpp:51-57, 109-113)
 *   silu / gelu -> multiply -> the matmuls reading the product     (model_qwen.cpp:114-116)
 * The run must be closed: every reader of the norm (of the product) is a grouped INT8 / INT4 matmul with
 * it as the left operand.  A node such a launch never writes (the norm's rows, the activation, the
 * product) is computed when its output is asked for.  fused_launches: fused launches issued so far. */
void   cactus_hip_graph_set_fusion(void* graph, int on);
size_t cactus_hip_graph_fused_launches(void* graph);

void*  cactus_hip_kvcache_create(size_t layers, size_t max_seq, size_t kv_heads, size_t head_dim);
void   cactus_hip_kvcache_destroy(void* cache);
int    cactus_hip_kvcache_set_window(void* cache, size_t window, size_t sink);
/* Ring mode (no counterpart in the reference, which memmoves the window once per layer and token,
 * engine_cache.cpp:276-303): with on != 0, tokens appended to a full window overwrite the oldest
 * slots of [sink, window) in place.  Slot sink + (ring_head + i) % (window - sink) then holds the
 * i-th oldest recent token; attention over the cache sees the same keys in rotated order.  Only
 * while the cache is empty. */
int    cactus_hip_kvcache_set_ring(void* cache, int on);
size_t cactus_hip_kvcache_ring_head(void* cache);
void   cactus_hip_kvcache_reset(void* cache);
/* k_nodes / v_nodes: one node id per layer holding the new FP16 K/V [seq_len, Hkv, D]. */
int    cactus_hip_kvcache_update(void* cache, void* graph, const size_t* k_nodes, const size_t* v_nodes,
                                 size_t seq_len);
size_t cactus_hip_kvcache_len(void* cache);
size_t cactus_hip_kvcache_total_len(void* cache);
const int8_t* cactus_hip_kvcache_keys(void* cache, size_t layer);
const int8_t* cactus_hip_kvcache_values(void* cache, size_t layer);
const float*  cactus_hip_kvcache_key_scales(void* cache, size_t layer);
const float*  cactus_hip_kvcache_value_scales(void* cache, size_t layer);

#ifdef __cplusplus
}
#endif
#endif /* CACTUS_HIP_GRAPH_H */
