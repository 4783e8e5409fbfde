// Host-side mirror of the slice of cactus/graph that drives the quantized hot path, with a
// device-buffer path added: the same builder calls (input / matmul / attention /
// attention_int8_hybrid / quantize_activations / set_grouped_scales / set_input / execute /
// get_output) and the same error behaviour as the reference's CactusGraph
// (cactus/graph/graph.h:386-500, graph_builder.cpp:61-81, 286-316, 560-596;
// dispatch in graph_ops_nn.cpp:129-231, 302-371), but every node buffer lives in HBM and the
// node kernels are the *_dev entry points of include/cactus_hip.h.
//
// Differences that are deliberate:
//   * BufferDesc carries a device pointer next to the host pointer and a residency flag; host
//     inputs are uploaded once per set_input, weights (set_external_input) once per pointer.
//   * the activation-quantization cache is keyed by node id, not by host pointer
//     (graph_ops_nn.cpp:15-17, 36-38 keys on the pointer and can go stale).
//   * packed INT4 weights stay packed in HBM (the reference unpacks them to INT8 at load,
//     graph_io.cpp:435-463); results are identical.
//   * KVCache keeps its INT8 payload and fp32 scales in HBM and quantizes appended tokens with
//     the device kernel; window/sink bookkeeping follows engine_cache.cpp:101-337.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace cactus_hip {

enum class Precision { INT8 = 0, FP16 = 1, FP32 = 2, INT4 = 3 };   // values as graph.h:99-104
enum class ComputeBackend { CPU = 0, NPU = 1, HIP = 2 };           // HIP is the additive enumerator
enum class OpType {
    INPUT, MATMUL, ATTENTION, ATTENTION_INT8_HYBRID, QUANTIZE_ACTIVATIONS,
    // the ops between the hot kernels of a decoder block
    ADD, ADD_CLIPPED, SUBTRACT, MULTIPLY, DIVIDE, SILU, GELU, RMS_NORM, ROPE, RESHAPE, EMBEDDING,
    CONV1D_CAUSAL,   // LFM2's short convolution
    SCALAR_ADD, SCALAR_SUBTRACT, SCALAR_MULTIPLY, SCALAR_DIVIDE,
    SLICE, CONCAT, INDEX, SAMPLE
};

size_t precision_size(Precision p);   // bytes per element (INT4 counted as 1, graph.h:128-136)

struct BufferDesc {
    std::vector<size_t> shape;
    size_t total_size = 0;   // elements
    size_t byte_size = 0;
    Precision precision = Precision::FP16;

    void* device_data = nullptr;     // HBM; owned unless external_device
    bool owns_device = false;
    bool device_valid = false;       // device_data holds the node's current value
    std::vector<char> host_copy;     // filled by get_output()

    // grouped weight quantization (graph.h:203-209)
    size_t group_size = 0, num_groups = 0;
    void* scales_device = nullptr;   // fp16 [N, num_groups]
    bool owns_scales = false;
    void* packed_int4_device = nullptr;   // uint8 [N, K/2]
    size_t packed_int4_size = 0;
    bool owns_packed = false;

    // per-row activation scales of a QUANTIZE_ACTIVATIONS output (graph.h:211-213)
    void* activation_scales_device = nullptr;
    size_t num_rows_for_activation_scales = 0;

    bool is_grouped_int8() const { return precision == Precision::INT8 && group_size > 0; }
    bool is_packed_int4() const { return packed_int4_device != nullptr && packed_int4_size > 0; }
    bool has_activation_scales() const {
        return activation_scales_device != nullptr && num_rows_for_activation_scales > 0;
    }
};

struct OpParams {
    float scale = 1.0f;
    bool pretransposed_rhs = false;
    size_t position_offset = 0;
    float epsilon = 1e-5f;   // RMS_NORM
    float theta = 0.0f;      // ROPE
    size_t dilation = 1;     // CONV1D_CAUSAL
    float scalar = 0.0f;     // SCALAR_*
    int axis = 0;            // SLICE, CONCAT, INDEX
    size_t slice_start = 0, slice_length = 0, index_value = 0;
    float temperature = 0.0f, top_p = 0.0f;   // SAMPLE
    size_t top_k = 0;
    uint32_t random_seed = 0;
    size_t window_size = 0;
    bool is_causal = true;
    Precision output_precision = Precision::INT8;
    ComputeBackend backend = ComputeBackend::HIP;
    // device pointers into the INT8 KV cache (graph.h:314-320 carries host pointers)
    const int8_t* cached_keys_int8 = nullptr;
    const int8_t* cached_values_int8 = nullptr;
    const float* cached_k_scales = nullptr;
    const float* cached_v_scales = nullptr;
    size_t cache_seq_len = 0;
    size_t num_kv_heads = 0;
    size_t head_dim = 0;
};

struct GraphNode {
    size_t id;
    OpType op_type;
    std::vector<size_t> input_ids;
    BufferDesc output_buffer;
    OpParams params;
};

class CactusGraph {
public:
    CactusGraph();
    ~CactusGraph();
    CactusGraph(const CactusGraph&) = delete;
    CactusGraph& operator=(const CactusGraph&) = delete;

    size_t input(const std::vector<size_t>& shape, Precision precision = Precision::INT8);
    size_t quantize_activations(size_t input);
    size_t matmul(size_t input1, size_t input2, bool pretransposed_rhs = false,
                  ComputeBackend backend = ComputeBackend::HIP);
    size_t attention(size_t query, size_t key, size_t value, float scale, bool is_causal = true,
                     ComputeBackend backend = ComputeBackend::HIP);
    size_t attention(size_t query, size_t key, size_t value, float scale, size_t position_offset,
                     ComputeBackend backend = ComputeBackend::HIP);
    size_t attention(size_t query, size_t key, size_t value, float scale, size_t position_offset,
                     size_t window_size, ComputeBackend backend = ComputeBackend::HIP);
    size_t attention_int8_hybrid(size_t query, size_t key_new, size_t value_new, float scale,
                                 size_t position_offset, const int8_t* cached_keys,
                                 const int8_t* cached_values, const float* k_scales, const float* v_scales,
                                 size_t cache_len, size_t num_kv_heads, size_t head_dim);

    // Decoder-block ops (graph.h:400-404, 416-417, 423, 443, 448-449; graph_builder.cpp:11-59,
    // 120-123, 253-261, 468-474, 524-539).  The binaries take equal shapes only (the reference
    // also broadcasts); reshape is a view of its input's buffer.
    size_t add(size_t input1, size_t input2);
    size_t add_clipped(size_t input1, size_t input2);
    size_t subtract(size_t input1, size_t input2);
    size_t multiply(size_t input1, size_t input2);
    size_t divide(size_t input1, size_t input2);
    size_t silu(size_t input);
    size_t gelu(size_t input);
    // graph.h:407-410; graph_builder.cpp:424-450
    size_t scalar_add(size_t input, float value);
    size_t scalar_subtract(size_t input, float value);
    size_t scalar_multiply(size_t input, float value);
    size_t scalar_divide(size_t input, float value);
    size_t reshape(size_t input, const std::vector<size_t>& new_shape);
    size_t embedding(size_t embedding_tensor, size_t indices);
    // graph.h:426-428, 444; graph_builder.cpp:125-160, 348-375, 499-522.  Along axis 0 slice and index are
    // views of their input's buffer (as in the reference), along other axes strided device copies.
    size_t slice(size_t input, int axis, size_t start, size_t length);
    size_t concat(size_t input1, size_t input2, int axis = 0);
    size_t index(size_t input, size_t index_value, int dim);
    // graph.h:446; graph_builder.cpp:396-420.  One token (uint32 in an FP32-sized node) from the last row of
    // logits [seq_len, vocab].  The reference seeds its generator from the clock per node; here the seed is an
    // argument (0: the clock) and the history of sampled tokens belongs to the graph, not the process.  The
    // logit bias the reference accepts never changes its result and is not taken.
    size_t sample(size_t logits, float temperature, float top_p, size_t top_k, uint32_t random_seed = 0);
    void reset_sample_history();
    size_t rms_norm(size_t input, size_t weight, float epsilon = 1e-5f);
    size_t rope(size_t input, float theta, size_t position_offset = 0,
                ComputeBackend backend = ComputeBackend::HIP);
    // graph.h:460; graph_builder.cpp:318-321.  input [N, L, C], weight [C, 1, K], FP16 both (the
    // reference also takes INT8 weights, which it dequantizes on the host per call).
    size_t conv1d_causal(size_t input, size_t weight, size_t kernel_size, size_t dilation = 1);

    // Host data: copied to HBM now (set_input) or once per distinct pointer and kept resident
    // (set_external_input: weights that outlive the graph's steps).
    void set_input(size_t node_id, const void* data, Precision precision);
    void set_external_input(size_t node_id, void* data, Precision precision);
    // Device data owned by the caller (already resident).
    void set_device_input(size_t node_id, void* device_ptr, Precision precision);
    // Group scales (fp16 [N, num_groups], host pointer; uploaded once per pointer).
    void set_grouped_scales(size_t node_id, size_t group_size, size_t num_groups, void* scales_ptr);
    // Packed INT4 payload for a weight node (uint8 [N, K/2], host pointer): what
    // MappedFile::load_into_graph sets (graph_io.cpp:470-495), kept packed on the device.
    void set_packed_int4(size_t node_id, const void* packed, size_t packed_bytes);

    // A CACT tensor file (graph_io.cpp:48-72, 470-566; written by python/src/tensor_io.py) as a
    // weight node: header parsed as the reference parses it, payload and group scales copied to
    // HBM once (INT4 stays packed) and owned by the node, the file unmapped again.  One node per
    // file name (the reference's weight_cache_).
    size_t mmap_weights(const std::string& filename);

    void execute();
    void* get_output(size_t node_id);          // host copy, synchronises
    void* get_device_output(size_t node_id);   // device pointer, no synchronisation
    const BufferDesc& get_output_buffer(size_t node_id) const;
    size_t get_node_count() const { return nodes_.size(); }
    // hipMalloc calls made for node buffers and scratch so far (stays flat once the pool is warm)
    size_t pool_device_allocations() const { return pool_.device_allocations(); }
    size_t pool_idle_bytes() const { return pool_.idle_bytes(); }
    // idle node buffers kept across soft resets (default 1 GiB); the excess is freed, largest first
    void set_pool_max_idle_bytes(size_t bytes) { pool_.set_max_idle_bytes(bytes); pool_.trim(); }
    // weights bound from now on get a tile-major twin for the decode matmuls (default on; costs a second copy)
    void set_streaming_twins(bool on) { streaming_twins_ = on; }
    void hard_reset();
    void soft_reset();   // drops every node that is not a weight (external) input
    // execute() hands runs of nodes to the fused launches of the C ABI (on by default):
    //   [add ->] rms_norm -> matmuls that read the normalized rows   (one launch for up to 4 rows)
    //   silu / gelu -> multiply -> matmuls that read the product
    // with the results of the separate nodes, bit for bit.  A node that a fused launch never materializes
    // (the norm's rows, the activation and its product) is computed when get_output() asks for it.
    void set_fusion(bool on) { fusion_ = on; }
    size_t fused_launches() const { return fused_launches_; }   // fused runs taken by execute() so far

    size_t add_node(OpType op_type, const std::vector<size_t>& inputs,
                    const std::vector<size_t>& output_shape, const OpParams& params = {});

    // Device buffers of node outputs and per-step scratch (activation quantization, transposes), kept
    // across execute() and soft_reset() and handed out again by size: a model that rebuilds the same
    // graph every token allocates on its first step only (the reference's BufferPool,
    // graph_execute.cpp:640-675).  All node kernels run on one stream, so a buffer released by one
    // step is safely reused by the next.  Emptied by hard_reset().
    class DevicePool {
    public:
        void* acquire(size_t bytes);
        void release(void* p);   // a pointer the pool did not hand out is freed
        void clear();
        void trim();             // frees idle buffers, largest first, until at most max_idle_bytes() stay idle
        size_t device_allocations() const { return allocations_; }
        size_t idle_bytes() const { return idle_bytes_; }
        size_t max_idle_bytes();
        void trim_to(size_t cap);
        void set_max_idle_bytes(size_t b) { max_idle_bytes_ = b; cap_set_ = true; }
    private:
        std::unordered_map<size_t, std::vector<void*>> free_;   // rounded size -> idle buffers
        std::unordered_map<void*, size_t> size_of_;             // every buffer handed out -> rounded size
        size_t allocations_ = 0;
        size_t idle_bytes_ = 0;
        size_t max_idle_bytes_ = size_t(1) << 30;   // idle buffers kept across soft resets; unless set: an eighth of the
        bool cap_set_ = false;                      // device's memory (at least 1 GiB), found at the first trim
    };

private:
    GraphNode& node(size_t id);
    const GraphNode& node(size_t id) const;
    void ensure_device(BufferDesc& b);
    void make_twin(BufferDesc& b);
    void* resident_copy(const void* host, size_t bytes);
    bool batchable_matmul(const GraphNode& n, bool* int4) const;
    bool fusable_matmul(const GraphNode& n, size_t lhs_id, bool* int4) const;
    bool consumer_matmuls(size_t producer_idx, std::vector<size_t>* int8_idx, std::vector<size_t>* int4_idx) const;
    bool try_fuse_norm(size_t norm_idx, std::vector<char>& done);
    bool try_fuse_activation(size_t mul_idx, std::vector<char>& done);
    void launch_produced(int kind, bool int4, const std::vector<size_t>& mm, const void* x, const void* x2,
                         const void* w, void* sum_out, float eps, size_t M, size_t K);
    void materialize(size_t idx);
    // the run rms_norm -> reshape -> rope on the q and k projections of a one-token attention_int8_hybrid node
    struct AttnChain {
        size_t q_norm, q_view, q_rope, k_norm, k_view, k_rope;   // node indices
        size_t q_src, k_src;                                     // the projections' rows (node ids)
    };
    bool match_attention_chain(size_t attn_idx, AttnChain* out, size_t* seq) const;
    bool match_decode_attention(size_t attn_idx, AttnChain* out) const;
    bool deferred_to_attention(size_t idx) const;
    bool try_fuse_decode_attention(size_t attn_idx, std::vector<char>& done);
    bool try_fuse_norm_rope(size_t attn_idx, std::vector<char>& done);
    void make_view(GraphNode& n);
    void compute_matmul_node(GraphNode& n);
    void compute_attention_node(GraphNode& n);
    void compute_attention_int8_hybrid_node(GraphNode& n);
    void compute_quantize_activations_node(GraphNode& n);
    size_t binary_node(OpType op, size_t input1, size_t input2);
    void compute_binary_node(GraphNode& n);
    void compute_activation_node(GraphNode& n);
    void compute_rms_norm_node(GraphNode& n);
    void compute_rope_node(GraphNode& n);
    void compute_embedding_node(GraphNode& n);
    void compute_conv1d_causal_node(GraphNode& n);
    size_t scalar_node(OpType op, size_t input, float value);
    void compute_scalar_node(GraphNode& n);
    void compute_slice_node(GraphNode& n);     // SLICE and INDEX
    void compute_concat_node(GraphNode& n);
    void compute_sample_node(GraphNode& n);

    DevicePool pool_;
    bool streaming_twins_ = true;
    std::vector<const void*> twins_;   // row-major weights this graph gave a twin
    std::vector<void*> retired_;   // resident copies replaced under the same host address
    bool fusion_ = true;
    size_t fused_launches_ = 0;
    std::vector<std::vector<size_t>> consumers_;   // per node index: indices of the nodes that read it (this execute())

    std::vector<std::unique_ptr<GraphNode>> nodes_;
    // nodes a soft reset took out of the graph, kept for the next pass's add_node: a decode loop builds and drops
    // the same ~700 nodes per token, and their allocations were a third of what a token costs the host
    std::vector<std::unique_ptr<GraphNode>> spare_nodes_;
    std::unordered_map<size_t, size_t> node_index_map_;   // the nodes a reset kept (ids below pass_base_id_)
    size_t next_node_id_ = 0;
    size_t pass_base_id_ = 0, pass_base_idx_ = 0;          // nodes with id >= pass_base_id_ sit at pass_base_idx_ + (id - pass_base_id_)
    bool find_index(size_t id, size_t* idx) const;
    size_t index_of(size_t id) const;
    std::unordered_map<size_t, bool> is_weight_;   // node id -> bound with set_external_input
    // host pointer -> resident device copy (weights, scales)
    struct Resident { void* dev; size_t bytes; };
    std::unordered_map<const void*, Resident> resident_;
    // activation quantization shared by the matmuls that read one FP16 node (keyed by node id)
    struct QuantCache { void* q = nullptr; void* scales = nullptr; size_t M = 0, K = 0; };
    std::unordered_map<size_t, QuantCache> quant_cache_;
    std::unordered_map<std::string, size_t> weight_cache_;   // file name -> node id
    int32_t* index_scratch_ = nullptr;   // int32 copy of an EMBEDDING node's indices + one flag word
    size_t index_scratch_count_ = 0;
    int* index_flag_ = nullptr;          // set by an EMBEDDING launch that met an index >= its vocabulary
    bool index_flag_pending_ = false;    // read back once, behind the pass's last launch (check_index_flag)
    size_t index_flag_vocab_ = 0;
    void check_index_flag();
    // SAMPLE: device history uint32[128], its count, the draw, the status word; workspace half[vocab]
    uint32_t* sample_state_ = nullptr;
    void* sample_work_ = nullptr;
    size_t sample_work_vocab_ = 0;
};

// Header of a CACT tensor file (graph_io.cpp:11-16, 497-566): 80 bytes, little endian: magic
// "CACT"@0, version@4, flags@8, alignment@12, ndim@16, dims 4 x u64@20, precision@52,
// data bytes@56, scales bytes@64, group size@72, group count@76; the header is padded to
// `alignment`, the fp16 scales follow, then the payload at the next multiple of `alignment`.
struct CactHeader {
    uint32_t version = 0, alignment = 1, ndim = 0, precision = 0, group_size = 0, num_groups = 0;
    uint64_t dims[4] = {0, 0, 0, 0};
    uint64_t data_bytes = 0, scales_bytes = 0;
    uint64_t scales_offset = 0, data_offset = 0;   // derived
};
// Throws std::runtime_error with the reference's messages (graph_io.cpp:498-500, 507-509, 561-563).
CactHeader parse_cact_header(const void* bytes, size_t file_size);

// The decoder-side fields of the reference's Config (cactus/engine/engine.h:43-116) with its
// defaults, filled from a model folder's config.txt as Config::from_json reads it
// (engine_model.cpp:403-539): `key = value` lines, '#' comments, unknown keys ignored, sampling
// defaults chosen by model type.  The vision / image preprocessing keys are accepted and skipped.
struct ModelConfig {
    uint32_t vocab_size = 151936, bos_token_id = 151643, eos_token_id = 151645, num_layers = 28;
    uint32_t hidden_dim = 1024, ffn_intermediate_dim = 3072, attention_heads = 16, attention_kv_heads = 8;
    uint32_t attention_head_dim = 128;
    float layer_norm_eps = 1e-6f, rope_theta = 1000000.0f;
    uint32_t num_experts = 0, num_shared_experts = 0, num_top_experts = 0, moe_every_n_layers = 0;
    bool tie_word_embeddings = true;
    int model_type = 0;      // QWEN 0, GEMMA 1, SMOL 2, NOMIC 3, LFM2 5, WHISPER 7 (engine.h:92)
    int model_variant = 0;   // DEFAULT 0, VLM 1, EXTRACT 2, RAG 3
    int precision = 2;       // INT8 0, FP16 1, FP32 2 (engine.h:104-105)
    float default_temperature = 0.6f, default_top_p = 0.95f;
    uint32_t default_top_k = 20;
    uint32_t conv_L_cache = 0;
    std::vector<std::string> layer_types;
};
ModelConfig parse_model_config(const std::string& text);
ModelConfig load_model_config(const std::string& path);   // throws when the file cannot be opened

// INT8 KV cache resident in HBM: per layer keys/values int8 [slot, Hkv, D] and fp32 scales
// [slot, Hkv, ceil(D/128)], slots 0..sink-1 = first tokens, the rest = most recent tokens
// (engine.h:324-381, engine_cache.cpp:11-51, 101-337).
struct KVCache {
    struct LayerCache {
        int8_t* keys = nullptr;
        int8_t* values = nullptr;
        float* key_scales = nullptr;
        float* value_scales = nullptr;
        size_t capacity_tokens = 0;   // allocated slots
        size_t size_tokens = 0;       // logical slots (the reference's keys.size() in tokens)
    };
    size_t num_layers = 0, max_seq_len = 0, num_kv_heads = 0, head_dim = 0;
    size_t window_size = 0, sink_size = 4;
    size_t current_seq_len = 0;   // tokens held (<= window)
    size_t total_seq_len = 0;     // tokens seen (RoPE / position offset)
    // Ring mode: once the window is full, appended tokens overwrite the oldest slots of the recent
    // region [sink, window) in place instead of sliding that region down (the reference's memmove,
    // engine_cache.cpp:276-303).  The slots then hold the same tokens as without the ring, rotated:
    // slot sink + (ring_head + i) % (window - sink) is the i-th oldest recent token.  Every cached key
    // is visible to every query of the hybrid attention, so the rotation only permutes its sums.
    bool ring = false;
    size_t ring_head = 0;
    std::vector<LayerCache> layer_caches;

    ~KVCache();
    void init(size_t layers, size_t max_seq, size_t kv_heads, size_t dim);
    void set_window_size(size_t window, size_t sink);
    void set_ring(bool on);       // only while the cache is empty
    void reset();
    bool is_empty() const { return current_seq_len == 0; }
    size_t get_effective_seq_len() const { return current_seq_len; }
    size_t get_total_seq_len() const { return total_seq_len; }
    const int8_t* get_keys_int8(size_t layer) const;
    const int8_t* get_values_int8(size_t layer) const;
    const float* get_key_scales(size_t layer) const;
    const float* get_value_scales(size_t layer) const;
    // Take seq_len new tokens per layer from graph nodes holding FP16 K/V, either the whole
    // sequence [old+new, Hkv, D] or only the new tokens [seq_len, Hkv, D]
    // (engine_cache.cpp:101-337).
    void update_from_graph(CactusGraph* gb, const std::vector<size_t>& k_nodes,
                           const std::vector<size_t>& v_nodes, size_t seq_len, size_t layers,
                           size_t kv_heads, size_t dim);

private:
    void ensure_capacity(LayerCache& c, size_t tokens);
};

}  // namespace cactus_hip
