// Device-resident mirror of the reference's graph dispatch for the hot path (see graph.h).
#include "graph.h"
#include <chrono>

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>

#include "../../include/cactus_hip.h"
#include "../../include/cactus_hip_graph.h"
#include "kernels.h"

namespace cactus_hip {

namespace {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void abi_check(int rc) {
    if (rc != CACTUS_HIP_OK) throw std::runtime_error(std::string("cactus_hip: ") + cactus_hip_last_error());
}
void* dev_alloc(size_t bytes) {
    void* p = nullptr;
    hip_check(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc");
    return p;
}
size_t product(const std::vector<size_t>& s) {
    size_t n = 1;
    for (size_t d : s) n *= d;
    return n;
}

}  // namespace

size_t precision_size(Precision p) {
    switch (p) {
        case Precision::INT8: return 1;
        case Precision::FP16: return 2;
        case Precision::FP32: return 4;
        case Precision::INT4: return 1;
    }
    return 1;
}

// ------------------------------------------------------------------------- DevicePool

// Buffers come in power-of-two sizes from 1 KiB (the reference's BufferPool buckets the same way, graph_core.cpp:5-19),
// so graphs of different sequence lengths share them.  Idle buffers are kept up to max_idle_bytes_; beyond that
// trim() frees the largest buckets first.  Every release happens behind the work that used the buffer on the
// graph's one stream, and hipFree synchronizes the device, so a freed idle buffer is never still being read.
static size_t pool_bucket(size_t bytes) {
    size_t b = 1024;
    while (b < bytes) b <<= 1;
    return b;
}

void* CactusGraph::DevicePool::acquire(size_t bytes) {
    const size_t rounded = pool_bucket(bytes ? bytes : 1);
    auto it = free_.find(rounded);
    if (it != free_.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        idle_bytes_ -= rounded;
        return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, rounded) != hipSuccess) {
        (void)hipGetLastError();
        trim_to(0);                // out of memory: give back everything idle, then try once more
        p = dev_alloc(rounded);
    }
    size_of_[p] = rounded;
    allocations_++;
    return p;
}

void CactusGraph::DevicePool::release(void* p) {
    if (!p) return;
    auto it = size_of_.find(p);
    if (it == size_of_.end()) {
        (void)hipFree(p);
        return;
    }
    free_[it->second].push_back(p);
    idle_bytes_ += it->second;
}

// Unless the caller set a cap: an eighth of the device's memory.  A 4096-token prompt of a 0.6B model as one chunk
// holds 5.6 GB of node outputs; under a 1 GiB cap every soft reset gave 300 buffers back with hipFree (58 ms, five
// times the 11 ms the prompt's kernels take) and the next prompt allocated them again.
size_t CactusGraph::DevicePool::max_idle_bytes() {
    if (!cap_set_) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 8 > max_idle_bytes_) max_idle_bytes_ = total_b / 8;
        else (void)hipGetLastError();
        cap_set_ = true;
    }
    return max_idle_bytes_;
}

void CactusGraph::DevicePool::trim() { trim_to(max_idle_bytes()); }

void CactusGraph::DevicePool::trim_to(size_t cap) {
    while (idle_bytes_ > cap) {
        size_t largest = 0;
        for (auto& kv : free_)
            if (!kv.second.empty() && kv.first > largest) largest = kv.first;
        if (largest == 0) break;
        auto& v = free_[largest];
        void* p = v.back();
        v.pop_back();
        size_of_.erase(p);
        idle_bytes_ -= largest;
        (void)hipFree(p);
    }
}

void CactusGraph::DevicePool::clear() {
    for (auto& kv : size_of_) (void)hipFree(kv.first);
    size_of_.clear();
    free_.clear();
    idle_bytes_ = 0;
}

// ------------------------------------------------------------------------- CactusGraph

CactusGraph::CactusGraph() = default;
CactusGraph::~CactusGraph() { hard_reset(); }

// Nodes added since the last reset sit behind the kept ones in id order, so their index is arithmetic; only the kept
// nodes (weights, whose ids are older) go through the map - a decode loop's ~700 nodes per token no longer cost a hash
// insertion each.
bool CactusGraph::find_index(size_t id, size_t* idx) const {
    if (id >= pass_base_id_) {
        const size_t i = pass_base_idx_ + (id - pass_base_id_);
        if (i >= nodes_.size()) return false;
        *idx = i;
        return true;
    }
    auto it = node_index_map_.find(id);
    if (it == node_index_map_.end()) return false;
    *idx = it->second;
    return true;
}
size_t CactusGraph::index_of(size_t id) const {
    size_t i = 0;
    if (!find_index(id, &i)) throw std::out_of_range("CactusGraph: no node with id " + std::to_string(id));
    return i;
}
GraphNode& CactusGraph::node(size_t id) { return *nodes_[index_of(id)]; }
const GraphNode& CactusGraph::node(size_t id) const { return *nodes_[index_of(id)]; }
const BufferDesc& CactusGraph::get_output_buffer(size_t node_id) const { return node(node_id).output_buffer; }

// graph_builder.cpp:572-596.  One deliberate difference: a QUANTIZE_ACTIVATIONS node keeps the
// INT8 precision it asks for (the reference's rule hands it its input's FP16 precision, which
// leaves that builder unusable: no model calls it).
size_t CactusGraph::add_node(OpType op_type, const std::vector<size_t>& inputs,
                             const std::vector<size_t>& output_shape, const OpParams& params) {
    std::unique_ptr<GraphNode> n;
    if (!spare_nodes_.empty()) {            // a node of an earlier pass: everything reset, the vectors' storage kept
        n = std::move(spare_nodes_.back());
        spare_nodes_.pop_back();
        std::vector<size_t> ids = std::move(n->input_ids), shp = std::move(n->output_buffer.shape);
        *n = GraphNode();
        n->input_ids = std::move(ids);
        n->output_buffer.shape = std::move(shp);
    } else {
        n = std::make_unique<GraphNode>();
    }
    n->id = next_node_id_;
    n->op_type = op_type;
    n->input_ids = inputs;
    n->params = params;
    const std::vector<size_t>& shape = output_shape.empty() && !inputs.empty() ? node(inputs[0]).output_buffer.shape : output_shape;
    Precision prec = params.output_precision;
    if (op_type != OpType::QUANTIZE_ACTIVATIONS && prec == Precision::INT8 && !inputs.empty())
        prec = node(inputs[0]).output_buffer.precision;
    n->output_buffer.shape.assign(shape.begin(), shape.end());
    n->output_buffer.precision = prec;
    n->output_buffer.total_size = product(shape);
    n->output_buffer.byte_size = n->output_buffer.total_size * precision_size(prec);
    const size_t id = next_node_id_++;
    nodes_.push_back(std::move(n));
    return id;
}

size_t CactusGraph::input(const std::vector<size_t>& shape, Precision precision) {
    OpParams p;
    p.output_precision = precision;
    return add_node(OpType::INPUT, {}, shape, p);
}

size_t CactusGraph::quantize_activations(size_t input) {   // graph_builder.cpp:560-570
    const auto& in = get_output_buffer(input);
    if (in.precision != Precision::FP16) throw std::invalid_argument("quantize_activations requires FP16 input");
    OpParams p;
    p.output_precision = Precision::INT8;
    return add_node(OpType::QUANTIZE_ACTIVATIONS, {input}, in.shape, p);
}

size_t CactusGraph::matmul(size_t input1, size_t input2, bool pretransposed_rhs, ComputeBackend backend) {
    const auto& lhs = get_output_buffer(input1);   // graph_builder.cpp:61-81
    const auto& rhs = get_output_buffer(input2);
    if (lhs.shape.size() != 2 || rhs.shape.size() != 2)
        throw std::invalid_argument("Matrix multiplication requires 2D tensors");
    const size_t M = lhs.shape[0], K = lhs.shape[1];
    const size_t rhs_K = pretransposed_rhs ? rhs.shape[1] : rhs.shape[0];
    const size_t N = pretransposed_rhs ? rhs.shape[0] : rhs.shape[1];
    if (K != rhs_K) throw std::invalid_argument("Matrix dimensions incompatible for multiplication");
    OpParams p;
    p.pretransposed_rhs = pretransposed_rhs;
    p.backend = backend;
    p.output_precision = Precision::FP16;   // every matmul branch writes FP16 (graph_ops_nn.cpp:154,184,218)
    return add_node(OpType::MATMUL, {input1, input2}, {M, N}, p);
}

size_t CactusGraph::attention(size_t q, size_t k, size_t v, float scale, bool is_causal, ComputeBackend backend) {
    OpParams p;
    p.scale = scale; p.is_causal = is_causal; p.backend = backend;
    return add_node(OpType::ATTENTION, {q, k, v}, {}, p);
}
size_t CactusGraph::attention(size_t q, size_t k, size_t v, float scale, size_t position_offset,
                              ComputeBackend backend) {
    OpParams p;
    p.scale = scale; p.position_offset = position_offset; p.backend = backend;
    return add_node(OpType::ATTENTION, {q, k, v}, {}, p);
}
size_t CactusGraph::attention(size_t q, size_t k, size_t v, float scale, size_t position_offset,
                              size_t window_size, ComputeBackend backend) {
    OpParams p;
    p.scale = scale; p.position_offset = position_offset; p.window_size = window_size; p.backend = backend;
    return add_node(OpType::ATTENTION, {q, k, v}, {}, p);
}
size_t CactusGraph::attention_int8_hybrid(size_t q, size_t key_new, size_t value_new, float scale,
                                          size_t position_offset, const int8_t* cached_keys,
                                          const int8_t* cached_values, const float* k_scales,
                                          const float* v_scales, size_t cache_len, size_t num_kv_heads,
                                          size_t head_dim) {
    OpParams p;   // graph_builder.cpp:301-316
    p.scale = scale; p.position_offset = position_offset;
    p.cached_keys_int8 = cached_keys; p.cached_values_int8 = cached_values;
    p.cached_k_scales = k_scales; p.cached_v_scales = v_scales;
    p.cache_seq_len = cache_len; p.num_kv_heads = num_kv_heads; p.head_dim = head_dim;
    return add_node(OpType::ATTENTION_INT8_HYBRID, {q, key_new, value_new}, {}, p);
}

// graph_builder.cpp:11-59.  Equal shapes only: the reference also broadcasts
// (BroadcastInfo::compute); the decoder blocks add and multiply tensors of one shape.
size_t CactusGraph::binary_node(OpType op, size_t input1, size_t input2) {
    const auto& lhs = get_output_buffer(input1);
    const auto& rhs = get_output_buffer(input2);
    if (lhs.shape != rhs.shape)
        throw std::invalid_argument("Binary operations on the HIP graph require equal shapes (no broadcasting)");
    return add_node(op, {input1, input2}, lhs.shape);
}
size_t CactusGraph::add(size_t a, size_t b) { return binary_node(OpType::ADD, a, b); }
size_t CactusGraph::add_clipped(size_t a, size_t b) { return binary_node(OpType::ADD_CLIPPED, a, b); }
size_t CactusGraph::subtract(size_t a, size_t b) { return binary_node(OpType::SUBTRACT, a, b); }
size_t CactusGraph::multiply(size_t a, size_t b) { return binary_node(OpType::MULTIPLY, a, b); }
size_t CactusGraph::divide(size_t a, size_t b) { return binary_node(OpType::DIVIDE, a, b); }
size_t CactusGraph::silu(size_t input) { return add_node(OpType::SILU, {input}, {}); }   // graph_builder.cpp:468-474
size_t CactusGraph::gelu(size_t input) { return add_node(OpType::GELU, {input}, {}); }

size_t CactusGraph::reshape(size_t input, const std::vector<size_t>& new_shape) {   // graph_builder.cpp:120-123
    if (product(new_shape) != get_output_buffer(input).total_size)
        throw std::invalid_argument("Reshape must keep the number of elements");
    return add_node(OpType::RESHAPE, {input}, new_shape);
}

size_t CactusGraph::slice(size_t input, int axis, size_t start, size_t length) {   // graph_builder.cpp:499-522
    const auto& in = get_output_buffer(input);
    if (in.shape.empty()) throw std::runtime_error("Cannot slice a scalar tensor");
    if (axis < 0 || static_cast<size_t>(axis) >= in.shape.size()) throw std::runtime_error("Invalid axis for slice operation");
    if (start + length > in.shape[axis]) throw std::runtime_error("Slice range extends beyond axis size");
    std::vector<size_t> shape = in.shape;
    shape[axis] = length;
    OpParams p;
    p.axis = axis; p.slice_start = start; p.slice_length = length; p.output_precision = in.precision;
    return add_node(OpType::SLICE, {input}, shape, p);
}

size_t CactusGraph::concat(size_t input1, size_t input2, int axis) {   // graph_builder.cpp:348-375
    const auto& b1 = get_output_buffer(input1);
    const auto& b2 = get_output_buffer(input2);
    if (b1.shape.size() != b2.shape.size()) throw std::runtime_error("Concat requires inputs with same number of dimensions");
    const int nd = (int)b1.shape.size();
    if (axis < 0) axis += nd;
    if (axis < 0 || axis >= nd) throw std::runtime_error("Invalid axis for concat operation");
    for (int i = 0; i < nd; i++)
        if (i != axis && b1.shape[i] != b2.shape[i])
            throw std::runtime_error("Concat inputs must have same shape except on concat axis");
    std::vector<size_t> shape = b1.shape;
    shape[axis] = b1.shape[axis] + b2.shape[axis];
    OpParams p;
    p.axis = axis;
    return add_node(OpType::CONCAT, {input1, input2}, shape, p);
}

size_t CactusGraph::index(size_t input, size_t index_value, int dim) {   // graph_builder.cpp:125-160
    const auto& in = get_output_buffer(input);
    if (in.shape.empty()) throw std::invalid_argument("Cannot index a scalar tensor");
    int d = dim < 0 ? dim + (int)in.shape.size() : dim;
    if (d < 0 || static_cast<size_t>(d) >= in.shape.size()) throw std::invalid_argument("Index dimension out of bounds");
    if (index_value >= in.shape[d])
        throw std::invalid_argument("Index value " + std::to_string(index_value) + " out of bounds for dimension " +
                                    std::to_string(d) + " with size " + std::to_string(in.shape[d]));
    std::vector<size_t> shape;
    for (size_t i = 0; i < in.shape.size(); i++)
        if ((int)i != d) shape.push_back(in.shape[i]);
    if (shape.empty()) shape = {1};
    OpParams p;
    p.axis = d; p.index_value = index_value; p.output_precision = in.precision;
    return add_node(OpType::INDEX, {input}, shape, p);
}

size_t CactusGraph::sample(size_t logits, float temperature, float top_p, size_t top_k, uint32_t random_seed) {
    if (get_output_buffer(logits).shape.empty()) throw std::runtime_error("Sample requires non-empty logits tensor");
    OpParams p;
    p.temperature = temperature; p.top_p = top_p; p.top_k = top_k;
    p.random_seed = random_seed ? random_seed
                                : (uint32_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    p.output_precision = Precision::FP32;
    return add_node(OpType::SAMPLE, {logits}, {1}, p);
}

void CactusGraph::reset_sample_history() {
    if (sample_state_) hip_check(hipMemset(sample_state_, 0, 132 * sizeof(uint32_t)), "hipMemset (sample history)");
}

size_t CactusGraph::embedding(size_t embedding_tensor, size_t indices) {   // graph_builder.cpp:524-539
    const auto& emb = get_output_buffer(embedding_tensor);
    if (emb.shape.size() != 2) throw std::runtime_error("Embedding tensor must be 2D [vocab_size, hidden_dim]");
    std::vector<size_t> shape = get_output_buffer(indices).shape;
    shape.push_back(emb.shape[1]);
    OpParams p;
    p.output_precision = Precision::FP16;
    return add_node(OpType::EMBEDDING, {embedding_tensor, indices}, shape, p);
}

size_t CactusGraph::rms_norm(size_t input, size_t weight, float epsilon) {   // graph_builder.cpp:253-256
    OpParams p;
    p.epsilon = epsilon;
    return add_node(OpType::RMS_NORM, {input, weight}, {}, p);
}

size_t CactusGraph::rope(size_t input, float theta, size_t position_offset, ComputeBackend backend) {   // :258-261
    OpParams p;
    p.theta = theta; p.position_offset = position_offset; p.backend = backend;
    return add_node(OpType::ROPE, {input}, {}, p);
}

size_t CactusGraph::scalar_node(OpType op, size_t input, float value) {   // graph_builder.cpp:424-450
    OpParams p;
    p.scalar = value;
    p.output_precision = get_output_buffer(input).precision;
    return add_node(op, {input}, {}, p);
}
size_t CactusGraph::scalar_add(size_t input, float value) { return scalar_node(OpType::SCALAR_ADD, input, value); }
size_t CactusGraph::scalar_subtract(size_t input, float value) { return scalar_node(OpType::SCALAR_SUBTRACT, input, value); }
size_t CactusGraph::scalar_multiply(size_t input, float value) { return scalar_node(OpType::SCALAR_MULTIPLY, input, value); }
size_t CactusGraph::scalar_divide(size_t input, float value) { return scalar_node(OpType::SCALAR_DIVIDE, input, value); }

size_t CactusGraph::conv1d_causal(size_t input, size_t weight, size_t, size_t dilation) {   // graph_builder.cpp:318-321
    OpParams p;
    p.dilation = dilation;
    return add_node(OpType::CONV1D_CAUSAL, {input, weight}, {}, p);
}

void CactusGraph::ensure_device(BufferDesc& b) {
    if (b.device_data == nullptr) {
        b.device_data = pool_.acquire(b.byte_size);
        b.owns_device = true;
    }
}

void* CactusGraph::resident_copy(const void* host, size_t bytes) {
    auto it = resident_.find(host);
    if (it != resident_.end() && it->second.bytes == bytes) return it->second.dev;
    if (it != resident_.end()) {   // nodes bound earlier may still point at the old copy: it stays until hard_reset
        retired_.push_back(it->second.dev);
        resident_.erase(it);
    }
    void* d = dev_alloc(bytes);
    hip_check(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D (resident)");
    resident_[host] = Resident{d, bytes};
    return d;
}

// A grouped weight that has reached the device gets its tile-major twin (cactus_hip_weights_pack_dev): the decode
// matmuls (M <= 16) of every later pass stream the twin, launches with more rows keep reading the row-major copy.
// Not being able to afford the second copy is not an error.
void CactusGraph::make_twin(BufferDesc& b) {
    if (!streaming_twins_ || b.shape.size() != 2 || b.group_size != 128 || b.scales_device == nullptr) return;
    const bool int4 = b.is_packed_int4();
    const void* w = int4 ? b.packed_int4_device : b.device_data;
    const size_t N = b.shape[0], K = b.shape[1];
    if (w == nullptr || K % (int4 ? 256 : 128) != 0 || b.num_groups != K / 128) return;
    if (!int4 && b.precision != Precision::INT8) return;
    if (cactus_hip_weights_is_packed_dev(w)) return;
    if (cactus_hip_weights_pack_dev(w, N, K, int4 ? 1 : 0, nullptr) == CACTUS_HIP_OK) twins_.push_back(w);
    else cactus_hip_clear_error();
}

void CactusGraph::set_input(size_t node_id, const void* data, Precision) {
    auto& n = node(node_id);
    if (n.op_type != OpType::INPUT) throw std::invalid_argument("Can only set data on input nodes");
    BufferDesc& b = n.output_buffer;
    if (b.device_data != nullptr && !b.owns_device) b.device_data = nullptr;
    ensure_device(b);
    hip_check(hipMemcpy(b.device_data, data, b.byte_size, hipMemcpyHostToDevice), "hipMemcpy H2D (input)");
    b.device_valid = true;
}

void CactusGraph::set_external_input(size_t node_id, void* data, Precision) {
    auto& n = node(node_id);
    if (n.op_type != OpType::INPUT) throw std::invalid_argument("Can only set data on input nodes");
    BufferDesc& b = n.output_buffer;
    if (b.owns_device && b.device_data) pool_.release(b.device_data);
    b.device_data = resident_copy(data, b.byte_size);
    b.owns_device = false;
    b.device_valid = true;
    is_weight_[node_id] = true;
    make_twin(b);
}

void CactusGraph::set_device_input(size_t node_id, void* device_ptr, Precision) {
    auto& n = node(node_id);
    if (n.op_type != OpType::INPUT) throw std::invalid_argument("Can only set data on input nodes");
    BufferDesc& b = n.output_buffer;
    if (b.owns_device && b.device_data) pool_.release(b.device_data);
    b.device_data = device_ptr;
    b.owns_device = false;
    b.device_valid = true;
}

void CactusGraph::set_grouped_scales(size_t node_id, size_t group_size, size_t num_groups, void* scales_ptr) {
    BufferDesc& b = node(node_id).output_buffer;   // graph_io.cpp:106-112
    if (b.shape.empty()) throw std::invalid_argument("set_grouped_scales on a shapeless node");
    b.group_size = group_size;
    b.num_groups = num_groups;
    b.scales_device = resident_copy(scales_ptr, b.shape[0] * num_groups * sizeof(uint16_t));
    b.owns_scales = false;
    make_twin(b);
}

void CactusGraph::set_packed_int4(size_t node_id, const void* packed, size_t packed_bytes) {
    BufferDesc& b = node(node_id).output_buffer;
    b.packed_int4_device = resident_copy(packed, packed_bytes);
    b.packed_int4_size = packed_bytes;
    b.device_valid = true;
    is_weight_[node_id] = true;
    make_twin(b);
}

ModelConfig parse_model_config(const std::string& text) {   // engine_model.cpp:410-538
    ModelConfig c;
    auto trim = [](std::string v) {
        const size_t a = v.find_first_not_of(" \t");
        if (a == std::string::npos) return std::string();
        const size_t b = v.find_last_not_of(" \t");
        return v.substr(a, b - a + 1);
    };
    auto u32 = [](const std::string& v) { return static_cast<uint32_t>(std::stoul(v)); };
    auto flag = [](const std::string& v) { return v == "true" || v == "1"; };
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) nl = text.size();
        const std::string line = text.substr(pos, nl - pos);
        pos = nl + 1;
        if (line.empty() || line[0] == '#') continue;
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        const std::string key = trim(line.substr(0, eq)), value = trim(line.substr(eq + 1));
        if (key == "vocab_size") c.vocab_size = u32(value);
        else if (key == "bos_token_id") c.bos_token_id = u32(value);
        else if (key == "eos_token_id") c.eos_token_id = u32(value);
        else if (key == "num_layers") c.num_layers = u32(value);
        else if (key == "hidden_dim") c.hidden_dim = u32(value);
        else if (key == "ffn_intermediate_dim") c.ffn_intermediate_dim = u32(value);
        else if (key == "attention_heads") c.attention_heads = u32(value);
        else if (key == "attention_kv_heads") c.attention_kv_heads = u32(value);
        else if (key == "attention_head_dim") c.attention_head_dim = u32(value);
        else if (key == "layer_norm_eps") c.layer_norm_eps = std::stof(value);
        else if (key == "rope_theta") c.rope_theta = std::stof(value);
        else if (key == "num_experts") c.num_experts = u32(value);
        else if (key == "num_shared_experts") c.num_shared_experts = u32(value);
        else if (key == "num_top_experts") c.num_top_experts = u32(value);
        else if (key == "moe_every_n_layers") c.moe_every_n_layers = u32(value);
        else if (key == "tie_word_embeddings") c.tie_word_embeddings = flag(value);
        else if (key == "conv_L_cache") c.conv_L_cache = u32(value);
        else if (key == "precision") c.precision = value == "INT8" ? 0 : (value == "FP16" ? 1 : 2);
        else if (key == "model_type") {
            if (value == "gemma" || value == "GEMMA") c.model_type = 1;
            else if (value == "lfm2" || value == "LFM2") c.model_type = 5;
            else if (value == "smol" || value == "SMOL" || value == "Smol") c.model_type = 2;
            else if (value == "bert" || value == "BERT") c.model_type = 3;
            else if (value == "whisper" || value == "WHISPER") c.model_type = 7;
            else c.model_type = 0;
        } else if (key == "model_variant") {
            std::string v = value;
            std::transform(v.begin(), v.end(), v.begin(), [](unsigned char ch) { return (char)std::tolower(ch); });
            c.model_variant = v == "vlm" ? 1 : (v == "extract" ? 2 : (v == "rag" ? 3 : 0));
        } else if (key == "layer_types") {   // ['conv', 'full_attention', ...] -> items
            c.layer_types.clear();
            std::string item;
            auto flush = [&] {
                const std::string it = trim(item);
                if (!it.empty()) c.layer_types.push_back(it);
                item.clear();
            };
            for (char ch : value) {
                if (ch == '[' || ch == ']' || ch == '\'' || ch == '"') continue;
                if (ch == ',') flush();
                else item.push_back(ch);
            }
            flush();
        }
    }
    switch (c.model_type) {   // engine_model.cpp:512-536
        case 1: c.default_temperature = 1.0f; c.default_top_p = 0.95f; c.default_top_k = 64; break;
        case 2: c.default_temperature = 0.2f; c.default_top_p = 0.95f; c.default_top_k = 20; break;
        case 5: c.default_temperature = 0.3f; c.default_top_p = 0.95f; c.default_top_k = 20; break;
        case 0: c.default_temperature = 0.6f; c.default_top_p = 0.95f; c.default_top_k = 20; break;
        case 7: c.default_temperature = 0.0f; c.default_top_p = 0.0f; c.default_top_k = 0; break;
        default: break;
    }
    return c;
}

ModelConfig load_model_config(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed to open config file: " + path);
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    return parse_model_config(text);
}

CactHeader parse_cact_header(const void* bytes, size_t file_size) {   // graph_io.cpp:497-566
    constexpr size_t HEADER_SIZE = 80;
    if (file_size < HEADER_SIZE) throw std::runtime_error("File too small: insufficient data for header");
    const char* p = static_cast<const char*>(bytes);
    auto u32 = [&](size_t off) { uint32_t v; std::memcpy(&v, p + off, 4); return v; };
    auto u64 = [&](size_t off) { uint64_t v; std::memcpy(&v, p + off, 8); return v; };
    if (u32(0) != 0x54434143u) throw std::runtime_error("Invalid tensor file: missing CACT magic number");
    CactHeader h;
    h.version = u32(4);
    h.alignment = u32(12) ? u32(12) : 1;
    h.ndim = u32(16);
    for (int i = 0; i < 4; i++) h.dims[i] = u64(20 + 8 * i);
    h.precision = u32(52);
    h.data_bytes = u64(56);
    h.scales_bytes = u64(64);
    h.group_size = u32(72);
    h.num_groups = u32(76);
    auto align = [&](uint64_t off) { const uint64_t r = off % h.alignment; return r ? off + (h.alignment - r) : off; };
    const uint64_t aligned_header = align(HEADER_SIZE);
    if (h.scales_bytes > 0) {
        h.scales_offset = aligned_header;
        h.data_offset = align(h.scales_offset + h.scales_bytes);
    } else {
        h.scales_offset = 0;
        h.data_offset = aligned_header;
    }
    if (h.data_offset + h.data_bytes > file_size)
        throw std::runtime_error("File corrupted: data extends beyond file size (v2)");
    return h;
}

size_t CactusGraph::mmap_weights(const std::string& filename) {   // graph_io.cpp:48-72, 470-495
    auto it = weight_cache_.find(filename);
    if (it != weight_cache_.end()) return it->second;
    const int fd = open(filename.c_str(), O_RDONLY);
    if (fd == -1) throw std::runtime_error("Cannot open file for mapping: " + filename);
    struct stat st;
    if (fstat(fd, &st) == -1) { close(fd); throw std::runtime_error("Cannot get file size: " + filename); }
    const size_t file_size = (size_t)st.st_size;
    void* map = mmap(nullptr, file_size ? file_size : 1, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (map == MAP_FAILED) throw std::runtime_error("Cannot map file: " + filename);
    struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n ? n : 1); } } unmap{map, file_size};
    const CactHeader h = parse_cact_header(map, file_size);
    std::vector<size_t> shape;
    for (uint32_t i = 0; i < 4; i++)
        if (i < h.ndim && h.dims[i] > 0) shape.push_back((size_t)h.dims[i]);
    if (h.precision > 3) throw std::runtime_error("Unknown precision in tensor file: " + filename);
    const bool int4 = h.precision == (uint32_t)Precision::INT4;
    const Precision prec = int4 ? Precision::INT8 : static_cast<Precision>(h.precision);   // effective_precision()
    // the file is validated as a whole before the graph gains a node: a matmul trusts these sizes
    const size_t elements = product(shape);
    const size_t need_bytes = int4 ? (elements + 1) / 2 : elements * precision_size(prec);
    if (h.data_bytes < need_bytes)
        throw std::runtime_error("File corrupted: payload smaller than the tensor: " + filename);
    const bool grouped = h.group_size > 0 && h.scales_bytes > 0 && (int4 || prec == Precision::INT8);
    if (grouped) {
        if (shape.empty() || h.scales_bytes < shape[0] * (size_t)h.num_groups * sizeof(uint16_t) ||
            h.scales_offset + h.scales_bytes > file_size)
            throw std::runtime_error("File corrupted: group scales smaller than [rows, groups]: " + filename);
    }
    const char* base = static_cast<const char*>(map);
    void* dev = dev_alloc(h.data_bytes);
    hipError_t ce = hipMemcpy(dev, base + h.data_offset, h.data_bytes, hipMemcpyHostToDevice);
    void* dev_scales = nullptr;
    if (ce == hipSuccess && grouped) {
        if (hipMalloc(&dev_scales, h.scales_bytes) != hipSuccess) ce = hipErrorOutOfMemory;
        else ce = hipMemcpy(dev_scales, base + h.scales_offset, h.scales_bytes, hipMemcpyHostToDevice);
    }
    if (ce != hipSuccess) {
        (void)hipFree(dev);
        if (dev_scales) (void)hipFree(dev_scales);
        hip_check(ce, "hipMemcpy H2D (weights)");
    }
    const size_t id = input(shape, prec);
    BufferDesc& b = node(id).output_buffer;
    if (int4) {
        b.packed_int4_device = dev;
        b.packed_int4_size = h.data_bytes;
        b.owns_packed = true;
    } else {
        b.device_data = dev;
        b.owns_device = true;
    }
    b.device_valid = true;
    if (grouped) {
        b.group_size = h.group_size;
        b.num_groups = h.num_groups;
        b.scales_device = dev_scales;
        b.owns_scales = true;
    }
    is_weight_[id] = true;
    weight_cache_[filename] = id;
    make_twin(node(id).output_buffer);
    return id;
}

// graph_ops_nn.cpp:129-231
void CactusGraph::compute_matmul_node(GraphNode& n) {
    BufferDesc& lhs = node(n.input_ids[0]).output_buffer;
    BufferDesc& rhs = node(n.input_ids[1]).output_buffer;
    const size_t M = lhs.shape[lhs.shape.size() - 2];
    const size_t K = lhs.shape[lhs.shape.size() - 1];
    const bool pre = n.params.pretransposed_rhs;
    const size_t N = pre ? rhs.shape[rhs.shape.size() - 2] : rhs.shape[rhs.shape.size() - 1];
    if (n.params.backend == ComputeBackend::NPU)
        throw std::runtime_error("NPU matrix multiplication not yet implemented");
    auto* out = static_cast<cactus_half_t*>(n.output_buffer.device_data);
    const bool lhs_prequant = lhs.precision == Precision::INT8 && lhs.has_activation_scales();

    const bool int4 = rhs.is_packed_int4();
    if (int4 || rhs.is_grouped_int8()) {
        if (!pre)
            throw std::runtime_error(int4 ? "INT4 matmul requires pretransposed weights"
                                          : "Group-wise INT8 matmul requires pretransposed weights");
        const int8_t* a8;
        const float* as;
        if (lhs_prequant) {
            a8 = static_cast<const int8_t*>(lhs.device_data);
            as = static_cast<const float*>(lhs.activation_scales_device);
        } else if (lhs.precision == Precision::FP16 && M <= 4) {
            // decode rows: the GEMV kernel quantizes the activations itself (one launch, and
            // cheaper than a shared quantization pass even when q/k/v read the same node)
            const auto* x = static_cast<const cactus_half_t*>(lhs.device_data);
            const auto* bs = static_cast<const cactus_half_t*>(rhs.scales_device);
            if (int4)
                abi_check(cactus_hip_matmul_int4_f16act_dev(x, static_cast<const uint8_t*>(rhs.packed_int4_device),
                                                            bs, out, M, K, N, rhs.group_size, nullptr));
            else
                abi_check(cactus_hip_matmul_int8_f16act_dev(x, static_cast<const int8_t*>(rhs.device_data), bs,
                                                            out, M, K, N, rhs.group_size, nullptr));
            return;
        } else if (lhs.precision == Precision::FP16) {
            // quantize once per FP16 node and step; q/k/v (and gate/up) share one input node
            QuantCache& qc = quant_cache_[n.input_ids[0]];
            if (qc.q == nullptr || qc.M != M || qc.K != K) {
                if (qc.q) { pool_.release(qc.q); pool_.release(qc.scales); }
                qc.q = pool_.acquire(M * K);
                qc.scales = pool_.acquire(M * sizeof(float));
                qc.M = M; qc.K = K;
                abi_check(cactus_hip_quantize_activations_dev(static_cast<const cactus_half_t*>(lhs.device_data),
                                                              static_cast<int8_t*>(qc.q),
                                                              static_cast<float*>(qc.scales), M, K, nullptr));
            }
            a8 = static_cast<const int8_t*>(qc.q);
            as = static_cast<const float*>(qc.scales);
        } else {
            throw std::runtime_error(int4 ? "INT4 matmul requires INT8 (pre-quantized) or FP16 activations"
                                          : "INT8 matmul requires INT8 (pre-quantized) or FP16 activations");
        }
        const auto* bs = static_cast<const cactus_half_t*>(rhs.scales_device);
        if (int4)
            abi_check(cactus_hip_matmul_int4_dev(a8, as, static_cast<const uint8_t*>(rhs.packed_int4_device), bs,
                                                 out, M, K, N, rhs.group_size, nullptr));
        else
            abi_check(cactus_hip_matmul_int8_dev(a8, as, static_cast<const int8_t*>(rhs.device_data), bs, out, M,
                                                 K, N, rhs.group_size, nullptr));
        return;
    }
    if (lhs.precision != Precision::FP16) throw std::runtime_error("FP16 matmul requires FP16 activations");
    const auto* a = static_cast<const cactus_half_t*>(lhs.device_data);
    const auto* b = static_cast<const cactus_half_t*>(rhs.device_data);
    if (pre) {
        abi_check(cactus_hip_matmul_f16_dev(a, b, out, M, K, N, nullptr));
    } else {
        // rhs is [K, N]: transpose into a scratch buffer, then the same kernel
        // (graph_ops_nn.cpp:222-229 does the same on the host)
        void* bt = pool_.acquire(N * K * 2);
        hipError_t e = launch_transpose_f16(static_cast<const uint16_t*>(rhs.device_data),
                                            static_cast<uint16_t*>(bt), K, N, nullptr);
        int rc = CACTUS_HIP_OK;
        if (e == hipSuccess) rc = cactus_hip_matmul_f16_dev(a, static_cast<const cactus_half_t*>(bt), out, M, K, N, nullptr);
        pool_.release(bt);   // whoever takes it next runs behind this matmul on the same stream
        hip_check(e, "transpose launch");
        abi_check(rc);
    }
}

// graph_ops_nn.cpp:78-127
void CactusGraph::compute_quantize_activations_node(GraphNode& n) {
    BufferDesc& in = node(n.input_ids[0]).output_buffer;
    if (in.precision != Precision::FP16) throw std::runtime_error("QUANTIZE_ACTIVATIONS requires FP16 input");
    if (in.shape.size() < 2) throw std::runtime_error("QUANTIZE_ACTIVATIONS requires at least 2D tensor");
    const size_t K = in.shape.back();
    const size_t M = in.total_size / K;
    BufferDesc& out = n.output_buffer;
    if (!out.has_activation_scales() || out.num_rows_for_activation_scales != M) {
        if (out.activation_scales_device) pool_.release(out.activation_scales_device);
        out.activation_scales_device = pool_.acquire(M * sizeof(float));
        out.num_rows_for_activation_scales = M;
    }
    abi_check(cactus_hip_quantize_activations_dev(static_cast<const cactus_half_t*>(in.device_data),
                                                  static_cast<int8_t*>(out.device_data),
                                                  static_cast<float*>(out.activation_scales_device), M, K, nullptr));
}

// graph_ops_nn.cpp:302-338
void CactusGraph::compute_attention_node(GraphNode& n) {
    if (n.params.backend == ComputeBackend::NPU)
        throw std::runtime_error("NPU attention operation not yet implemented");
    if (n.input_ids.size() < 3)
        throw std::runtime_error("Attention operation requires 3 inputs (query, key, value), got " +
                                 std::to_string(n.input_ids.size()) + " inputs");
    const BufferDesc& q = node(n.input_ids[0]).output_buffer;
    const BufferDesc& k = node(n.input_ids[1]).output_buffer;
    const BufferDesc& v = node(n.input_ids[2]).output_buffer;
    if (q.shape.size() < 4)
        throw std::runtime_error("Attention operation requires 4D tensors [batch, seq_len, num_heads, head_dim], got " +
                                 std::to_string(q.shape.size()) + "D tensor");
    if (q.precision != Precision::FP16) throw std::runtime_error("Attention operation only supports FP16 precision");
    abi_check(cactus_hip_attention_f16_dev(
        static_cast<const cactus_half_t*>(q.device_data), static_cast<const cactus_half_t*>(k.device_data),
        static_cast<const cactus_half_t*>(v.device_data), static_cast<cactus_half_t*>(n.output_buffer.device_data),
        q.shape[0], q.shape[1], k.shape[1], q.shape[2], k.shape[2], q.shape[3], n.params.scale, nullptr,
        n.params.position_offset, n.params.window_size, n.params.is_causal ? 1 : 0, nullptr));
}

// graph_ops_nn.cpp:340-371
void CactusGraph::compute_attention_int8_hybrid_node(GraphNode& n) {
    const BufferDesc& q = node(n.input_ids[0]).output_buffer;
    const BufferDesc& kn = node(n.input_ids[1]).output_buffer;
    const BufferDesc& vn = node(n.input_ids[2]).output_buffer;
    if (q.shape.size() < 4) throw std::runtime_error("ATTENTION_INT8_HYBRID requires 4D query tensor");
    abi_check(cactus_hip_attention_hybrid_int8_fp16_dev(
        static_cast<const cactus_half_t*>(q.device_data), n.params.cached_keys_int8, n.params.cached_values_int8,
        n.params.cached_k_scales, n.params.cached_v_scales, static_cast<const cactus_half_t*>(kn.device_data),
        static_cast<const cactus_half_t*>(vn.device_data), static_cast<cactus_half_t*>(n.output_buffer.device_data),
        q.shape[0], q.shape[1], n.params.cache_seq_len, kn.shape[1], q.shape[2], n.params.num_kv_heads,
        n.params.head_dim, n.params.scale, n.params.position_offset, 1, CACTUS_KV_QUANT_GROUP_SIZE, nullptr));
}

// graph_ops_math.cpp:110-153 without the broadcasting branch
void CactusGraph::compute_binary_node(GraphNode& n) {
    const BufferDesc& lhs = node(n.input_ids[0]).output_buffer;
    const BufferDesc& rhs = node(n.input_ids[1]).output_buffer;
    if (lhs.precision != Precision::FP16) throw std::runtime_error("Binary operations only support FP16 precision");
    const cactus_half_t* a = static_cast<const cactus_half_t*>(lhs.device_data);
    const cactus_half_t* b = static_cast<const cactus_half_t*>(rhs.device_data);
    cactus_half_t* o = static_cast<cactus_half_t*>(n.output_buffer.device_data);
    const size_t count = n.output_buffer.total_size;
    switch (n.op_type) {
        case OpType::ADD: abi_check(cactus_hip_add_f16_dev(a, b, o, count, nullptr)); break;
        case OpType::ADD_CLIPPED: abi_check(cactus_hip_add_f16_clipped_dev(a, b, o, count, nullptr)); break;
        case OpType::SUBTRACT: abi_check(cactus_hip_subtract_f16_dev(a, b, o, count, nullptr)); break;
        case OpType::MULTIPLY: abi_check(cactus_hip_multiply_f16_dev(a, b, o, count, nullptr)); break;
        default: abi_check(cactus_hip_divide_f16_dev(a, b, o, count, nullptr)); break;
    }
}

// graph_ops_tensor.cpp:138-224 (slice), 350-395 (index): `count` entries from `first` along the axis,
// one strided device copy (axis 0 never gets here: it is a view).
void CactusGraph::compute_slice_node(GraphNode& n) {
    const BufferDesc& in = node(n.input_ids[0]).output_buffer;
    const size_t axis = (size_t)n.params.axis;
    const bool is_index = n.op_type == OpType::INDEX;
    const size_t first = is_index ? n.params.index_value : n.params.slice_start;
    const size_t count = is_index ? 1 : (n.params.slice_length ? n.params.slice_length : in.shape[axis] - first);
    size_t outer = 1, inner = precision_size(in.precision);
    for (size_t i = 0; i < axis; i++) outer *= in.shape[i];
    for (size_t i = axis + 1; i < in.shape.size(); i++) inner *= in.shape[i];
    if (outer == 0 || count == 0 || inner == 0) return;
    hip_check(hipMemcpy2DAsync(n.output_buffer.device_data, count * inner,
                               static_cast<const char*>(in.device_data) + first * inner, in.shape[axis] * inner,
                               count * inner, outer, hipMemcpyDeviceToDevice, nullptr), "slice copy");
}

// graph_ops_sample.cpp:8-39: the last row of the logits through cactus_sample_f16.
void CactusGraph::compute_sample_node(GraphNode& n) {
    const BufferDesc& lg = node(n.input_ids[0]).output_buffer;
    if (lg.shape.size() != 2) throw std::runtime_error("Sample expects 2D logits tensor [seq_len, vocab_size]");
    if (lg.precision != Precision::FP16) throw std::runtime_error("Sample supports FP16 logits on the device");
    const size_t seq_len = lg.shape[0], vocab = lg.shape[1];
    if (seq_len == 0) throw std::runtime_error("Sample requires non-empty logits tensor");
    if (!sample_state_) {   // [0,128) history, [128] count, [129] draw, [130] status
        sample_state_ = static_cast<uint32_t*>(dev_alloc(132 * sizeof(uint32_t)));
        hip_check(hipMemset(sample_state_, 0, 132 * sizeof(uint32_t)), "hipMemset (sample state)");
    }
    if (sample_work_vocab_ < vocab) {
        if (sample_work_) (void)hipFree(sample_work_);
        sample_work_ = dev_alloc(vocab * sizeof(cactus_half_t));
        sample_work_vocab_ = vocab;
    }
    const float draw = cactus_hip_sample_uniform(n.params.random_seed);
    hip_check(hipMemcpy(sample_state_ + 129, &draw, sizeof draw, hipMemcpyHostToDevice), "hipMemcpy H2D (draw)");
    hip_check(hipMemset(sample_state_ + 130, 0, sizeof(int)), "hipMemset (sample status)");
    abi_check(cactus_hip_sample_f16_dev(static_cast<const cactus_half_t*>(lg.device_data) + (seq_len - 1) * vocab,
                                        static_cast<uint32_t*>(n.output_buffer.device_data), 1, vocab,
                                        n.params.temperature, n.params.top_p, n.params.top_k,
                                        reinterpret_cast<const float*>(sample_state_ + 129), sample_state_,
                                        sample_state_ + 128, static_cast<cactus_half_t*>(sample_work_),
                                        reinterpret_cast<int*>(sample_state_ + 130), nullptr));
    int status = 0;
    hip_check(hipMemcpy(&status, sample_state_ + 130, sizeof status, hipMemcpyDeviceToHost), "hipMemcpy D2H (sample status)");
    if (status != 0) throw std::runtime_error("Sample: more logits survive top-k than the device sampler holds");
}

// graph_ops_tensor.cpp:333-348
void CactusGraph::compute_concat_node(GraphNode& n) {
    const BufferDesc& a = node(n.input_ids[0]).output_buffer;
    const BufferDesc& b = node(n.input_ids[1]).output_buffer;
    if (a.precision != Precision::FP16 || b.precision != Precision::FP16)
        throw std::runtime_error("Concat operation only supports FP16 precision");
    const size_t axis = (size_t)n.params.axis;
    size_t outer = 1, inner = precision_size(a.precision);
    for (size_t i = 0; i < axis; i++) outer *= a.shape[i];
    for (size_t i = axis + 1; i < a.shape.size(); i++) inner *= a.shape[i];
    const size_t wa = a.shape[axis] * inner, wb = b.shape[axis] * inner;
    char* out = static_cast<char*>(n.output_buffer.device_data);
    if (outer == 0 || wa + wb == 0) return;
    if (wa) hip_check(hipMemcpy2DAsync(out, wa + wb, a.device_data, wa, wa, outer, hipMemcpyDeviceToDevice, nullptr), "concat copy");
    if (wb) hip_check(hipMemcpy2DAsync(out + wa, wa + wb, b.device_data, wb, wb, outer, hipMemcpyDeviceToDevice, nullptr), "concat copy");
}

// graph_ops_math.cpp:66-93 (the FP16 branch)
void CactusGraph::compute_scalar_node(GraphNode& n) {
    const BufferDesc& in = node(n.input_ids[0]).output_buffer;
    if (in.precision != Precision::FP16) throw std::runtime_error("Scalar operations only support FP16 precision");
    const int op = n.op_type == OpType::SCALAR_ADD ? 0 : n.op_type == OpType::SCALAR_SUBTRACT ? 1
                 : n.op_type == OpType::SCALAR_MULTIPLY ? 2 : 3;
    abi_check(cactus_hip_scalar_op_f16_dev(static_cast<const cactus_half_t*>(in.device_data),
                                           static_cast<cactus_half_t*>(n.output_buffer.device_data),
                                           n.output_buffer.total_size, n.params.scalar, op, nullptr));
}

// graph_ops_math.cpp:158-180
void CactusGraph::compute_activation_node(GraphNode& n) {
    const BufferDesc& in = node(n.input_ids[0]).output_buffer;
    if (in.precision != Precision::FP16) throw std::runtime_error("Activation operations only support FP16 precision");
    const cactus_half_t* x = static_cast<const cactus_half_t*>(in.device_data);
    cactus_half_t* o = static_cast<cactus_half_t*>(n.output_buffer.device_data);
    if (n.op_type == OpType::SILU) abi_check(cactus_hip_silu_f16_dev(x, o, n.output_buffer.total_size, nullptr));
    else abi_check(cactus_hip_gelu_f16_dev(x, o, n.output_buffer.total_size, nullptr));
}

// graph_ops_nn.cpp:233-251
void CactusGraph::compute_rms_norm_node(GraphNode& n) {
    const BufferDesc& in = node(n.input_ids[0]).output_buffer;
    const BufferDesc& w = node(n.input_ids[1]).output_buffer;
    if (in.shape.size() != 2)
        throw std::runtime_error("RMS normalization requires 2D input tensor [batch_size, dims], got " +
                                 std::to_string(in.shape.size()) + "D tensor");
    if (in.precision != Precision::FP16) throw std::runtime_error("RMS normalization only supports FP16 precision");
    if (w.total_size < in.shape[1]) throw std::runtime_error("RMS normalization weight is shorter than dims");
    abi_check(cactus_hip_rms_norm_f16_dev(static_cast<const cactus_half_t*>(in.device_data),
                                          static_cast<const cactus_half_t*>(w.device_data),
                                          static_cast<cactus_half_t*>(n.output_buffer.device_data), in.shape[0],
                                          in.shape[1], n.params.epsilon, nullptr));
}

// graph_ops_nn.cpp:459-530 (the FP16-weight branch)
void CactusGraph::compute_conv1d_causal_node(GraphNode& n) {
    if (n.params.backend == ComputeBackend::NPU)
        throw std::runtime_error("NPU causal convolution operation not yet implemented");
    const BufferDesc& x = node(n.input_ids[0]).output_buffer;
    const BufferDesc& w = node(n.input_ids[1]).output_buffer;
    if (x.shape.size() != 3) throw std::runtime_error("Causal conv requires 3D input [batch, seq_len, in_channels]");
    if (w.shape.size() != 3) throw std::runtime_error("Weight must be 3D");
    if (n.params.dilation < 1) throw std::runtime_error("dilation must be >= 1");
    if (w.shape[1] != 1 || w.shape[0] != x.shape[2])
        throw std::runtime_error("Only depthwise causal convolution is supported currently");
    if (w.precision != Precision::FP16 || x.precision != Precision::FP16)
        throw std::runtime_error("Depthwise causal conv supports FP16 weights and inputs on the device");
    abi_check(cactus_hip_conv1d_causal_depthwise_f16_dev(static_cast<const cactus_half_t*>(x.device_data),
                                                         static_cast<const cactus_half_t*>(w.device_data),
                                                         static_cast<cactus_half_t*>(n.output_buffer.device_data),
                                                         x.shape[0], x.shape[1], x.shape[2], w.shape[2],
                                                         n.params.dilation, nullptr));
}

// graph_ops_nn.cpp:253-277
void CactusGraph::compute_rope_node(GraphNode& n) {
    if (n.params.backend == ComputeBackend::NPU) throw std::runtime_error("NPU RoPE operation not yet implemented");
    const BufferDesc& in = node(n.input_ids[0]).output_buffer;
    if (in.shape.size() < 4)
        throw std::runtime_error("RoPE operation requires 4D tensor with shape [batch, seq_len, num_heads, head_dim], got " +
                                 std::to_string(in.shape.size()) + "D tensor");
    if (in.precision != Precision::FP16 || n.output_buffer.precision != Precision::FP16)
        throw std::runtime_error("RoPE operation only supports FP16 precision");
    abi_check(cactus_hip_rope_f16_dev(static_cast<const cactus_half_t*>(in.device_data),
                                      static_cast<cactus_half_t*>(n.output_buffer.device_data), in.shape[0],
                                      in.shape[1], in.shape[2], in.shape[3], n.params.position_offset,
                                      n.params.theta, nullptr));
}

// graph_ops_tensor.cpp:226-320.  Indices arrive as FP32 or INT8 values (:257-275), are widened
// to int32 on the device, and an index outside the table raises the reference's exception
// (:261-263) after the lookup kernel has reported it.
void CactusGraph::compute_embedding_node(GraphNode& n) {
    const BufferDesc& emb = node(n.input_ids[0]).output_buffer;
    const BufferDesc& idx = node(n.input_ids[1]).output_buffer;
    const size_t vocab = emb.shape[0], hidden = emb.shape[1], count = idx.total_size;
    if (idx.precision != Precision::FP32 && idx.precision != Precision::INT8)
        throw std::runtime_error("Embedding indices must be FP32 or INT8");
    if (index_scratch_count_ < count + 1) {
        if (index_scratch_) (void)hipFree(index_scratch_);
        index_scratch_ = static_cast<int32_t*>(dev_alloc((count + 1) * sizeof(int32_t)));
        index_scratch_count_ = count + 1;
    }
    // The flag is read back once, when the pass's launches are all enqueued (check_index_flag): a read here would hold
    // the host until the lookup has run while the GPU waits for the launches behind it.
    if (!index_flag_) {
        index_flag_ = static_cast<int*>(dev_alloc(sizeof(int)));
        hip_check(hipMemsetAsync(index_flag_, 0, sizeof(int), nullptr), "hipMemsetAsync");
    }
    int* flag = index_flag_;
    abi_check(cactus_hip_indices_to_int32_dev(idx.device_data, idx.precision == Precision::FP32 ? 2 : 0,
                                              index_scratch_, count, nullptr));
    cactus_half_t* out = static_cast<cactus_half_t*>(n.output_buffer.device_data);
    if (emb.precision == Precision::INT8) {
        if (!emb.is_grouped_int8() || emb.scales_device == nullptr)
            throw std::runtime_error("INT8 embedding tables need group scales on the HIP graph");
        abi_check(cactus_hip_embedding_int8_dev(static_cast<const int8_t*>(emb.device_data),
                                                static_cast<const cactus_half_t*>(emb.scales_device), index_scratch_,
                                                out, count, vocab, hidden, emb.group_size, flag, nullptr));
    } else if (emb.precision == Precision::FP16) {
        abi_check(cactus_hip_embedding_f16_dev(static_cast<const cactus_half_t*>(emb.device_data), index_scratch_, out,
                                               count, vocab, hidden, flag, nullptr));
    } else {
        throw std::runtime_error("Embedding table must be INT8 or FP16");
    }
    index_flag_pending_ = true;
    index_flag_vocab_ = vocab;
}

void CactusGraph::check_index_flag() {
    if (!index_flag_pending_) return;
    index_flag_pending_ = false;
    int bad = 0;
    hip_check(hipMemcpy(&bad, index_flag_, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy D2H (index flag)");
    if (bad) {
        hip_check(hipMemsetAsync(index_flag_, 0, sizeof(int), nullptr), "hipMemsetAsync");
        throw std::runtime_error("Embedding index out of bounds: >= " + std::to_string(index_flag_vocab_));
    }
}

// A matmul node the batched decode launch can take: grouped INT8 or packed INT4 weights,
// pretransposed, FP16 activations with at most 4 rows, inputs already computed.
bool CactusGraph::batchable_matmul(const GraphNode& n, bool* int4) const {
    if (n.op_type != OpType::MATMUL || n.params.backend == ComputeBackend::NPU || !n.params.pretransposed_rhs)
        return false;
    const BufferDesc& lhs = node(n.input_ids[0]).output_buffer;
    const BufferDesc& rhs = node(n.input_ids[1]).output_buffer;
    if (!lhs.device_valid || !rhs.device_valid) return false;
    if (lhs.precision != Precision::FP16 || lhs.shape.size() != 2 || lhs.shape[0] > 4) return false;
    if (!rhs.is_packed_int4() && !rhs.is_grouped_int8()) return false;
    if (rhs.group_size != 128 || rhs.scales_device == nullptr) return false;
    *int4 = rhs.is_packed_int4();
    return true;
}

// A matmul node that can take its FP16 rows from a producer fused into its launch: grouped INT8 or packed
// INT4 weights (group 128), pretransposed, left operand = node `lhs_id`, weights resident.
bool CactusGraph::fusable_matmul(const GraphNode& n, size_t lhs_id, bool* int4) const {
    if (n.op_type != OpType::MATMUL || n.params.backend == ComputeBackend::NPU || !n.params.pretransposed_rhs)
        return false;
    if (n.input_ids[0] != lhs_id || n.input_ids[1] == lhs_id) return false;
    const BufferDesc& rhs = node(n.input_ids[1]).output_buffer;
    if (!rhs.device_valid) return false;
    if (!rhs.is_packed_int4() && !rhs.is_grouped_int8()) return false;
    if (rhs.group_size != 128 || rhs.scales_device == nullptr) return false;
    *int4 = rhs.is_packed_int4();
    return true;
}

// Every reader of node `producer_idx` is a fusable matmul (and there is at least one): their indices by precision.
bool CactusGraph::consumer_matmuls(size_t producer_idx, std::vector<size_t>* int8_idx, std::vector<size_t>* int4_idx) const {
    const std::vector<size_t>& cons = consumers_[producer_idx];
    if (cons.empty()) return false;
    const size_t id = nodes_[producer_idx]->id;
    for (size_t c : cons) {
        bool i4 = false;
        if (!fusable_matmul(*nodes_[c], id, &i4)) return false;
        (i4 ? int4_idx : int8_idx)->push_back(c);
    }
    return int8_idx->size() <= 16 && int4_idx->size() <= 16;
}

void CactusGraph::launch_produced(int kind, bool int4, const std::vector<size_t>& mm, const void* x, const void* x2,
                                  const void* w, void* sum_out, float eps, size_t M, size_t K) {
    std::vector<cactus_hip_matmul_weights> wt;
    for (size_t c : mm) {
        GraphNode& m = *nodes_[c];
        const BufferDesc& rhs = node(m.input_ids[1]).output_buffer;
        ensure_device(m.output_buffer);
        cactus_hip_matmul_weights e;
        e.B = int4 ? rhs.packed_int4_device : rhs.device_data;
        e.B_scales = static_cast<const cactus_half_t*>(rhs.scales_device);
        e.C = static_cast<cactus_half_t*>(m.output_buffer.device_data);
        e.N = rhs.shape[0];
        wt.push_back(e);
    }
    const auto* hx = static_cast<const cactus_half_t*>(x);
    const auto* hx2 = static_cast<const cactus_half_t*>(x2);
    int rc;
    if (kind == 1) {
        auto fn = int4 ? cactus_hip_matmul_int4_norm_f16act_dev : cactus_hip_matmul_int8_norm_f16act_dev;
        rc = fn(hx, hx2, static_cast<const cactus_half_t*>(w), static_cast<cactus_half_t*>(sum_out), eps, wt.data(),
                wt.size(), M, K, 128, nullptr);
    } else if (kind == 2) {
        auto fn = int4 ? cactus_hip_matmul_int4_silu_f16act_dev : cactus_hip_matmul_int8_silu_f16act_dev;
        rc = fn(hx, hx2, wt.data(), wt.size(), M, K, 128, nullptr);
    } else {
        auto fn = int4 ? cactus_hip_matmul_int4_gelu_f16act_dev : cactus_hip_matmul_int8_gelu_f16act_dev;
        rc = fn(hx, hx2, wt.data(), wt.size(), M, K, 128, nullptr);
    }
    abi_check(rc);
    fused_launches_++;
}

// nodes_[norm_idx] is an RMS_NORM whose readers are all fusable matmuls: one launch per weight precision
// computes them from the norm's input; when that input is a plain ADD of two computed nodes that has not
// run yet, the add goes into the same launch (its sum is written out: the residual stream reads it later).
bool CactusGraph::try_fuse_norm(size_t norm_idx, std::vector<char>& done) {
    GraphNode& r = *nodes_[norm_idx];
    if (r.op_type != OpType::RMS_NORM) return false;
    const size_t src_idx = index_of(r.input_ids[0]);
    GraphNode& src = *nodes_[src_idx];
    const BufferDesc& w = node(r.input_ids[1]).output_buffer;
    const BufferDesc& shape_of = src.output_buffer;
    if (shape_of.shape.size() != 2 || shape_of.precision != Precision::FP16 || !w.device_valid) return false;
    const size_t M = shape_of.shape[0], K = shape_of.shape[1];
    if (M == 0 || K == 0 || K % 128 != 0 || K > 16384 || w.total_size < K || w.precision != Precision::FP16) return false;
    std::vector<size_t> m8, m4;
    if (!consumer_matmuls(norm_idx, &m8, &m4)) return false;
    const void *x = nullptr, *x2 = nullptr;
    void* sum_out = nullptr;
    bool with_add = false;
    if (!done[src_idx]) {
        // the add in front, not yet computed: same shapes on both sides (no broadcast), operands computed
        if (src.op_type != OpType::ADD || src.input_ids.size() != 2) return false;
        const BufferDesc& a = node(src.input_ids[0]).output_buffer;
        const BufferDesc& b = node(src.input_ids[1]).output_buffer;
        if (!a.device_valid || !b.device_valid || a.precision != Precision::FP16 || b.precision != Precision::FP16 ||
            a.total_size != shape_of.total_size || b.total_size != shape_of.total_size)
            return false;
        ensure_device(src.output_buffer);
        if (src.output_buffer.device_data == a.device_data || src.output_buffer.device_data == b.device_data) return false;
        x = a.device_data; x2 = b.device_data; sum_out = src.output_buffer.device_data;
        with_add = true;
    } else {
        if (!src.output_buffer.device_valid) return false;
        x = src.output_buffer.device_data;
    }
    if (!m8.empty()) launch_produced(1, false, m8, x, x2, w.device_data, sum_out, r.params.epsilon, M, K);
    if (!m4.empty()) launch_produced(1, true, m4, x, x2, w.device_data, sum_out, r.params.epsilon, M, K);
    if (with_add) { src.output_buffer.device_valid = true; done[src_idx] = 1; }
    done[norm_idx] = 1;      // never materialized: get_output() computes it on demand
    for (size_t c : m8) { nodes_[c]->output_buffer.device_valid = true; done[c] = 1; }
    for (size_t c : m4) { nodes_[c]->output_buffer.device_valid = true; done[c] = 1; }
    return true;
}

// nodes_[mul_idx] is MULTIPLY(act(gate), up) with act = SILU or GELU read by nothing else, and every reader
// of the product is a fusable matmul.
bool CactusGraph::try_fuse_activation(size_t mul_idx, std::vector<char>& done) {
    GraphNode& m = *nodes_[mul_idx];
    if (m.op_type != OpType::MULTIPLY || m.input_ids.size() != 2) return false;
    const size_t act_idx = index_of(m.input_ids[0]);
    GraphNode& act = *nodes_[act_idx];
    if (act.op_type != OpType::SILU && act.op_type != OpType::GELU) return false;
    if (done[act_idx] || consumers_[act_idx].size() != 1) return false;
    const BufferDesc& gate = node(act.input_ids[0]).output_buffer;
    const BufferDesc& up = node(m.input_ids[1]).output_buffer;
    if (!gate.device_valid || !up.device_valid || gate.precision != Precision::FP16 || up.precision != Precision::FP16)
        return false;
    if (gate.shape.size() != 2 || gate.total_size != up.total_size || m.output_buffer.total_size != gate.total_size) return false;
    const size_t M = gate.shape[0], K = gate.shape[1];
    if (M == 0 || K == 0 || K % 128 != 0 || K > 16384) return false;
    std::vector<size_t> m8, m4;
    if (!consumer_matmuls(mul_idx, &m8, &m4)) return false;
    const int kind = act.op_type == OpType::SILU ? 2 : 3;
    if (!m8.empty()) launch_produced(kind, false, m8, gate.device_data, up.device_data, nullptr, nullptr, 0.0f, M, K);
    if (!m4.empty()) launch_produced(kind, true, m4, gate.device_data, up.device_data, nullptr, nullptr, 0.0f, M, K);
    done[act_idx] = 1;
    done[mul_idx] = 1;
    for (size_t c : m8) { nodes_[c]->output_buffer.device_valid = true; done[c] = 1; }
    for (size_t c : m4) { nodes_[c]->output_buffer.device_valid = true; done[c] = 1; }
    return true;
}

// A node a fused launch skipped (RMS_NORM, SILU / GELU, MULTIPLY; RESHAPE and ROPE in front of a fused decode
// attention), computed now from its inputs.
void CactusGraph::materialize(size_t idx) {
    GraphNode& n = *nodes_[idx];
    if (n.output_buffer.device_valid) return;
    if (n.op_type != OpType::RMS_NORM && n.op_type != OpType::SILU && n.op_type != OpType::GELU &&
        n.op_type != OpType::MULTIPLY && n.op_type != OpType::RESHAPE && n.op_type != OpType::ROPE)
        return;
    for (size_t in : n.input_ids) {
        materialize(index_of(in));
        if (!node(in).output_buffer.device_valid) return;
    }
    if (n.op_type == OpType::RESHAPE) {
        make_view(n);
        return;
    }
    ensure_device(n.output_buffer);
    if (n.op_type == OpType::RMS_NORM) compute_rms_norm_node(n);
    else if (n.op_type == OpType::MULTIPLY) compute_binary_node(n);
    else if (n.op_type == OpType::ROPE) compute_rope_node(n);
    else compute_activation_node(n);
    n.output_buffer.device_valid = true;
}

// RESHAPE: a view - same bytes, new shape
void CactusGraph::make_view(GraphNode& n) {
    BufferDesc& in = node(n.input_ids[0]).output_buffer;
    BufferDesc& out = n.output_buffer;
    if (out.owns_device && out.device_data) pool_.release(out.device_data);
    out.device_data = in.device_data;
    out.owns_device = false;
    out.device_valid = true;
}

// nodes_[attn_idx] is an ATTENTION or ATTENTION_INT8_HYBRID over the new tokens of one sequence whose q and k inputs
// are each rope(reshape(rms_norm(reshape(projection)))) - per-head norm and RoPE of the projections' rows, as every
// decoder of the reference builds them (model_qwen.cpp:63-84) - read by nothing else in the graph, with equal
// epsilon, theta and position on both sides.  *seq: the new tokens.
bool CactusGraph::match_attention_chain(size_t attn_idx, AttnChain* out, size_t* seq) const {
    const GraphNode& a = *nodes_[attn_idx];
    if ((a.op_type != OpType::ATTENTION_INT8_HYBRID && a.op_type != OpType::ATTENTION) || a.input_ids.size() < 3) return false;
    const OpParams& p = a.params;
    const BufferDesc& q = node(a.input_ids[0]).output_buffer;
    const BufferDesc& kb = node(a.input_ids[1]).output_buffer;
    if (q.shape.size() != 4 || kb.shape.size() != 4 || q.shape[0] != 1 || q.shape[1] == 0 || kb.shape[1] != q.shape[1]) return false;
    const size_t S = q.shape[1], D = q.shape[3], Hkv = kb.shape[2];
    if (Hkv == 0 || q.shape[2] % Hkv != 0 || kb.shape[3] != D) return false;
    auto chain = [&](size_t rope_id, size_t heads, size_t* norm_i, size_t* view_i, size_t* rope_i, size_t* src_id) {
        size_t rope_idx = 0;
        if (!find_index(rope_id, &rope_idx)) return false;
        const GraphNode& rope = *nodes_[rope_idx];
        if (rope.op_type != OpType::ROPE || rope.params.backend == ComputeBackend::NPU) return false;
        const std::vector<size_t>& rs = rope.output_buffer.shape;
        if (rs.size() != 4 || rs[0] != 1 || rs[1] != S || rs[2] != heads || rs[3] != D) return false;
        if (rope.output_buffer.precision != Precision::FP16 || consumers_[rope_idx].size() != 1) return false;
        const size_t vi = index_of(rope.input_ids[0]);
        const GraphNode& view = *nodes_[vi];
        if (view.op_type != OpType::RESHAPE || consumers_[vi].size() != 1) return false;
        const size_t ni = index_of(view.input_ids[0]);
        const GraphNode& norm = *nodes_[ni];
        if (norm.op_type != OpType::RMS_NORM || consumers_[ni].size() != 1) return false;
        const BufferDesc& in = node(norm.input_ids[0]).output_buffer;
        const BufferDesc& w = node(norm.input_ids[1]).output_buffer;
        if (in.shape.size() != 2 || in.shape[0] != S * heads || in.shape[1] != D || in.precision != Precision::FP16) return false;
        if (w.precision != Precision::FP16 || w.total_size < D || !w.device_valid) return false;
        *norm_i = ni; *view_i = vi; *rope_i = rope_idx; *src_id = norm.input_ids[0];
        return true;
    };
    AttnChain c;
    if (!chain(a.input_ids[0], q.shape[2], &c.q_norm, &c.q_view, &c.q_rope, &c.q_src)) return false;
    if (!chain(a.input_ids[1], Hkv, &c.k_norm, &c.k_view, &c.k_rope, &c.k_src)) return false;
    const OpParams& qr = nodes_[c.q_rope]->params;
    const OpParams& kr = nodes_[c.k_rope]->params;
    if (qr.theta != kr.theta || qr.position_offset != kr.position_offset || qr.position_offset != p.position_offset) return false;
    if (nodes_[c.q_norm]->params.epsilon != nodes_[c.k_norm]->params.epsilon) return false;
    if (out) *out = c;
    if (seq) *seq = S;
    return true;
}

// ... an ATTENTION_INT8_HYBRID over ONE new token, on a shape the fused decode kernel takes
bool CactusGraph::match_decode_attention(size_t attn_idx, AttnChain* out) const {
    const GraphNode& a = *nodes_[attn_idx];
    if (a.op_type != OpType::ATTENTION_INT8_HYBRID || a.input_ids.size() < 3) return false;
    const OpParams& p = a.params;
    const size_t D = p.head_dim, Hkv = p.num_kv_heads;
    if (p.cache_seq_len == 0 || !p.cached_keys_int8 || !p.cached_values_int8 || !p.cached_k_scales || !p.cached_v_scales)
        return false;
    if (D != 64 && D != 128 && D != 256) return false;
    const BufferDesc& q = node(a.input_ids[0]).output_buffer;
    const BufferDesc& kb = node(a.input_ids[1]).output_buffer;
    if (q.shape.size() != 4 || Hkv == 0 || q.shape[2] % Hkv != 0 || q.shape[2] / Hkv > 16 || q.shape[3] != D) return false;
    if (kb.shape.size() != 4 || kb.shape[2] != Hkv) return false;
    size_t S = 0;
    AttnChain c;
    if (!match_attention_chain(attn_idx, &c, &S) || S != 1) return false;
    const BufferDesc& v = node(a.input_ids[2]).output_buffer;
    if (v.precision != Precision::FP16 || v.total_size != Hkv * D) return false;
    if (out) *out = c;
    return true;
}

// RMS_NORM / RESHAPE / ROPE node idx sits in such a run: it is left for the attention node to take
bool CactusGraph::deferred_to_attention(size_t idx) const {
    size_t i = idx;
    auto is_attn = [&](size_t j) { return nodes_[j]->op_type == OpType::ATTENTION_INT8_HYBRID || nodes_[j]->op_type == OpType::ATTENTION; };
    for (int hop = 0; hop < 3; hop++) {       // norm -> view -> rope -> attention
        if (consumers_[i].size() != 1) return false;
        i = consumers_[i][0];
        if (is_attn(i)) break;
    }
    AttnChain c;
    if (!is_attn(i) || !match_attention_chain(i, &c, nullptr)) return false;
    return idx == c.q_norm || idx == c.q_view || idx == c.q_rope || idx == c.k_norm || idx == c.k_view || idx == c.k_rope;
}

// One launch for the run; the rotated key node is written by it (the KV cache is fed from that node), the
// other five nodes stay unmaterialized until somebody asks for their output.
bool CactusGraph::try_fuse_decode_attention(size_t attn_idx, std::vector<char>& done) {
    AttnChain c;
    if (!match_decode_attention(attn_idx, &c)) return false;
    for (size_t i : {c.q_norm, c.q_view, c.q_rope, c.k_norm, c.k_view, c.k_rope})
        if (done[i]) return false;      // computed on its own already: the plain path
    GraphNode& a = *nodes_[attn_idx];
    const BufferDesc& qs = node(c.q_src).output_buffer;
    const BufferDesc& ks = node(c.k_src).output_buffer;
    const BufferDesc& v = node(a.input_ids[2]).output_buffer;
    if (!qs.device_valid || !ks.device_valid || !v.device_valid) return false;
    GraphNode& krope = *nodes_[c.k_rope];
    ensure_device(a.output_buffer);
    ensure_device(krope.output_buffer);
    const OpParams& p = a.params;
    const int rc = cactus_hip_attention_decode_fused_single_f16_dev(
        static_cast<const cactus_half_t*>(qs.device_data), static_cast<const cactus_half_t*>(ks.device_data),
        static_cast<const cactus_half_t*>(v.device_data),
        static_cast<const cactus_half_t*>(node(nodes_[c.q_norm]->input_ids[1]).output_buffer.device_data),
        static_cast<const cactus_half_t*>(node(nodes_[c.k_norm]->input_ids[1]).output_buffer.device_data),
        p.cached_keys_int8, p.cached_values_int8, p.cached_k_scales, p.cached_v_scales,
        static_cast<cactus_half_t*>(a.output_buffer.device_data), static_cast<cactus_half_t*>(krope.output_buffer.device_data),
        p.cache_seq_len, p.position_offset, node(a.input_ids[0]).output_buffer.shape[2], p.num_kv_heads, p.head_dim,
        nodes_[c.q_norm]->params.epsilon, nodes_[c.q_rope]->params.theta, p.scale, CACTUS_KV_QUANT_GROUP_SIZE, nullptr);
    if (rc != CACTUS_HIP_OK) return false;      // a shape or alignment the kernel does not take: the plain path
    fused_launches_++;
    for (size_t i : {c.q_norm, c.q_view, c.q_rope, c.k_norm, c.k_view}) done[i] = 1;   // not valid: computed on demand
    krope.output_buffer.device_valid = true;
    done[c.k_rope] = 1;
    a.output_buffer.device_valid = true;
    done[attn_idx] = 1;
    return true;
}

// The same run in front of an attention over several new tokens (a prompt or a chunk of one): the two per-head norms
// and the two RoPEs in one launch (cactus_hip_qk_norm_rope_f16_dev: the arithmetic of the four separate calls); the two
// rotated nodes are written, the norms and views stay unmaterialized until somebody asks for them; the attention
// node then runs as it is.
bool CactusGraph::try_fuse_norm_rope(size_t attn_idx, std::vector<char>& done) {
    AttnChain c;
    size_t S = 0;
    if (!match_attention_chain(attn_idx, &c, &S)) return false;
    for (size_t i : {c.q_norm, c.q_view, c.q_rope, c.k_norm, c.k_view, c.k_rope})
        if (done[i]) return false;
    const GraphNode& a = *nodes_[attn_idx];
    const BufferDesc& qs = node(c.q_src).output_buffer;
    const BufferDesc& ks = node(c.k_src).output_buffer;
    if (!qs.device_valid || !ks.device_valid) return false;
    GraphNode& qrope = *nodes_[c.q_rope];
    GraphNode& krope = *nodes_[c.k_rope];
    ensure_device(qrope.output_buffer);
    ensure_device(krope.output_buffer);
    const std::vector<size_t>& qsh = qrope.output_buffer.shape;
    const int rc = cactus_hip_qk_norm_rope_f16_dev(
        static_cast<const cactus_half_t*>(qs.device_data), static_cast<const cactus_half_t*>(ks.device_data),
        static_cast<const cactus_half_t*>(node(nodes_[c.q_norm]->input_ids[1]).output_buffer.device_data),
        static_cast<const cactus_half_t*>(node(nodes_[c.k_norm]->input_ids[1]).output_buffer.device_data),
        static_cast<cactus_half_t*>(qrope.output_buffer.device_data), static_cast<cactus_half_t*>(krope.output_buffer.device_data),
        1, S, qsh[2], krope.output_buffer.shape[2], qsh[3], nodes_[c.q_norm]->params.epsilon, a.params.position_offset,
        qrope.params.theta, nullptr);
    if (rc != CACTUS_HIP_OK) return false;      // a shape the launch does not take: the plain path
    fused_launches_++;
    for (size_t i : {c.q_norm, c.q_view, c.k_norm, c.k_view}) done[i] = 1;     // not valid: computed on demand
    qrope.output_buffer.device_valid = krope.output_buffer.device_valid = true;
    done[c.q_rope] = done[c.k_rope] = 1;
    return true;
}

void CactusGraph::execute() {
    for (auto& kv : quant_cache_) {
        if (kv.second.q) { pool_.release(kv.second.q); pool_.release(kv.second.scales); }
    }
    quant_cache_.clear();
    // Every computed node is stale until this pass reaches it: a graph may be executed again after
    // set_input without a reset, and "is this input valid yet" is what orders the batched matmul runs.
    for (auto& up : nodes_)
        if (up->op_type != OpType::INPUT) up->output_buffer.device_valid = false;
    std::vector<char> done(nodes_.size(), 0);
    if (consumers_.size() < nodes_.size()) consumers_.resize(nodes_.size());     // the lists keep their storage from pass to pass
    for (size_t i = 0; i < nodes_.size(); i++) consumers_[i].clear();
    for (size_t idx = 0; idx < nodes_.size(); idx++)
        for (size_t in : nodes_[idx]->input_ids) {
            size_t pi = 0;
            if (find_index(in, &pi)) consumers_[pi].push_back(idx);
        }
    for (size_t idx = 0; idx < nodes_.size(); idx++) {
        GraphNode& n = *nodes_[idx];
        if (done[idx]) continue;       // taken by a fused launch further up
        if (n.op_type == OpType::INPUT) {
            if (!n.output_buffer.device_valid)
                throw std::runtime_error("Input node " + std::to_string(n.id) + " has no data");
            done[idx] = 1;
            continue;
        }
        if (fusion_) {
            // an ADD read by a norm whose rows only feed matmuls: add + norm + matmuls in one launch
            if (n.op_type == OpType::ADD) {
                bool taken = false;
                for (size_t c : consumers_[idx])
                    if (nodes_[c]->op_type == OpType::RMS_NORM && nodes_[c]->input_ids[0] == n.id && !done[c] &&
                        try_fuse_norm(c, done)) {
                        taken = true;
                        break;
                    }
                if (taken) continue;
            }
            if (n.op_type == OpType::RMS_NORM && node(n.input_ids[0]).output_buffer.device_valid && try_fuse_norm(idx, done))
                continue;
            // an activation read only by a multiply: wait for the multiply, which may take both into its readers' launch
            if ((n.op_type == OpType::SILU || n.op_type == OpType::GELU) && consumers_[idx].size() == 1 &&
                nodes_[consumers_[idx][0]]->op_type == OpType::MULTIPLY && nodes_[consumers_[idx][0]]->input_ids[0] == n.id &&
                node(n.input_ids[0]).output_buffer.device_valid)
                continue;               // not done: the multiply decides
            if (n.op_type == OpType::MULTIPLY) {
                const size_t a_idx = index_of(n.input_ids[0]);
                if (!done[a_idx]) {     // the activation deferred above
                    if (node(n.input_ids[1]).output_buffer.device_valid && try_fuse_activation(idx, done)) continue;
                    materialize(a_idx);
                    done[a_idx] = 1;
                }
            }
            // per-head norm and RoPE in front of a one-token attention over the cache: the attention node takes
            // them into its launch
            if ((n.op_type == OpType::RMS_NORM || n.op_type == OpType::RESHAPE || n.op_type == OpType::ROPE) &&
                deferred_to_attention(idx))
                continue;               // not done: the attention node decides
            if (n.op_type == OpType::ATTENTION_INT8_HYBRID || n.op_type == OpType::ATTENTION) {
                const size_t q_idx = index_of(n.input_ids[0]);
                if (!done[q_idx]) {     // the run deferred above
                    if (n.op_type == OpType::ATTENTION_INT8_HYBRID && try_fuse_decode_attention(idx, done)) continue;
                    if (!try_fuse_norm_rope(idx, done))
                    for (int s = 0; s < 2; s++) {           // the plain path: compute what was left for this node
                        const size_t r = index_of(n.input_ids[s]);
                        const size_t v = index_of(nodes_[r]->input_ids[0]);
                        const size_t m = index_of(nodes_[v]->input_ids[0]);
                        materialize(r);
                        done[m] = done[v] = done[r] = 1;
                    }
                }
            }
        }
        for (size_t in : n.input_ids)
            if (!node(in).output_buffer.device_valid)
                throw std::runtime_error("Node " + std::to_string(n.id) + " reads a node without data");

        // Consecutive decode matmuls that do not depend on each other (q/k/v, gate/up) go out as
        // one launch; a node that reads a member of the run ends it (its input is not valid yet).
        bool int4 = false;
        if (batchable_matmul(n, &int4)) {
            std::vector<cactus_hip_matmul_problem> run;
            size_t last = idx;
            for (size_t j = idx; j < nodes_.size() && run.size() < 16; j++) {
                GraphNode& m = *nodes_[j];
                bool m4 = false;
                if (done[j] || !batchable_matmul(m, &m4) || m4 != int4) break;
                BufferDesc& lhs = node(m.input_ids[0]).output_buffer;
                BufferDesc& rhs = node(m.input_ids[1]).output_buffer;
                ensure_device(m.output_buffer);
                cactus_hip_matmul_problem pr;
                pr.X = static_cast<const cactus_half_t*>(lhs.device_data);
                pr.B = int4 ? rhs.packed_int4_device : rhs.device_data;
                pr.B_scales = static_cast<const cactus_half_t*>(rhs.scales_device);
                pr.C = static_cast<cactus_half_t*>(m.output_buffer.device_data);
                pr.M = lhs.shape[0];
                pr.K = lhs.shape[1];
                pr.N = rhs.shape[0];
                run.push_back(pr);
                last = j;
            }
            if (run.size() > 1) {
                abi_check(int4 ? cactus_hip_matmul_int4_f16act_batch_dev(run.data(), run.size(), 128, nullptr)
                               : cactus_hip_matmul_int8_f16act_batch_dev(run.data(), run.size(), 128, nullptr));
                for (size_t j = idx; j <= last; j++) { nodes_[j]->output_buffer.device_valid = true; done[j] = 1; }
                idx = last;
                continue;
            }
        }

        if (n.op_type == OpType::RESHAPE) {
            make_view(n);
            continue;
        }
        if ((n.op_type == OpType::SLICE || n.op_type == OpType::INDEX) && n.params.axis == 0) {
            // a view of the input's rows (graph_ops_tensor.cpp:154-168, 361-366)
            BufferDesc& in = node(n.input_ids[0]).output_buffer;
            BufferDesc& out = n.output_buffer;
            const size_t first = n.op_type == OpType::INDEX ? n.params.index_value : n.params.slice_start;
            const size_t row_bytes = in.shape[0] ? in.byte_size / in.shape[0] : 0;
            if (out.owns_device && out.device_data) pool_.release(out.device_data);
            out.device_data = static_cast<char*>(in.device_data) + first * row_bytes;
            out.owns_device = false;
            out.device_valid = true;
            continue;
        }
        ensure_device(n.output_buffer);
        switch (n.op_type) {
            case OpType::ADD: case OpType::ADD_CLIPPED: case OpType::SUBTRACT: case OpType::MULTIPLY:
            case OpType::DIVIDE: compute_binary_node(n); break;
            case OpType::SILU: case OpType::GELU: compute_activation_node(n); break;
            case OpType::RMS_NORM: compute_rms_norm_node(n); break;
            case OpType::ROPE: compute_rope_node(n); break;
            case OpType::EMBEDDING: compute_embedding_node(n); break;
            case OpType::CONV1D_CAUSAL: compute_conv1d_causal_node(n); break;
            case OpType::SCALAR_ADD: case OpType::SCALAR_SUBTRACT: case OpType::SCALAR_MULTIPLY:
            case OpType::SCALAR_DIVIDE: compute_scalar_node(n); break;
            case OpType::SLICE: case OpType::INDEX: compute_slice_node(n); break;
            case OpType::CONCAT: compute_concat_node(n); break;
            case OpType::SAMPLE: compute_sample_node(n); break;
            case OpType::MATMUL: compute_matmul_node(n); break;
            case OpType::ATTENTION: compute_attention_node(n); break;
            case OpType::ATTENTION_INT8_HYBRID: compute_attention_int8_hybrid_node(n); break;
            case OpType::QUANTIZE_ACTIVATIONS: compute_quantize_activations_node(n); break;
            default: break;
        }
        n.output_buffer.device_valid = true;
        done[idx] = 1;
    }
    check_index_flag();
}

void* CactusGraph::get_device_output(size_t node_id) {
    BufferDesc& b = node(node_id).output_buffer;
    if (!b.device_valid) {      // a node a fused launch skipped
        materialize(index_of(node_id));
        check_index_flag();
    }
    return b.device_data;
}

void* CactusGraph::get_output(size_t node_id) {   // graph_execute.cpp:223-229; synchronises
    BufferDesc& b = node(node_id).output_buffer;
    if (!b.device_valid) {      // a node a fused launch skipped
        materialize(index_of(node_id));
        check_index_flag();
    }
    if (!b.device_valid || b.device_data == nullptr) return nullptr;
    b.host_copy.resize(b.byte_size);
    hip_check(hipMemcpy(b.host_copy.data(), b.device_data, b.byte_size, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    return b.host_copy.data();
}

// node buffers and scratch go back to the pool (a weight file's payload, which the pool did not hand
// out, is freed by it)
static void free_node(GraphNode& n, CactusGraph::DevicePool& pool) {
    BufferDesc& b = n.output_buffer;
    if (b.owns_device && b.device_data) pool.release(b.device_data);
    if (b.activation_scales_device) pool.release(b.activation_scales_device);
    if (b.owns_scales && b.scales_device) (void)hipFree(b.scales_device);
    if (b.owns_packed && b.packed_int4_device) (void)hipFree(b.packed_int4_device);
    b.scales_device = nullptr;
    b.packed_int4_device = nullptr;
    b.device_data = nullptr;
    b.activation_scales_device = nullptr;
}

void CactusGraph::soft_reset() {   // graph_execute.cpp:640-675: weights survive, everything else goes
    std::vector<std::unique_ptr<GraphNode>> keep;
    for (auto& up : nodes_) {
        if (is_weight_.count(up->id)) {
            keep.push_back(std::move(up));
        } else {
            free_node(*up, pool_);
            if (spare_nodes_.size() < 4096) {
                if (!up->output_buffer.host_copy.empty()) std::vector<char>().swap(up->output_buffer.host_copy);   // no output copy outlives its pass
                spare_nodes_.push_back(std::move(up));
            }
        }
    }
    nodes_ = std::move(keep);
    // the kept nodes' places (normally the same as after the last reset: nothing to insert); what this pass adds is
    // found by arithmetic
    if (node_index_map_.size() > nodes_.size()) node_index_map_.clear();      // (a kept node went: start over)
    for (size_t i = 0; i < nodes_.size(); i++) {
        auto it = node_index_map_.find(nodes_[i]->id);
        if (it == node_index_map_.end()) node_index_map_[nodes_[i]->id] = i;
        else it->second = i;
    }
    if (node_index_map_.size() != nodes_.size()) {                            // stale entries: rebuild
        node_index_map_.clear();
        for (size_t i = 0; i < nodes_.size(); i++) node_index_map_[nodes_[i]->id] = i;
    }
    pass_base_id_ = next_node_id_;
    pass_base_idx_ = nodes_.size();
    for (auto& kv : quant_cache_)
        if (kv.second.q) { pool_.release(kv.second.q); pool_.release(kv.second.scales); }
    quant_cache_.clear();
    pool_.trim();
}

void CactusGraph::hard_reset() {
    for (const void* w : twins_) (void)cactus_hip_weights_release_packed_dev(w);
    twins_.clear();
    for (auto& up : nodes_) free_node(*up, pool_);
    nodes_.clear();
    spare_nodes_.clear();
    node_index_map_.clear();
    pass_base_id_ = 0;
    pass_base_idx_ = 0;
    is_weight_.clear();
    weight_cache_.clear();
    for (auto& kv : resident_) (void)hipFree(kv.second.dev);
    resident_.clear();
    for (void* p : retired_) (void)hipFree(p);
    retired_.clear();
    for (auto& kv : quant_cache_)
        if (kv.second.q) { pool_.release(kv.second.q); pool_.release(kv.second.scales); }
    quant_cache_.clear();
    (void)hipDeviceSynchronize();   // pooled buffers may still be read by kernels in flight
    pool_.clear();
    if (index_scratch_) (void)hipFree(index_scratch_);
    index_scratch_ = nullptr;
    index_scratch_count_ = 0;
    if (index_flag_) (void)hipFree(index_flag_);
    index_flag_ = nullptr;
    index_flag_pending_ = false;
    if (sample_state_) (void)hipFree(sample_state_);
    if (sample_work_) (void)hipFree(sample_work_);
    sample_state_ = nullptr;
    sample_work_ = nullptr;
    sample_work_vocab_ = 0;
    next_node_id_ = 0;
}

// ------------------------------------------------------------------------------ KVCache

KVCache::~KVCache() {
    for (auto& c : layer_caches) {
        if (c.keys) (void)hipFree(c.keys);
        if (c.values) (void)hipFree(c.values);
        if (c.key_scales) (void)hipFree(c.key_scales);
        if (c.value_scales) (void)hipFree(c.value_scales);
    }
}

void KVCache::init(size_t layers, size_t max_seq, size_t kv_heads, size_t dim) {   // engine_cache.cpp:11-23
    num_layers = layers;
    max_seq_len = max_seq;
    num_kv_heads = kv_heads;
    head_dim = dim;
    layer_caches.resize(layers);
    current_seq_len = 0;
    total_seq_len = 0;
}

// Grows a layer's device buffers.  Growth is geometric (at least 256 tokens, at least double, never past
// max_seq_len unless a single request asks for more): one token per step would otherwise re-allocate and copy
// every layer's four buffers on every decode step.
void KVCache::ensure_capacity(LayerCache& c, size_t tokens) {
    if (c.capacity_tokens >= tokens) return;
    size_t want = std::max<size_t>(std::max<size_t>(256, 2 * c.capacity_tokens), tokens);
    if (max_seq_len >= tokens) want = std::min(want, max_seq_len);
    tokens = want;
    const size_t G = (head_dim + CACTUS_KV_QUANT_GROUP_SIZE - 1) / CACTUS_KV_QUANT_GROUP_SIZE;
    const size_t eb = tokens * num_kv_heads * head_dim, sb = tokens * num_kv_heads * G * sizeof(float);
    auto grow = [&](void* old, size_t old_bytes, size_t new_bytes) {
        void* p = dev_alloc(new_bytes);
        if (old) {
            hip_check(hipMemcpy(p, old, old_bytes, hipMemcpyDeviceToDevice), "hipMemcpy D2D (grow)");
            (void)hipFree(old);
        }
        return p;
    };
    const size_t oe = c.capacity_tokens * num_kv_heads * head_dim, os = c.capacity_tokens * num_kv_heads * G * sizeof(float);
    c.keys = static_cast<int8_t*>(grow(c.keys, oe, eb));
    c.values = static_cast<int8_t*>(grow(c.values, oe, eb));
    c.key_scales = static_cast<float*>(grow(c.key_scales, os, sb));
    c.value_scales = static_cast<float*>(grow(c.value_scales, os, sb));
    c.capacity_tokens = tokens;
}

// engine_cache.cpp:25-46: the reference sizes every layer to the window here, zero-fills the
// payload and sets every scale to 1.0.  The logical size this leaves behind (size_tokens) is
// what the whole-sequence branch of update_from_graph compares against to decide whether the
// sink tokens still have to be written (:169), so it is tracked the same way.
void KVCache::set_window_size(size_t window, size_t sink) {
    window_size = window;
    sink_size = sink;
    if (num_kv_heads > 0 && head_dim > 0 && window_size > 0) {
        const size_t G = (head_dim + CACTUS_KV_QUANT_GROUP_SIZE - 1) / CACTUS_KV_QUANT_GROUP_SIZE;
        const size_t eb = window_size * num_kv_heads * head_dim, ns = window_size * num_kv_heads * G;
        const std::vector<float> ones(ns, 1.0f);
        for (auto& c : layer_caches) {
            ensure_capacity(c, window_size);
            hip_check(hipMemset(c.keys, 0, eb), "hipMemset (kv keys)");
            hip_check(hipMemset(c.values, 0, eb), "hipMemset (kv values)");
            hip_check(hipMemcpy(c.key_scales, ones.data(), ns * sizeof(float), hipMemcpyHostToDevice), "kv scales");
            hip_check(hipMemcpy(c.value_scales, ones.data(), ns * sizeof(float), hipMemcpyHostToDevice), "kv scales");
            c.size_tokens = window_size;
        }
    }
}

void KVCache::set_ring(bool on) {
    if (current_seq_len != 0) throw std::runtime_error("KVCache: ring mode can only change while the cache is empty");
    ring = on;
    ring_head = 0;
}

void KVCache::reset() {
    current_seq_len = 0;
    total_seq_len = 0;
    ring_head = 0;
}

const int8_t* KVCache::get_keys_int8(size_t layer) const {
    return (current_seq_len == 0 || layer >= num_layers) ? nullptr : layer_caches[layer].keys;
}
const int8_t* KVCache::get_values_int8(size_t layer) const {
    return (current_seq_len == 0 || layer >= num_layers) ? nullptr : layer_caches[layer].values;
}
const float* KVCache::get_key_scales(size_t layer) const {
    return (current_seq_len == 0 || layer >= num_layers) ? nullptr : layer_caches[layer].key_scales;
}
const float* KVCache::get_value_scales(size_t layer) const {
    return (current_seq_len == 0 || layer >= num_layers) ? nullptr : layer_caches[layer].value_scales;
}

// engine_cache.cpp:101-337.  A layer's K/V node holds either every token so far
// (old + new, :136-233) or only the new ones (:234-329); the first test wins when both match.
// Tokens are quantized on the device straight into their slots.  Once the window is full the
// recent region slides down by seq_len tokens (the reference memmoves, :284-301); an appended
// chunk that does not fit beside the sink tokens makes the reference's unsigned
// tokens_to_shift wrap (:276) and is rejected here instead.
void KVCache::update_from_graph(CactusGraph* gb, const std::vector<size_t>& k_nodes,
                                const std::vector<size_t>& v_nodes, size_t seq_len, size_t layers,
                                size_t kv_heads, size_t dim) {
    const size_t old_seq_len = current_seq_len;
    const size_t new_total_len = old_seq_len + seq_len;
    const size_t ept = kv_heads * dim;
    const size_t G = (dim + CACTUS_KV_QUANT_GROUP_SIZE - 1) / CACTUS_KV_QUANT_GROUP_SIZE;
    const size_t spt = kv_heads * G;
    const bool sliding = window_size > 0 && new_total_len > window_size;
    if (layers > layer_caches.size() || k_nodes.size() < layers || v_nodes.size() < layers)
        throw std::invalid_argument("KVCache: fewer layer caches or K/V nodes than layers");
    if (kv_heads != num_kv_heads || dim != head_dim)
        throw std::invalid_argument("KVCache: kv_heads/head_dim differ from init()");
    if (sliding && sink_size > window_size)
        throw std::runtime_error("KVCache: sink larger than the window");

    // Quantization jobs are collected and launched together at the end, the equally sized ones of all layers in
    // one launch (a decode step: K and V of every layer): they write disjoint rows, and every copy that moves
    // older rows out of their way has been queued on the same stream by then.
    std::vector<std::pair<size_t, cactus_hip::KvQuantJob>> jobs;
    auto quant = [&](const cactus_half_t* src, size_t src_token, LayerCache& c, bool keys, size_t dst_token,
                     size_t tokens) {
        if (tokens == 0) return;
        int8_t* d = (keys ? c.keys : c.values) + dst_token * ept;
        float* sc = (keys ? c.key_scales : c.value_scales) + dst_token * spt;
        jobs.push_back({tokens, cactus_hip::KvQuantJob{src + src_token * ept, d, sc}});
    };
    auto flush_jobs = [&]() {
        std::stable_sort(jobs.begin(), jobs.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        std::vector<cactus_hip::KvQuantJob> same;
        for (size_t i = 0; i < jobs.size();) {
            same.clear();
            size_t j = i;
            for (; j < jobs.size() && jobs[j].first == jobs[i].first; j++) same.push_back(jobs[j].second);
            hip_check(cactus_hip::launch_kv_quant_jobs(same.data(), same.size(), jobs[i].first, kv_heads, dim,
                                                       CACTUS_KV_QUANT_GROUP_SIZE, nullptr), "kv_quant launch");
            i = j;
        }
        jobs.clear();
    };

    // validate before any state changes
    std::vector<int> mode(layers, 0);   // 0 skip, 1 whole sequence, 2 new tokens only
    for (size_t l = 0; l < layers; l++) {
        const BufferDesc& kb = gb->get_output_buffer(k_nodes[l]);
        const BufferDesc& vb = gb->get_output_buffer(v_nodes[l]);
        if (!gb->get_device_output(k_nodes[l]) || !gb->get_device_output(v_nodes[l])) continue;
        if (kb.precision != Precision::FP16 || vb.precision != Precision::FP16)
            throw std::runtime_error("KVCache: K/V nodes must be FP16");
        if (kb.total_size == new_total_len * ept && vb.total_size == new_total_len * ept) mode[l] = 1;
        else if (kb.total_size == seq_len * ept && vb.total_size == seq_len * ept) mode[l] = 2;
        if (mode[l] == 2 && sliding && seq_len + sink_size > window_size)
            throw std::runtime_error("KVCache: appended chunk does not fit the window beside the sink tokens");
    }

    total_seq_len += seq_len;
    bool any = false;
    void* scratch = nullptr;
    size_t next_ring_head = ring_head;
    for (size_t l = 0; l < layers; l++) {
        if (!mode[l]) continue;
        any = true;
        auto* ksrc = static_cast<const cactus_half_t*>(gb->get_device_output(k_nodes[l]));
        auto* vsrc = static_cast<const cactus_half_t*>(gb->get_device_output(v_nodes[l]));
        LayerCache& c = layer_caches[l];
        if (mode[l] == 1) {
            if (!sliding) {
                ensure_capacity(c, new_total_len);
                quant(ksrc, 0, c, true, 0, new_total_len);
                quant(vsrc, 0, c, false, 0, new_total_len);
                c.size_tokens = new_total_len;
            } else {
                ensure_capacity(c, window_size);
                const size_t remaining_window = window_size - sink_size;
                const size_t skip_tokens = new_total_len - window_size;
                const bool first_slide = c.size_tokens != window_size;
                if (first_slide) {
                    quant(ksrc, 0, c, true, 0, sink_size);
                    quant(vsrc, 0, c, false, 0, sink_size);
                }
                quant(ksrc, sink_size + skip_tokens, c, true, sink_size, remaining_window);
                quant(vsrc, sink_size + skip_tokens, c, false, sink_size, remaining_window);
                c.size_tokens = window_size;
                next_ring_head = 0;   // rewritten in temporal order
            }
            continue;
        }
        if (sliding && ring) {
            // Overwrite the oldest recent slots in place.  The write position is the end of the held
            // tokens while the window is still filling, the ring head afterwards; it wraps from the
            // end of the window back to the first slot after the sink.
            ensure_capacity(c, window_size);
            c.size_tokens = window_size;
            const size_t recent = window_size - sink_size;
            size_t pos = old_seq_len < window_size ? old_seq_len : sink_size + ring_head;
            size_t done = 0;
            while (done < seq_len) {
                const size_t n = std::min(seq_len - done, window_size - pos);
                quant(ksrc, done, c, true, pos, n);
                quant(vsrc, done, c, false, pos, n);
                done += n;
                pos += n;
                if (pos == window_size) pos = sink_size;
            }
            next_ring_head = (pos - sink_size) % recent;
            continue;
        }
        size_t dst_token;
        if (!sliding) {
            ensure_capacity(c, new_total_len);
            dst_token = old_seq_len;
            c.size_tokens = new_total_len;
        } else {
            ensure_capacity(c, window_size);
            c.size_tokens = window_size;
            const size_t tokens_to_shift = window_size - sink_size - seq_len;
            if (tokens_to_shift > 0 && old_seq_len > sink_size) {
                const size_t shift_src = old_seq_len - tokens_to_shift;
                if (shift_src > sink_size) {
                    // overlapping move toward lower addresses: stage through scratch
                    const size_t eb = tokens_to_shift * ept, sb = tokens_to_shift * spt * sizeof(float);
                    if (!scratch) scratch = dev_alloc(std::max(eb, sb));
                    auto move = [&](void* base, size_t dst_off, size_t src_off, size_t bytes) {
                        hip_check(hipMemcpyAsync(scratch, static_cast<char*>(base) + src_off, bytes,
                                                 hipMemcpyDeviceToDevice, nullptr), "kv shift");
                        hip_check(hipMemcpyAsync(static_cast<char*>(base) + dst_off, scratch, bytes,
                                                 hipMemcpyDeviceToDevice, nullptr), "kv shift");
                    };
                    move(c.keys, sink_size * ept, shift_src * ept, eb);
                    move(c.values, sink_size * ept, shift_src * ept, eb);
                    move(c.key_scales, sink_size * spt * sizeof(float), shift_src * spt * sizeof(float), sb);
                    move(c.value_scales, sink_size * spt * sizeof(float), shift_src * spt * sizeof(float), sb);
                }
            }
            dst_token = window_size - seq_len;
        }
        quant(ksrc, 0, c, true, dst_token, seq_len);
        quant(vsrc, 0, c, false, dst_token, seq_len);
    }
    flush_jobs();
    if (scratch) {
        (void)hipStreamSynchronize(nullptr);
        (void)hipFree(scratch);
    }
    if (any) {
        current_seq_len = sliding ? window_size : new_total_len;
        ring_head = next_ring_head;
    }
}

}  // namespace cactus_hip

// ------------------------------------------------------------------------- C ABI wrappers
using cactus_hip::CactusGraph;
using cactus_hip::KVCache;
using cactus_hip::Precision;

namespace cactus_hip { void set_last_error(const char* msg); }

template <class R, class F>
static R guarded(R failval, F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        cactus_hip::set_last_error(e.what());
        return failval;
    }
}
#define G_(g) static_cast<CactusGraph*>(g)
#define K_(c) static_cast<KVCache*>(c)

extern "C" {

void* cactus_hip_graph_create(void) { return new CactusGraph(); }
void cactus_hip_graph_destroy(void* g) { delete G_(g); }

long cactus_hip_graph_input(void* g, const size_t* shape, size_t ndim, int precision) {
    return guarded<long>(-1, [&] {
        return (long)G_(g)->input(std::vector<size_t>(shape, shape + ndim), static_cast<Precision>(precision));
    });
}
long cactus_hip_graph_matmul(void* g, size_t a, size_t b, int pretransposed_rhs) {
    return guarded<long>(-1, [&] { return (long)G_(g)->matmul(a, b, pretransposed_rhs != 0); });
}
long cactus_hip_graph_quantize_activations(void* g, size_t input) {
    return guarded<long>(-1, [&] { return (long)G_(g)->quantize_activations(input); });
}
long cactus_hip_graph_attention(void* g, size_t q, size_t k, size_t v, float scale, size_t position_offset,
                                size_t window_size, int is_causal) {
    return guarded<long>(-1, [&] {
        if (!is_causal) return (long)G_(g)->attention(q, k, v, scale, false);
        return (long)G_(g)->attention(q, k, v, scale, position_offset, window_size);
    });
}
long cactus_hip_graph_attention_int8_hybrid(void* g, size_t q, size_t key_new, size_t value_new, float scale,
                                            size_t position_offset, const int8_t* cached_keys,
                                            const int8_t* cached_values, const float* k_scales,
                                            const float* v_scales, size_t cache_len, size_t num_kv_heads,
                                            size_t head_dim) {
    return guarded<long>(-1, [&] {
        return (long)G_(g)->attention_int8_hybrid(q, key_new, value_new, scale, position_offset, cached_keys,
                                                  cached_values, k_scales, v_scales, cache_len, num_kv_heads,
                                                  head_dim);
    });
}
long cactus_hip_graph_binary(void* g, int op, size_t a, size_t b) {
    return guarded<long>(-1, [&] {
        switch (op) {
            case 0: return (long)G_(g)->add(a, b);
            case 1: return (long)G_(g)->subtract(a, b);
            case 2: return (long)G_(g)->multiply(a, b);
            case 3: return (long)G_(g)->divide(a, b);
            case 4: return (long)G_(g)->add_clipped(a, b);
            default: throw std::invalid_argument("binary op must be 0..4");
        }
    });
}
long cactus_hip_graph_scalar_op(void* g, int op, size_t input, float value) {
    return guarded<long>(-1, [&] {
        switch (op) {
            case 0: return (long)G_(g)->scalar_add(input, value);
            case 1: return (long)G_(g)->scalar_subtract(input, value);
            case 2: return (long)G_(g)->scalar_multiply(input, value);
            case 3: return (long)G_(g)->scalar_divide(input, value);
            default: throw std::invalid_argument("scalar op must be 0..3");
        }
    });
}
long cactus_hip_graph_silu(void* g, size_t input) { return guarded<long>(-1, [&] { return (long)G_(g)->silu(input); }); }
long cactus_hip_graph_gelu(void* g, size_t input) { return guarded<long>(-1, [&] { return (long)G_(g)->gelu(input); }); }
long cactus_hip_graph_reshape(void* g, size_t input, const size_t* shape, size_t ndim) {
    return guarded<long>(-1, [&] { return (long)G_(g)->reshape(input, std::vector<size_t>(shape, shape + ndim)); });
}
long cactus_hip_graph_slice(void* g, size_t input, int axis, size_t start, size_t length) {
    return guarded<long>(-1, [&] { return (long)G_(g)->slice(input, axis, start, length); });
}
long cactus_hip_graph_concat(void* g, size_t input1, size_t input2, int axis) {
    return guarded<long>(-1, [&] { return (long)G_(g)->concat(input1, input2, axis); });
}
long cactus_hip_graph_index(void* g, size_t input, size_t index_value, int dim) {
    return guarded<long>(-1, [&] { return (long)G_(g)->index(input, index_value, dim); });
}
long cactus_hip_graph_sample(void* g, size_t logits, float temperature, float top_p, size_t top_k, uint32_t seed) {
    return guarded<long>(-1, [&] { return (long)G_(g)->sample(logits, temperature, top_p, top_k, seed); });
}
int cactus_hip_graph_reset_sample_history(void* g) {
    return guarded<int>(-1, [&] { G_(g)->reset_sample_history(); return 0; });
}
long cactus_hip_graph_embedding(void* g, size_t table, size_t indices) {
    return guarded<long>(-1, [&] { return (long)G_(g)->embedding(table, indices); });
}
long cactus_hip_graph_conv1d_causal(void* g, size_t input, size_t weight, size_t kernel_size, size_t dilation) {
    return guarded<long>(-1, [&] { return (long)G_(g)->conv1d_causal(input, weight, kernel_size, dilation); });
}
long cactus_hip_graph_rms_norm(void* g, size_t input, size_t weight, float epsilon) {
    return guarded<long>(-1, [&] { return (long)G_(g)->rms_norm(input, weight, epsilon); });
}
long cactus_hip_graph_rope(void* g, size_t input, float theta, size_t position_offset) {
    return guarded<long>(-1, [&] { return (long)G_(g)->rope(input, theta, position_offset); });
}
long cactus_hip_graph_mmap_weights(void* g, const char* filename) {
    return guarded<long>(-1, [&] { return (long)G_(g)->mmap_weights(filename ? filename : ""); });
}
int cactus_hip_graph_node_info(void* g, size_t node, size_t* shape4, size_t* ndim, int* precision, size_t* group_size,
                               size_t* num_groups, int* packed_int4) {
    return guarded<int>(-1, [&] {
        const cactus_hip::BufferDesc& b = G_(g)->get_output_buffer(node);
        if (b.shape.size() > 4) throw std::runtime_error("node has more than 4 dimensions");
        for (size_t i = 0; i < b.shape.size(); i++) shape4[i] = b.shape[i];
        *ndim = b.shape.size();
        *precision = (int)b.precision;
        *group_size = b.group_size;
        *num_groups = b.num_groups;
        *packed_int4 = b.is_packed_int4() ? 1 : 0;
        return 0;
    });
}
int cactus_hip_cact_parse_header(const void* bytes, size_t size, uint64_t* out12) {   // 14 values, see the header
    return guarded<int>(-1, [&] {
        const cactus_hip::CactHeader h = cactus_hip::parse_cact_header(bytes, size);
        const uint64_t v[12] = {h.version, h.alignment, h.ndim, h.dims[0], h.dims[1], h.dims[2], h.dims[3], h.precision,
                                h.data_bytes, h.scales_bytes, h.group_size, h.num_groups};
        for (int i = 0; i < 12; i++) out12[i] = v[i];
        out12[12] = h.scales_offset;
        out12[13] = h.data_offset;
        return 0;
    });
}
int cactus_hip_model_config_load(const char* path, cactus_hip_model_config* out) {
    return guarded<int>(-1, [&] {
        if (!path || !out) throw std::runtime_error("model_config_load: null argument");
        const cactus_hip::ModelConfig c = cactus_hip::load_model_config(path);
        memset(out, 0, sizeof *out);
        out->vocab_size = c.vocab_size; out->bos_token_id = c.bos_token_id; out->eos_token_id = c.eos_token_id;
        out->num_layers = c.num_layers; out->hidden_dim = c.hidden_dim;
        out->ffn_intermediate_dim = c.ffn_intermediate_dim; out->attention_heads = c.attention_heads;
        out->attention_kv_heads = c.attention_kv_heads; out->attention_head_dim = c.attention_head_dim;
        out->layer_norm_eps = c.layer_norm_eps; out->rope_theta = c.rope_theta;
        out->num_experts = c.num_experts; out->num_shared_experts = c.num_shared_experts;
        out->num_top_experts = c.num_top_experts; out->moe_every_n_layers = c.moe_every_n_layers;
        out->tie_word_embeddings = c.tie_word_embeddings ? 1 : 0;
        out->model_type = c.model_type; out->model_variant = c.model_variant; out->precision = c.precision;
        out->default_temperature = c.default_temperature; out->default_top_p = c.default_top_p;
        out->default_top_k = c.default_top_k; out->conv_L_cache = c.conv_L_cache;
        out->num_layer_types = (uint32_t)c.layer_types.size();
        std::string joined;
        for (size_t i = 0; i < c.layer_types.size(); i++) joined += (i ? "," : "") + c.layer_types[i];
        if (joined.size() >= sizeof out->layer_types) throw std::runtime_error("model_config_load: layer_types too long");
        memcpy(out->layer_types, joined.c_str(), joined.size() + 1);
        return 0;
    });
}
int cactus_hip_graph_set_input(void* g, size_t node, const void* data, int precision) {
    return guarded<int>(-1, [&] { G_(g)->set_input(node, data, static_cast<Precision>(precision)); return 0; });
}
int cactus_hip_graph_set_external_input(void* g, size_t node, void* data, int precision) {
    return guarded<int>(-1, [&] {
        G_(g)->set_external_input(node, data, static_cast<Precision>(precision));
        return 0;
    });
}
int cactus_hip_graph_set_grouped_scales(void* g, size_t node, size_t group_size, size_t num_groups, void* scales) {
    return guarded<int>(-1, [&] { G_(g)->set_grouped_scales(node, group_size, num_groups, scales); return 0; });
}
int cactus_hip_graph_set_packed_int4(void* g, size_t node, const void* packed, size_t bytes) {
    return guarded<int>(-1, [&] { G_(g)->set_packed_int4(node, packed, bytes); return 0; });
}
int cactus_hip_graph_execute(void* g) {
    return guarded<int>(-1, [&] { G_(g)->execute(); return 0; });
}
int cactus_hip_graph_get_output(void* g, size_t node, void* dst, size_t bytes) {
    return guarded<int>(-1, [&] {
        void* p = G_(g)->get_output(node);
        if (!p) throw std::runtime_error("node has no output");
        if (bytes > G_(g)->get_output_buffer(node).byte_size) throw std::runtime_error("output smaller than request");
        std::memcpy(dst, p, bytes);
        return 0;
    });
}
void* cactus_hip_graph_get_device_output(void* g, size_t node) {
    return guarded<void*>(nullptr, [&] { return G_(g)->get_device_output(node); });
}
int cactus_hip_graph_soft_reset(void* g) {
    return guarded<int>(-1, [&] { G_(g)->soft_reset(); return 0; });
}
int cactus_hip_graph_hard_reset(void* g) {
    return guarded<int>(-1, [&] { G_(g)->hard_reset(); return 0; });
}
size_t cactus_hip_graph_node_count(void* g) { return G_(g)->get_node_count(); }
size_t cactus_hip_graph_pool_allocations(void* g) { return G_(g)->pool_device_allocations(); }
size_t cactus_hip_graph_pool_idle_bytes(void* g) { return G_(g)->pool_idle_bytes(); }
void cactus_hip_graph_set_pool_max_idle_bytes(void* g, size_t bytes) { G_(g)->set_pool_max_idle_bytes(bytes); }
void cactus_hip_graph_set_streaming_twins(void* g, int on) { G_(g)->set_streaming_twins(on != 0); }
void cactus_hip_graph_set_fusion(void* g, int on) { G_(g)->set_fusion(on != 0); }
size_t cactus_hip_graph_fused_launches(void* g) { return G_(g)->fused_launches(); }

void* cactus_hip_kvcache_create(size_t layers, size_t max_seq, size_t kv_heads, size_t head_dim) {
    auto* c = new KVCache();
    c->init(layers, max_seq, kv_heads, head_dim);
    return c;
}
void cactus_hip_kvcache_destroy(void* c) { delete K_(c); }
int cactus_hip_kvcache_set_window(void* c, size_t window, size_t sink) {
    return guarded<int>(-1, [&] { K_(c)->set_window_size(window, sink); return 0; });
}
int cactus_hip_kvcache_set_ring(void* c, int on) {
    return guarded<int>(-1, [&] { K_(c)->set_ring(on != 0); return 0; });
}
size_t cactus_hip_kvcache_ring_head(void* c) { return K_(c)->ring_head; }
void cactus_hip_kvcache_reset(void* c) { K_(c)->reset(); }
int cactus_hip_kvcache_update(void* c, void* g, const size_t* k_nodes, const size_t* v_nodes, size_t seq_len) {
    return guarded<int>(-1, [&] {
        KVCache* kc = K_(c);
        std::vector<size_t> k(k_nodes, k_nodes + kc->num_layers), v(v_nodes, v_nodes + kc->num_layers);
        kc->update_from_graph(G_(g), k, v, seq_len, kc->num_layers, kc->num_kv_heads, kc->head_dim);
        return 0;
    });
}
size_t cactus_hip_kvcache_len(void* c) { return K_(c)->get_effective_seq_len(); }
size_t cactus_hip_kvcache_total_len(void* c) { return K_(c)->get_total_seq_len(); }
const int8_t* cactus_hip_kvcache_keys(void* c, size_t layer) { return K_(c)->get_keys_int8(layer); }
const int8_t* cactus_hip_kvcache_values(void* c, size_t layer) { return K_(c)->get_values_int8(layer); }
const float* cactus_hip_kvcache_key_scales(void* c, size_t layer) { return K_(c)->get_key_scales(layer); }
const float* cactus_hip_kvcache_value_scales(void* c, size_t layer) { return K_(c)->get_value_scales(layer); }

}  // extern "C"
