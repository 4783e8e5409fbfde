// A C++ host decoding a BATCH of independent sequences through the library's step driver
// (include/cactus_hip_steps.h: cactus_hip_decode_step_dev), the path a compiled engine takes for many sequences per
// GPU.  The reference decodes one sequence per graph (cactus/models/model_qwen.cpp:74-76 reshape to batch 1); here
// every step runs the same op sequence (model_qwen.cpp:51-172, engine_model.cpp:221-253) for all sequences at once,
// with token ids, positions and the INT8 KV caches resident on the device.
//
//   qwen_batched_decode <model_dir> <batch> <steps> <first token> [<first token> ...]
//
// The model folder is in the reference's layout (config.txt + CACT tensor files as model_qwen.cpp:17-44 names
// them); sequence b starts from first token number b % count at position 0.  Weights get their tile-major twins when
// the batch is small enough for the streaming kernels.  Prints one JSON line: every sequence's greedy tokens and the
// milliseconds per step (all sequences) after two warm-up steps.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "cactus_hip.h"
#include "cactus_hip_graph.h"
#include "cactus_hip_steps.h"

namespace {

[[noreturn]] void die(const std::string& what) {
    std::fprintf(stderr, "qwen_batched_decode: %s: %s\n", what.c_str(), cactus_hip_last_error());
    std::exit(1);
}
void ok(int rc, const char* what) {
    if (rc != 0) die(what);
}
void* dev(size_t bytes) {
    void* p = cactus_hip_malloc(bytes);
    if (!p) die("cactus_hip_malloc");
    return p;
}
void* upload(const void* host, size_t bytes) {
    void* p = dev(bytes);
    ok(cactus_hip_memcpy_h2d(p, host, bytes, nullptr), "memcpy_h2d");
    return p;
}

std::vector<char> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) die("cannot open " + path);
    std::vector<char> buf((size_t)f.tellg());
    f.seekg(0);
    f.read(buf.data(), (std::streamsize)buf.size());
    return buf;
}

// a quantized matrix file -> device payload + scales (the header is parsed by the library, graph_io.cpp:497-566)
cactus_hip_qmatrix load_matrix(const std::string& path, bool pack) {
    const std::vector<char> buf = slurp(path);
    uint64_t h[14];
    ok(cactus_hip_cact_parse_header(buf.data(), buf.size(), h), path.c_str());
    const uint64_t ndim = h[2], precision = h[7], data_bytes = h[8], scales_bytes = h[9], group = h[10];
    if (ndim != 2 || group != 128 || (precision != 0 && precision != 3)) die(path + ": not a group-128 INT8 / INT4 matrix");
    cactus_hip_qmatrix m{};
    m.N = (size_t)h[3];
    m.K = (size_t)h[4];
    m.int4 = precision == 3;
    m.B = upload(buf.data() + h[13], (size_t)data_bytes);
    m.scales = static_cast<const cactus_half_t*>(upload(buf.data() + h[12], (size_t)scales_bytes));
    if (pack) ok(cactus_hip_weights_pack_dev(m.B, m.N, m.K, m.int4, nullptr), "weights_pack");
    return m;
}
const cactus_half_t* load_vector(const std::string& path) {
    const std::vector<char> buf = slurp(path);
    uint64_t h[14];
    ok(cactus_hip_cact_parse_header(buf.data(), buf.size(), h), path.c_str());
    if (h[7] != 1) die(path + ": not an FP16 vector");
    return static_cast<const cactus_half_t*>(upload(buf.data() + h[13], (size_t)h[8]));
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: qwen_batched_decode <model_dir> <batch> <steps> <first token> [...]\n");
        return 2;
    }
    const std::string dir = argv[1];
    const size_t batch = (size_t)std::atol(argv[2]), steps = (size_t)std::atol(argv[3]);
    std::vector<int32_t> first;
    for (int i = 4; i < argc; i++) first.push_back((int32_t)std::atol(argv[i]));
    if (batch == 0 || steps == 0) return 2;
    ok(cactus_hip_init(0), "init");

    cactus_hip_model_config cfg{};
    ok(cactus_hip_model_config_load((dir + "/config.txt").c_str(), &cfg), "config.txt");
    const bool pack = batch <= 16;      // the streaming kernels take these batches
    cactus_hip_decoder_weights w{};
    w.hidden = cfg.hidden_dim; w.q_heads = cfg.attention_heads; w.kv_heads = cfg.attention_kv_heads;
    w.head_dim = cfg.attention_head_dim; w.ffn = cfg.ffn_intermediate_dim; w.vocab = cfg.vocab_size; w.layers = cfg.num_layers;
    w.eps = cfg.layer_norm_eps; w.theta = cfg.rope_theta;
    std::vector<cactus_hip_block_weights> blocks(cfg.num_layers);
    for (uint32_t i = 0; i < cfg.num_layers; i++) {
        const std::string p = dir + "/layer_" + std::to_string(i) + "_";
        cactus_hip_block_weights& L = blocks[i];
        L.kind = 0;
        L.q = load_matrix(p + "attn_q.weights", pack);
        L.k = load_matrix(p + "attn_k.weights", pack);
        L.v = load_matrix(p + "attn_v.weights", pack);
        L.o = load_matrix(p + "attn_output.weights", pack);
        L.gate = load_matrix(p + "ffn_gate.weights", pack);
        L.up = load_matrix(p + "ffn_up.weights", pack);
        L.down = load_matrix(p + "ffn_down.weights", pack);
        L.in_norm = load_vector(p + "input_norm.weights");
        L.post_norm = load_vector(p + "post_attn_norm.weights");
        L.q_norm = load_vector(p + "attn_q_norm.weights");
        L.k_norm = load_vector(p + "attn_k_norm.weights");
    }
    w.blocks = blocks.data();
    w.embedding = load_matrix(dir + "/token_embeddings.weights", pack);
    if (!cfg.tie_word_embeddings) w.lm_head = load_matrix(dir + "/output_weight.weights", pack);
    w.final_norm = load_vector(dir + "/output_norm.weights");

    const size_t slots = steps + 4, G = (w.head_dim + 127) / 128;
    cactus_hip_decode_state st{};
    st.batch = batch; st.slots = slots; st.window = 0; st.sink = 0; st.max_position = slots;
    const size_t cache_bytes = w.layers * batch * slots * w.kv_heads * w.head_dim;
    const size_t scale_bytes = w.layers * batch * slots * w.kv_heads * G * sizeof(float);
    st.k_cache = static_cast<int8_t*>(dev(cache_bytes));
    st.v_cache = static_cast<int8_t*>(dev(cache_bytes));
    st.k_scales = static_cast<float*>(dev(scale_bytes));
    st.v_scales = static_cast<float*>(dev(scale_bytes));
    std::vector<int32_t> zeros(batch, 0), toks(batch);
    for (size_t b = 0; b < batch; b++) toks[b] = first[b % first.size()];
    st.positions = static_cast<int32_t*>(upload(zeros.data(), batch * 4));
    st.cache_lens = static_cast<int32_t*>(upload(zeros.data(), batch * 4));
    st.tokens = static_cast<int32_t*>(upload(toks.data(), batch * 4));
    ok(cactus_hip_rope_reserve(w.head_dim, w.theta, slots + 1), "rope_reserve");
    void* workspace = dev(cactus_hip_decode_step_workspace_bytes(&w, batch));
    cactus_half_t* logits = static_cast<cactus_half_t*>(dev(batch * w.vocab * sizeof(cactus_half_t)));

    std::vector<std::vector<int32_t>> out(batch);
    double timed_ms = 0.0;
    size_t timed = 0;
    for (size_t s = 0; s < steps; s++) {
        const auto t0 = std::chrono::steady_clock::now();
        ok(cactus_hip_decode_step_dev(&w, &st, workspace, logits, 1, nullptr), "decode_step");
        ok(cactus_hip_memcpy_d2h(toks.data(), st.tokens, batch * 4, nullptr), "memcpy_d2h");   // synchronizes
        const auto t1 = std::chrono::steady_clock::now();
        if (s >= 2) {
            timed_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
            timed++;
        }
        for (size_t b = 0; b < batch; b++) out[b].push_back(toks[b]);
    }
    std::printf("{\"batch\": %zu, \"steps\": %zu, \"ms_per_step\": %.4f, \"tokens\": [", batch, steps, timed ? timed_ms / (double)timed : 0.0);
    for (size_t b = 0; b < batch; b++) {
        std::printf("%s[", b ? ", " : "");
        for (size_t s = 0; s < out[b].size(); s++) std::printf("%s%d", s ? ", " : "", out[b][s]);
        std::printf("]");
    }
    std::printf("]}\n");
    std::fflush(stdout);
#ifdef EXIT_WITHOUT_TEARDOWN      // sanitizer builds: the ROCm ASan runtime can trip over itself while the HIP runtime unloads
    std::_Exit(0);
#endif
    return 0;
}
