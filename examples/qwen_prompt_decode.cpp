// A C++ host that takes ONE prompt through the library's step drivers (include/cactus_hip_steps.h): the whole prompt
// with cactus_hip_prefill_dev (Model::prefill's chunk loop, engine_model.cpp:170-219, with several chunks in flight), then
// greedy decode steps with cactus_hip_decode_step_dev on the SAME cache buffers (Model::decode, engine_model.cpp:221-253):
// a prefill state is laid out as a decode state of batch 1 whose slot count is the prefill's capacity.
//
//   qwen_prompt_decode <model_dir> <steps> <chunk> <chunks in flight> <prompt token> [<prompt token> ...]
//
// The model folder is in the reference's layout (config.txt + CACT tensor files as model_qwen.cpp:17-44 names them).
// Prints one JSON line: the prompt's greedy continuation (steps + 1 tokens: the prefill's choice, then one per decode
// step) and the milliseconds the prefill took.
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
    std::fprintf(stderr, "qwen_prompt_decode: %s: %s\n", what.c_str(), cactus_hip_last_error());
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
    if (argc < 6) {
        std::fprintf(stderr, "usage: qwen_prompt_decode <model_dir> <steps> <chunk> <chunks in flight> <prompt token> [...]\n");
        return 2;
    }
    const std::string dir = argv[1];
    const size_t steps = (size_t)std::atol(argv[2]), chunk = (size_t)std::atol(argv[3]), lanes = (size_t)std::atol(argv[4]);
    std::vector<int32_t> prompt;
    for (int i = 5; i < argc; i++) prompt.push_back((int32_t)std::atol(argv[i]));
    if (chunk == 0 || lanes == 0 || prompt.empty()) return 2;
    ok(cactus_hip_init(0), "init");

    cactus_hip_model_config cfg{};
    ok(cactus_hip_model_config_load((dir + "/config.txt").c_str(), &cfg), "config.txt");
    const bool pack = true;             // the decode steps of one sequence run on the streaming kernels
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

    // one set of buffers for both phases: capacity = prompt + the tokens to come
    const size_t n = prompt.size(), capacity = n + steps + 1, G = (w.head_dim + 127) / 128;
    const size_t cache_bytes = w.layers * capacity * w.kv_heads * w.head_dim;
    const size_t scale_bytes = w.layers * capacity * w.kv_heads * G * sizeof(float);
    cactus_hip_prefill_state ps{};
    ps.capacity = capacity;
    ps.k_cache = static_cast<int8_t*>(dev(cache_bytes));
    ps.v_cache = static_cast<int8_t*>(dev(cache_bytes));
    ps.k_scales = static_cast<float*>(dev(scale_bytes));
    ps.v_scales = static_cast<float*>(dev(scale_bytes));
    ps.conv_state = nullptr;
    ok(cactus_hip_rope_reserve(w.head_dim, w.theta, capacity + 1), "rope_reserve");
    const size_t nchunks = (n + chunk - 1) / chunk;
    std::vector<int32_t> starts(nchunks);
    for (size_t c = 0; c < nchunks; c++) starts[c] = (int32_t)(c * chunk);
    int32_t* tokens_dev = static_cast<int32_t*>(upload(prompt.data(), n * 4));
    int32_t* starts_dev = static_cast<int32_t*>(upload(starts.data(), nchunks * 4));
    void* pf_ws = dev(cactus_hip_prefill_workspace_bytes(&w, chunk, lanes));
    cactus_half_t* logits = static_cast<cactus_half_t*>(dev(w.vocab * sizeof(cactus_half_t)));
    int32_t* tok_dev = static_cast<int32_t*>(dev(4));

    int32_t tok = 0;
    const auto t0 = std::chrono::steady_clock::now();
    ok(cactus_hip_prefill_dev(&w, &ps, tokens_dev, n, chunk, starts_dev, pf_ws, lanes, logits, tok_dev, nullptr), "prefill");
    ok(cactus_hip_memcpy_d2h(&tok, tok_dev, 4, nullptr), "memcpy_d2h");        // synchronizes
    const double prefill_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    // the decode state on the same caches
    cactus_hip_decode_state st{};
    st.batch = 1; st.slots = capacity; st.window = 0; st.sink = 0; st.max_position = capacity;
    st.k_cache = ps.k_cache; st.v_cache = ps.v_cache; st.k_scales = ps.k_scales; st.v_scales = ps.v_scales;
    const int32_t pos = (int32_t)n;
    st.positions = static_cast<int32_t*>(upload(&pos, 4));
    st.cache_lens = static_cast<int32_t*>(upload(&pos, 4));
    st.tokens = tok_dev;                       // in: the prefill's choice; out: each step's
    void* workspace = dev(cactus_hip_decode_step_workspace_bytes(&w, 1));
    std::vector<int32_t> out{tok};
    for (size_t s = 0; s < steps; s++) {
        ok(cactus_hip_decode_step_dev(&w, &st, workspace, logits, 1, nullptr), "decode_step");
        ok(cactus_hip_memcpy_d2h(&tok, tok_dev, 4, nullptr), "memcpy_d2h");
        out.push_back(tok);
    }
    std::printf("{\"prompt_tokens\": %zu, \"chunk\": %zu, \"chunks_in_flight\": %zu, \"prefill_ms\": %.3f, \"tokens\": [", n, chunk,
                lanes, prefill_ms);
    for (size_t s = 0; s < out.size(); s++) std::printf("%s%d", s ? ", " : "", out[s]);
    std::printf("]}\n");
    std::fflush(stdout);
#ifdef EXIT_WITHOUT_TEARDOWN      // sanitizer builds: the ROCm ASan runtime can trip over itself while the HIP runtime unloads
    std::_Exit(0);
#endif
    return 0;
}
