// A C++ host driving the library the way the reference's engine drives its graph layer: the Qwen forward pass
// (cactus/models/model_qwen.cpp:17-172) rebuilt node by node on a graph for every token, the INT8 KV cache fed from
// the k / v nodes (cactus/engine/engine_cache.cpp:186-266), greedy sampling as a graph node, soft reset between
// steps (Model::prefill / Model::decode, cactus/engine/engine_model.cpp:170-253) - all through the C ABI of
// include/cactus_hip_graph.h.  The model folder is in the reference's layout (config.txt + CACT tensor files named
// as model_qwen.cpp names them).
//
//   qwen_graph_decode <model_dir> <decode_steps> <fusion 0|1> <prompt token> [<prompt token> ...]
//
// Prints one JSON line: the greedy tokens, the wall-clock milliseconds of the prompt step (one chunk, first run of
// every kernel included) and per decode token, split into building
// the graph, execute() + token read-back, cache update and soft reset (the first two decode steps are warm-up and
// are not timed).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cactus_hip.h"
#include "cactus_hip_graph.h"

namespace {

[[noreturn]] void die(const char* what) {
    std::fprintf(stderr, "qwen_graph_decode: %s: %s\n", what, cactus_hip_last_error());
    std::exit(1);
}
size_t ok(long id, const char* what) {
    if (id < 0) die(what);
    return (size_t)id;
}
void ok(int rc, const char* what) {
    if (rc != 0) die(what);
}

struct Layer {
    size_t q, k, v, o, in_norm, q_norm, k_norm, gate, up, down, post_norm;
};

struct Model {
    void* g = nullptr;
    void* cache = nullptr;
    cactus_hip_model_config cfg{};
    size_t emb = 0, out_norm = 0, out_w = 0;
    std::vector<Layer> layers;
    size_t pos = 0;

    void load(const std::string& dir) {
        ok(cactus_hip_model_config_load((dir + "/config.txt").c_str(), &cfg), "config.txt");
        g = cactus_hip_graph_create();
        if (!g) die("graph_create");
        auto w = [&](const std::string& name) { return ok(cactus_hip_graph_mmap_weights(g, (dir + "/" + name).c_str()), name.c_str()); };
        emb = w("token_embeddings.weights");
        out_norm = w("output_norm.weights");
        out_w = cfg.tie_word_embeddings ? emb : w("output_weight.weights");
        layers.resize(cfg.num_layers);
        for (uint32_t i = 0; i < cfg.num_layers; i++) {
            const std::string p = "layer_" + std::to_string(i) + "_";
            Layer& L = layers[i];
            L.q = w(p + "attn_q.weights");
            L.k = w(p + "attn_k.weights");
            L.v = w(p + "attn_v.weights");
            L.o = w(p + "attn_output.weights");
            L.in_norm = w(p + "input_norm.weights");
            L.q_norm = w(p + "attn_q_norm.weights");
            L.k_norm = w(p + "attn_k_norm.weights");
            L.gate = w(p + "ffn_gate.weights");
            L.up = w(p + "ffn_up.weights");
            L.down = w(p + "ffn_down.weights");
            L.post_norm = w(p + "post_attn_norm.weights");
        }
        cache = cactus_hip_kvcache_create(cfg.num_layers, 4096, cfg.attention_kv_heads, cfg.attention_head_dim);
        if (!cache) die("kvcache_create");
    }

    size_t reshape(size_t n, std::initializer_list<size_t> shape) {
        std::vector<size_t> s(shape);
        return ok(cactus_hip_graph_reshape(g, n, s.data(), s.size()), "reshape");
    }

    // the nodes of one forward pass over `tokens` at the current position (model_qwen.cpp:51-172); returns the
    // sampled-token node
    size_t build(const std::vector<uint32_t>& tokens, std::vector<size_t>& ks, std::vector<size_t>& vs) {
        const size_t S = tokens.size(), Hq = cfg.attention_heads, Hkv = cfg.attention_kv_heads, D = cfg.attention_head_dim;
        const float eps = cfg.layer_norm_eps, theta = cfg.rope_theta, scale = 1.0f / std::sqrt((float)D);
        std::vector<float> ids(tokens.begin(), tokens.end());
        const size_t idx = ok(cactus_hip_graph_input(g, &S, 1, 2), "input");
        ok(cactus_hip_graph_set_input(g, idx, ids.data(), 2), "set_input");
        size_t h = ok(cactus_hip_graph_embedding(g, emb, idx), "embedding");
        const size_t cached = cactus_hip_kvcache_len(cache);
        ks.clear();
        vs.clear();
        for (uint32_t li = 0; li < cfg.num_layers; li++) {
            const Layer& L = layers[li];
            const size_t n1 = ok(cactus_hip_graph_rms_norm(g, h, L.in_norm, eps), "rms_norm");
            size_t q = ok(cactus_hip_graph_matmul(g, n1, L.q, 1), "matmul q");
            size_t k = ok(cactus_hip_graph_matmul(g, n1, L.k, 1), "matmul k");
            size_t v = ok(cactus_hip_graph_matmul(g, n1, L.v, 1), "matmul v");
            q = reshape(ok(cactus_hip_graph_rms_norm(g, reshape(q, {S * Hq, D}), L.q_norm, eps), "q_norm"), {1, S, Hq, D});
            k = reshape(ok(cactus_hip_graph_rms_norm(g, reshape(k, {S * Hkv, D}), L.k_norm, eps), "k_norm"), {1, S, Hkv, D});
            v = reshape(v, {1, S, Hkv, D});
            q = ok(cactus_hip_graph_rope(g, q, theta, pos), "rope q");
            k = ok(cactus_hip_graph_rope(g, k, theta, pos), "rope k");
            ks.push_back(k);
            vs.push_back(v);
            size_t attn;
            if (cached == 0)
                attn = ok(cactus_hip_graph_attention(g, q, k, v, scale, pos, 0, 1), "attention");
            else
                attn = ok(cactus_hip_graph_attention_int8_hybrid(g, q, k, v, scale, pos, cactus_hip_kvcache_keys(cache, li),
                                                                 cactus_hip_kvcache_values(cache, li),
                                                                 cactus_hip_kvcache_key_scales(cache, li),
                                                                 cactus_hip_kvcache_value_scales(cache, li), cached, Hkv, D),
                          "attention_int8_hybrid");
            h = ok(cactus_hip_graph_binary(g, 0, h, ok(cactus_hip_graph_matmul(g, reshape(attn, {S, Hq * D}), L.o, 1), "matmul o")), "add");
            const size_t n2 = ok(cactus_hip_graph_rms_norm(g, h, L.post_norm, eps), "post_norm");
            const size_t gate = ok(cactus_hip_graph_silu(g, ok(cactus_hip_graph_matmul(g, n2, L.gate, 1), "matmul gate")), "silu");
            const size_t act = ok(cactus_hip_graph_binary(g, 2, gate, ok(cactus_hip_graph_matmul(g, n2, L.up, 1), "matmul up")), "multiply");
            h = ok(cactus_hip_graph_binary(g, 0, h, ok(cactus_hip_graph_matmul(g, act, L.down, 1), "matmul down")), "add");
        }
        const size_t last = reshape(ok(cactus_hip_graph_index(g, ok(cactus_hip_graph_rms_norm(g, h, out_norm, eps), "out_norm"), S - 1, 0), "index"),
                                    {1, (size_t)cfg.hidden_dim});          // engine_model.cpp:241-246
        const size_t logits = ok(cactus_hip_graph_matmul(g, last, out_w, 1), "lm_head");
        return ok(cactus_hip_graph_sample(g, logits, 0.0f, 0.0f, 0, 1), "sample");   // temperature 0: greedy (kernel_nn.cpp:451)
    }
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s <model_dir> <decode_steps> <fusion 0|1> <prompt token> [...]\n", argv[0]);
        return 2;
    }
    const int steps = std::atoi(argv[2]);
    const int fusion = std::atoi(argv[3]);
    std::vector<uint32_t> prompt;
    for (int i = 4; i < argc; i++) prompt.push_back((uint32_t)std::strtoul(argv[i], nullptr, 10));
    ok(cactus_hip_init(0), "init");
    Model m;
    m.load(argv[1]);
    cactus_hip_graph_set_fusion(m.g, fusion);

    std::vector<size_t> ks, vs;
    std::vector<uint32_t> out;
    double t_build = 0, t_exec = 0, t_update = 0, t_reset = 0, t_all = 0, prefill_ms = 0, prefill_parts[4] = {0, 0, 0, 0};
    int timed = 0;
    std::vector<uint32_t> feed = prompt;
    // CACTUS_PREFILL_CHUNK=n: the prompt in chunks of n tokens, a graph each, as Model::prefill walks it
    // (engine_model.cpp:170-219; get_prefill_chunk_size() is 256 there); default: the prompt as one graph
    const char* chunk_env = std::getenv("CACTUS_PREFILL_CHUNK");
    const size_t chunk = chunk_env ? (size_t)std::atol(chunk_env) : 0;
    double chunk_parts[3] = {0, 0, 0};          // build, execute + cache update, soft reset of the timed run's chunks
    auto run_chunks = [&](bool timed_run) {     // all chunks but the last; returns the milliseconds they took
        const auto c0 = std::chrono::steady_clock::now();
        size_t done = 0;
        while (chunk && prompt.size() - done > chunk) {
            std::vector<uint32_t> part(prompt.begin() + (long)done, prompt.begin() + (long)(done + chunk));
            const auto b0 = std::chrono::steady_clock::now();
            (void)m.build(part, ks, vs);
            const auto b1 = std::chrono::steady_clock::now();
            ok(cactus_hip_graph_execute(m.g), "execute");
            ok(cactus_hip_kvcache_update(m.cache, m.g, ks.data(), vs.data(), part.size()), "kvcache_update");
            const auto b2 = std::chrono::steady_clock::now();
            ok(cactus_hip_graph_soft_reset(m.g), "soft_reset");
            const auto b3 = std::chrono::steady_clock::now();
            if (timed_run) {
                chunk_parts[0] += std::chrono::duration<double, std::milli>(b1 - b0).count();
                chunk_parts[1] += std::chrono::duration<double, std::milli>(b2 - b1).count();
                chunk_parts[2] += std::chrono::duration<double, std::milli>(b3 - b2).count();
            }
            m.pos += part.size();
            done += chunk;
        }
        feed.assign(prompt.begin() + (long)done, prompt.end());
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
    };
    double chunks_ms = 0;
    // step -1: the prompt once untimed (first use of every kernel, buffers, RoPE rows), then the cache is emptied
    // and the run proper starts; step 0: the prompt
    for (int step = prompt.size() > 1 ? -1 : 0; step <= steps; step++) {
        const bool timing = step > 2;
        if (step <= 0 && prompt.size() > 1) chunks_ms = run_chunks(step == 0);
        const auto t0 = std::chrono::steady_clock::now();
        const size_t tok_node = m.build(feed, ks, vs);
        const auto t1 = std::chrono::steady_clock::now();
        ok(cactus_hip_graph_execute(m.g), "execute");
        uint32_t tok = 0;
        ok(cactus_hip_graph_get_output(m.g, tok_node, &tok, sizeof tok), "get_output");
        const auto t2 = std::chrono::steady_clock::now();
        ok(cactus_hip_kvcache_update(m.cache, m.g, ks.data(), vs.data(), feed.size()), "kvcache_update");
        const auto t3 = std::chrono::steady_clock::now();
        ok(cactus_hip_graph_soft_reset(m.g), "soft_reset");
        const auto t4 = std::chrono::steady_clock::now();
        if (step == 0) {
            prefill_ms = chunks_ms + std::chrono::duration<double, std::milli>(t4 - t0).count();
            prefill_parts[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
            prefill_parts[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
            prefill_parts[2] = std::chrono::duration<double, std::milli>(t3 - t2).count();
            prefill_parts[3] = std::chrono::duration<double, std::milli>(t4 - t3).count();
        }
        if (timing) {
            t_build += std::chrono::duration<double, std::milli>(t1 - t0).count();
            t_exec += std::chrono::duration<double, std::milli>(t2 - t1).count();
            t_update += std::chrono::duration<double, std::milli>(t3 - t2).count();
            t_reset += std::chrono::duration<double, std::milli>(t4 - t3).count();
            t_all += std::chrono::duration<double, std::milli>(t4 - t0).count();
            timed++;
        }
        if (step < 0) {
            cactus_hip_kvcache_reset(m.cache);
            m.pos = 0;
            continue;
        }
        m.pos += feed.size();
        out.push_back(tok);
        feed.assign(1, tok);
    }
    std::printf("{\"tokens\": [");
    for (size_t i = 0; i < out.size(); i++) std::printf("%s%u", i ? ", " : "", out[i]);
    const double n = timed ? (double)timed : 1.0;
    std::printf("], \"prompt_tokens\": %zu, \"prefill_ms\": %.3f, \"timed_steps\": %d, \"ms_per_token\": %.4f, \"build_ms\": %.4f, \"execute_ms\": %.4f, \"cache_update_ms\": %.4f, "
                "\"soft_reset_ms\": %.4f, \"fused_launches\": %zu, \"pool_allocations\": %zu, "
                "\"prefill_build_execute_update_reset_ms\": [%.3f, %.3f, %.3f, %.3f], \"earlier_chunks_build_execute_reset_ms\": [%.3f, %.3f, %.3f]}\n",
                prompt.size(), prefill_ms, timed, t_all / n, t_build / n, t_exec / n, t_update / n, t_reset / n, cactus_hip_graph_fused_launches(m.g),
                cactus_hip_graph_pool_allocations(m.g), prefill_parts[0], prefill_parts[1], prefill_parts[2], prefill_parts[3], chunk_parts[0],
                chunk_parts[1], chunk_parts[2]);
    cactus_hip_kvcache_destroy(m.cache);
    cactus_hip_graph_destroy(m.g);
    std::fflush(stdout);
#ifdef EXIT_WITHOUT_TEARDOWN      // sanitizer builds: the ROCm ASan runtime can trip over itself while the HIP runtime unloads
    std::_Exit(0);
#endif
    return 0;
}
