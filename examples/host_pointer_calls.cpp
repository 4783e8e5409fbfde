// The host-pointer drop-in family from a plain C++ host: every entry that takes the reference's argument list
// (cactus/kernel/kernel.h:20-24, 41-104) is called on host buffers, the weight matmuls several times over the same
// weights (the library keeps them resident between calls), once more after the weights were rewritten in place and
// once after cactus_hip_release_weights.  Prints one JSON line of checksums; meant for runs under host sanitizers
// (scripts/build_host_sanitized.sh) and as a smoke test that the family links from C++.
//   host_pointer_calls [rounds]
#include <cactus_hip.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

uint32_t g_state = 12345u;
uint32_t next_u32() {
    g_state = g_state * 1664525u + 1013904223u;
    return g_state >> 8;
}
// a binary16 in about [-1, 1): exponent 0x2C..0x3B, random mantissa and sign
cactus_half_t rand_half() {
    const uint32_t r = next_u32();
    return (cactus_half_t)(((r & 1u) << 15) | ((0x2Cu + (r >> 1) % 15u) << 10) | ((r >> 8) & 0x3FFu));
}
int8_t rand_i8() { return (int8_t)((int)(next_u32() % 255u) - 127); }

uint64_t checksum(const void* p, size_t bytes) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

bool ok(const char* what) {
    const char* e = cactus_hip_last_error();
    if (e && *e) {
        std::fprintf(stderr, "%s: %s\n", what, e);
        return false;
    }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3;
    if (cactus_hip_init(0) != CACTUS_HIP_OK) {
        std::fprintf(stderr, "init: %s\n", cactus_hip_last_error());
        return 1;
    }
    uint64_t sum = 0;
    const size_t gs = 128;
    // weight matmuls: one row, a few rows, a tile-kernel batch; INT8 and packed INT4; ragged N
    const size_t shapes[][3] = {{1, 1024, 1000}, {16, 512, 272}, {300, 256, 528}};
    for (const auto& sh : shapes) {
        const size_t M = sh[0], K = sh[1], N = sh[2], G = K / gs;
        std::vector<int8_t> A(M * K), B8(N * K);
        std::vector<uint8_t> B4(N * K / 2);
        std::vector<float> As(M);
        std::vector<cactus_half_t> Bs(N * G), C(M * N), X(M * K), W16(N * K);
        for (auto& v : A) v = rand_i8();
        for (auto& v : B8) v = rand_i8();
        for (auto& v : B4) v = (uint8_t)next_u32();
        for (auto& v : As) v = 0.001f + (float)(next_u32() % 100u) * 1e-4f;
        for (auto& v : Bs) v = (cactus_half_t)(0x2000u + (next_u32() & 0x3FFu));
        for (auto& v : X) v = rand_half();
        for (auto& v : W16) v = rand_half();
        for (int r = 0; r < rounds; r++) {
            cactus_hip_matmul_int8(A.data(), As.data(), B8.data(), Bs.data(), C.data(), M, K, N, gs);
            if (!ok("matmul_int8")) return 1;
            sum ^= checksum(C.data(), C.size() * 2);
            cactus_hip_matmul_int4(A.data(), As.data(), B4.data(), Bs.data(), C.data(), M, K, N, gs);
            if (!ok("matmul_int4")) return 1;
            sum ^= checksum(C.data(), C.size() * 2);
            if (r == 0) {       // rewritten in place: the resident copy must be refreshed
                for (size_t i = 0; i < B8.size(); i += 97) B8[i] = rand_i8();
                for (size_t i = 0; i < B4.size(); i += 89) B4[i] = (uint8_t)next_u32();
            }
            if (r == 1) {
                cactus_hip_release_weights(B8.data());
                cactus_hip_release_weights(B4.data());
            }
        }
        cactus_hip_matmul_f16(X.data(), W16.data(), C.data(), M, K, N);
        if (!ok("matmul_f16")) return 1;
        sum ^= checksum(C.data(), C.size() * 2);
        cactus_hip_release_weights(B8.data());
        cactus_hip_release_weights(B4.data());
        cactus_hip_release_weights(W16.data());
    }
    // quantizers
    {
        const size_t n = 5000 + 3;
        std::vector<cactus_half_t> src(n);
        for (auto& v : src) v = rand_half();
        std::vector<int8_t> q(n);
        const float mx = cactus_hip_fp16_max_abs(src.data(), n);
        if (!ok("fp16_max_abs")) return 1;
        cactus_hip_fp16_to_int8(src.data(), q.data(), n, mx > 0 ? mx / 127.0f : 1.0f);
        if (!ok("fp16_to_int8")) return 1;
        sum ^= checksum(q.data(), n);
        std::vector<uint8_t> packed((n + 1) / 2);
        for (auto& v : packed) v = (uint8_t)next_u32();
        std::vector<int8_t> un(n);
        cactus_hip_unpack_int4_to_int8(packed.data(), un.data(), n);
        if (!ok("unpack_int4_to_int8")) return 1;
        sum ^= checksum(un.data(), n);
    }
    // attention: FP16 prompt, then one decode token over the quantized cache
    {
        const size_t S = 40, Hq = 8, Hkv = 4, D = 128, cache = 37;
        std::vector<cactus_half_t> Q(S * Hq * D), Kf(S * Hkv * D), Vf(S * Hkv * D), O(S * Hq * D);
        for (auto& v : Q) v = rand_half();
        for (auto& v : Kf) v = rand_half();
        for (auto& v : Vf) v = rand_half();
        cactus_hip_attention_f16(Q.data(), Kf.data(), Vf.data(), O.data(), 1, S, S, Hq, Hkv, D, 0.0f, nullptr, 0, 0, 1);
        if (!ok("attention_f16")) return 1;
        sum ^= checksum(O.data(), O.size() * 2);
        std::vector<int8_t> Kq(cache * Hkv * D), Vq(cache * Hkv * D);
        std::vector<float> Ks(cache * Hkv), Vs(cache * Hkv);
        cactus_hip_quantize_kv_fp16_to_int8(Kf.data(), Kq.data(), Ks.data(), cache, Hkv, D, gs);
        cactus_hip_quantize_kv_fp16_to_int8(Vf.data(), Vq.data(), Vs.data(), cache, Hkv, D, gs);
        if (!ok("quantize_kv")) return 1;
        std::vector<cactus_half_t> q1(Hq * D), k1(Hkv * D), v1(Hkv * D), o1(Hq * D);
        for (auto& v : q1) v = rand_half();
        for (auto& v : k1) v = rand_half();
        for (auto& v : v1) v = rand_half();
        cactus_hip_attention_hybrid_int8_fp16(q1.data(), Kq.data(), Vq.data(), Ks.data(), Vs.data(), k1.data(), v1.data(), o1.data(), 1,
                                              1, cache, 1, Hq, Hkv, D, 0.0f, cache, 1, gs);
        if (!ok("attention_hybrid_int8_fp16")) return 1;
        sum ^= checksum(o1.data(), o1.size() * 2);
    }
    // block ops
    {
        const size_t rows = 7, dims = 1024, n = rows * dims;
        std::vector<cactus_half_t> a(n), b(n), w(dims), o(n);
        for (auto& v : a) v = rand_half();
        for (auto& v : b) v = rand_half();
        for (auto& v : w) v = rand_half();
        cactus_hip_rms_norm_f16(a.data(), w.data(), o.data(), rows, dims, 1e-6f);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_rope_f16(a.data(), o.data(), 1, rows, 8, 128, 5, 1e6f);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_silu_f16(a.data(), o.data(), n);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_gelu_f16(a.data(), o.data(), n);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_add_f16(a.data(), b.data(), o.data(), n);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_add_f16_clipped(a.data(), b.data(), o.data(), n);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_subtract_f16(a.data(), b.data(), o.data(), n);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_multiply_f16(a.data(), b.data(), o.data(), n);
        sum ^= checksum(o.data(), n * 2);
        cactus_hip_divide_f16(a.data(), b.data(), o.data(), n);
        sum ^= checksum(o.data(), n * 2);
        if (!ok("block ops")) return 1;
    }
    std::printf("{\"host_pointer_calls\": \"ok\", \"rounds\": %d, \"checksum\": \"%016llx\"}\n", rounds, (unsigned long long)sum);
    std::fflush(stdout);
#ifdef EXIT_WITHOUT_TEARDOWN      // sanitizer builds: the ROCm ASan runtime can trip over itself while the HIP runtime unloads
    std::_Exit(0);
#endif
    return 0;
}
