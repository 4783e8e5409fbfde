// Standalone timing harness for the M<=16 streaming matmul variants (hipEvent timed, rotating
// weight buffers larger than the Infinity Cache).  Build: see scripts/build_bench.sh.
//   gemv_bench M K N int4 iters variant [variant ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../cactus_amd/csrc/kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(uint32_t* p, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = x;
    }
}
__global__ void fill_scales(uint16_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0x2000 + (uint16_t)((i * 37) & 0x3FF);   // ~0.008 .. 0.016
}

typedef int v4i_ __attribute__((ext_vector_type(4)));
// calibration 1: linear 16 B/lane stream, grid-stride, XOR-reduce (one dword store per wave)
__global__ __launch_bounds__(256) void stream_linear(const v4i_* __restrict__ p, size_t n16, int* out, int nt) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    v4i_ acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n16; i += 4 * stride) {
        v4i_ a, b, c, d;
        if (nt) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride);
                  c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride); }
        else { a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride]; }
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= p[i];
    int r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x12345678) out[0] = r;
}
// calibration 2: the GEMV's pattern -- a wave reads 16 rows x 64 B per instruction, W waves of a
// workgroup take interleaved 128-B groups of the same 16 rows -- without MFMA/LDS
template <int W, int P>
__global__ __launch_bounds__(W * 64) void stream_rows(const uint8_t* __restrict__ B, int K, int* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const uint8_t* bp = B + (size_t)(blockIdx.x * 16 + r) * K + q * 16;
    const int G = K / 128;
    v4i_ acc = {0, 0, 0, 0};
    for (int g = w; g < G; g += W * P) {
        v4i_ x[P][2];
#pragma unroll
        for (int p = 0; p < P; p++) if (g + p * W < G) {
            x[p][0] = __builtin_nontemporal_load((const v4i_*)(bp + (size_t)(g + p * W) * 128));
            x[p][1] = __builtin_nontemporal_load((const v4i_*)(bp + (size_t)(g + p * W) * 128 + 64));
        }
#pragma unroll
        for (int p = 0; p < P; p++) if (g + p * W < G) acc ^= x[p][0] ^ x[p][1];
    }
    int rr = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (rr == 0x12345678) out[0] = rr;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: gemv_bench M K N int4 iters variant...\n"); return 2; }
    const size_t M = atoi(argv[1]), K = atoi(argv[2]), N = atoi(argv[3]);
    const bool int4 = atoi(argv[4]) != 0;
    const int iters = atoi(argv[5]);
    const size_t wbytes = int4 ? N * K / 2 : N * K;
    size_t nbuf = (600u << 20) / wbytes + 1;
    if (nbuf < 2) nbuf = 2;
    if (nbuf > 64) nbuf = 64;
    std::vector<uint8_t*> W(nbuf);
    for (size_t i = 0; i < nbuf; i++) {
        CK(hipMalloc(&W[i], wbytes));
        fill_kernel<<<1024, 256>>>((uint32_t*)W[i], wbytes / 4, (uint32_t)i * 977u + 1u);
    }
    uint16_t* Bs; CK(hipMalloc(&Bs, N * (K / 128) * 2));
    fill_scales<<<(unsigned)((N * (K / 128) + 255) / 256), 256>>>(Bs, N * (K / 128));
    int8_t* A; CK(hipMalloc(&A, M * K));
    fill_kernel<<<64, 256>>>((uint32_t*)A, M * K / 4, 12345u);
    // FP16 activations for the fused-quantization variants (2xxxx): halves in [-1, 1)
    int8_t* X; CK(hipMalloc(&X, 2 * M * K));
    {
        std::vector<uint16_t> hx(M * K);
        uint32_t st = 99u;
        for (auto& h : hx) { st = st * 1664525u + 1013904223u; h = (uint16_t)(0x3000u + ((st >> 9) & 0x0BFFu)) | (uint16_t)((st >> 1) & 0x8000u); }
        CK(hipMemcpy(X, hx.data(), 2 * M * K, hipMemcpyHostToDevice));
    }
    float* As; CK(hipMalloc(&As, M * 4));
    std::vector<float> has(M, 0.01f);
    CK(hipMemcpy(As, has.data(), M * 4, hipMemcpyHostToDevice));
    uint16_t* C; CK(hipMalloc(&C, M * N * 2));
    std::vector<uint16_t> ref(M * N), got(M * N);
    CK(hipDeviceSynchronize());
    const double bytes = (double)wbytes + 2.0 * N * (K / 128) + M * K + 4 * M + 2.0 * M * N;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int* dout; CK(hipMalloc(&dout, 64));
    bool have_ref = false;
    for (int a = 6; a < argc; a++) {
        int variant = atoi(argv[a]);
        cactus_hip::g_skinny_ablate = 0;
        if (variant >= 100000) { cactus_hip::g_skinny_ablate = variant / 100000; variant %= 100000; }
        if (variant > 0 && variant < 100) {
            auto run = [&](int i) {
                const uint8_t* wp = W[i % nbuf];
                if (variant == 1) stream_linear<<<2048, 256>>>((const v4i_*)wp, wbytes / 16, dout, 0);
                else if (variant == 2) stream_linear<<<2048, 256>>>((const v4i_*)wp, wbytes / 16, dout, 1);
                else if (variant == 3) stream_linear<<<8192, 256>>>((const v4i_*)wp, wbytes / 16, dout, 1);
                else if (variant == 4) stream_rows<8, 4><<<(unsigned)(N / 16), 512>>>(wp, (int)(int4 ? K / 2 : K), dout);
                else if (variant == 5) stream_rows<4, 4><<<(unsigned)(N / 16), 256>>>(wp, (int)(int4 ? K / 2 : K), dout);
                else if (variant == 6) stream_rows<8, 8><<<(unsigned)(N / 16), 512>>>(wp, (int)(int4 ? K / 2 : K), dout);
            };
            for (int i = 0; i < 5; i++) run(i);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; i++) run(i + 5);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("calibration %d: %.2f us  %.1f GB/s (weights only)\n", variant, us, (double)wbytes / us / 1e3);
            continue;
        }
        const bool fq = variant >= 20000 && variant < 30000;
        const int8_t* Ain = fq ? X : A;
        CK(cactus_hip::launch_matmul_skinny_variant(variant, int4, Ain, As, W[0], Bs, C, nullptr, M, K, N, 0));
        CK(hipMemcpy(got.data(), C, M * N * 2, hipMemcpyDeviceToHost));
        if (!have_ref) { ref = got; have_ref = true; }
        const bool same = memcmp(ref.data(), got.data(), M * N * 2) == 0;
        for (int i = 0; i < 5; i++)
            CK(cactus_hip::launch_matmul_skinny_variant(variant, int4, Ain, As, W[i % nbuf], Bs, C, nullptr, M, K, N, 0));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++)
            CK(cactus_hip::launch_matmul_skinny_variant(variant, int4, Ain, As, W[(i + 5) % nbuf], Bs, C, nullptr, M, K, N, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        if (cactus_hip::g_skinny_ablate & 16) {
            std::vector<unsigned long long> tr(512);
            CK(cactus_hip::skinny_read_trace(tr.data()));
            for (int tl : {0, 1, 31, 63}) {
                printf("  tile %2d phases (cycles from entry):", tl);
                for (int ph = 1; ph < 7; ph++) printf(" %lld", (long long)(tr[tl * 8 + ph] - tr[tl * 8]));
                printf("   entry-rel-to-tile0 %lld\n", (long long)(tr[tl * 8] - tr[0]));
            }
        }
        printf("M=%zu K=%zu N=%zu %s variant=%d ablate=%d  %.2f us  %.1f GB/s  (%.1f%% of 8 TB/s)  %s\n", M, K, N,
               int4 ? "int4" : "int8", variant, cactus_hip::g_skinny_ablate, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0,
               same ? "same" : "DIFFERENT");
    }
    return 0;
}
