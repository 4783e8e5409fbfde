// Issue-cost microbenchmark for gfx950: an INT8 MFMA followed by the GEMM's per-group rescale work
// (v_cvt_f32_i32 of an older accumulator + v_fma_f32), for the two INT8 MFMA shapes, at one and two waves
// per SIMD.  Prints shader cycles per 16384 multiply-accumulates (one 16x16x64, half a 32x32x32) with NV VALU
// instructions behind them.  Modes: the rescale as v_cvt_f32_i32 + v_fma_f32 per output (nvalu = 4 is the GEMM's
// work per 16x16x64), as two conversions + one v_pk_fma_f32 per output pair (nvalu = 3), and as one v_pk_add_f32
// (accumulators started at the bits of 1.5 * 2^23, minus that constant: exact) + one v_pk_fma_f32 per pair (nvalu = 2).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix scripts/mfma_valu_mix.hip && ./mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int NV, int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, 8};
    v4i acc[8];
    v16i big[4];
    for (int i = 0; i < 8; i++) acc[i] = v4i{i, i, i, i};
    for (int i = 0; i < 4; i++)
        for (int e = 0; e < 16; e++) big[i][e] = i + e;
    typedef float v2f __attribute__((ext_vector_type(2)));
    float f[8], run[8];
    v2f f2[8], run2[8];
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x + i; run[i] = i * 0.5f; f2[i] = v2f{f[i], f[i]}; run2[i] = v2f{run[i], run[i]}; }
    const float s = 1.0009765625f;
    const v2f s2 = {s, s}, magic = {-12582912.0f, -12582912.0f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (SHAPE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[u], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const int x = (u * NV + v) & 7;
                    if (MODE == 0) {
                        if (v & 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(run[x]) : "v"(f[(x + 3) & 7]), "v"(s));
                        else asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[x]) : "v"(acc[(u + 4) & 7][(v >> 1) & 3]));
                    } else if (MODE == 1) {      // 2 cvt + 1 pk_fma per 2 outputs: NV counts instructions, 3 per pair
                        if (v % 3 == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(run2[x]) : "v"(f2[(x + 3) & 7]), "v"(s2));
                        else asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f2[x][v % 3]) : "v"(acc[(u + 4) & 7][v & 3]));
                    } else {                     // 1 pk_add + 1 pk_fma per 2 outputs
                        const v4i& src = acc[(u + 4) & 7];
                        if (v & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(run2[x]) : "v"(f2[(x + 3) & 7]), "v"(s2));
                        else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f2[x]) : "v"(__builtin_shufflevector(src, src, 0, 1)), "v"(magic));
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                big[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, big[u], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 2 * NV; v++) {
                    const int x = (u * 2 * NV + v) & 7;
                    if (MODE == 0) {
                        if (v & 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(run[x]) : "v"(f[(x + 3) & 7]), "v"(s));
                        else asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[x]) : "v"(big[(u + 2) & 3][(v >> 1) & 15]));
                    } else if (MODE == 1) {
                        if (v % 3 == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(run2[x]) : "v"(f2[(x + 3) & 7]), "v"(s2));
                        else asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f2[x][v % 3]) : "v"(big[(u + 2) & 3][v & 15]));
                    } else {
                        const v16i& src = big[(u + 2) & 3];
                        if (v & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(run2[x]) : "v"(f2[(x + 3) & 7]), "v"(s2));
                        else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f2[x]) : "v"(__builtin_shufflevector(src, src, 2, 3)), "v"(magic));
                    }
                }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 8; i++) sum += f[i] + run[i] + acc[i][0] + f2[i][0] + f2[i][1] + run2[i][0] + run2[i][1];
    for (int i = 0; i < 4; i++) sum += big[i][0];
    if ((threadIdx.x & 63) == 0) out[2 + (threadIdx.x >> 6)] = t1 - t0;     // per wave: the SIMD's arbiter favours the oldest
    if (sum == 123.456f) out[1] = 1;
}

template <int SHAPE, int NV, int MODE>
static void run(unsigned long long* d, int threads) {
    const int iters = 2000;
    k<SHAPE, NV, MODE><<<1, threads>>>(d, iters);
    k<SHAPE, NV, MODE><<<1, threads>>>(d, iters);
    hipDeviceSynchronize();
    unsigned long long h[10];
    hipMemcpy(h, d, 80, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < threads / 64; w++) { lo = h[2 + w] < lo ? h[2 + w] : lo; hi = h[2 + w] > hi ? h[2 + w] : hi; }
    printf("%s  %s  waves/SIMD=%d  nvalu=%d : %6.1f cycles per SIMD and (16384 MACs + %d VALU)   [fastest wave %.1f, slowest %.1f per own group]\n",
           SHAPE ? "32x32x32" : "16x16x64", MODE == 0 ? "cvt+fma        " : MODE == 1 ? "2cvt+pk_fma    " : "pk_add+pk_fma  ", threads / 256, NV, (double)hi / (iters * 8.0) / (threads / 256), NV,
           (double)lo / (iters * 8.0), (double)hi / (iters * 8.0));
}

#define ALLNV(SHAPE, T) run<SHAPE, 0, 0>(d, T); run<SHAPE, 2, 0>(d, T); run<SHAPE, 4, 0>(d, T); run<SHAPE, 6, 0>(d, T); \
    run<SHAPE, 3, 1>(d, T); run<SHAPE, 1, 2>(d, T); run<SHAPE, 2, 2>(d, T); run<SHAPE, 4, 2>(d, T);

int main() {
    unsigned long long* d;
    hipMalloc(&d, 128);
    for (int T : {256, 512}) {
        ALLNV(0, T)
        ALLNV(1, T)
    }
    return 0;
}
