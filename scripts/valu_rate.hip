// Issue-cost microbenchmark for gfx950: cycles per instruction of a few VALU ops, alone and
// placed behind an INT8 MFMA, one wave per SIMD and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate scripts/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

template <int MODE, int NV, bool MFMA>
__global__ void k(unsigned long long* out, int iters) {
    v4i a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
    v4i acc[8];
    for (int i = 0; i < 8; i++) acc[i] = v4i{i, i, i, i};
    float f[8], g[8];
    int iv[8];
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x + i; g[i] = i * 0.5f; iv[i] = threadIdx.x * 3 + i; }
    const float s = 1.0009765625f;
    const unsigned sh = 0x3C003C00u;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MFMA) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const int x = (u * NV + v) & 7;
                if (MODE == 0) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[x]) : "v"(iv[x]));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[x]) : "v"(g[x]), "v"(s));
                if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(f[x]) : "v"(g[x]), "v"(sh));
                if (MODE == 3) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[x]) : "v"(g[x]), "v"(s));
                if (MODE == 4) asm volatile("v_add_u32 %0, %1, %2" : "=v"(iv[x]) : "v"(iv[(x + 1) & 7]), "v"(x));
                if (MODE == 5) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[x]) : "v"(acc[(u + 4) & 7][v & 3]));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 8; i++) sum += f[i] + iv[i] + acc[i][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (sum == 123.456f) out[1] = 1;
}

template <int MODE, int NV, bool MFMA>
static void run(const char* name, unsigned long long* d, int threads) {
    const int iters = 2000;
    k<MODE, NV, MFMA><<<1, threads>>>(d, iters);
    k<MODE, NV, MFMA><<<1, threads>>>(d, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-14s mfma=%d nvalu=%d threads=%3d : %6.1f cycles per (mfma + %d valu) group\n", name, (int)MFMA, NV, threads,
           (double)h / (iters * 8.0), NV);
}

#define ALLNV(MODE, NAME, T)                                                                         \
    run<MODE, 1, false>(NAME, d, T); run<MODE, 4, false>(NAME, d, T);                                \
    run<MODE, 0, true>(NAME, d, T); run<MODE, 1, true>(NAME, d, T); run<MODE, 2, true>(NAME, d, T);  \
    run<MODE, 3, true>(NAME, d, T); run<MODE, 4, true>(NAME, d, T);

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    for (int T : {64, 256, 512}) {
        ALLNV(0, "cvt_f32_i32", T)
        ALLNV(1, "fma_f32", T)
        ALLNV(2, "fma_mix_f32", T)
        ALLNV(3, "sub_f32", T)
        ALLNV(5, "cvt(acc)", T)
    }
    return 0;
}
