// Measured matrix-core peak on gfx950: a grid that fills every SIMD with waves that do nothing but
// independent MFMAs (8 accumulator chains per wave), timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_rate scripts/mfma_rate.hip && ./mfma_rate
// Prints dense TOP/s for v_mfma_i32_16x16x64_i8 (the W8A8 / W4A8 GEMM's instruction) and TFLOP/s for
// v_mfma_f32_16x16x32_f16 (attention), at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <bool F16>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
    v4i acc[8];
    v4f facc[8];
    for (int i = 0; i < 8; i++) { acc[i] = v4i{i, i, i, i}; facc[i] = v4f{0.f, 0.f, 0.f, 0.f}; }
    const v8h ha = __builtin_bit_cast(v8h, a), hb = __builtin_bit_cast(v8h, b);
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (F16) facc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, facc[u], 0, 0, 0);
            else acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[u], 0, 0, 0);
        }
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += F16 ? facc[i][0] : (float)acc[i][0];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // shader cycles and constant-rate (100 MHz) ticks of the loop
        out[1] = (float)(c1 - c0);
        out[2] = (float)(r1 - r0);
    }
}

template <bool F16>
void run(int wgs_per_cu, int cus) {
    float* out;
    hipMalloc(&out, 16);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<F16><<<cus * wgs_per_cu, 256>>>(out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<F16><<<cus * wgs_per_cu, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)cus * wgs_per_cu * 4 * iters * 8 * (F16 ? 16.0 * 16 * 32 * 2 : 16.0 * 16 * 64 * 2);
    float h[4];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%s  %d wave(s) per SIMD: %.3f ms  %.1f T%s/s   shader clock under this load %.0f MHz, workgroup 0: %.1f cycles per MFMA\n",
           F16 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_i32_16x16x64_i8 ", wgs_per_cu, ms, ops / (ms * 1e-3) / 1e12,
           F16 ? "FLOP" : "OP", h[1] / h[2] * 100.0, h[1] / (iters * 8.0));
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    for (int w : {1, 2, 4}) run<false>(w, p.multiProcessorCount);
    for (int w : {1, 2, 4}) run<true>(w, p.multiProcessorCount);
    return 0;
}
