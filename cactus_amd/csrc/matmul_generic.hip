// Shape-agnostic grouped INT8/INT4 matmul: one thread per output element, any group size.
// Correctness path for the shapes the tiled kernels do not take (group_size != 128, operands
// that are not 16-byte aligned); same arithmetic order as the reference
// (kernel_gemm.cpp:170-173, 311-324): exact INT32 group sum, fp32 fma chain over ascending g.
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

template <bool INT4>
__global__ __launch_bounds__(256) void matmul_generic_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ A_scales, const uint8_t* __restrict__ B,
    const uint16_t* __restrict__ B_scales, uint16_t* __restrict__ C, int32_t* __restrict__ gsums,
    size_t M, size_t K, size_t N, size_t gs) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N) return;
    const size_t m = idx / N, n = idx % N;
    const size_t G = K / gs;
    const int8_t* a = A + m * K;
    float run = 0.0f;
    for (size_t g = 0; g < G; g++) {
        int32_t acc = 0;
        for (size_t k = g * gs; k < (g + 1) * gs; k++) {
            int b;
            if constexpr (INT4) {
                const uint8_t byte = B[n * (K / 2) + k / 2];
                const int nib = (k & 1) ? (byte >> 4) : (byte & 0x0F);
                b = nib > 7 ? nib - 16 : nib;
            } else {
                b = (int)reinterpret_cast<const int8_t*>(B)[n * K + k];
            }
            acc += (int)a[k] * b;
        }
        if (gsums != nullptr) gsums[idx * G + g] = acc;
        if (C != nullptr) run = fmaf((float)acc, h2f(B_scales[n * G + g]), run);
    }
    if (C != nullptr) C[idx] = f2h_mul(run, A_scales[m]);
}

hipError_t launch_matmul_generic(bool int4, const int8_t* A, const float* A_scales, const void* B,
                                 const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                                 size_t K, size_t N, size_t gs, hipStream_t st) {
    const dim3 grid((unsigned)((M * N + 255) / 256)), block(256);
    if (int4)
        matmul_generic_kernel<true><<<grid, block, 0, st>>>(A, A_scales, (const uint8_t*)B, B_scales, C,
                                                           gsums, M, K, N, gs);
    else
        matmul_generic_kernel<false><<<grid, block, 0, st>>>(A, A_scales, (const uint8_t*)B, B_scales, C,
                                                            gsums, M, K, N, gs);
    return hipGetLastError();
}

}  // namespace cactus_hip
