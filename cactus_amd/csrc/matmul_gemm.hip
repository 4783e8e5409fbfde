// Prefill-shaped (M > 16) W8A8 / W4A8 matmul on the INT8 matrix cores.
//
// Roofline: MFMA (INT8) once M >= ~256; algorithmic ops = 2*M*N*K.
//
// One 256-thread workgroup (4 waves, 2x2) owns a 128(m) x 128(n) output tile and walks K one
// quantization group (128 values) per step.  Both operand tiles are staged global -> VGPR ->
// LDS as 128 rows x 128 bytes with the 16-byte chunk index XOR-swizzled by (row & 7), which
// makes the ds_read_b128 operand reads of v_mfma_i32_16x16x64_i8 conflict-free.  The loads of
// step t+1 are issued before the MFMAs of step t and written to LDS after them.
//
// Per step each wave runs 32 MFMAs into INT32 accumulators that start at zero, so the group
// sum is exact; the accumulators are then folded into fp32 running sums in group order,
//   running = fma(float(acc), float(B_scale[n,g]), running)     (kernel_gemm.cpp:311-316)
// and the output is half(running * A_scale[m]) (kernel_gemm.cpp:319-324): the same order of
// floating-point operations as the reference, for every M.
//
// INT4 weights stay packed in HBM and are split while staging: low nibbles (even k) fill
// chunks 0-3 of an LDS row, high nibbles (odd k) chunks 4-7, each nibble moved to the top of
// its byte (value * 16, sign included); the activation tile is de-interleaved the same way
// and the factor 16 is folded into the scale.
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 128;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <bool INT4>
__global__ __launch_bounds__(256) void gemm_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ A_scales, const uint8_t* __restrict__ B,
    const uint16_t* __restrict__ B_scales, uint16_t* __restrict__ C, int32_t* __restrict__ gsums,
    int M, int K, int N, int G) {
    __shared__ __attribute__((aligned(16))) uint8_t sA[GM_BM * GM_BK];
    __shared__ __attribute__((aligned(16))) uint8_t sB[GM_BN * GM_BK];

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    const size_t brow_bytes = INT4 ? (size_t)K / 2 : (size_t)K;

    v4i pa[4], pb[INT4 ? 2 : 4];
    float psc[4];

    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int idx = p * 256 + t, row = idx >> 3, ch = idx & 7;
            const int gm = min(m0 + row, M - 1);
            pa[p] = *reinterpret_cast<const v4i*>(A + (size_t)gm * K + (size_t)kt * 128 + ch * 16);
        }
        if constexpr (INT4) {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int idx = p * 256 + t, row = idx >> 2, ch = idx & 3;
                const int gn = min(n0 + row, N - 1);
                pb[p] = *reinterpret_cast<const v4i*>(B + (size_t)gn * brow_bytes + (size_t)kt * 64 + ch * 16);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int idx = p * 256 + t, row = idx >> 3, ch = idx & 7;
                const int gn = min(n0 + row, N - 1);
                pb[p] = *reinterpret_cast<const v4i*>(B + (size_t)gn * brow_bytes + (size_t)kt * 128 + ch * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gn = min(n0 + wn * 64 + j * 16 + r, N - 1);
            const float s = (B_scales != nullptr) ? h2f(B_scales[(size_t)gn * G + kt]) : 0.0f;
            psc[j] = INT4 ? s * 0.0625f : s;
        }
    };

    auto lds_store = [&]() {
        if constexpr (INT4) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int idx = p * 256 + t, row = idx >> 3, ch = idx & 7;
                // 16 activation bytes k = 16ch..16ch+15 -> 8 even (k/2 = 8ch..) and 8 odd bytes
                uint2 ev, od;
                ev.x = __builtin_amdgcn_perm((uint32_t)pa[p][1], (uint32_t)pa[p][0], 0x06040200u);
                ev.y = __builtin_amdgcn_perm((uint32_t)pa[p][3], (uint32_t)pa[p][2], 0x06040200u);
                od.x = __builtin_amdgcn_perm((uint32_t)pa[p][1], (uint32_t)pa[p][0], 0x07050301u);
                od.y = __builtin_amdgcn_perm((uint32_t)pa[p][3], (uint32_t)pa[p][2], 0x07050301u);
                *reinterpret_cast<uint2*>(sA + swz(row, ch >> 1) + (ch & 1) * 8) = ev;
                *reinterpret_cast<uint2*>(sA + swz(row, 4 + (ch >> 1)) + (ch & 1) * 8) = od;
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int idx = p * 256 + t, row = idx >> 2, ch = idx & 3;
                v4i lo, hi;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t x = (uint32_t)pb[p][i];
                    lo[i] = (int)((x & 0x0F0F0F0Fu) << 4);
                    hi[i] = (int)(x & 0xF0F0F0F0u);
                }
                *reinterpret_cast<v4i*>(sB + swz(row, ch)) = lo;
                *reinterpret_cast<v4i*>(sB + swz(row, 4 + ch)) = hi;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int idx = p * 256 + t, row = idx >> 3, ch = idx & 7;
                *reinterpret_cast<v4i*>(sA + swz(row, ch)) = pa[p];
                *reinterpret_cast<v4i*>(sB + swz(row, ch)) = pb[p];
            }
        }
    };

    float run[4][4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) run[i][j][e] = 0.0f;

    gload(0);
    lds_store();
    float sc[4] = {psc[0], psc[1], psc[2], psc[3]};
    __syncthreads();

    for (int kt = 0; kt < G; kt++) {
        if (kt + 1 < G) gload(kt + 1);

        v4i acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = v4i{0, 0, 0, 0};

#pragma unroll
        for (int s = 0; s < 2; s++) {
            v4i fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                fa[i] = *reinterpret_cast<const v4i*>(sA + swz(wm * 64 + i * 16 + r, 4 * s + q));
#pragma unroll
            for (int j = 0; j < 4; j++)
                fb[j] = *reinterpret_cast<const v4i*>(sB + swz(wn * 64 + j * 16 + r, 4 * s + q));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }

        if (gsums != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int m = m0 + wm * 64 + i * 16 + q * 4 + e, n = n0 + wn * 64 + j * 16 + r;
                        if (m < M && n < N)
                            gsums[((size_t)m * N + n) * G + kt] = INT4 ? (acc[i][j][e] >> 4) : acc[i][j][e];
                    }
        }

#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) run[i][j][e] = fmaf((float)acc[i][j][e], sc[j], run[i][j][e]);

        __syncthreads();
        if (kt + 1 < G) {
            lds_store();
#pragma unroll
            for (int j = 0; j < 4; j++) sc[j] = psc[j];
            __syncthreads();
        }
    }

    if (C == nullptr) return;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int m = m0 + wm * 64 + i * 16 + q * 4 + e;
            if (m >= M) continue;
            const float as = A_scales[m];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = n0 + wn * 64 + j * 16 + r;
                if (n < N) C[(size_t)m * N + n] = f2h_mul(run[i][j][e], as);
            }
        }
}

hipError_t launch_matmul_gemm(bool int4, const int8_t* A, const float* A_scales, const void* B,
                              const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                              size_t K, size_t N, hipStream_t st) {
    const int G = (int)(K / 128);
    const dim3 grid((unsigned)((N + GM_BN - 1) / GM_BN), (unsigned)((M + GM_BM - 1) / GM_BM)), block(256);
    if (int4)
        gemm_kernel<true><<<grid, block, 0, st>>>(A, A_scales, (const uint8_t*)B, B_scales, C, gsums, (int)M,
                                                  (int)K, (int)N, G);
    else
        gemm_kernel<false><<<grid, block, 0, st>>>(A, A_scales, (const uint8_t*)B, B_scales, C, gsums, (int)M,
                                                   (int)K, (int)N, G);
    return hipGetLastError();
}

}  // namespace cactus_hip
