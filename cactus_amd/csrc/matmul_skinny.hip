// Decode-shaped (M <= 16) W8A8 / W4A8 matmul: a pure weight stream.
//
// Roofline: HBM.  Algorithmic bytes per call = N*K (INT8) or N*K/2 (INT4) weight bytes
// + 2*N*K/128 scale bytes + M*K + 4*M activation bytes + 2*M*N output bytes.
//
// Layout.  One workgroup (8 waves) owns 16 consecutive weight rows n0..n0+15 and walks K in
// chunks of 32 groups (4096 values).  Inside a chunk wave w takes groups w, w+8, w+16, w+24,
// so the 8 waves stream disjoint 128-value slices of the same 16 rows: every dwordx4 load of
// a wave covers 16 rows x 64 contiguous bytes and goes straight to VGPRs (no LDS round trip:
// nothing is shared between waves).  The INT32 dot products come from
// v_mfma_i32_16x16x64_i8 with the activations as the row operand (rows m >= M are zero) and
// the 16 weight rows as the column operand; two MFMAs cover one 128-wide group, which makes
// the group sum exact and does the cross-lane reduction for free.  INT4 weights stay packed
// in HBM: one 16-byte load holds 32 values, the low nibbles feed the first MFMA and the high
// nibbles the second, each moved to the top of its byte (value*16, sign included) so no
// sign-extension arithmetic is needed; the factor 16 is folded into the scale (exactly, it
// is a power of two).  The activation operand is permuted to match (even / odd k).
//
// Any consistent permutation of k inside a group leaves an integer sum unchanged, so the
// lane->k mapping is chosen for coalescing, not for the MFMA's nominal k order.
//
// The fp32 part follows the reference order exactly (kernel_gemm.cpp:311-324):
//   running = fma(float(group_sum_g), float(B_scale[n,g]), running)   g ascending
//   C = half(running * A_scale[m])
// Waves drop their converted group sums into LDS; after one barrier per chunk, thread
// (m, n) of the workgroup runs the g-ordered fma chain.  LDS is double buffered so the
// chain of chunk c overlaps the streaming of chunk c+1.
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

constexpr int SK_WAVES = 8;
constexpr int SK_GPW = 4;
constexpr int SK_S = SK_WAVES * SK_GPW;  // groups per chunk
constexpr int SK_TN = 16;                // weight rows per workgroup
constexpr int SK_THREADS = SK_WAVES * 64;

__device__ __forceinline__ v4i ldg_nt(const void* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
}
__device__ __forceinline__ v4i ldg(const void* p) { return *reinterpret_cast<const v4i*>(p); }

template <bool INT4>
struct ChunkRegs {
    v4i b[SK_GPW][INT4 ? 1 : 2];
    v4i a[SK_GPW][2];
};

template <bool INT4>
__global__ __launch_bounds__(SK_THREADS) void skinny_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ A_scales,
    const uint8_t* __restrict__ B, const uint16_t* __restrict__ B_scales,
    uint16_t* __restrict__ C, int32_t* __restrict__ gsums, int M, int K, int N, int G) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];  // [2][SK_S][M][16]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15;   // MFMA row (activation row m) for A, column (weight row) for B
    const int q = lane >> 4;   // k block of the MFMA operand
    const int n0 = blockIdx.x * SK_TN;
    const int brow = min(n0 + r, N - 1);
    const size_t row_bytes = INT4 ? (size_t)K / 2 : (size_t)K;
    const uint8_t* bptr = B + (size_t)brow * row_bytes;
    const int8_t* aptr = A + (size_t)min(r, M - 1) * K;
    const bool a_live = r < M;
    const v4i zero4 = {0, 0, 0, 0};

    auto load_chunk = [&](int c, ChunkRegs<INT4>& R) {
#pragma unroll
        for (int j = 0; j < SK_GPW; j++) {
            const int g = c * SK_S + j * SK_WAVES + w;
            if (g < G) {
                if constexpr (INT4) {
                    R.b[j][0] = ldg_nt(bptr + (size_t)g * 64 + q * 16);
                    const int8_t* ap = aptr + (size_t)g * 128 + q * 32;
                    R.a[j][0] = a_live ? ldg(ap) : zero4;
                    R.a[j][1] = a_live ? ldg(ap + 16) : zero4;
                } else {
                    const uint8_t* bp = bptr + (size_t)g * 128 + q * 16;
                    R.b[j][0] = ldg_nt(bp);
                    R.b[j][1] = ldg_nt(bp + 64);
                    const int8_t* ap = aptr + (size_t)g * 128 + q * 16;
                    R.a[j][0] = a_live ? ldg(ap) : zero4;
                    R.a[j][1] = a_live ? ldg(ap + 64) : zero4;
                }
            }
        }
    };

    const int nchunks = (G + SK_S - 1) / SK_S;
    const int t = threadIdx.x;
    const bool chain = (t < M * 16) && C != nullptr;  // C == nullptr: group sums only
    const int cm = t >> 4, cn = t & 15;
    const uint16_t* sc_ptr = B_scales + (size_t)min(n0 + cn, N - 1) * G;
    float run = 0.0f;

    ChunkRegs<INT4> cur, nxt;
    load_chunk(0, cur);

    for (int c = 0; c < nchunks; c++) {
        if (c + 1 < nchunks) load_chunk(c + 1, nxt);

        // scales of this chunk for the chain thread (issued early, consumed after the barrier)
        const int ng = min(SK_S, G - c * SK_S);
        float sc[SK_S];
        if (chain) {
#pragma unroll
            for (int gl = 0; gl < SK_S; gl++) {
                const float s = (gl < ng) ? h2f(sc_ptr[c * SK_S + gl]) : 0.0f;
                sc[gl] = INT4 ? s * 0.0625f : s;
            }
        }

        float* buf = sk_lds + (size_t)(c & 1) * SK_S * M * 16;
#pragma unroll
        for (int j = 0; j < SK_GPW; j++) {
            const int gl = j * SK_WAVES + w;
            const int g = c * SK_S + gl;
            if (g < G) {
                v4i acc = zero4;
                if constexpr (INT4) {
                    v4i blo, bhi, ae, ao;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t x = (uint32_t)cur.b[j][0][i];
                        blo[i] = (int)((x & 0x0F0F0F0Fu) << 4);
                        bhi[i] = (int)(x & 0xF0F0F0F0u);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t lo = (uint32_t)cur.a[j][i >> 1][(2 * i) & 3];
                        const uint32_t hi = (uint32_t)cur.a[j][i >> 1][((2 * i) & 3) + 1];
                        ae[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x06040200u);
                        ao[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x07050301u);
                    }
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ae, blo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ao, bhi, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[j][0], cur.b[j][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[j][1], cur.b[j][1], acc, 0, 0, 0);
                }
                // result layout: lane holds column n = lane & 15, rows m = 4*(lane >> 4) + i
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int m = q * 4 + i;
                    if (m < M) {
                        buf[(gl * M + m) * 16 + r] = (float)acc[i];
                        if (gsums != nullptr && n0 + r < N)
                            gsums[((size_t)m * N + n0 + r) * G + g] = INT4 ? (acc[i] >> 4) : acc[i];
                    }
                }
            }
        }
        __syncthreads();
        if (chain) {
#pragma unroll
            for (int gl = 0; gl < SK_S; gl++)
                if (gl < ng) run = fmaf(buf[(gl * M + cm) * 16 + cn], sc[gl], run);
        }
        cur = nxt;
    }

    if (chain && n0 + cn < N)
        C[(size_t)cm * N + n0 + cn] = f2h_mul(run, A_scales[cm]);
}

hipError_t launch_matmul_skinny(bool int4, const int8_t* A, const float* A_scales, const void* B,
                                const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                                size_t K, size_t N, hipStream_t st) {
    const int G = (int)(K / 128);
    const size_t lds = (size_t)2 * SK_S * M * 16 * sizeof(float);
    const dim3 grid((unsigned)((N + SK_TN - 1) / SK_TN)), block(SK_THREADS);
    if (int4)
        skinny_kernel<true><<<grid, block, lds, st>>>(A, A_scales, (const uint8_t*)B, B_scales, C, gsums,
                                                      (int)M, (int)K, (int)N, G);
    else
        skinny_kernel<false><<<grid, block, lds, st>>>(A, A_scales, (const uint8_t*)B, B_scales, C, gsums,
                                                       (int)M, (int)K, (int)N, G);
    return hipGetLastError();
}

}  // namespace cactus_hip
