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
#include <cstring>
#include <type_traits>
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

// ---------------------------------------------------------------------------------------
// Pipelined kernel (the one used for outputs; the kernel above remains for the INT32 group-sum
// output and for K so long that the scale panel does not fit LDS).
//
// 512 threads = 8 waves as 2 (m) x 4 (n); each wave owns 64 x 32 outputs = 4 x 2 MFMA tiles, so
// every SIMD hosts two waves and one wave's fp32 rescale (VALU) overlaps the other's MFMAs.
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR staging) into a ring of
// four stages; tile t+3 is requested while tile t is consumed (96 KB in flight per CU), counted s_waitcnt vmcnt
// and raw s_barriers order the hand-off (no vmcnt(0) in the loop).  The LDS image of
// a tile is lane-linear per load instruction, so the bank swizzle (16-byte slot XOR (row>>1)&7
// for 128-byte rows, XOR {0,2,3,1}[(row>>2)&3] for the 64-byte rows of packed INT4: both are
// conflict-free for ds_read_b128's lane groups) is applied to the
// per-lane SOURCE address and again when fragments are read.  The B-scale panel of the tile
// (128 columns x G groups, fp16) is staged once, transposed to [group][column].
// INT4 stays packed through LDS: a lane's 16 packed bytes give the low-nibble operand of the
// first MFMA and the high-nibble operand of the second (nibbles moved to the top of the byte),
// with the activation fragment split into even / odd bytes by v_perm.
// Workgroup ids are remapped so that the m-tiles sharing one weight panel run on one XCD (one
// L2) back to back.
constexpr int G2_STAGE = 32768;     // bytes per ring stage (A 16 KB + B up to 16 KB)
constexpr int G2_STAGES = 4;

// Timing experiments only (results are wrong when nonzero): bit 0 no LDS-DMA requests in the loop,
// bit 1 no rescale, bit 2 one MFMA per slab instead of 8, bit 3 no fragment reads, bit 4 no barrier.
#ifndef G2_TIMING
#define G2_TIMING 0
#endif

// Benchmarks only (-DGEMM_TRACE): per-phase shader-clock totals of wave 0 of the first 64 tiles.
#ifdef GEMM_TRACE
__device__ unsigned long long g_gemm_trace[64 * 8];
#define G2_T(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define G2_T(var)
#endif

typedef __attribute__((address_space(1))) const void* gptr_t;

// A wave-uniform pointer pinned in scalar registers, so that `p + 32-bit lane offset` selects the
// scalar-base addressing form and advancing p costs no vector instructions.
template <typename T>
__device__ __forceinline__ const T* scalar_ptr(const T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}
typedef __attribute__((address_space(3))) void* lptr_t;

// Several weight matrices that multiply the same activations (q/k/v, gate/up) in one launch: the
// n-tiles of all of them are laid side by side; bn_end[i] = running count of n-tiles.
constexpr int G2_MAXB = 8;
struct GemmBatch {
    int count;                 // 0: the single problem of the plain arguments
    int bn_end[G2_MAXB];
    const uint8_t* B[G2_MAXB];
    const uint16_t* Bs[G2_MAXB];
    uint16_t* C[G2_MAXB];
    int N[G2_MAXB];
};

// NI: 16-row MFMA tiles per wave along m (4: 128-row workgroup tile, 2: 64-row tile for launches whose
// 128-row tiles would leave most of the chip idle, 1: 32-row tile when 64-row tiles still number fewer
// than half the CUs -- a workgroup then re-reads its weight tile for a quarter of the rows, from L2).
template <bool INT4, int NI = 4>
__global__ __launch_bounds__(512) void gemm2_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ A_scales, const uint8_t* __restrict__ B_arg,
    const uint16_t* __restrict__ B_scales_arg, uint16_t* __restrict__ C_arg, int M, int K, int N_arg, int G, int nbm,
    int nwg, const GemmBatch bt) {
    const uint8_t* __restrict__ B = B_arg;
    const uint16_t* __restrict__ B_scales = B_scales_arg;
    uint16_t* __restrict__ C = C_arg;
    int N = N_arg;
    extern __shared__ __attribute__((aligned(16))) uint8_t g2_lds[];
    uint16_t* sc16 = reinterpret_cast<uint16_t*>(g2_lds + G2_STAGES * G2_STAGE);   // [G][128]
    G2_T(tr_entry);

    const int t = threadIdx.x;
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int r = lane & 15, q = lane >> 4;

    // XCD-aware, bijective remap of the workgroup id (ids congruent mod 8 share an XCD)
    const int orig = blockIdx.x, xcd = orig & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wg = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
    const int bm = wg % nbm;
    int bn = wg / nbm;
    if (bt.count) {
        int pi = 0;
        while (pi + 1 < bt.count && bn >= bt.bn_end[pi]) pi++;
        bn -= pi ? bt.bn_end[pi - 1] : 0;
        B = bt.B[pi]; B_scales = bt.Bs[pi]; C = bt.C[pi]; N = bt.N[pi];
    }
    constexpr int BM = 32 * NI;          // rows of the workgroup tile
    const int m0 = bm * BM, n0 = bn * GM_BN;
    const size_t brow_bytes = INT4 ? (size_t)K / 2 : (size_t)K;

    // Source of this thread's pieces (a piece = one wave instruction = 1 KB): a workgroup-uniform
    // base that advances with kt (scalar registers) plus a 32-bit per-lane offset inside the tile's
    // rows, which is the addressing form global_load_lds takes without any per-step vector math.
    // Rows past M / N are clamped to the last row (their results are never stored).
    const int8_t* a_base = A + (size_t)m0 * K;
    const uint8_t* b_base = B + (size_t)n0 * brow_bytes;
    unsigned a_src[2], b_src[2];
    int a_dst[2], b_dst[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int pi = w + 8 * j;
        const int row = pi * 8 + (lane >> 3), slot = lane & 7, ch = slot ^ ((row >> 1) & 7);
        a_src[j] = (unsigned)min(min(row, BM - 1), M - 1 - m0) * (unsigned)K + ch * 16;   // piece j >= NI/2: unused
        a_dst[j] = pi * 1024;
        if constexpr (!INT4) {
            b_src[j] = (unsigned)min(row, N - 1 - n0) * (unsigned)brow_bytes + ch * 16;
            b_dst[j] = 16384 + pi * 1024;
        }
    }
    if constexpr (INT4) {
        const int row = w * 16 + (lane >> 2), slot = lane & 3, ch = slot ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);
        b_src[0] = (unsigned)min(row, N - 1 - n0) * (unsigned)brow_bytes + ch * 16;
        b_dst[0] = 16384 + w * 1024;
        b_src[1] = 0; b_dst[1] = 0;
    }
    auto issue_tile = [&](int kt) {
        uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
        const int8_t* ak = scalar_ptr(a_base + (size_t)kt * 128);
        const uint8_t* bk = scalar_ptr(b_base + (size_t)kt * (INT4 ? 64 : 128));
#pragma unroll
        for (int j = 0; j < (NI + 1) / 2; j++)   // NI = 1: rows past the tile are clamped duplicates
            __builtin_amdgcn_global_load_lds((gptr_t)(ak + a_src[j]), (lptr_t)(st + a_dst[j]), 16, 0, 0);
        if constexpr (INT4) {
            __builtin_amdgcn_global_load_lds((gptr_t)(bk + b_src[0]), (lptr_t)(st + b_dst[0]), 16, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++)
                __builtin_amdgcn_global_load_lds((gptr_t)(bk + b_src[j]), (lptr_t)(st + b_dst[j]), 16, 0, 0);
        }
    };

    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f run[NI][2][2];   // [..][e / 2] holds elements e, e + 1: folded with v_pk_fma_f32 (two exact fmas)
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 2; e++) run[i][j][e] = v2f{0.0f, 0.0f};

    // fragment read offsets inside a stage
    int fa_off[NI][2], fb_off[2][2];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int row = wm * (16 * NI) + i * 16 + r;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int ch = INT4 ? 2 * q + s : 4 * s + q;
            fa_off[i][s] = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int row = wn * 32 + j * 16 + r;
        if constexpr (INT4) {
            fb_off[j][0] = 16384 + row * 64 + ((q ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
            fb_off[j][1] = fb_off[j][0];
        } else {
#pragma unroll
            for (int s = 0; s < 2; s++) fb_off[j][s] = 16384 + row * 128 + (((4 * s + q) ^ ((row >> 1) & 7)) << 4);
        }
    }

    // Each K step has two phases separated by barriers: LOAD (read this wave's fragments of tile
    // kt from LDS into registers) and MATH (16 MFMAs on registers only, the LDS-DMA requests of
    // tile kt+3 issued under them, then the fp32 rescale).  The two waves that share a SIMD (w and w+4, i.e. the two m-halves)
    // run one barrier apart, so one is in LOAD while the other is in MATH: the MFMA pipe is fed
    // by one wave at a time while the other's LDS reads and LDS-DMA requests issue underneath.
    // Barrier b: half 0 enters LOAD(kt) at b = 2kt and MATH(kt) at 2kt+1, half 1 one later.
    //  - tile kt is read after barrier 2kt (half 0) / 2kt+1 (half 1); every wave has waited for
    //    its own pieces of tile kt before it arrives at barrier 2kt (half 0: end of MATH(kt-1),
    //    half 1: end of LOAD(kt-1));
    //  - tile kt+3 reuses the stage of tile kt-1, whose last reader (half 1, LOAD(kt-1)) drained
    //    its reads before barrier 2kt, and is requested in MATH(kt), after barrier 2kt+1.
    v4i fa[NI][2], fb[2][2];
    float sc[2];
    auto load_phase = [&](int kt) {
        const uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if constexpr (INT4) {
                const v4i x = *reinterpret_cast<const v4i*>(st + fb_off[j][0]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    fb[j][0][e] = (int)(((uint32_t)x[e] & 0x0F0F0F0Fu) << 4);
                    fb[j][1][e] = (int)((uint32_t)x[e] & 0xF0F0F0F0u);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; s++) fb[j][s] = *reinterpret_cast<const v4i*>(st + fb_off[j][s]);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const v4i c0 = *reinterpret_cast<const v4i*>(st + fa_off[i][0]);
            const v4i c1 = *reinterpret_cast<const v4i*>(st + fa_off[i][1]);
            if constexpr (INT4) {   // bytes 32q..32q+31 of the group: even bytes -> s = 0, odd -> s = 1
                fa[i][0] = v4i{(int)__builtin_amdgcn_perm((uint32_t)c0[1], (uint32_t)c0[0], 0x06040200u),
                               (int)__builtin_amdgcn_perm((uint32_t)c0[3], (uint32_t)c0[2], 0x06040200u),
                               (int)__builtin_amdgcn_perm((uint32_t)c1[1], (uint32_t)c1[0], 0x06040200u),
                               (int)__builtin_amdgcn_perm((uint32_t)c1[3], (uint32_t)c1[2], 0x06040200u)};
                fa[i][1] = v4i{(int)__builtin_amdgcn_perm((uint32_t)c0[1], (uint32_t)c0[0], 0x07050301u),
                               (int)__builtin_amdgcn_perm((uint32_t)c0[3], (uint32_t)c0[2], 0x07050301u),
                               (int)__builtin_amdgcn_perm((uint32_t)c1[1], (uint32_t)c1[0], 0x07050301u),
                               (int)__builtin_amdgcn_perm((uint32_t)c1[3], (uint32_t)c1[2], 0x07050301u)};
            } else {
                fa[i][0] = c0;
                fa[i][1] = c1;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float sv = h2f(sc16[kt * GM_BN + wn * 32 + j * 16 + r]);
            sc[j] = INT4 ? sv * 0.0625f : sv;
        }
    };
    auto math_phase = [&](int kt) {
        v4i acc[NI][2];
        __builtin_amdgcn_s_setprio(1);   // the wave feeding the MFMA pipe goes first on its SIMD
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i][0], fb[j][0], v4i{0, 0, 0, 0}, 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
            }
        // tile kt+3 is requested here, under the MFMAs: its stage (tile kt-1) was released at the
        // barrier that opened this phase at the latest
        __builtin_amdgcn_s_setprio(0);
        if (kt + 3 < G) issue_tile(kt + 3);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const v2f a2 = v2f{(float)acc[i][j][2 * e], (float)acc[i][j][2 * e + 1]};
                    run[i][j][e] = __builtin_elementwise_fma(a2, v2f{sc[j], sc[j]}, run[i][j][e]);
                }
    };
    // this wave's pieces of tile j have landed; `ahead` later tiles (as many as exist) stay in flight
    auto wait_tile = [&](int j, int ahead = 2) {
        if (j >= G) return;
        const int later = min(ahead, G - 1 - j);
        // requests per tile and wave: NI/2 for A, 2 (INT8) or 1 (INT4) for B
        if (later == 2) {
            if constexpr (INT4 && NI == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (INT4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (later == 1) {
            if constexpr (INT4 && NI == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if constexpr (INT4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    issue_tile(0);
    if (G > 1) issue_tile(1);
    if (G > 2) issue_tile(2);
    // scale panel -> LDS, transposed to [group][column], behind the first tile requests (its plain
    // loads make the compiler drain them too, which the wait below would do anyway)
    if ((G & 7) == 0 && (reinterpret_cast<uintptr_t>(B_scales) & 15) == 0) {
        for (int idx = t; idx < GM_BN * (G >> 3); idx += 512) {      // 8 consecutive groups of one column
            const int n = idx / (G >> 3), g8 = (idx - n * (G >> 3)) * 8;
            const v4i v = *reinterpret_cast<const v4i*>(B_scales + (size_t)min(n0 + n, N - 1) * G + g8);
#pragma unroll
            for (int e = 0; e < 8; e++)
                sc16[(g8 + e) * GM_BN + n] = (uint16_t)(((uint32_t)v[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
        }
    } else {
        for (int idx = t; idx < GM_BN * G; idx += 512) {
            const int n = idx / G, g = idx - n * G;
            sc16[g * GM_BN + n] = B_scales[(size_t)min(n0 + n, N - 1) * G + g];
        }
    }
    wait_tile(0);
    barrier();   // scale panel and tile 0 visible to every wave
#ifdef GEMM_TRACE
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tr_start = __builtin_readcyclecounter();
#endif
    if constexpr (!INT4) {
        // INT8: one software-pipelined stream per wave.  A K step is two 64-deep slabs; while the 8
        // MFMAs of one slab run, the wave's LDS reads of the next slab are in flight and the fp32
        // rescale of the PREVIOUS group's accumulators (a second register set) issues in the MFMA
        // shadows, so neither the LDS latency nor the VALU work sits between MFMAs.  One barrier
        // per K step, between the slabs: before it every wave has waited for its own pieces of
        // tile kt+1 and has consumed every fragment of tile kt-1, so after it tile kt+1 is
        // readable and the stage of tile kt-1 can take the requests of tile kt+3.
        v4i f0a[NI], f0b[2], f1a[NI], f1b[2];
        auto read_slab = [&](int kt, int sl, v4i (&xa)[NI], v4i (&xb)[2]) {
            const uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
#pragma unroll
            for (int j = 0; j < 2; j++) xb[j] = *reinterpret_cast<const v4i*>(st + fb_off[j][sl]);
#pragma unroll
            for (int i = 0; i < NI; i++) xa[i] = *reinterpret_cast<const v4i*>(st + fa_off[i][sl]);
        };
        // plain v_fma_f32 on purpose: beside MFMAs a packed fp32 fma costs far more than the two
        // scalar ones it replaces (this file is built without SLP vectorization for the same reason)
        auto rescale_half = [&](int half, v4i (&ap)[NI][2], const float (&sp)[2]) {
#pragma unroll
            for (int i = (NI * half) / 2; i < (NI * (half + 1)) / 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        run[i][j][e >> 1][e & 1] = __builtin_fmaf((float)ap[i][j][e], sp[j], run[i][j][e >> 1][e & 1]);
        };
        // Issue order inside a half step, pinned with sched_barriers (nothing is scheduled across
        // one): the LDS-DMA requests, the first MFMA (its operands were requested half a step ago,
        // so the wait in front of it is short), the LDS reads of the next slab, then the other 7
        // MFMAs with 4 VALU of the rescale (2 conversions, 2 fmas) behind each.
        auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
        auto shadow = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
            for (int x = 0; x < 2 * NI - 1; x++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            fence();
        };
        // one K step: MFMAs into `ac`, rescale of `ap` (the previous group, raw fp16 scales `sp`);
        // `steady`: tiles kt+1..kt+3 exist, so the body is one branch-free block
        auto step = [&](auto steady, int kt, v4i (&ac)[NI][2], v4i (&ap)[NI][2], const unsigned (&sp)[2], unsigned (&sn)[2]) {
            constexpr bool ST = decltype(steady)::value;
            fence();
            ac[0][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f0a[0], f0b[0], v4i{0, 0, 0, 0}, 0, 0, 0);
            fence();
            if (!(G2_TIMING & 8)) read_slab(kt, 1, f1a, f1b);
#pragma unroll
            for (int j = 0; j < 2; j++) sn[j] = sc16[kt * GM_BN + wn * 32 + j * 16 + r];
            fence();
            const float spf[2] = {h2f((uint16_t)sp[0]), h2f((uint16_t)sp[1])};
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    if ((i | j) && !(G2_TIMING & 4)) ac[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f0a[i], f0b[j], v4i{0, 0, 0, 0}, 0, 0, 0);
            if (!(G2_TIMING & 2)) rescale_half(0, ap, spf);
            shadow();
            if constexpr (G2_TIMING & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (ST && NI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (ST) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else wait_tile(kt + 1, 1);
            if (!(G2_TIMING & 16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            fence();
            if (!(G2_TIMING & 1) && (ST || kt + 3 < G)) issue_tile(kt + 3);
            fence();
            ac[0][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f1a[0], f1b[0], ac[0][0], 0, 0, 0);
            fence();
            if (!(G2_TIMING & 8) && (ST || kt + 1 < G)) read_slab(kt + 1, 0, f0a, f0b);
            fence();
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    if ((i | j) && !(G2_TIMING & 4)) ac[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f1a[i], f1b[j], ac[i][j], 0, 0, 0);
            if (!(G2_TIMING & 2)) rescale_half(1, ap, spf);
            shadow();
        };
        v4i accx[NI][2], accy[NI][2];
        unsigned scx[2] = {0, 0}, scy[2] = {0, 0};   // raw fp16 scales, widened when used
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) accy[i][j] = v4i{0, 0, 0, 0};
        read_slab(0, 0, f0a, f0b);
        int kt = 0;
        for (; kt + 4 < G; kt += 2) {
            step(std::true_type{}, kt, accx, accy, scy, scx);       // first pass: accy = 0, scy = 0 leaves `run` at +0
            step(std::true_type{}, kt + 1, accy, accx, scx, scy);
        }
        for (; kt + 1 < G; kt += 2) {
            step(std::false_type{}, kt, accx, accy, scy, scx);
            step(std::false_type{}, kt + 1, accy, accx, scx, scy);
        }
        if (kt < G) {
            step(std::false_type{}, kt, accx, accy, scy, scx);
            const float lf[2] = {h2f((uint16_t)scx[0]), h2f((uint16_t)scx[1])};
            rescale_half(0, accx, lf);
            rescale_half(1, accx, lf);
        } else {
            const float lf[2] = {h2f((uint16_t)scy[0]), h2f((uint16_t)scy[1])};
            rescale_half(0, accy, lf);
            rescale_half(1, accy, lf);
        }
    } else {
        if (wm == 1) barrier();                 // half 1 runs one barrier behind
        for (int kt = 0; kt < G; kt++) {
            G2_T(t0);
            barrier();
            G2_T(t1);
            G2_T(t2);
            load_phase(kt);
            if (wm == 1) wait_tile(kt + 1, 1);   // half 1 has requested up to tile kt+2 so far
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G2_T(t3);
            barrier();
            G2_T(t4);
            math_phase(kt);
            if (wm == 0) wait_tile(kt + 1, 2);
    #ifdef GEMM_TRACE
            asm volatile("s_nop 0" ::: "memory");
            const unsigned long long t5 = __builtin_readcyclecounter();
            tr[0] += t1 - t0; tr[1] += t2 - t1; tr[2] += t3 - t2; tr[3] += t4 - t3; tr[4] += t5 - t4;
    #endif
        }
    #ifdef GEMM_TRACE
        if (lane == 0 && wg < 8) {
            const int slot = wg * 8 + w;
            g_gemm_trace[slot * 8 + 0] = __builtin_readcyclecounter() - tr_start;
            g_gemm_trace[slot * 8 + 6] = tr_start - tr_entry;
            g_gemm_trace[slot * 8 + 7] = __builtin_readcyclecounter();   // loop end (absolute)
            for (int i = 0; i < 5; i++) g_gemm_trace[slot * 8 + 1 + i] = tr[i];
        }
    #endif
        if (wm == 0) barrier();                 // same barrier count for both halves
    }

    // Output: half(running * A_scale[m]).  The lane layout of the accumulators (4 consecutive rows
    // of one column) would give 2-byte stores; the tile goes through LDS (the ring is idle now)
    // and leaves as 16-byte row segments.
    const bool vec_store = (N & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    if (vec_store) {
        constexpr int CS = GM_BN + 8;   // halves per LDS row (16-byte aligned, rows 4 banks apart)
        uint16_t* ct = reinterpret_cast<uint16_t*>(g2_lds);
        barrier();                      // every wave is done with the ring
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int lm = wm * (16 * NI) + i * 16 + q * 4 + e;
                const float as = A_scales[min(m0 + lm, M - 1)];
#pragma unroll
                for (int j = 0; j < 2; j++)
                    ct[lm * CS + wn * 32 + j * 16 + r] = f2h_mul(run[i][j][e >> 1][e & 1], as);
            }
        barrier();
#pragma unroll
        for (int p = 0; p < NI; p++) {
            const int idx = p * 512 + t, lm = idx >> 4, c8 = (idx & 15) * 8;
            const int m = m0 + lm, n = n0 + c8;
            if (m < M && n < N)   // N % 8 == 0: a segment is inside or outside as a whole
                *reinterpret_cast<v4i*>(C + (size_t)m * N + n) = *reinterpret_cast<const v4i*>(ct + lm * CS + c8);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int m = m0 + wm * (16 * NI) + i * 16 + q * 4 + e;
                if (m >= M) continue;
                const float as = A_scales[m];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int n = n0 + wn * 32 + j * 16 + r;
                    if (n < N) C[(size_t)m * N + n] = f2h_mul(run[i][j][e >> 1][e & 1], as);
                }
            }
    }
#ifdef GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && wg < 8) {
        const int slot = wg * 8 + w;
        g_gemm_trace[slot * 8 + 7] = __builtin_readcyclecounter() - g_gemm_trace[slot * 8 + 7];
    }
#endif
}

// more than 64 KB of dynamic LDS has to be allowed once per device
template <bool INT4, int NI>
static hipError_t gemm2_allow_lds() {
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2_kernel<INT4, NI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    return hipSuccess;
}

// 32-row tiles when even 64-row tiles would leave more than half of the CUs without a workgroup
static bool gemm2_tiny_tiles(size_t M, size_t nbn) {
    return ((M + 63) / 64) * nbn < 128;
}

// 64-row tiles when 128-row tiles would not give every CU a workgroup
static bool gemm2_small_tiles(size_t M, size_t nbn) {
#ifdef G2_NO_SMALL_TILES
    return false;
#endif
    return ((M + GM_BM - 1) / GM_BM) * nbn < 256;
}

template <bool INT4>
static hipError_t launch_gemm2(const int8_t* A, const float* A_scales, const void* B, const uint16_t* B_scales,
                               uint16_t* C, size_t M, size_t K, size_t N, int G, hipStream_t st) {
    const size_t lds = (size_t)G2_STAGES * G2_STAGE + (size_t)G * GM_BN * sizeof(uint16_t);
    const int nbn = (int)((N + GM_BN - 1) / GM_BN);
    GemmBatch none;
    none.count = 0;
    if (gemm2_tiny_tiles(M, (size_t)nbn)) {
        hipError_t et = gemm2_allow_lds<INT4, 1>();
        if (et != hipSuccess) return et;
        const int nbm = (int)((M + 31) / 32), nwg = nbm * nbn;
        gemm2_kernel<INT4, 1><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(A, A_scales, (const uint8_t*)B, B_scales,
                                                                          C, (int)M, (int)K, (int)N, G, nbm, nwg, none);
        return hipGetLastError();
    }
    if (gemm2_small_tiles(M, (size_t)nbn)) {
        hipError_t es = gemm2_allow_lds<INT4, 2>();
        if (es != hipSuccess) return es;
        const int nbm = (int)((M + 63) / 64), nwg = nbm * nbn;
        gemm2_kernel<INT4, 2><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(A, A_scales, (const uint8_t*)B, B_scales,
                                                                          C, (int)M, (int)K, (int)N, G, nbm, nwg, none);
        return hipGetLastError();
    }
    hipError_t ea = gemm2_allow_lds<INT4, 4>();
    if (ea != hipSuccess) return ea;
    const int nbm = (int)((M + GM_BM - 1) / GM_BM);
    const int nwg = nbm * nbn;
    gemm2_kernel<INT4, 4><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(A, A_scales, (const uint8_t*)B, B_scales, C,
                                                                      (int)M, (int)K, (int)N, G, nbm, nwg, none);
    return hipGetLastError();
}

template <bool INT4>
static hipError_t launch_gemm2_batch(const int8_t* A, const float* A_scales, const GemmProblem* pr, size_t count,
                                     size_t M, size_t K, int G, hipStream_t st) {
    const size_t lds = (size_t)G2_STAGES * G2_STAGE + (size_t)G * GM_BN * sizeof(uint16_t);
    GemmBatch bt;
    memset(&bt, 0, sizeof bt);
    bt.count = (int)count;
    size_t nbn_all = 0;
    for (size_t i = 0; i < count; i++) nbn_all += (pr[i].N + GM_BN - 1) / GM_BN;
    const bool small = gemm2_small_tiles(M, nbn_all);
    const bool tiny = gemm2_tiny_tiles(M, nbn_all);
    hipError_t e = small ? gemm2_allow_lds<INT4, 2>() : gemm2_allow_lds<INT4, 4>();
    if (e != hipSuccess) return e;
    if (tiny && (e = gemm2_allow_lds<INT4, 1>()) != hipSuccess) return e;
    const int nbm = tiny ? (int)((M + 31) / 32) : small ? (int)((M + 63) / 64) : (int)((M + GM_BM - 1) / GM_BM);
    size_t nbn = 0;
    for (size_t i = 0; i < count; i++) {
        nbn += (pr[i].N + GM_BN - 1) / GM_BN;
        if (nbn * nbm >= (size_t(1) << 30)) return hipErrorInvalidValue;
        bt.bn_end[i] = (int)nbn;
        bt.B[i] = static_cast<const uint8_t*>(pr[i].B);
        bt.Bs[i] = pr[i].Bs;
        bt.C[i] = pr[i].C;
        bt.N[i] = (int)pr[i].N;
    }
    const int nwg = nbm * (int)nbn;
    if (nwg == 0) return hipSuccess;
    if (tiny) {
        gemm2_kernel<INT4, 1><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(A, A_scales, nullptr, nullptr, nullptr,
                                                                          (int)M, (int)K, 0, G, nbm, nwg, bt);
        return hipGetLastError();
    }
    if (small) {
        gemm2_kernel<INT4, 2><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(A, A_scales, nullptr, nullptr, nullptr,
                                                                          (int)M, (int)K, 0, G, nbm, nwg, bt);
        return hipGetLastError();
    }
    gemm2_kernel<INT4, 4><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(A, A_scales, nullptr, nullptr, nullptr, (int)M,
                                                                      (int)K, 0, G, nbm, nwg, bt);
    return hipGetLastError();
}

#ifdef GEMM_TRACE
extern "C" int cactus_hip_debug_gemm_trace(unsigned long long* out512) {
    return (int)hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 512);
}
#endif

// Up to 8 weight matrices against one quantized activation matrix, one launch.  false: the shape
// is outside what the pipelined kernel takes (the caller launches them one by one).
bool gemm_batch_ok(size_t count, size_t K) {
    const size_t G = K / 128;
    return count >= 1 && count <= (size_t)G2_MAXB && K % 128 == 0 &&
           (size_t)G2_STAGES * G2_STAGE + G * GM_BN * 2 <= 160 * 1024;
}
hipError_t launch_matmul_gemm_batch(bool int4, const int8_t* A, const float* A_scales, const GemmProblem* pr,
                                    size_t count, size_t M, size_t K, hipStream_t st) {
    if (!gemm_batch_ok(count, K)) return hipErrorInvalidValue;
    const int G = (int)(K / 128);
    return int4 ? launch_gemm2_batch<true>(A, A_scales, pr, count, M, K, G, st)
                : launch_gemm2_batch<false>(A, A_scales, pr, count, M, K, G, st);
}

hipError_t launch_matmul_gemm(bool int4, const int8_t* A, const float* A_scales, const void* B,
                              const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                              size_t K, size_t N, hipStream_t st) {
    const int G = (int)(K / 128);
    if (gsums == nullptr && C != nullptr && (size_t)G2_STAGES * G2_STAGE + (size_t)G * GM_BN * 2 <= 160 * 1024 &&
        ((M + GM_BM - 1) / GM_BM) * ((N + GM_BN - 1) / GM_BN) < (size_t(1) << 30)) {
        return int4 ? launch_gemm2<true>(A, A_scales, B, B_scales, C, M, K, N, G, st)
                    : launch_gemm2<false>(A, A_scales, B, B_scales, C, M, K, N, G, st);
    }
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
