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
// A workgroup owns a (32*NI) x 128 output tile and walks K one quantization group (128 values) per step.
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR staging) into a ring of four 32 KB
// stages (activations 16 KB + weights 16 KB), ordered by counted s_waitcnt vmcnt and ONE raw s_barrier per
// step (no vmcnt(0) in the loop).  The LDS image of a tile is lane-linear per request, so the bank swizzle
// (16-byte slot XOR (row >> 1) & 7 on 128-byte rows: conflict-free for ds_read_b128's lane groups) is
// applied to the per-lane SOURCE address and again when fragments are read.  The weight-scale panel of the
// tile (128 columns x G groups, fp16) is staged once, transposed to [group][column].  Workgroup ids are
// remapped so that the m-tiles sharing one weight panel run on one XCD (one L2) back to back.
//
// 768 threads = 8 compute waves (2 (m) x 4 (n), 16*NI x 32 outputs each) + 4 loader waves, one per SIMD.
// Why the roles: an LDS-DMA request holds the issuing wave for 60-180 cycles, and when every wave both
// loads and computes, the two waves of a SIMD stall on their requests at the same point of the step (right
// behind the barrier), serial with the MFMA / rescale stream.  A loader wave does nothing but request
// tiles: its stalls sit in the memory pipe while the SIMD's two compute waves own the vector issue slot.
// Loader p requests its share of the pieces (8 rows x 128 B) of both tiles, waits (counted vmcnt) for its
// pieces of tile kt+1, meets the compute waves at barrier kt and then requests tile kt+4 into the stage of
// tile kt, whose last reads every compute wave completed (lgkmcnt(0)) before that barrier: a tile has
// three whole steps to arrive.
//
// Compute waves keep ONE accumulator set.  A step is two 64-deep slabs; in the first, every 16x16 tile
// converts its finished group sum (the previous group) into temporaries, starts the new group in the
// same registers (MFMA with C = 0) and folds the temporaries into the fp32 running sum,
//   running = fma(float(acc), float(B_scale[n,g]), running)     (kernel_gemm.cpp:311-316),
// the second slab is MFMAs only.  The matrix pipe and the vector ALU share a SIMD's issue slot on this
// part (a 16x16x64 INT8 MFMA holds it for about 8 cycles, a VALU instruction for about 4.7), so nothing is
// gained by a second accumulator set that would let the rescale trail the MFMAs, and the registers saved
// (138 instead of 204) keep three waves on a SIMD.
//
// Packed INT4 weights: see the loader's INT4 branch -- they are unpacked once per tile by the loaders into
// the INT8 image (values x 16, the exact 1/16 folded into the scale), so the compute loop is the same.
constexpr int G2_STAGE = 32768;     // bytes per ring stage (A 16 KB + B 16 KB)
constexpr int G2_STAGES = 4;

// Benchmarks only (-DGEMM_TRACE): per-phase shader-clock totals of the compute waves of the first 8 tiles.
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

// NI: 16-row MFMA tiles per compute wave along m (4: 128-row workgroup tile, 2: 64-row tile for launches
// whose 128-row tiles would leave most of the chip idle, 1: 32-row tile when 64-row tiles still number
// fewer than half the CUs -- a workgroup then re-reads its weight tile for a quarter of the rows, from L2).
// Timing experiments only (results are wrong when nonzero): bit 0 no tile requests after the first four,
// bit 1 no rescale, bit 2 no MFMAs, bit 3 no fragment reads in the loop.
#ifndef G3_TIMING
#define G3_TIMING 0
#endif
#ifndef G3_LOADERS
#define G3_LOADERS 4
#endif
constexpr int G3_LW = G3_LOADERS;                 // loader waves per workgroup
constexpr int G3_THREADS = 512 + 64 * G3_LW;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NJ: 16-column MFMA tiles per compute wave along n (2: 128-column workgroup tile; 1: 64 columns, for launches of
// few rows whose 128-column tiles would leave most CUs without a workgroup -- the weight stream of a small batch is
// bound by how many CUs pull on it).
// F16: both operands are binary16 (cactus_matmul_f16, kernel_gemm.cpp:17-126, with fp32 accumulation on the matrix
// cores): a 128-byte tile row is 64 halves, the ring, the swizzle and the fragment chunks are the INT8 kernel's, K and
// the pointers are in bytes, there are no scales, and the loop is MFMAs (v_mfma_f32_16x16x32_f16) and reads only.
template <bool INT4, int NI, int NJ, bool F16 = false>
__global__ __launch_bounds__(G3_THREADS) void gemm3_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ A_scales, const uint8_t* __restrict__ B_arg,
    const uint16_t* __restrict__ B_scales_arg, uint16_t* __restrict__ C_arg, int M, int K, int N_arg, int G, int nbm,
    int nwg, const GemmBatch bt) {
    const uint8_t* __restrict__ B = B_arg;
    const uint16_t* __restrict__ B_scales = B_scales_arg;
    uint16_t* __restrict__ C = C_arg;
    int N = N_arg;
    extern __shared__ __attribute__((aligned(16))) uint8_t g2_lds[];
    uint16_t* sc16 = reinterpret_cast<uint16_t*>(g2_lds + G2_STAGES * G2_STAGE);   // [G][128]

    const int t = threadIdx.x;
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);

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
    constexpr int BM = 32 * NI, BN = 64 * NJ;
    const int m0 = bm * BM, n0 = bn * BN;

    auto barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // scale panel -> LDS, transposed to [group][column]; every wave helps
    auto stage_scales = [&]() {
        if ((G & 7) == 0 && (reinterpret_cast<uintptr_t>(B_scales) & 15) == 0) {
            for (int idx = t; idx < BN * (G >> 3); idx += G3_THREADS) {
                const int n = idx / (G >> 3), g8 = (idx - n * (G >> 3)) * 8;
                const v4i v = *reinterpret_cast<const v4i*>(B_scales + (size_t)min(n0 + n, N - 1) * G + g8);
#pragma unroll
                for (int e = 0; e < 8; e++)
                    sc16[(g8 + e) * BN + n] = (uint16_t)(((uint32_t)v[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
            }
        } else {
            for (int idx = t; idx < BN * G; idx += G3_THREADS) {
                const int n = idx / G, g = idx - n * G;
                sc16[g * BN + n] = B_scales[(size_t)min(n0 + n, N - 1) * G + g];
            }
        }
    };
    const bool vec_store = (N & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;

    if (w >= 8) {
        // ---- loader wave ----
        const int p = w - 8;
        constexpr int PA = 4 * NI / G3_LW, PB = BN / 8 / G3_LW;   // pieces (8 rows x 128 B) of the two tiles per loader
        constexpr int R = PA + PB;                            // requests per tile and loader wave
        const int8_t* a_base = A + (size_t)m0 * K;
        unsigned a_src[PA];
        int a_dst[PA];
#pragma unroll
        for (int j = 0; j < PA; j++) {
            const int pi = p * PA + j, row = pi * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
            a_src[j] = (unsigned)min(row, M - 1 - m0) * (unsigned)K + ch * 16;   // rows past M: clamped, never stored
            a_dst[j] = pi * 1024;
        }
        auto issue_a = [&](int kt) {
            uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
            const int8_t* ak = scalar_ptr(a_base + (size_t)kt * 128);
#pragma unroll
            for (int j = 0; j < PA; j++)
                __builtin_amdgcn_global_load_lds((gptr_t)(ak + a_src[j]), (lptr_t)(st + a_dst[j]), 16, 0, 0);
        };
        if constexpr (INT4) {
            // Packed weights are unpacked HERE, once per tile, instead of by every compute wave on every
            // fragment.  A loader owns a quarter of the tile's weight rows: it requests their packed bytes (pieces of
            // 16 rows x 64 bytes) by LDS-DMA into the upper half of the 2 KB a piece's rows take up unpacked, and when
            // they have landed reads its own 16 bytes per lane back, turns each packed word into two words of
            // INT8 values x 16 in k order (nibble to the top of its byte: the sign comes for free and the exact
            // 1/16 goes into the scale) and stores them where an INT8 tile would lie.  Only the loader itself
            // touches that memory before the barrier, and a wave's LDS operations execute in order, so the
            // overlap of a piece's staging bytes with its unpacked rows is harmless.  The compute waves run the
            // INT8 loop unchanged.  The LDS accesses are inline asm: beside LDS-DMA the compiler puts a
            // vmcnt(0) in front of LDS accesses it can see, which would drain the three tiles in flight.
            constexpr int PQ = BN / 16 / G3_LW;  // packed pieces (16 rows x 64 B) per tile and loader
            constexpr int R = PA + PQ;          // requests per tile and loader wave
            const uint8_t* b_base = B + (size_t)n0 * (size_t)(K / 2);
            unsigned q_src[PQ];
            int q_tmp[PQ], q_dst[PQ][2];
#pragma unroll
            for (int j = 0; j < PQ; j++) {
                // eight consecutive lanes (one cycle of a 16-byte LDS store) take rows r and r ^ 2, whose swizzles
                // differ in the low bit: their eight 16-byte slots fall on distinct banks of the 128-byte store row
                const int l4 = lane >> 2, rsel = (l4 & ~3) | ((l4 & 1) << 1) | ((l4 >> 1) & 1);
                const int pi = p * PQ + j, row = pi * 16 + rsel, c4 = lane & 3, sw = (row >> 1) & 7;
                q_src[j] = (unsigned)min(row, N - 1 - n0) * (unsigned)(K / 2) + c4 * 16;
                q_tmp[j] = 16384 + pi * 2048 + 1024;
                q_dst[j][0] = 16384 + row * 128 + (((2 * c4) ^ sw) << 4);
                q_dst[j][1] = 16384 + row * 128 + (((2 * c4 + 1) ^ sw) << 4);
            }
            const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lptr_t)g2_lds);   // LDS byte address of the ring
            auto issue_tile = [&](int kt) {
                uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
                const uint8_t* bk = scalar_ptr(b_base + (size_t)kt * 64);
                issue_a(kt);
#pragma unroll
                for (int j = 0; j < PQ; j++)
                    __builtin_amdgcn_global_load_lds((gptr_t)(bk + q_src[j]), (lptr_t)(st + q_tmp[j]), 16, 0, 0);
            };
            auto wait_later = [&](int later) {
                if (later >= 3) wait_vmcnt<3 * R>();
                else if (later == 2) wait_vmcnt<2 * R>();
                else if (later == 1) wait_vmcnt<R>();
                else wait_vmcnt<0>();
            };
            auto unpack_tile = [&](int kt) {   // this wave's packed pieces of tile kt have landed
                const unsigned st = lds0 + (unsigned)((kt % G2_STAGES) * G2_STAGE);
                v4i x[PQ];
#pragma unroll
                for (int j = 0; j < PQ; j++) {
                    const unsigned a = st + (unsigned)q_tmp[j] + (unsigned)lane * 16u;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(x[j]) : "v"(a) : "memory");
                }
                if constexpr (PQ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1])::"memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0])::"memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < PQ; j++) {
                    v4i o[2];
#pragma unroll
                    for (int i = 0; i < 4; i++) {   // packed byte b = elements 2b (low nibble), 2b + 1 (high)
                        const uint32_t v = (uint32_t)x[j][i];
                        const uint32_t lo = (v & 0x0F0F0F0Fu) << 4, hi = v & 0xF0F0F0F0u;
                        o[i >> 1][2 * (i & 1)] = (int)__builtin_amdgcn_perm(hi, lo, 0x05010400u);
                        o[i >> 1][2 * (i & 1) + 1] = (int)__builtin_amdgcn_perm(hi, lo, 0x07030602u);
                    }
                    const unsigned d0 = st + (unsigned)q_dst[j][0], d1 = st + (unsigned)q_dst[j][1];
                    asm volatile("ds_write_b128 %0, %1" ::"v"(d0), "v"(o[0]) : "memory");
                    asm volatile("ds_write_b128 %0, %1" ::"v"(d1), "v"(o[1]) : "memory");
                }
            };
            static_assert(PQ == 1 || PQ == 2, "loader count");
            __builtin_amdgcn_s_setprio(3);
            stage_scales();   // first: its ordinary loads are waited for with vmcnt(0)
            issue_tile(0);
            if (G > 1) issue_tile(1);
            if (G > 2) issue_tile(2);
            if (G > 3) issue_tile(3);
            wait_later(min(3, G - 1));
            unpack_tile(0);
            barrier();
            int kt = 0;
            for (; kt + 4 < G; kt++) {          // tiles kt+1 .. kt+3 are outstanding, tile kt+4 follows
                wait_vmcnt<2 * R>();
                unpack_tile(kt + 1);
                barrier();
                issue_tile(kt + 4);             // into the stage of tile kt
            }
            for (; kt < G; kt++) {
                if (kt + 1 < G) {
                    wait_later(G - 2 - kt);
                    unpack_tile(kt + 1);
                }
                barrier();
            }
            if (vec_store) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
        const uint8_t* b_base = B + (size_t)n0 * K;
        unsigned b_src[PB];
        int b_dst[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int pi = p * PB + j, row = pi * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
            b_src[j] = (unsigned)min(row, N - 1 - n0) * (unsigned)K + ch * 16;
            b_dst[j] = 16384 + pi * 1024;
        }
        auto issue_tile = [&](int kt) {
            uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
            const uint8_t* bk = scalar_ptr(b_base + (size_t)kt * 128);
            issue_a(kt);
#pragma unroll
            for (int j = 0; j < PB; j++)
                __builtin_amdgcn_global_load_lds((gptr_t)(bk + b_src[j]), (lptr_t)(st + b_dst[j]), 16, 0, 0);
        };
        // this wave's pieces of a tile have landed when at most `later` younger tiles are outstanding
        auto wait_later = [&](int later) {
            if (later >= 3) wait_vmcnt<3 * R>();
            else if (later == 2) wait_vmcnt<2 * R>();
            else if (later == 1) wait_vmcnt<R>();
            else wait_vmcnt<0>();
        };
        __builtin_amdgcn_s_setprio(3);
        issue_tile(0);
        if (G > 1) issue_tile(1);
        if (G > 2) issue_tile(2);
        if (G > 3) issue_tile(3);
        if constexpr (!F16) stage_scales();
        wait_later(min(3, G - 1));
        barrier();
        int kt = 0;
        for (; kt + 4 < G; kt++) {          // tiles kt+1 .. kt+3 are outstanding, tile kt+4 follows
            if (!(G3_TIMING & 1)) wait_vmcnt<2 * R>();
            __builtin_amdgcn_s_barrier();
            if (!(G3_TIMING & 1)) issue_tile(kt + 4);   // into the stage of tile kt
        }
        for (; kt < G; kt++) {
            if (kt + 1 < G) wait_later(G - 2 - kt);
            __builtin_amdgcn_s_barrier();
        }
        if (vec_store) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---- compute wave ----
    const int wm = w >> 2, wn = w & 3;
    const int r = lane & 15, q = lane >> 4;
    int fa_off[NI][2], fb_off[NJ][2];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int row = wm * (16 * NI) + i * 16 + r;
#pragma unroll
        for (int sl = 0; sl < 2; sl++) fa_off[i][sl] = row * 128 + (((4 * sl + q) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int row = wn * (16 * NJ) + j * 16 + r;
#pragma unroll
        for (int sl = 0; sl < 2; sl++) fb_off[j][sl] = 16384 + row * 128 + (((4 * sl + q) ^ ((row >> 1) & 7)) << 4);
    }
    float run[NI][NJ][4];
    v4i acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            acc[i][j] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; e++) run[i][j][e] = 0.0f;
        }
    v4i f0a[NI], f0b[NJ], f1a[NI], f1b[NJ];   // fragments of the first / second slab of a step
    // fragment read number x of a slab: the weight fragments first, then the activation fragments
    auto read_one = [&](const uint8_t* st, int sl, int x, v4i (&xa)[NI], v4i (&xb)[NJ]) {
        if (x < NJ) xb[x] = *reinterpret_cast<const v4i*>(st + fb_off[x][sl]);
        else if (x < NJ + NI) xa[x - NJ] = *reinterpret_cast<const v4i*>(st + fa_off[x - NJ][sl]);
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    constexpr int NT = NJ * NI;                      // MFMAs per slab
    constexpr int RPS = (NJ + NI + NT - 1) / NT;     // fragment reads issued behind each MFMA

    if constexpr (F16) {
        v4f facc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) facc[i][j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        barrier();   // tile 0 visible
#pragma unroll
        for (int x = 0; x < NJ + NI; x++) read_one(g2_lds, 0, x, f0a, f0b);
        for (int kt = 0; kt < G; kt++) {
            const uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
            const uint8_t* stn = g2_lds + ((kt + 1) % G2_STAGES) * G2_STAGE;   // past the last tile: a finished stage, values unused
            fence();
#pragma unroll
            for (int x = 0; x < NT; x++) {
                const int i = x / NJ, j = x % NJ;
                facc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, f0a[i]), __builtin_bit_cast(v8h, f0b[j]),
                                                                    facc[i][j], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < RPS; y++) read_one(st, 1, x * RPS + y, f1a, f1b);
            }
#pragma unroll
            for (int x = 0; x < NT; x++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPS, 0);
            }
            fence();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tile kt are complete
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            fence();
#pragma unroll
            for (int x = 0; x < NT; x++) {
                const int i = x / NJ, j = x % NJ;
                facc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, f1a[i]), __builtin_bit_cast(v8h, f1b[j]),
                                                                    facc[i][j], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < RPS; y++) read_one(stn, 0, x * RPS + y, f0a, f0b);
            }
#pragma unroll
            for (int x = 0; x < NT; x++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPS, 0);
            }
            fence();
        }
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) run[i][j][e] = facc[i][j][e];
    } else {
    stage_scales();
    barrier();   // scale panel and tile 0 visible
#pragma unroll
    for (int x = 0; x < NJ + NI; x++) read_one(g2_lds, 0, x, f0a, f0b);
    float sprev[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) sprev[j] = 0.0f;   // scales of the group whose sums sit in acc (first pass: acc = 0 leaves run at +0)
#ifdef GEMM_TRACE
    unsigned long long tr[3] = {0, 0, 0};
    const unsigned long long tr_c0 = clock64(), tr_r0 = wall_clock64();
#endif
    // The body of a step is one basic block whose issue order is pinned.  First slab: every tile converts its
    // finished group sum, restarts in the same registers (MFMA with C = 0) and folds the converted values
    // into the running sum; second slab: MFMAs only.  The LDS reads of the next slab are spread one behind
    // each MFMA (issued in a burst behind the barrier, the eight waves' reads queue up at the LDS and every
    // wave sits on its last read while the matrix pipe idles).
    for (int kt = 0; kt < G; kt++) {
        const uint8_t* st = g2_lds + (kt % G2_STAGES) * G2_STAGE;
        const uint8_t* stn = g2_lds + ((kt + 1) % G2_STAGES) * G2_STAGE;   // past the last tile: a finished stage, values unused
        G2_T(t0);
        fence();
        unsigned sraw[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) sraw[j] = sc16[kt * BN + wn * (16 * NJ) + j * 16 + r];
#pragma unroll
        for (int x = 0; x < NT; x++) {
            const int i = x / NJ, j = x % NJ;
            float tmp[4];
            if (!(G3_TIMING & 2)) {
#pragma unroll
                for (int e = 0; e < 4; e++) tmp[e] = (float)acc[i][j][e];
            }
            if (!(G3_TIMING & 4) || x == 0)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f0a[i], f0b[j], v4i{0, 0, 0, 0}, 0, 0, 0);
            if (!(G3_TIMING & 8)) {
#pragma unroll
                for (int y = 0; y < RPS; y++) read_one(st, 1, x * RPS + y, f1a, f1b);
            }
            if (!(G3_TIMING & 2)) {
#pragma unroll
                for (int e = 0; e < 4; e++) run[i][j][e] = __builtin_fmaf(tmp[e], sprev[j], run[i][j][e]);
            }
        }
#pragma unroll
        for (int x = 0; x < NT; x++) {
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // conversions of tile x
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // its MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, RPS, 0);   // fragment reads
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // fmas of tile x
        }
        fence();
        G2_T(t1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tile kt are complete
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        G2_T(t2);
        fence();
#pragma unroll
        for (int x = 0; x < NT; x++) {
            const int i = x / NJ, j = x % NJ;
            if (!(G3_TIMING & 4) || x == 0)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f1a[i], f1b[j], acc[i][j], 0, 0, 0);
            if (!(G3_TIMING & 8)) {
#pragma unroll
                for (int y = 0; y < RPS; y++) read_one(stn, 0, x * RPS + y, f0a, f0b);
            }
        }
        // widened once per step behind an empty asm, so that the folds above are plain v_fma_f32 (the
        // compiler otherwise folds the conversion into every fma as v_fma_mix_f32)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            sprev[j] = INT4 ? h2f((uint16_t)sraw[j]) * 0.0625f : h2f((uint16_t)sraw[j]);   // x 1/16: exact
            asm volatile("" : "+v"(sprev[j]));
        }
#pragma unroll
        for (int x = 0; x < NT; x++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RPS, 0);
        }
        fence();
#ifdef GEMM_TRACE
        const unsigned long long t3 = __builtin_readcyclecounter();
        tr[0] += t1 - t0; tr[1] += t2 - t1; tr[2] += t3 - t2;
#endif
    }
#ifdef GEMM_TRACE
    if (lane == 0 && wg < 8) {
        const int slot = wg * 8 + w;
        g_gemm_trace[slot * 8 + 0] = clock64() - tr_c0;
        g_gemm_trace[slot * 8 + 1] = wall_clock64() - tr_r0;
        for (int i = 0; i < 3; i++) g_gemm_trace[slot * 8 + 2 + i] = tr[i];
    }
#endif
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) run[i][j][e] = __builtin_fmaf((float)acc[i][j][e], sprev[j], run[i][j][e]);
    }

    // Output: half(running * A_scale[m]) (F16: half(sum)), through LDS so that it leaves as 16-byte row segments.
    if (vec_store) {
        constexpr int CS = BN + 8;
        uint16_t* ct = reinterpret_cast<uint16_t*>(g2_lds);
        barrier();                      // every wave is done with the ring
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int lm = wm * (16 * NI) + i * 16 + q * 4 + e;
                const float as = F16 ? 1.0f : A_scales[min(m0 + lm, M - 1)];
#pragma unroll
                for (int j = 0; j < NJ; j++) ct[lm * CS + wn * (16 * NJ) + j * 16 + r] = f2h_mul(run[i][j][e], as);
            }
        barrier();
        constexpr int SPR = BN / 8;                     // 16-byte segments per row
#pragma unroll
        for (int p = 0; p < (BM * SPR + 511) / 512; p++) {
            const int idx = p * 512 + t, lm = idx / SPR, c8 = (idx % SPR) * 8;
            const int m = m0 + lm, n = n0 + c8;
            if (lm < BM && m < M && n < N)
                *reinterpret_cast<v4i*>(C + (size_t)m * N + n) = *reinterpret_cast<const v4i*>(ct + lm * CS + c8);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int m = m0 + wm * (16 * NI) + i * 16 + q * 4 + e;
                if (m >= M) continue;
                const float as = F16 ? 1.0f : A_scales[m];
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const int n = n0 + wn * (16 * NJ) + j * 16 + r;
                    if (n < N) C[(size_t)m * N + n] = f2h_mul(run[i][j][e], as);
                }
            }
    }
}

// more than 64 KB of dynamic LDS has to be allowed once per device
template <bool INT4, int NI, int NJ, bool F16 = false>
static hipError_t gemm3_allow_lds() {
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_kernel<INT4, NI, NJ, F16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    return hipSuccess;
}

template <bool INT4, int NI, int NJ, bool F16 = false>
static hipError_t launch_gemm3(const int8_t* A, const float* A_scales, const uint8_t* B, const uint16_t* B_scales,
                               uint16_t* C, int M, int K, int N, int G, int nwg, const GemmBatch& bt, hipStream_t st) {
    hipError_t e = gemm3_allow_lds<INT4, NI, NJ, F16>();
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)G2_STAGES * G2_STAGE + (F16 ? 0 : (size_t)G * (64 * NJ) * sizeof(uint16_t));
    const int nbm = (M + 32 * NI - 1) / (32 * NI);
    gemm3_kernel<INT4, NI, NJ, F16><<<dim3((unsigned)nwg), dim3(G3_THREADS), lds, st>>>(A, A_scales, B, B_scales, C, M, K, N,
                                                                                  G, nbm, nwg, bt);
    return hipGetLastError();
}

// Workgroup tile for M rows against weight matrices of the given widths: among 128x128, 64x128, 64x64, 32x128 and
// 32x64 (rows x columns) the one that gives the most workgroups without exceeding one per CU (between equals the
// smaller tile, then the taller); 128x128 when every tile gives more.  Measured over decode- and prefill-step
// shapes (scripts/tile_sweep.sh): a launch is fastest with as many CUs pulling on the operands as it can get, and
// slower again once the workgroups no longer fit the chip at once (M = 256, K = 1024: N = 4096 7.1 us at 64x128 =
// 128 workgroups, 6.1 at 64x64 = 256, 8.6 at 32x64 = 512; N = 6144 7.5 us at 64x128 = 192, 9.9 at 64x64 = 384).
struct GemmTile { int rows, cols; };
static GemmTile gemm3_tile(size_t M, const size_t* Ns, size_t count) {
#ifdef G3_FORCE_ROWS
    return {G3_FORCE_ROWS, G3_FORCE_COLS};
#endif
    constexpr size_t NCU = 256;
    static const GemmTile cand[5] = {{128, 128}, {64, 128}, {64, 64}, {32, 128}, {32, 64}};
    GemmTile best = cand[0];
    size_t best_wg = 0;
    for (const GemmTile& t : cand) {
        size_t nbn = 0;
        for (size_t i = 0; i < count; i++) nbn += (Ns[i] + t.cols - 1) / t.cols;
        const size_t wg = ((M + t.rows - 1) / t.rows) * nbn;
        if (wg <= NCU && (wg > best_wg || (wg == best_wg && t.rows * t.cols < best.rows * best.cols))) {
            best = t;
            best_wg = wg;
        }
    }
    return best;   // no tile fits the chip at once: 128 x 128
}

// bt.count == 0: the single problem B / B_scales / C / N; else the problems of bt side by side (nbn column tiles of
// tl.cols in all, as bt.bn_end counts them)
template <bool INT4>
static hipError_t launch_gemm3_any(const int8_t* A, const float* A_scales, const uint8_t* B, const uint16_t* B_scales,
                                   uint16_t* C, size_t M, size_t K, size_t N, int G, GemmTile tl, size_t nbn,
                                   const GemmBatch& bt, hipStream_t st) {
    const size_t nwg = ((M + tl.rows - 1) / tl.rows) * nbn;
    if (nwg == 0) return hipSuccess;
    if (nwg >= (size_t(1) << 30)) return hipErrorInvalidValue;
    const int m = (int)M, k = (int)K, n = (int)N, g = (int)nwg;
    if (tl.rows == 32 && tl.cols == 64) return launch_gemm3<INT4, 1, 1>(A, A_scales, B, B_scales, C, m, k, n, G, g, bt, st);
    if (tl.rows == 32) return launch_gemm3<INT4, 1, 2>(A, A_scales, B, B_scales, C, m, k, n, G, g, bt, st);
    if (tl.rows == 64 && tl.cols == 64) return launch_gemm3<INT4, 2, 1>(A, A_scales, B, B_scales, C, m, k, n, G, g, bt, st);
    if (tl.rows == 64) return launch_gemm3<INT4, 2, 2>(A, A_scales, B, B_scales, C, m, k, n, G, g, bt, st);
    return launch_gemm3<INT4, 4, 2>(A, A_scales, B, B_scales, C, m, k, n, G, g, bt, st);
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
    GemmBatch bt;
    memset(&bt, 0, sizeof bt);
    bt.count = (int)count;
    size_t Ns[G2_MAXB];
    for (size_t i = 0; i < count; i++) Ns[i] = pr[i].N;
    const GemmTile tl = gemm3_tile(M, Ns, count);
    size_t nbn = 0;
    for (size_t i = 0; i < count; i++) {
        nbn += (pr[i].N + tl.cols - 1) / tl.cols;
        if (nbn >= (size_t(1) << 30)) return hipErrorInvalidValue;
        bt.bn_end[i] = (int)nbn;
        bt.B[i] = static_cast<const uint8_t*>(pr[i].B);
        bt.Bs[i] = pr[i].Bs;
        bt.C[i] = pr[i].C;
        bt.N[i] = (int)pr[i].N;
    }
    return int4 ? launch_gemm3_any<true>(A, A_scales, nullptr, nullptr, nullptr, M, K, 0, G, tl, nbn, bt, st)
                : launch_gemm3_any<false>(A, A_scales, nullptr, nullptr, nullptr, M, K, 0, G, tl, nbn, bt, st);
}

// FP16 x FP16^T through the tile kernel: K a multiple of 64 halves, 16-byte aligned operands; false: not this shape
bool launch_matmul_f16_tiles(const uint16_t* a, const uint16_t* bT, uint16_t* c, size_t M, size_t K, size_t N,
                             hipStream_t st, hipError_t* err) {
    if (K == 0 || K % 64 != 0 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(bT)) & 15) != 0 ||
        K >= (size_t(1) << 29) || N >= (size_t(1) << 30) || M >= (size_t(1) << 30))
        return false;
    const GemmTile tl = gemm3_tile(M, &N, 1);
    const size_t nbn = (N + tl.cols - 1) / tl.cols, nwg = ((M + tl.rows - 1) / tl.rows) * nbn;
    if (nwg == 0 || nwg >= (size_t(1) << 30)) return false;
    GemmBatch none;
    memset(&none, 0, sizeof none);
    const int8_t* A = reinterpret_cast<const int8_t*>(a);
    const uint8_t* B = reinterpret_cast<const uint8_t*>(bT);
    const int m = (int)M, kb = (int)(2 * K), n = (int)N, G = (int)(2 * K / 128), g = (int)nwg;
    if (tl.rows == 32 && tl.cols == 64) *err = launch_gemm3<false, 1, 1, true>(A, nullptr, B, nullptr, c, m, kb, n, G, g, none, st);
    else if (tl.rows == 32) *err = launch_gemm3<false, 1, 2, true>(A, nullptr, B, nullptr, c, m, kb, n, G, g, none, st);
    else if (tl.rows == 64 && tl.cols == 64) *err = launch_gemm3<false, 2, 1, true>(A, nullptr, B, nullptr, c, m, kb, n, G, g, none, st);
    else if (tl.rows == 64) *err = launch_gemm3<false, 2, 2, true>(A, nullptr, B, nullptr, c, m, kb, n, G, g, none, st);
    else *err = launch_gemm3<false, 4, 2, true>(A, nullptr, B, nullptr, c, m, kb, n, G, g, none, st);
    return true;
}

hipError_t launch_matmul_gemm(bool int4, const int8_t* A, const float* A_scales, const void* B,
                              const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                              size_t K, size_t N, hipStream_t st) {
    const int G = (int)(K / 128);
    if (gsums == nullptr && C != nullptr && (size_t)G2_STAGES * G2_STAGE + (size_t)G * GM_BN * 2 <= 160 * 1024 &&
        ((M + 31) / 32) * ((N + GM_BN - 1) / GM_BN) < (size_t(1) << 30)) {
        GemmBatch none;
        memset(&none, 0, sizeof none);
        const uint8_t* Bp = static_cast<const uint8_t*>(B);
        const GemmTile tl = gemm3_tile(M, &N, 1);
        const size_t nbn = (N + tl.cols - 1) / tl.cols;
        return int4 ? launch_gemm3_any<true>(A, A_scales, Bp, B_scales, C, M, K, N, G, tl, nbn, none, st)
                    : launch_gemm3_any<false>(A, A_scales, Bp, B_scales, C, M, K, N, G, tl, nbn, none, st);
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
