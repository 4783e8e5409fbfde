// Decode-shaped (M <= 16) W8A8 / W4A8 matmul: a pure weight stream.
//
// Roofline: HBM.  Algorithmic bytes per call = N*K (INT8) or N*K/2 (INT4) weight bytes
// + 2*N*K/128 scale bytes + M*K + 4*M activation bytes + 2*M*N output bytes.
//
// Layout.  One workgroup of W waves owns 16 consecutive weight rows n0..n0+15.  The row is cut
// into units of 128 bytes (one quantization group of INT8, two of packed INT4); wave w streams
// units w, w+W, w+2W, ... so the waves read disjoint slices and every 128-byte line is fetched
// by exactly one wave.  A unit is two dwordx4 loads per lane (16 rows x 64 contiguous bytes
// each) that go straight to VGPRs: nothing is shared between waves, so there is no LDS staging
// of weights.  Each wave keeps a ring of P units in flight; the steady-state loop is
// branch-free so the compiler can retire loads with counted s_waitcnt vmcnt(N) and a slot is
// refilled as soon as its MFMAs are issued.
//
// The INT32 dot products come from v_mfma_i32_16x16x64_i8 with the activations as the row
// operand (rows m >= M are zeroed) and the 16 weight rows as the column operand; two MFMAs
// cover one group, which makes the group sum exact and does the cross-lane reduction for free.
// INT4 weights stay packed in HBM: a 16-byte load holds 32 values, the low nibbles feed the
// first MFMA of a group and the high nibbles the second, each moved to the top of its byte
// (value*16, sign included) so no sign-extension arithmetic is needed; the factor 16 is folded
// into the scale (exactly: a power of two).  The activation operand is permuted to match (even
// / odd k).  Any consistent permutation of k inside a group leaves an integer sum unchanged,
// so the lane->k mapping is chosen for coalescing, not for the MFMA's nominal k order.
//
// The fp32 part follows the reference order exactly (kernel_gemm.cpp:311-324):
//   running = fma(float(group_sum_g), float(B_scale[n,g]), running)   g ascending
//   C = half(running * A_scale[m])
// Waves drop their converted group sums into LDS; thread (m, n) of the workgroup then runs the
// g-ordered fma chain with the tile's scales, which are copied to LDS behind the first weight
// loads.  K is processed in chunks only when the M*G group sums exceed the LDS budget.
#include <algorithm>
#include "common.hip.h"
#include "kernels.h"
#include "block_math.hip.h"

#include <cstring>

namespace cactus_hip {

constexpr int SK_TN = 16;  // weight rows per workgroup

// Benchmarks only (ablate bit 4): shader-clock stamps of the phases of the first 64 tiles.
__device__ unsigned long long g_sk_trace[64 * 8];
#define SK_STAMP(ph)                                                                   \
    do {                                                                               \
        if ((ablate & 16) && threadIdx.x == 0 && tile < 64)                            \
            g_sk_trace[tile * 8 + (ph)] = __builtin_readcyclecounter();                \
    } while (0)

// Diagnostic builds only (-DFQ_TRACE, scripts/build_fq_trace.sh): wall-clock stamps (100 MHz, the same clock on every
// CU and across launches) of the phases of the fused-quantization body, for up to FQT_WGS workgroups of each of the
// last FQT_LAUNCHES launches.  A launch learns its index from a counter that thread 0 of every workgroup raises when
// the workgroup is done (launches on one stream run one after the other).
#ifdef FQ_TRACE
constexpr int FQT_LAUNCHES = 512, FQT_WGS = 64, FQT_PH = 8;
__device__ unsigned long long g_fq_trace[FQT_LAUNCHES * FQT_WGS * FQT_PH];
static int g_fq_trace_next = 0;     // host: every enqueued launch takes the next index (a captured launch keeps its own)
#define FQ_STAMP(ph)                                                                                   \
    do {                                                                                               \
        if (t == 0 && fqt_slot >= 0) g_fq_trace[fqt_slot * FQT_PH + (ph)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define FQ_STAMP(ph) do { } while (0)
#endif

__device__ __forceinline__ v4i ldg(const void* p) { return *reinterpret_cast<const v4i*>(p); }

// Raw buffer loads: one SGPR descriptor per operand, a 32-bit per-lane byte offset, hardware
// range check (offsets >= num_records return 0 without touching memory, which is how rows
// beyond N / M and units beyond K are padded).  AUX 2 = non-temporal (weights are read once).
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr unsigned SK_OOB = 0x80000000u;   // added to an offset to push it out of range
template <int AUX>
__device__ __forceinline__ v4i buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int imm) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + imm), 0, AUX);
    return v4i{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}

// the same with a wave-uniform byte offset in an SGPR (added after the range check of voff: callers only add it to
// offsets of whole units that lie inside the buffer)
template <int AUX>
__device__ __forceinline__ v4i buf_load_s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, AUX);
    return v4i{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}

// Tile-major weights ("packed twin", pack_weights_kernel below).  Row-major [N, K] makes every wave instruction of
// the stream touch 16 separate 64-byte segments, one per weight row.  The twin stores tile t = rows 16 t .. 16 t + 15
// as one contiguous run of K / 128 units (INT4: K / 256) of 2 KB; a unit is two 1 KB halves, and the 16-byte piece of
// lane l = 16 q + r in half h is row r's bytes [128 u + 64 h + 16 q, + 16) - exactly what that lane feeds the MFMA -
// so a wave instruction reads 1 KB of consecutive bytes and a workgroup walks one contiguous stretch of memory
// (measured on a read-only kernel over 240 MB: 6.7 TB/s in this shape against 5.3-5.5 in 16-row units).  Rows past N
// are zero rows of the last tile.  The kernels take `pk` at run time: it only changes three address terms.
struct PkAddr {
    unsigned ustride;   // bytes from one unit of a lane's stream to the next
    unsigned half;      // bytes from the first to the second load of a unit
};
__device__ __forceinline__ PkAddr pk_addr(bool pk) { return pk ? PkAddr{2048u, 1024u} : PkAddr{128u, 64u}; }

template <bool INT4, bool ALDS>
struct UnitRegs {
    v4i b[2];
    v4i a[ALDS ? 1 : (INT4 ? 4 : 2)];   // ALDS: the activation operand is read from LDS at use
};

// INT4:  packed weights.   W: waves per workgroup.   P: ring depth (units in flight per wave).
// ALDS:  activations (M*K bytes) are staged in LDS once per workgroup and the MFMA row operand
//        is read from there; otherwise it is loaded from global memory (L2) with the ring.
// GS:    also write the INT32 group sums (parity checks); C may then be null.
template <bool INT4, int W, int P, bool ALDS, bool GS>
__device__ __forceinline__ void skinny_body(
    const int8_t* __restrict__ A, const float* __restrict__ A_scales,
    const uint8_t* __restrict__ B, const uint16_t* __restrict__ B_scales,
    uint16_t* __restrict__ C, int32_t* __restrict__ gsums, int M, int K, int N, int G, int SC,
    int ablate, int tile, bool pk) {
    constexpr int GPU_ = INT4 ? 2 : 1;   // groups per unit
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    // Rows are padded by 4 floats: 16-byte aligned for the chain's ds_read_b128, and the 16
    // lanes that write one group land on 8 different banks.
    const int SCP = SC + 4, GP = ((G + 3) & ~3) + 4;
    float* gbuf = sk_lds;                                                        // [M*16][SCP] group sums
    float* scl = sk_lds + (size_t)M * 16 * SCP;                                  // [16][GP] fp32 scales
    int8_t* lds_a = reinterpret_cast<int8_t*>(scl + SK_TN * GP);                 // [M][K] (ALDS)

    SK_STAMP(0);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15;   // MFMA row (activation row m) for A, column (weight row) for B
    const int q = lane >> 4;   // k block of the MFMA operand
    const int n0 = tile * SK_TN;
    const unsigned row_bytes = INT4 ? (unsigned)K / 2 : (unsigned)K;
    const PkAddr pa = pk_addr(pk);
    const unsigned nrows = pk ? (unsigned)((N + 15) & ~15) : (unsigned)N;   // the twin holds whole tiles
    const __amdgpu_buffer_rsrc_t brsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(B), 0, (int)(nrows * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(A), 0, M * K, 0x00020000);
    // per-lane byte offsets; rows past the end are pushed out of range and read as zero
    const unsigned boff = pk ? ((unsigned)n0 < nrows ? (unsigned)n0 * row_bytes + lane * 16 : SK_OOB)
                             : ((n0 + r < N) ? (unsigned)(n0 + r) * row_bytes + q * 16 : SK_OOB);
    const unsigned aoff = (r < M) ? (unsigned)r * K + (INT4 ? q * 32 : q * 16) : SK_OOB;
    const int arow = min(r, M - 1);
    const int8_t* aptr_lds = lds_a + (size_t)arow * K + (INT4 ? q * 32 : q * 16);
    const bool a_live = r < M;
    const v4i zero4 = {0, 0, 0, 0};
    const bool want_c = C != nullptr;
    const int nunits = G / GPU_;   // units per weight row

    // u: unit index in the row; units >= nunits are read as zeros (no memory traffic)
    auto load_unit = [&](int u, UnitRegs<INT4, ALDS>& R) {
        const unsigned uo = (u < nunits) ? (unsigned)u * pa.ustride : SK_OOB;
        R.b[0] = buf_load<2>(brsrc, boff + uo, 0);
        R.b[1] = buf_load_s<2>(brsrc, boff + uo, pa.half);
        if constexpr (!ALDS) {
            const unsigned ao = (u < nunits) ? (unsigned)u * (INT4 ? 256u : 128u) : SK_OOB;
            if constexpr (INT4) {
                R.a[0] = buf_load<0>(arsrc, aoff + ao, 0);
                R.a[1] = buf_load<0>(arsrc, aoff + ao, 16);
                R.a[2] = buf_load<0>(arsrc, aoff + ao, 128);
                R.a[3] = buf_load<0>(arsrc, aoff + ao, 144);
            } else {
                R.a[0] = buf_load<0>(arsrc, aoff + ao, 0);
                R.a[1] = buf_load<0>(arsrc, aoff + ao, 64);
            }
        }
    };

    // store one group's 16x16 INT32 tile: lane holds column n = lane & 15, rows m = 4*(lane>>4)+i
    auto put_group = [&](int g, int gl, int ng, v4i acc) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = q * 4 + i;
            if (m < M && gl < ng && !(ablate & 2)) {
                gbuf[(m * 16 + r) * SCP + gl] = (float)acc[i];
                if constexpr (GS) {
                    if (n0 + r < N) gsums[((size_t)m * N + n0 + r) * G + g] = INT4 ? (acc[i] >> 4) : acc[i];
                }
            }
        }
    };

    // u: unit index inside the chunk; g0: first group of the chunk
    auto compute_unit = [&](int u, int g0, int ng, const UnitRegs<INT4, ALDS>& R) {
        if constexpr (INT4) {
            v4i a[4];
            if constexpr (ALDS) {
                const int8_t* ap = aptr_lds + (size_t)min(g0 / 2 + u, nunits - 1) * 256;
                a[0] = ldg(ap); a[1] = ldg(ap + 16); a[2] = ldg(ap + 128); a[3] = ldg(ap + 144);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] = R.a[i];
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {   // the unit's two groups: 64 packed bytes each
                v4i blo, bhi, ae, ao;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t x = (uint32_t)R.b[h][i];
                    blo[i] = (int)((x & 0x0F0F0F0Fu) << 4);
                    bhi[i] = (int)(x & 0xF0F0F0F0u);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t lo = (uint32_t)a[2 * h + (i >> 1)][(2 * i) & 3];
                    const uint32_t hi = (uint32_t)a[2 * h + (i >> 1)][((2 * i) & 3) + 1];
                    ae[i] = (!ALDS || a_live) ? (int)__builtin_amdgcn_perm(hi, lo, 0x06040200u) : 0;
                    ao[i] = (!ALDS || a_live) ? (int)__builtin_amdgcn_perm(hi, lo, 0x07050301u) : 0;
                }
                v4i acc = zero4;
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ae, blo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ao, bhi, acc, 0, 0, 0);
                put_group(g0 + 2 * u + h, 2 * u + h, ng, acc);
            }
        } else {
            v4i a0, a1;
            if constexpr (ALDS) {
                const int8_t* ap = aptr_lds + (size_t)min(g0 + u, nunits - 1) * 128;
                a0 = ldg(ap); a1 = ldg(ap + 64);
            } else {
                a0 = R.a[0]; a1 = R.a[1];
            }
            if constexpr (ALDS) {
                a0 = a_live ? a0 : zero4;
                a1 = a_live ? a1 : zero4;
            }
            v4i acc = zero4;
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, R.b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, R.b[1], acc, 0, 0, 0);
            put_group(g0 + u, u, ng, acc);
        }
    };

    // chain element e = t + k*W*64 -> (m = e / 16, n = e % 16); NR elements per thread
    constexpr int NR = (256 + W * 64 - 1) / (W * 64);
    float run[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) run[k] = 0.0f;

    for (int g0 = 0; g0 < G; g0 += SC) {
        const int ng = min(SC, G - g0);
        const int nu = ng / GPU_;                    // units in this chunk
        const int ubase = g0 / GPU_;                 // first unit of the chunk
        UnitRegs<INT4, ALDS> ring[P];

        // activations -> LDS (ALDS runs as a single chunk): requested first, written to LDS
        // after the first weight loads have been issued
        constexpr int AST = 4;   // 16-byte pieces per thread held in registers
        const int npieces = M * K / 16;
        v4i a_stage[AST];
        if constexpr (ALDS) {
#pragma unroll
            for (int j = 0; j < AST; j++)   // pieces past the end read as zeros (no branch)
                a_stage[j] = buf_load<0>(arsrc, (unsigned)(t + j * W * 64) * 16u, 0);
        }
        // every wave runs the same number of rounds; units past the chunk read as zeros
        const int nu_end = ubase + nu;               // one past the chunk's last unit
        auto chunk_unit = [&](int ul) { return (ul < nu) ? ubase + ul : nunits; };
#pragma unroll
        for (int p = 0; p < P; p++) load_unit(chunk_unit(w + p * W), ring[p]);
        (void)nu_end;

        // tile scales: requested behind the first weight loads, parked in registers, and written
        // to LDS (as fp32, x 1/16 for INT4) only after the stream, so no wait sits in the ring
        const bool sc_fast = (n0 + SK_TN <= N) && ((G & 7) == 0) && (SK_TN * G / 8 <= W * 64) &&
                             ((reinterpret_cast<uintptr_t>(B_scales) & 15) == 0);
        const bool sc_mine = g0 == 0 && want_c && !(ablate & 4) && sc_fast && t < SK_TN * G / 8;
        const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(B_scales), 0, (int)min((size_t)N * G * 2, (size_t)0x7FFFFFFF), 0x00020000);
        const v4i sc_regs = buf_load<0>(srsrc, sc_mine ? (unsigned)(((size_t)n0 * G + (size_t)t * 8) * 2) : SK_OOB, 0);
        if (g0 == 0 && want_c && !(ablate & 4) && !sc_fast) {
            for (int idx = t; idx < SK_TN * G; idx += W * 64) {
                const int row = idx / G, g = idx - row * G;
                const float s = h2f(B_scales[(size_t)min(n0 + row, N - 1) * G + g]);
                scl[row * GP + g] = INT4 ? s * 0.0625f : s;
            }
        }
        if constexpr (ALDS) {   // LDS writes visible, weight loads stay in flight
#pragma unroll
            for (int j = 0; j < AST; j++) {
                const int i16 = t + j * W * 64;
                if (i16 < npieces) *reinterpret_cast<v4i*>(lds_a + (size_t)i16 * 16) = a_stage[j];
            }
            for (int i16 = t + AST * W * 64; i16 < npieces; i16 += W * 64)
                *reinterpret_cast<v4i*>(lds_a + (size_t)i16 * 16) = ldg(A + (size_t)i16 * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }

        SK_STAMP(3);
        const int rounds = (nu + W * P - 1) / (W * P);
        for (int rd = 0; rd + 1 < rounds; rd++) {    // no VMEM under a branch: counted waits
#pragma unroll
            for (int p = 0; p < P; p++) {
                const int ul = w + (rd * P + p) * W; // unit index inside the chunk
                compute_unit(ul, g0, ng, ring[p]);
                load_unit(chunk_unit(ul + P * W), ring[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < P; p++)                  // last round: nothing left to request
            compute_unit(w + ((rounds - 1) * P + p) * W, g0, ng, ring[p]);
        if (sc_mine) {   // 8 consecutive scales of one row (G % 8 == 0)
            const int row = (t * 8) / G, g = (t * 8) - row * G;
            float* dst = scl + row * GP + g;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t x = (uint32_t)sc_regs[i];
                const float lo = h2f((uint16_t)(x & 0xFFFFu)), hi = h2f((uint16_t)(x >> 16));
                dst[2 * i] = INT4 ? lo * 0.0625f : lo;
                dst[2 * i + 1] = INT4 ? hi * 0.0625f : hi;
            }
        }
        SK_STAMP(4);
        __syncthreads();
        SK_STAMP(5);
        if (want_c && !(ablate & 1)) {
#pragma unroll
            for (int k = 0; k < NR; k++) {
                const int e = t + k * W * 64;
                if (e < M * 16) {
                    const float* sp = scl + (e & 15) * GP + g0;   // g0 and GP are multiples of 4
                    const float* gp = gbuf + (size_t)e * SCP;
                    int gl = 0;
                    // 16 groups per batch: eight ds_read_b128, then the serial fma chain
                    for (; gl + 16 <= ng; gl += 16) {
                        v4f gv[4], sv[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            gv[u] = *reinterpret_cast<const v4f*>(gp + gl + 4 * u);
                            sv[u] = *reinterpret_cast<const v4f*>(sp + gl + 4 * u);
                        }
#pragma unroll
                        for (int u = 0; u < 16; u++) run[k] = fmaf(gv[u >> 2][u & 3], sv[u >> 2][u & 3], run[k]);
                    }
                    for (; gl < ng; gl++) run[k] = fmaf(gp[gl], sp[gl], run[k]);
                }
            }
        }
        if (g0 + SC < G) __syncthreads();
    }

    SK_STAMP(6);
    if (want_c) {
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int e = t + k * W * 64;
            const int cm = e >> 4, cn = e & 15;
            if (e < M * 16 && n0 + cn < N) {
                C[(size_t)cm * N + n0 + cn] = f2h_mul(run[k], A_scales[cm]);
            }
        }
    }
}

template <bool INT4, int W, int P, bool ALDS, bool GS>
__global__ __launch_bounds__(W * 64) void skinny_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ A_scales,
    const uint8_t* __restrict__ B, const uint16_t* __restrict__ B_scales,
    uint16_t* __restrict__ C, int32_t* __restrict__ gsums, int M, int K, int N, int G, int SC,
    int ablate, int pk) {
    skinny_body<INT4, W, P, ALDS, GS>(A, A_scales, B, B_scales, C, gsums, M, K, N, G, SC, ablate,
                                          (int)blockIdx.x, pk != 0);
}

// ---------------------------------------------------------------------------------------
// Fused-quantization body (FP16 activations, M <= 4), one tile of 16 * CT weight rows.
//
// The wave group quantizes the activation rows into LDS -- absmax, scale = max(absmax/127, 1e-10),
// q = rne(clamp(x * (1/scale))), exactly as act_quant_rowwise_kernel -- while its first ring of
// weight loads is in flight, streams the tile, then runs the fp32 chain.  (Walking several tiles
// per wave group, with the next tile's loads under the epilogue, measured slower at every tile
// count and cost 30 registers: 4 instead of 5 waves per SIMD.)
//
// PRO: where the activation rows come from.
//   0  X as given.
//   1  rms_norm(X [+ x2]) * w -- the residual add and norm in front of q/k/v or gate/up
//      (model_qwen.cpp:51-53, 125-127) with the arithmetic of rms_norm_kernel: the summed rows go
//      to LDS, the 16-lane groups of wave 0 run the accumulation chain of one row each, every
//      thread then normalizes its own pieces in place.  Tile 0 also writes the sum to sum_out.
//   2  silu(X) * x2    3  gelu(X) * x2 -- the activation in front of the down projection
//      (model_qwen.cpp:114-116), element-wise on the way in (unary_kernel<OP, true>'s arithmetic).
// Every workgroup recomputes the row: a few hundred nanoseconds against a launch of its own.
struct FqPro {
    const uint16_t* x2 = nullptr;
    const uint16_t* w = nullptr;
    uint16_t* sum_out = nullptr;
    float eps = 0.0f;
    int chains = 0;     // PRO 1: the LDS region holds a second copy of the summed rows, chain by chain (see fq_body)
};

__device__ __forceinline__ v4i add8_bits(v4i a, v4i b) {
    v4i o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t lo = h2bits(bits2h((uint16_t)((uint32_t)a[e] & 0xFFFFu)) + bits2h((uint16_t)((uint32_t)b[e] & 0xFFFFu)));
        const uint32_t hi = h2bits(bits2h((uint16_t)((uint32_t)a[e] >> 16)) + bits2h((uint16_t)((uint32_t)b[e] >> 16)));
        o[e] = (int)(lo | (hi << 16));
    }
    return o;
}
__device__ __forceinline__ v4i rms_out8_bits(v4i x, v4i w, float inv, h16 inv_h) {
    v4i o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t lo = rms_out((uint16_t)((uint32_t)x[e] & 0xFFFFu), (uint16_t)((uint32_t)w[e] & 0xFFFFu), inv, inv_h, true);
        const uint32_t hi = rms_out((uint16_t)((uint32_t)x[e] >> 16), (uint16_t)((uint32_t)w[e] >> 16), inv, inv_h, true);
        o[e] = (int)(lo | (hi << 16));
    }
    return o;
}
template <int OP>
__device__ __forceinline__ v4i act_mul8_bits(v4i g, v4i u) {
    v4i o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t lo = h2bits(bits2h(unary_one<OP>((uint16_t)((uint32_t)g[e] & 0xFFFFu), true)) * bits2h((uint16_t)((uint32_t)u[e] & 0xFFFFu)));
        const uint32_t hi = h2bits(bits2h(unary_one<OP>((uint16_t)((uint32_t)g[e] >> 16), true)) * bits2h((uint16_t)((uint32_t)u[e] >> 16)));
        o[e] = (int)(lo | (hi << 16));
    }
    return o;
}

// WT (wave tiles): the W waves share the activation rows but each streams a 16-row tile of its own (tile
// tl * W + w) over the whole of K and runs the fp32 chain in registers as the group sums come out of the
// MFMAs - no LDS hand-over, no barrier after the prologue.  For thousands of tiles with short rows (the
// vocabulary projection: 9496 tiles of 1024 bytes per row): each wave then has its whole tile in flight at
// once.  Rows of several ring rounds stream faster split over the waves (the default), whose reads of one
// row are adjacent.  Needs K % 1024 == 0 (a round is 8 groups = one 16-byte chunk of scales).
#ifndef SK_A_PRESPLIT
#define SK_A_PRESPLIT 1
#endif
template <bool INT4, int W, int P, int CT, int PRO = 0, bool WT = false>
__device__ __forceinline__ void fq_body(const uint16_t* __restrict__ X, float* __restrict__ scales_out,
                                        const uint8_t* __restrict__ B, const uint16_t* __restrict__ B_scales,
                                        uint16_t* __restrict__ C, int M, int K, int N, int tl,
                                        int t, float* __restrict__ sk_lds, bool pk, const FqPro pro = FqPro(),
                                        int trace_id = -1) {
    // t: thread index among the W waves that share the tile; sk_lds: their LDS region.  A workgroup
    // may hold several such wave groups (tiles of one problem, same barrier sequence).
    constexpr int GPU_ = INT4 ? 2 : 1;   // groups per unit
    const int G = K / 128;
    constexpr int step = W * GPU_;
    const int SC = ((G + step - 1) / step) * step;
    const int SCP = SC + 4, GP = ((G + 3) & ~3) + 4;
    constexpr int TR = SK_TN * CT;   // weight rows per tile: CT MFMA column tiles share each activation fragment
    static_assert(!WT || (CT == 1 && PRO == 0 && P * GPU_ == 8), "wave tiles: one column tile, 8 groups per ring round");
    float* gbuf = sk_lds;                                                        // [CT][M*16][SCP] group sums
    float* scl = sk_lds + (WT ? 0 : (size_t)CT * M * 16 * SCP);                  // [TR][GP] fp32 scales
    int8_t* lds_a = reinterpret_cast<int8_t*>(scl + (WT ? 0 : TR * GP));         // [M][K]
    float* wmax = reinterpret_cast<float*>(lds_a + (((size_t)M * K + 15) & ~(size_t)15));   // [W][4]
    float* s_inv = wmax + W * 4;                                                            // [4]      (PRO 1)
    uint16_t* lds_x = reinterpret_cast<uint16_t*>(s_inv + 4);                               // [M][K]   (PRO 1)
    // [M][16][K / 16 + 8] (PRO 1, pro.chains): the same rows, the 16 accumulation chains of the norm's sum of squares
    // each contiguous (elements c, c + 16, c + 32 ...), so that a chain's lane reads 8 elements per LDS instruction; the
    // 16 bytes between chains put the 16 lanes' reads on different banks
    uint16_t* lds_t = lds_x + (size_t)M * K;

#ifdef FQ_TRACE
    int fqt_slot = -1;
    const unsigned long long fqt_t0 = __builtin_amdgcn_s_memrealtime();    // before anything else of the body
    {
        const int wg = (int)blockIdx.x, nb = (int)gridDim.x;
        const int stride = nb > FQT_WGS ? nb / FQT_WGS : 1;     // FQT_WGS workgroups spread over the grid
        if ((int)threadIdx.x == 0 && trace_id >= 0 && wg % stride == 0 && wg / stride < FQT_WGS)
            fqt_slot = (trace_id % FQT_LAUNCHES) * FQT_WGS + wg / stride;
    }
    auto fqt_finish = [&]() {
        if ((int)threadIdx.x == 0 && fqt_slot >= 0) g_fq_trace[fqt_slot * FQT_PH + 5] = __builtin_amdgcn_s_memrealtime();
    };
    if (t == 0 && fqt_slot >= 0) g_fq_trace[fqt_slot * FQT_PH + 0] = fqt_t0;
    FQ_STAMP(6);
#endif
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int wtile = WT ? tl * W + w : tl;       // the tile whose weights this wave reads
    const unsigned row_bytes = INT4 ? (unsigned)K / 2 : (unsigned)K;
    const PkAddr pa = pk_addr(pk);
    const unsigned nrows = pk ? (unsigned)((N + 15) & ~15) : (unsigned)N;   // the twin holds whole tiles
    const __amdgpu_buffer_rsrc_t brsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(B), 0, (int)(nrows * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X), 0, 2 * M * K, 0x00020000);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(B_scales), 0, (int)min((size_t)N * G * 2, (size_t)0x7FFFFFFF), 0x00020000);
    const int arow = min(r, M - 1);
    const int8_t* aptr_lds = lds_a + (size_t)arow * K + (INT4 ? q * 32 : q * 16);
    const v4i zero4 = {0, 0, 0, 0};
    const int nunits = G / GPU_;
    const bool sc_fast_shape = ((G & 7) == 0) && (TR * G / 8 <= CT * W * 64) &&
                               ((reinterpret_cast<uintptr_t>(B_scales) & 15) == 0);

    struct Unit { v4i b[CT][2]; };
    struct Boff { unsigned o[CT]; };
    struct ScRegs { v4i v[CT]; };
    Unit ring[P];
    Boff boff;   // a row past N is out of range and reads zeros
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int n = wtile * TR + ct * SK_TN + r;
        const unsigned nt0 = (unsigned)(wtile * TR + ct * SK_TN);
        boff.o[ct] = pk ? (nt0 < nrows ? nt0 * row_bytes + lane * 16 : SK_OOB)
                        : ((n < N) ? (unsigned)n * row_bytes + q * 16 : SK_OOB);
    }
    auto load_unit = [&](int u, const Boff& bo, Unit& R) {
        const unsigned uo = (u < nunits) ? (unsigned)u * pa.ustride : SK_OOB;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            R.b[ct][0] = buf_load<2>(brsrc, bo.o[ct] + uo, 0);
            R.b[ct][1] = buf_load_s<2>(brsrc, bo.o[ct] + uo, pa.half);
        }
    };
    // this thread's chunks of 8 scales of the tile (fast path: chunk t + j*W*64), else out of range
    const int n0 = wtile * TR;
    const bool fast = sc_fast_shape && n0 + TR <= N;
    // wave tiles: the fp32 chains of rows m = 4 q + i, and the lane's row of scales, 8 groups (16 bytes) per
    // ring round, requested one round ahead
    float run[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    v4i sc_cur = zero4;
    auto load_scale_chunk = [&](int chunk) {
        const int n = n0 + r;
        return buf_load<0>(srsrc, (n < N && chunk * 8 < G) ? (unsigned)(((size_t)n * G + (size_t)chunk * 8) * 2) : SK_OOB, 0);
    };
    auto load_scales = [&]() {
        ScRegs sr;
#pragma unroll
        for (int j = 0; j < CT; j++) {
            const int ch = t + j * W * 64;
            const unsigned off = (fast && ch < TR * G / 8) ? (unsigned)(((size_t)tl * TR * G + (size_t)ch * 8) * 2) : SK_OOB;
            sr.v[j] = buf_load<0>(srsrc, off, 0);
        }
        return sr;
    };
    auto put_group = [&](int ct, int gl, v4i acc, int j8) {
        if constexpr (WT) {      // the chain step of this group; j8 = gl % 8 (a constant after unrolling) picks its scale
            const uint32_t x = (uint32_t)sc_cur[j8 >> 1];
            float sv = h2f((uint16_t)((j8 & 1) ? (x >> 16) : (x & 0xFFFFu)));
            if (INT4) sv *= 0.0625f;
            run[0] = fmaf((float)acc[0], sv, run[0]);
            if (M > 1) {            // tests the whole wave shares: one row is the common case
                run[1] = fmaf((float)acc[1], sv, run[1]);
                if (M > 2) run[2] = fmaf((float)acc[2], sv, run[2]);
                if (M > 3) run[3] = fmaf((float)acc[3], sv, run[3]);
            }
            return;
        }
        // M <= 4: the rows m = 4 q + i that exist sit in the lanes of q = 0; one lane test per group, then tests on M
        // that the whole wave shares (no conversion or store is issued for a row that does not exist)
        if (q == 0 && gl < G) {
            float* dst = gbuf + ((size_t)(ct * M) * 16 + r) * SCP + gl;
            dst[0] = (float)acc[0];
            if (M > 1) dst[16 * SCP] = (float)acc[1];
            if (M > 2) dst[32 * SCP] = (float)acc[2];
            if (M > 3) dst[48 * SCP] = (float)acc[3];
        }
    };
    auto compute_unit = [&](int u, const Unit& R, int pslot) {
        if constexpr (INT4) {
            const int8_t* ap = aptr_lds + (size_t)min(u, nunits - 1) * 256;
            v4i a[4] = {ldg(ap), ldg(ap + 16), ldg(ap + 128), ldg(ap + 144)};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                v4i ae, ao;
                // (MFMA rows m >= M hold whatever row M - 1 gives: no one reads them, so nothing is zeroed)
                if constexpr (SK_A_PRESPLIT) {      // the spans in LDS are [even | odd] already
                    ae = a[2 * h];
                    ao = a[2 * h + 1];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t lo = (uint32_t)a[2 * h + (i >> 1)][(2 * i) & 3];
                        const uint32_t hi = (uint32_t)a[2 * h + (i >> 1)][((2 * i) & 3) + 1];
                        ae[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x06040200u);
                        ao[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x07050301u);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    v4i blo, bhi;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t x = (uint32_t)R.b[ct][h][i];
                        blo[i] = (int)((x & 0x0F0F0F0Fu) << 4);
                        bhi[i] = (int)(x & 0xF0F0F0F0u);
                    }
                    v4i acc = zero4;
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ae, blo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ao, bhi, acc, 0, 0, 0);
                    put_group(ct, 2 * u + h, acc, 2 * pslot + h);
                }
            }
        } else {
            const int8_t* ap = aptr_lds + (size_t)min(u, nunits - 1) * 128;
            const v4i a0 = ldg(ap), a1 = ldg(ap + 64);
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                v4i acc = zero4;
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, R.b[ct][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, R.b[ct][1], acc, 0, 0, 0);
                put_group(ct, u, acc, pslot);
            }
        }
    };

    // ---- requests of the first tile, then the activation rows -> LDS
    constexpr int AST = 2;   // 16-byte pieces per thread held in registers (an M = 1 row of K <= 8192 fits)
    const int npieces = M * K / 8;
    v4i a_stage[AST];
    v4i b_stage[PRO ? AST : 1], w_stage[PRO == 1 ? AST : 1];
    const __amdgpu_buffer_rsrc_t x2rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pro.x2), 0, pro.x2 ? 2 * M * K : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pro.w), 0, pro.w ? 2 * K : 0, 0x00020000);
    const int kp = K / 8;    // pieces per row
    auto wpiece = [&](int i16) {   // piece of the norm weight row under piece i16 (M <= 4: no division)
        int c = i16;
        c -= (M > 1 && c >= kp) ? kp : 0;
        c -= (M > 2 && c >= kp) ? kp : 0;
        c -= (M > 3 && c >= kp) ? kp : 0;
        return c;
    };
#pragma unroll
    for (int j = 0; j < AST; j++) {
        const int i16 = t + j * W * 64;
        a_stage[j] = buf_load<0>(xrsrc, (unsigned)i16 * 16u, 0);
        if constexpr (PRO != 0) b_stage[j] = buf_load<0>(x2rsrc, (unsigned)i16 * 16u, 0);   // no x2: zeros
        if constexpr (PRO == 1) w_stage[j] = buf_load<0>(wrsrc, i16 < npieces ? (unsigned)wpiece(i16) * 16u : SK_OOB, 0);
    }
#pragma unroll
    for (int p = 0; p < P; p++) load_unit(WT ? p : w + p * W, boff, ring[p]);
    const ScRegs sc_regs = WT ? ScRegs() : load_scales();
    v4i sc_next = WT ? load_scale_chunk(0) : zero4;
    if constexpr (PRO == 2 || PRO == 3) {
#pragma unroll
        for (int j = 0; j < AST; j++) a_stage[j] = act_mul8_bits<PRO == 2 ? OP_SILU : OP_GELU>(a_stage[j], b_stage[j]);
    }
    if constexpr (PRO == 1) {
        const char* Xb = reinterpret_cast<const char*>(X);
        const char* X2b = reinterpret_cast<const char*>(pro.x2);
        const char* Wb = reinterpret_cast<const char*>(pro.w);
        auto stage_sum = [&](int i16, v4i x, v4i d) {
            const v4i sv = pro.x2 ? add8_bits(x, d) : x;
            *reinterpret_cast<v4i*>(lds_x + (size_t)i16 * 8) = sv;
            if (pro.sum_out != nullptr && tl == 0) *reinterpret_cast<v4i*>(pro.sum_out + (size_t)i16 * 8) = sv;
            if (pro.chains) {       // elements k0 .. k0 + 7 of a row: position k0 / 16 of the chains k0 % 16 .. + 7
                const int e0 = i16 * 8;
                const int row = (M > 1 && e0 >= K) + (M > 2 && e0 >= 2 * K) + (M > 3 && e0 >= 3 * K);
                const int k0 = e0 - row * K, cl = (K >> 4) + 8;
                uint16_t* dst = lds_t + (size_t)(row * 16 + (k0 & 15)) * cl + (k0 >> 4);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t wd = (uint32_t)sv[j >> 1];
                    dst[j * cl] = (uint16_t)((j & 1) ? (wd >> 16) : (wd & 0xFFFFu));
                }
            }
        };
#pragma unroll
        for (int j = 0; j < AST; j++) {
            const int i16 = t + j * W * 64;
            if (i16 < npieces) stage_sum(i16, a_stage[j], b_stage[j]);
        }
#pragma unroll 1
        for (int i16 = t + AST * W * 64; i16 < npieces; i16 += W * 64)
            stage_sum(i16, ldg(Xb + (size_t)i16 * 16), pro.x2 ? ldg(X2b + (size_t)i16 * 16) : zero4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (w == 0 && (lane >> 4) < M) {      // one 16-lane group per row
            const uint16_t* xr = lds_x + (size_t)(lane >> 4) * K;
            const float inv = pro.chains
                ? row_inv_rms_chain(lds_t + (size_t)((lane >> 4) * 16 + (lane & 15)) * ((K >> 4) + 8), K, pro.eps, lane & 15, lane & ~15)
                : row_inv_rms([&](int i) -> uint16_t { return xr[i]; }, K, pro.eps, lane & 15, lane & ~15);
            if ((lane & 15) == 0) s_inv[lane >> 4] = inv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        auto norm_piece = [&](int i16, v4i wv) {
            const int row = (M > 1 && i16 >= kp) + (M > 2 && i16 >= 2 * kp) + (M > 3 && i16 >= 3 * kp);
            const float inv = s_inv[row];
            v4i* px = reinterpret_cast<v4i*>(lds_x + (size_t)i16 * 8);
            const v4i y = rms_out8_bits(*px, wv, inv, (h16)inv);
            *px = y;
            return y;
        };
#pragma unroll
        for (int j = 0; j < AST; j++) {
            const int i16 = t + j * W * 64;
            if (i16 < npieces) a_stage[j] = norm_piece(i16, w_stage[j]);
            else a_stage[j] = zero4;
        }
#pragma unroll 1
        for (int i16 = t + AST * W * 64; i16 < npieces; i16 += W * 64) norm_piece(i16, ldg(Wb + (size_t)wpiece(i16) * 16));
    }
    // the finished activation piece i16 beyond the register stage
    auto late_piece = [&](int i16) -> v4i {
        if constexpr (PRO == 1) {
            return *reinterpret_cast<const v4i*>(lds_x + (size_t)i16 * 8);
        } else if constexpr (PRO == 2 || PRO == 3) {
            return act_mul8_bits<PRO == 2 ? OP_SILU : OP_GELU>(ldg(reinterpret_cast<const char*>(X) + (size_t)i16 * 16),
                                                               ldg(reinterpret_cast<const char*>(pro.x2) + (size_t)i16 * 16));
        } else {
            return ldg(reinterpret_cast<const char*>(X) + (size_t)i16 * 16);
        }
    };
    {
        float mx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto row_of = [&](int i16) {   // M <= 4: compares instead of a division
            const int e = i16 * 8;
            return min((M > 1 && e >= K) + (M > 2 && e >= 2 * K) + (M > 3 && e >= 3 * K), 3);
        };
        auto fold = [&](int i16, v4i x) {
            const int row = row_of(i16);
            const float v = absmax8_bits(x);
#pragma unroll
            for (int m = 0; m < 4; m++) mx[m] = (row == m) ? fmaxf(mx[m], v) : mx[m];
        };
#pragma unroll
        for (int j = 0; j < AST; j++) fold(t + j * W * 64, a_stage[j]);
#pragma unroll 1
        for (int i16 = t + AST * W * 64; i16 < npieces; i16 += W * 64) fold(i16, late_piece(i16));
#pragma unroll
        for (int m = 0; m < 4; m++)
            if (m < M) mx[m] = wave_max(mx[m]);
        FQ_STAMP(2);    // the activation rows have arrived (and, with a producer prologue, have been produced)
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < 4; m++) wmax[w * 4 + m] = mx[m];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float inv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (m < M) {
                float rmax = wmax[m];
#pragma unroll
                for (int ww = 1; ww < W; ww++) rmax = fmaxf(rmax, wmax[ww * 4 + m]);
                float sc = rmax / 127.0f;
                if (sc < 1e-10f) sc = 1e-10f;
                inv[m] = 1.0f / sc;
            }
        }
        auto quant_piece = [&](int i16, v4i x) {
            const int row = row_of(i16);
            const float iv = row == 0 ? inv[0] : (row == 1 ? inv[1] : (row == 2 ? inv[2] : inv[3]));
            const uint2 qv = quant8_bits(x, iv);
            if constexpr (INT4 && SK_A_PRESPLIT) {
                // a packed weight byte holds elements 2j (low nibble) and 2j + 1 of its row, so the MFMAs take the even
                // and the odd elements of 32 activations separately: every 32-byte span is stored as [16 even | 16 odd]
                // once here instead of being split by every wave at every unit (K % 32 == 0: a piece stays in its row)
                int8_t* span = lds_a + (((size_t)i16 * 8) & ~(size_t)31);
                const int p0 = (i16 * 8) & 31;
                *reinterpret_cast<uint32_t*>(span + (p0 >> 1)) = __builtin_amdgcn_perm(qv.y, qv.x, 0x06040200u);
                *reinterpret_cast<uint32_t*>(span + 16 + (p0 >> 1)) = __builtin_amdgcn_perm(qv.y, qv.x, 0x07050301u);
            } else {
                *reinterpret_cast<uint2*>(lds_a + (size_t)i16 * 8) = qv;
            }
        };
#pragma unroll
        for (int j = 0; j < AST; j++) {
            const int i16 = t + j * W * 64;
            if (i16 < npieces) quant_piece(i16, a_stage[j]);
        }
#pragma unroll 1
        for (int i16 = t + AST * W * 64; i16 < npieces; i16 += W * 64) quant_piece(i16, late_piece(i16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    FQ_STAMP(1);
    if constexpr (WT) {
        const int rounds = nunits / P;               // K % 1024 == 0: whole rounds of 8 groups
        for (int rd = 0; rd + 1 < rounds; rd++) {
            sc_cur = sc_next;
            sc_next = load_scale_chunk(rd + 1);
#pragma unroll
            for (int p = 0; p < P; p++) {
                const int ul = rd * P + p;
                compute_unit(ul, ring[p], p);
                load_unit(ul + P, boff, ring[p]);
            }
        }
        sc_cur = sc_next;
#pragma unroll
        for (int p = 0; p < P; p++) compute_unit((rounds - 1) * P + p, ring[p], p);
        // rows m = 4 q + i of the accumulator: the lanes of q = 0 hold the M <= 4 real ones
        if (q == 0 && n0 + r < N) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i < M) {
                    float rmax = wmax[i];
#pragma unroll
                    for (int ww = 1; ww < W; ww++) rmax = fmaxf(rmax, wmax[ww * 4 + i]);
                    float as = rmax / 127.0f;
                    if (as < 1e-10f) as = 1e-10f;
                    if (scales_out != nullptr && wtile == 0 && r == 0) scales_out[i] = as;
                    C[(size_t)i * N + n0 + r] = f2h_mul(run[i], as);
                }
            }
        }
#ifdef FQ_TRACE
        fqt_finish();
#endif
        return;
    }
    const int rounds = (nunits + W * P - 1) / (W * P);
    for (int rd = 0; rd + 1 < rounds; rd++) {    // no VMEM under a branch: counted waits
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int ul = w + (rd * P + p) * W;
            compute_unit(ul, ring[p], p);
            load_unit(ul + P * W, boff, ring[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++) compute_unit(w + ((rounds - 1) * P + p) * W, ring[p], p);

    FQ_STAMP(3);
    // the tile's scales -> LDS as fp32 (x 1/16 for INT4)
    if (fast) {
#pragma unroll
        for (int j = 0; j < CT; j++) {
            const int ch = t + j * W * 64;
            if (ch < TR * G / 8) {
                const int row = (ch * 8) / G, g = (ch * 8) - row * G;
                float* dst = scl + row * GP + g;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t x = (uint32_t)sc_regs.v[j][i];
                    const float lo = h2f((uint16_t)(x & 0xFFFFu)), hi = h2f((uint16_t)(x >> 16));
                    dst[2 * i] = INT4 ? lo * 0.0625f : lo;
                    dst[2 * i + 1] = INT4 ? hi * 0.0625f : hi;
                }
            }
        }
    } else {
        for (int idx = t; idx < TR * G; idx += W * 64) {
            const int row = idx / G, g = idx - row * G;
            const float sv = h2f(B_scales[(size_t)min(n0 + row, N - 1) * G + g]);
            scl[row * GP + g] = INT4 ? sv * 0.0625f : sv;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    FQ_STAMP(4);
    // wave ct runs the chains of column tile ct (W >= CT): thread (m, n) walks its groups in order
    const int ct = w, e = lane;
    if (ct < CT && e < M * 16) {
        const float* sp = scl + (ct * SK_TN + (e & 15)) * GP;
        const float* gp = gbuf + ((size_t)ct * M * 16 + e) * SCP;
        float run = 0.0f;
        int gl = 0;
        for (; gl + 16 <= G; gl += 16) {   // eight ds_read_b128, then the serial fma chain
            v4f gv[4], sv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                gv[u] = *reinterpret_cast<const v4f*>(gp + gl + 4 * u);
                sv[u] = *reinterpret_cast<const v4f*>(sp + gl + 4 * u);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) run = fmaf(gv[u >> 2][u & 3], sv[u >> 2][u & 3], run);
        }
        for (; gl < G; gl++) run = fmaf(gp[gl], sp[gl], run);
        const int cm = e >> 4, cn = e & 15;
        float rmax = wmax[cm];
#pragma unroll
        for (int ww = 1; ww < W; ww++) rmax = fmaxf(rmax, wmax[ww * 4 + cm]);
        float as = rmax / 127.0f;
        if (as < 1e-10f) as = 1e-10f;
        if (scales_out != nullptr && tl == 0 && ct == 0 && cn == 0) scales_out[cm] = as;
        if (n0 + ct * SK_TN + cn < N) C[(size_t)cm * N + n0 + ct * SK_TN + cn] = f2h_mul(run, as);
    }
#ifdef FQ_TRACE
    fqt_finish();
#endif
}

template <bool INT4, int W, int P, int CT>
__global__ __launch_bounds__(W * 64) void skinny_fq_kernel(const uint16_t* __restrict__ X,
                                                          float* __restrict__ scales_out,
                                                          const uint8_t* __restrict__ B,
                                                          const uint16_t* __restrict__ B_scales,
                                                          uint16_t* __restrict__ C, int M, int K, int N, int pk) {
    extern __shared__ __attribute__((aligned(16))) float fq_lds[];
    fq_body<INT4, W, P, CT>(X, scales_out, B, B_scales, C, M, K, N, (int)blockIdx.x, (int)threadIdx.x, fq_lds, (pk & 1) != 0,
                            FqPro(), (pk >> 1) - 1);
}

// Several independent FP16-activation matmuls (M <= 4 each) in one launch: the projections
// that read one activation row (q/k/v, gate/up), or any set of small decode matmuls whose
// individual launches would each be latency-bound.  Workgroup -> (problem, 16-row tile) through
// the running tile counts in the argument block; every problem runs the fused-quantization
// body with its own K.
constexpr int SK_MAXB = 16;
struct SkinnyBatch {
    int count;
    int wt[SK_MAXB];                // sub-tiled launch: waves per tile (1, 2 or 4), per problem
    int tile_end[SK_MAXB];          // running sum of workgroups per problem
    int lds_floats[SK_MAXB];        // sub-tiled launch: floats of LDS per wave group
    const void* X[SK_MAXB];
    const void* B[SK_MAXB];
    const uint16_t* Bs[SK_MAXB];
    uint16_t* C[SK_MAXB];
    int M[SK_MAXB], K[SK_MAXB], N[SK_MAXB];
    unsigned pk_mask;               // bit i: problem i's B is the tile-major twin
    int trace_id;                   // diagnostic builds: the launch's index in the stamp table (-1: none)
    FqPro pro;                      // producer prologue operands (shared by every problem of the launch)
};

template <bool INT4, int W, int P, int CT>
#ifndef SK_WPE
#define SK_WPE 0
#endif
#ifndef SK_BATCH_P
#define SK_BATCH_P 4
#endif
// Ring depth of the batched launches on packed INT4 weights.  A wave walks its units in whole rounds of the ring, and
// a packed row has half the units: K = 1024 split over two waves is 2 units per wave, which a ring of 4 pads with two
// empty ones (their loads return zeros, their MFMAs still run).  16-shape INT4 sweep by depth 1 / 2 / 3 / 4 / 8:
// 25.4 / 25.2 / 29.5 / 27.9 / 39.4 us (INT8 at 2 / 3 / 4: 39.8 / 40.2 / 39.8).
#ifndef SK_BATCH_P4
#define SK_BATCH_P4 2
#endif
#if SK_WPE
#define SK_WPE_ATTR __attribute__((amdgpu_waves_per_eu(SK_WPE, SK_WPE)))
#else
#define SK_WPE_ATTR
#endif
__global__ __launch_bounds__(W * 64) SK_WPE_ATTR void skinny_batch_kernel(const SkinnyBatch bt) {
    const int blk = (int)blockIdx.x;
    int pi = 0;
    while (pi + 1 < bt.count && blk >= bt.tile_end[pi]) pi++;
    const int tile = blk - (pi ? bt.tile_end[pi - 1] : 0);
    extern __shared__ __attribute__((aligned(16))) float fq_lds[];
    fq_body<INT4, W, P, CT>(reinterpret_cast<const uint16_t*>(bt.X[pi]), nullptr,
                        reinterpret_cast<const uint8_t*>(bt.B[pi]), bt.Bs[pi], bt.C[pi], bt.M[pi], bt.K[pi],
                        bt.N[pi], tile, (int)threadIdx.x, fq_lds, (bt.pk_mask >> pi) & 1u, FqPro(), bt.trace_id);
}

// The batched launch with wave tiles: a workgroup of 4 waves quantizes its problem's activation rows once
// and every wave streams one of 4 neighbouring tiles.  tile_end counts workgroups (4 tiles each).
template <bool INT4>
__global__ __launch_bounds__(256) void skinny_batch_wave_kernel(const SkinnyBatch bt) {
    const int blk = (int)blockIdx.x;
    int pi = 0;
    while (pi + 1 < bt.count && blk >= bt.tile_end[pi]) pi++;
    const int tile4 = blk - (pi ? bt.tile_end[pi - 1] : 0);
    extern __shared__ __attribute__((aligned(16))) float fq_lds[];
    fq_body<INT4, 4, INT4 ? 4 : 8, 1, 0, true>(reinterpret_cast<const uint16_t*>(bt.X[pi]), nullptr,
                                    reinterpret_cast<const uint8_t*>(bt.B[pi]), bt.Bs[pi], bt.C[pi], bt.M[pi],
                                    bt.K[pi], bt.N[pi], tile4, (int)threadIdx.x, fq_lds, (bt.pk_mask >> pi) & 1u, FqPro(), bt.trace_id);
}

// The same launch with a workgroup of 4 waves split into 4 / wt wave groups, each on its own tile:
// a short row (few units) cannot feed 4 waves, and with one wave per tile four times as many
// independent tiles are in flight per CU, which is what hides the per-tile latencies of small K.
// wt (waves per tile: 1, 2 or 4) is per problem; the wave groups of a workgroup work on tiles of
// one problem, so they execute the same barriers.  A wave group past the problem's last tile runs
// the body on an out-of-range tile: its loads return zeros and it stores nothing.
template <bool INT4, int P, int PRO = 0>
__global__ __launch_bounds__(256) void skinny_batch_sub_kernel(const SkinnyBatch bt) {
    extern __shared__ __attribute__((aligned(16))) float fq_lds[];
    const int blk = (int)blockIdx.x;
    int pi = 0;
    while (pi + 1 < bt.count && blk >= bt.tile_end[pi]) pi++;
    const int wt = bt.wt[pi];
    const int wave = (int)threadIdx.x >> 6;
    const int sub = wave / wt, per = 4 / wt;
    const int t = (int)threadIdx.x - sub * wt * 64;
    const int tile = (blk - (pi ? bt.tile_end[pi - 1] : 0)) * per + sub;
    float* lds = fq_lds + (size_t)sub * bt.lds_floats[pi];
    const uint16_t* X = reinterpret_cast<const uint16_t*>(bt.X[pi]);
    const uint8_t* B = reinterpret_cast<const uint8_t*>(bt.B[pi]);
    const bool pk = (bt.pk_mask >> pi) & 1u;
    if (wt == 1) fq_body<INT4, 1, P, 1, PRO>(X, nullptr, B, bt.Bs[pi], bt.C[pi], bt.M[pi], bt.K[pi], bt.N[pi], tile, t, lds, pk, bt.pro, bt.trace_id);
    else if (wt == 2) fq_body<INT4, 2, P, 1, PRO>(X, nullptr, B, bt.Bs[pi], bt.C[pi], bt.M[pi], bt.K[pi], bt.N[pi], tile, t, lds, pk, bt.pro, bt.trace_id);
    else fq_body<INT4, 4, P, 1, PRO>(X, nullptr, B, bt.Bs[pi], bt.C[pi], bt.M[pi], bt.K[pi], bt.N[pi], tile, t, lds, pk, bt.pro, bt.trace_id);
}

// Row-major weights -> the tile-major twin (layout above); one thread per 16-byte piece.  rows16: rows of the twin
// (N rounded up to 16); pieces of rows past N are zeros.
__global__ __launch_bounds__(256) void pack_weights_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                           size_t N, size_t rows16, size_t row_bytes) {
    const size_t units = row_bytes / 128;
    const size_t total = rows16 * row_bytes / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t lane = i & 63, h = (i >> 6) & 1, u = (i >> 7) % units, tile = (i >> 7) / units;
        const size_t r = lane & 15, q = lane >> 4, n = tile * 16 + r;
        v4i v = {0, 0, 0, 0};
        if (n < N) v = *reinterpret_cast<const v4i*>(src + n * row_bytes + u * 128 + h * 64 + q * 16);
        *reinterpret_cast<v4i*>(dst + i * 16) = v;
    }
}

hipError_t launch_pack_weights(size_t row_bytes, const void* B, void* packed, size_t N, hipStream_t st) {
    if (row_bytes == 0 || row_bytes % 128 != 0 || N == 0) return hipErrorInvalidValue;
    const size_t rows16 = (N + 15) & ~size_t(15);
    const size_t total = rows16 * row_bytes / 16;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 65536);
    pack_weights_kernel<<<grid, 256, 0, st>>>(static_cast<const uint8_t*>(B), static_cast<uint8_t*>(packed), N, rows16,
                                              row_bytes);
    return hipGetLastError();
}

#ifdef FQ_TRACE
extern "C" int cactus_hip_debug_fq_trace(unsigned long long* out, unsigned int* launches) {
    *launches = (unsigned int)g_fq_trace_next;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fq_trace), sizeof(unsigned long long) * FQT_LAUNCHES * FQT_WGS * FQT_PH);
}
#endif

int g_skinny_ablate = 0;   // benchmarks only: bit0 skip chain, bit1 skip LDS writes, bit2 skip scale staging,
                           // bit4 record phase stamps (read back with skinny_read_trace)
hipError_t skinny_read_trace(unsigned long long* out512) {
    return hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_sk_trace), sizeof(unsigned long long) * 512);
}

namespace {

template <bool INT4, int W, int P, bool ALDS>
hipError_t launch_cfg(const int8_t* A, const float* A_scales_c, const void* B, const uint16_t* B_scales,
                      uint16_t* C, int32_t* gsums, size_t M, size_t K, size_t N, hipStream_t st, bool pk = false) {
    const float* A_scales = A_scales_c;
    const int G = (int)(K / 128);
    constexpr int GPU_ = INT4 ? 2 : 1;
    if (INT4 && (G & 1)) return hipErrorInvalidValue;   // packed units hold two groups
    if (((N + 15) & ~size_t(15)) * (INT4 ? K / 2 : K) > (size_t(1) << 31) || M * K > (size_t(1) << 31)) return hipErrorInvalidValue;
    // chunk of groups whose sums fit a 16 KB (M <= 4) / 32 KB LDS budget: a multiple of W units
    const int step = W * GPU_;
#ifndef SK_GS_KB
#define SK_GS_KB 72
#endif
    int SC = (int)(((M > 4 ? SK_GS_KB : 16) * 1024) / (M * 64));
    SC = (SC / step) * step;
    if (SC < step) SC = step;
    while (SC % 4) SC += step;   // chunk starts stay 16-byte aligned in the fp32 LDS rows
    const int Gr = ((G + step - 1) / step) * step;
    if (SC > Gr) SC = Gr;
    size_t lds = ((size_t)M * 16 * (SC + 4) + (size_t)SK_TN * (((G + 3) & ~3) + 4)) * sizeof(float);
    if (ALDS) lds += ((M * K + 15) & ~(size_t)15);
    if (ALDS && (SC < Gr || lds > 64 * 1024)) return hipErrorInvalidValue;   // single chunk only
    const dim3 grid((unsigned)((N + SK_TN - 1) / SK_TN)), block(W * 64);
    if (lds > 64 * 1024) {   // more than 64 KB of dynamic LDS has to be allowed once per device
        static bool attr_set[2][64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[gsums != nullptr][dev]) {
            const void* fn = gsums != nullptr ? reinterpret_cast<const void*>(skinny_kernel<INT4, W, P, ALDS, true>)
                                              : reinterpret_cast<const void*>(skinny_kernel<INT4, W, P, ALDS, false>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_set[gsums != nullptr][dev] = true;
        }
    }
    if (gsums != nullptr)
        skinny_kernel<INT4, W, P, ALDS, true><<<grid, block, lds, st>>>(
            A, A_scales, (const uint8_t*)B, B_scales, C, gsums, (int)M, (int)K, (int)N, G, SC, g_skinny_ablate, (int)pk);
    else
        skinny_kernel<INT4, W, P, ALDS, false><<<grid, block, lds, st>>>(
            A, A_scales, (const uint8_t*)B, B_scales, C, gsums, (int)M, (int)K, (int)N, G, SC, g_skinny_ablate, (int)pk);
    return hipGetLastError();
}

}  // namespace

#ifndef SK_FQ_UPW4
#define SK_FQ_UPW4 4
#endif
#ifndef SK_FQ_UPW8
#define SK_FQ_UPW8 8
#endif
static int skinny_waves(bool int4, size_t K) {
    const size_t nu = K / (int4 ? 256 : 128);
    int W = 1;
    while (W < 8 && (size_t)W * (int4 ? SK_FQ_UPW4 : SK_FQ_UPW8) <= nu) W *= 2;
    return W;
}

bool skinny_fused_quant_ok(bool int4, size_t M, size_t K) {
    const size_t G = K / 128;
    const size_t lds = (M * 16 * (G + 8 + 4) + 16 * (G + 8)) * sizeof(float) + M * K + 256;
    return M >= 1 && M <= 4 && K % 128 == 0 && (!int4 || K % 256 == 0) && lds <= 60 * 1024;
}

static size_t fq_lds_bytes(bool int4, int W, int CT, size_t M, size_t K) {
    const size_t G = K / 128, step = (size_t)W * (int4 ? 2 : 1);
    const size_t SC = ((G + step - 1) / step) * step;
    return ((size_t)CT * M * 16 * (SC + 4) + (size_t)CT * SK_TN * (((G + 3) & ~(size_t)3) + 4)) * sizeof(float) +
           ((M * K + 15) & ~(size_t)15) + (size_t)W * 4 * sizeof(float);
}

// Column tiles per tile.  2 (32 weight rows per workgroup, -DSK_CT=2) halves the quantization
// prologues per byte streamed but needs 180 registers (2 waves per SIMD instead of 4) and
// measured slower on the 16-shape sweep: 3.5 TB/s at best (ring depth 2) against 4.3 at 1.
#ifndef SK_CT
#define SK_CT 1
#endif

// Waves per tile and MFMA column tiles per tile for `rows16` groups of 16 weight rows whose longest
// row is K.  Waves: as many as the row can feed (about 4 units each), but with plenty of tiles fewer
// waves per tile keep more independent tiles in flight and shorten the waits at the tile barriers
// (16-shape sweep, 3840 tiles: 8 waves 3.5 TB/s, 4 waves 4.1, 2 waves 4.0, 1 wave 3.4).
static void fq_shape(bool int4, size_t kmax, size_t rows16, int* W_out, int* ct_out) {
    int W = skinny_waves(int4, kmax);
    int ct = SK_CT;
    if (W < ct) ct = 1;
    while (W > ct && (rows16 / ct) * (size_t)W > 16384) W /= 2;
    *W_out = W;
    *ct_out = ct;
}

int g_skinny_ct = 1;    // benchmarks only: column tiles per tile of the 2xxxx variants

static hipError_t launch_fq(bool int4, int W, int ct, const uint16_t* X, float* scales_out, const void* B,
                            const uint16_t* B_scales, uint16_t* C, size_t M, size_t K, size_t N, hipStream_t st,
                            bool pk = false) {
    if (!skinny_fused_quant_ok(int4, M, K)) return hipErrorInvalidValue;
    if (((N + 15) & ~size_t(15)) * (int4 ? K / 2 : K) > (size_t(1) << 31)) return hipErrorInvalidValue;
    if (ct != SK_CT || W < 2 || fq_lds_bytes(int4, W, ct, M, K) > 64 * 1024) ct = 1;
    const size_t lds = fq_lds_bytes(int4, W, ct, M, K);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const size_t tiles = (N + SK_TN * ct - 1) / (SK_TN * ct);
    const dim3 grid((unsigned)tiles), block(W * 64);
    int pk_arg = pk ? 1 : 0;        // bit 0: tile-major twin; diagnostic builds: (stamp-table index + 1) above it
#ifdef FQ_TRACE
    pk_arg |= (g_fq_trace_next++ + 1) << 1;
#endif
// Ring depth of the single fused-quantization launches (units in flight per wave) and the least units a wave takes
// before a tile gets another wave.  M = 1, tile-major twins, by (depth, least units): INT4 8192^2 (4, 8) 10.7 us,
// (2, 8) 9.85, (2, 4) 9.8, (1, 4) 9.8, (3, 6) 11.4; 2048 x 8192 6.5 / 6.2 / 5.5 / 5.2 / 7.2; INT8 8192^2 (4, 8) 13.5,
// (2, 8) 13.0, (2, 4) 12.95; 2048 x 8192 5.9 / 5.55 / 6.0.
#ifndef SK_FQ_P4
#define SK_FQ_P4 2
#endif
#ifndef SK_FQ_P8
#define SK_FQ_P8 2
#endif
#define FQ_LAUNCH(I4, WW, CC)                                                                                     \
    skinny_fq_kernel<I4, WW, (I4 ? SK_FQ_P4 : SK_FQ_P8), CC><<<grid, block, lds, st>>>(X, scales_out, (const uint8_t*)B, B_scales, C,       \
                                                             (int)M, (int)K, (int)N, pk_arg)
#define FQ_CASE(WW)                                                        \
    case WW:                                                                \
        if (ct == 2 && WW >= 2) {                                           \
            if (int4) FQ_LAUNCH(true, WW, (WW >= 2 ? SK_CT : 1));           \
            else FQ_LAUNCH(false, WW, (WW >= 2 ? SK_CT : 1));               \
        } else {                                                            \
            if (int4) FQ_LAUNCH(true, WW, 1);                               \
            else FQ_LAUNCH(false, WW, 1);                                   \
        }                                                                   \
        break;
    switch (W) {
        FQ_CASE(1) FQ_CASE(2) FQ_CASE(4) FQ_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef FQ_CASE
#undef FQ_LAUNCH
    return hipGetLastError();
}

// variant: 0 = default choice; otherwise [1 if ALDS]*10000 + W*100 + P (benchmarks only).
hipError_t launch_matmul_skinny_variant(int variant, bool int4, const int8_t* A, const float* A_scales,
                                        const void* B, const uint16_t* B_scales, uint16_t* C, int32_t* gsums,
                                        size_t M, size_t K, size_t N, hipStream_t st) {
#define SK_CASE(WW, PP)                                                                                \
    case WW * 100 + PP:                                                                                 \
        return int4 ? launch_cfg<true, WW, PP, false>(A, A_scales, B, B_scales, C, gsums, M, K, N, st, pk)  \
                    : launch_cfg<false, WW, PP, false>(A, A_scales, B, B_scales, C, gsums, M, K, N, st, pk); \
    case 10000 + WW * 100 + PP:                                                                         \
        return int4 ? launch_cfg<true, WW, PP, true>(A, A_scales, B, B_scales, C, gsums, M, K, N, st, pk)   \
                    : launch_cfg<false, WW, PP, true>(A, A_scales, B, B_scales, C, gsums, M, K, N, st, pk);
    bool pk = false;                           // + 40000: B is the tile-major twin
    if (variant >= 40000 && variant < 80000) { pk = true; variant -= 40000; }
    if (variant >= 20000 && variant < 30000)   // fused quantization: A holds FP16 activations; W from the code
        return launch_fq(int4, (variant - 20000) / 100, g_skinny_ct, reinterpret_cast<const uint16_t*>(A), nullptr, B,
                         B_scales, C, M, K, N, st, pk);
    switch (variant) {
        SK_CASE(4, 2) SK_CASE(4, 4) SK_CASE(4, 8) SK_CASE(8, 2) SK_CASE(8, 4) SK_CASE(8, 8)
        SK_CASE(16, 2) SK_CASE(16, 4) SK_CASE(2, 4) SK_CASE(2, 8) SK_CASE(8, 1) SK_CASE(4, 1) SK_CASE(2, 2)
        SK_CASE(16, 1) SK_CASE(1, 4) SK_CASE(1, 8) SK_CASE(1, 2) SK_CASE(1, 1) SK_CASE(2, 1)
        default: return hipErrorInvalidValue;
    }
#undef SK_CASE
}

// FP16 activations in, quantized inside the kernel (M <= 4 and skinny_fused_quant_ok only).
// One wave per tile pays for launches of 2048 tiles or more whose rows are at most 1024 (INT4: 2048) bytes:
// measured on M = 1, N = 151936, K = 1024: 0.64 of 8 TB/s against 0.61 (INT4: 0.53 / 0.48); N = 32768, K = 4096
// INT4: 0.52 / 0.45; K = 8192: 0.53 against 0.63 the other way round.
static bool wave_tiles_pay(bool int4, size_t K, size_t tiles) {
    return tiles >= 2048 && K % 1024 == 0 && (int4 ? K <= 4096 : K <= 1024);
}

hipError_t launch_matmul_skinny_fq(bool int4, const uint16_t* X, float* scales_out, const void* B,
                                   const uint16_t* B_scales, uint16_t* C, size_t M, size_t K, size_t N,
                                   hipStream_t st, bool packed) {
    if (scales_out == nullptr && wave_tiles_pay(int4, K, (N + SK_TN - 1) / SK_TN)) {
        const SkinnyProblem one{X, B, B_scales, C, M, K, N, packed};     // wave tiles (the batched launch decides)
        return launch_matmul_skinny_fq_batch(int4, &one, 1, st);
    }
    int W, ct;
    fq_shape(int4, K, (N + SK_TN - 1) / SK_TN, &W, &ct);
    return launch_fq(int4, W, ct, X, scales_out, B, B_scales, C, M, K, N, st, packed);
}

// Batched fused launch: every problem must satisfy skinny_fused_quant_ok and the alignment the
// single launch needs; count <= SK_MAXB (the caller splits longer lists).
// Waves per tile of the wave-group launches, by the bytes of a weight row: a short row cannot feed 4 waves, and with
// fewer waves per tile more independent tiles are in flight per CU.  Measured on the 16-shape sweeps from HBM (four /
// six weight sets in turn): INT8 rows of 1024 bytes on one wave and of 2048 on two 39.3 us against 40.4 with two and
// four; packed INT4 the other way (26.6 us against 27.2 - 27.7).
#ifndef SK_SUB_K1
#define SK_SUB_K1 1024   // INT8 weight rows up to this many bytes take one wave per tile
#endif
#ifndef SK_SUB_K2
#define SK_SUB_K2 2048   // ... up to this, two waves per tile
#endif
#ifndef SK_SUB_K1_4
#define SK_SUB_K1_4 0    // the same for packed INT4 rows
#endif
#ifndef SK_SUB_K2_4
#define SK_SUB_K2_4 1024
#endif
// total_tiles: 16-row tiles of the whole launch; one wave per tile only pays when there are thousands (the launches of a
// batch-1 decode step, a few hundred tiles each, take 1.13 ms per step with it against 1.04)
static int sub_waves(bool int4, size_t K, size_t total_tiles) {
    const size_t kb = int4 ? K / 2 : K;
    const size_t k1 = int4 ? SK_SUB_K1_4 : SK_SUB_K1, k2 = int4 ? SK_SUB_K2_4 : SK_SUB_K2;
    return (kb <= k1 && total_tiles >= 2048) ? 1 : (kb <= k2 ? 2 : 4);
}

hipError_t launch_matmul_skinny_fq_batch(bool int4, const SkinnyProblem* pr, size_t count, hipStream_t st) {
    if (count == 0) return hipSuccess;
    if (count > (size_t)SK_MAXB) return hipErrorInvalidValue;
    SkinnyBatch bt;
    memset(&bt, 0, sizeof bt);
    bt.count = (int)count;
    bt.trace_id = -1;
#ifdef FQ_TRACE
    bt.trace_id = g_fq_trace_next++;
#endif
    size_t tiles = 0, lds = 0, kmax = 0;
    for (size_t i = 0; i < count; i++)
        if (pr[i].K > kmax) kmax = pr[i].K;
    size_t total_tiles = 0;
    for (size_t i = 0; i < count; i++) total_tiles += (pr[i].N + SK_TN - 1) / SK_TN;
    int W, ct;
    fq_shape(int4, kmax, total_tiles, &W, &ct);
#ifdef SK_BATCH_W
    W = SK_BATCH_W;
#endif
    if (ct == 2)
        for (size_t i = 0; i < count; i++)
            if (W < 2 || fq_lds_bytes(int4, W, 2, pr[i].M, pr[i].K) > 64 * 1024) ct = 1;
    // Longest rows first: a workgroup's run time grows with K, so the long ones start first and
    // the short ones fill the tail of the launch (16-shape sweep: 51 us against 56 in K*N order).
    size_t order[SK_MAXB];
    for (size_t i = 0; i < count; i++) order[i] = i;
    std::stable_sort(order, order + count, [&](size_t a, size_t b) { return pr[a].K > pr[b].K; });
    for (size_t i = 0; i < count; i++) {
        const SkinnyProblem& q = pr[order[i]];
        if (!skinny_fused_quant_ok(int4, q.M, q.K)) return hipErrorInvalidValue;
        if (((q.N + 15) & ~size_t(15)) * (int4 ? q.K / 2 : q.K) > (size_t(1) << 31)) return hipErrorInvalidValue;
        if (q.packed) bt.pk_mask |= 1u << i;
        const size_t l = fq_lds_bytes(int4, W, ct, q.M, q.K);
        if (l > 64 * 1024) return hipErrorInvalidValue;
        if (l > lds) lds = l;
        tiles += (q.N + SK_TN * ct - 1) / (SK_TN * ct);
        if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
        bt.tile_end[i] = (int)tiles;
        bt.X[i] = q.X; bt.B[i] = q.B; bt.Bs[i] = q.Bs; bt.C[i] = q.C;
        bt.M[i] = (int)q.M; bt.K[i] = (int)q.K; bt.N[i] = (int)q.N;
    }
    if (tiles == 0) return hipSuccess;
    // Thousands of tiles: one wave per tile, the fp32 chain in registers (no LDS hand-over, no tile barrier)
    {
        bool wave_tiles = true;
        size_t blocks = 0, lds_w = 0;
        for (size_t i = 0; wave_tiles && i < count; i++) {
            const SkinnyProblem& q = pr[order[i]];
            wave_tiles = wave_tiles_pay(int4, q.K, total_tiles) && (reinterpret_cast<uintptr_t>(q.Bs) & 15) == 0;
            lds_w = std::max(lds_w, ((q.M * q.K + 15) & ~(size_t)15) + (4 * 4 + 4) * sizeof(float) + 16);
            blocks += ((q.N + SK_TN - 1) / SK_TN + 3) / 4;
            bt.tile_end[i] = (int)blocks;
        }
        if (wave_tiles && lds_w <= 64 * 1024) {
            if (int4) skinny_batch_wave_kernel<true><<<dim3((unsigned)blocks), dim3(256), lds_w, st>>>(bt);
            else skinny_batch_wave_kernel<false><<<dim3((unsigned)blocks), dim3(256), lds_w, st>>>(bt);
            return hipGetLastError();
        }
        size_t tt = 0;       // not taken: back to one tile per workgroup
        for (size_t i = 0; i < count; i++) {
            tt += (pr[order[i]].N + SK_TN * ct - 1) / (SK_TN * ct);
            bt.tile_end[i] = (int)tt;
        }
    }
    // Four waves per tile: the workgroup of 4 waves is cut into wave groups per problem, so that
    // short rows (few units) get fewer waves per tile and more tiles in flight.
    if (W == 4 && ct == 1) {
        size_t blocks = 0, lds_sub = 0;
        for (size_t i = 0; i < count; i++) {
            const SkinnyProblem& q = pr[order[i]];
            const int wt = sub_waves(int4, q.K, total_tiles);
            const size_t l = (fq_lds_bytes(int4, wt, 1, q.M, q.K) + 15) & ~(size_t)15;
            bt.wt[i] = wt;
            bt.lds_floats[i] = (int)(l / sizeof(float));
            lds_sub = std::max(lds_sub, l * (4 / wt));
            const size_t ptiles = (q.N + SK_TN - 1) / SK_TN, per = 4 / wt;
            blocks += (ptiles + per - 1) / per;
            bt.tile_end[i] = (int)blocks;
        }
        if (lds_sub <= 64 * 1024) {
            if (int4) skinny_batch_sub_kernel<true, SK_BATCH_P4><<<dim3((unsigned)blocks), dim3(256), lds_sub, st>>>(bt);
            else skinny_batch_sub_kernel<false, SK_BATCH_P><<<dim3((unsigned)blocks), dim3(256), lds_sub, st>>>(bt);
            return hipGetLastError();
        }
        size_t tt = 0;   // does not fit: one tile per workgroup
        for (size_t i = 0; i < count; i++) {
            tt += (pr[order[i]].N + SK_TN - 1) / SK_TN;
            bt.tile_end[i] = (int)tt;
        }
    }
    const dim3 grid((unsigned)tiles), block(W * 64);
#define FQB_CASE(WW)                                                                                      \
    case WW:                                                                                               \
        if (ct == 2 && WW >= 2) {                                                                          \
            if (int4) skinny_batch_kernel<true, WW, SK_BATCH_P, (WW >= 2 ? SK_CT : 1)><<<grid, block, lds, st>>>(bt);  \
            else skinny_batch_kernel<false, WW, SK_BATCH_P, (WW >= 2 ? SK_CT : 1)><<<grid, block, lds, st>>>(bt);      \
        } else {                                                                                           \
            if (int4) skinny_batch_kernel<true, WW, SK_BATCH_P, 1><<<grid, block, lds, st>>>(bt);          \
            else skinny_batch_kernel<false, WW, SK_BATCH_P, 1><<<grid, block, lds, st>>>(bt);              \
        }                                                                                                  \
        break;
    switch (W) {
        FQB_CASE(1) FQB_CASE(2) FQB_CASE(4) FQB_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef FQB_CASE
    return hipGetLastError();
}

// The batched fused launch with a producer prologue: every problem reads the same activation rows,
// which each workgroup derives from the operands in front of the matmul.  kind 1: rms_norm(X [+ x2])
// * w (the sum also goes to sum_out), 2: silu(X) * x2, 3: gelu(X) * x2.  Always the wave-group
// kernel (per-problem waves per tile).
hipError_t launch_matmul_skinny_fq_batch_pro(bool int4, int kind, const SkinnyProblem* pr, size_t count,
                                             const uint16_t* x2, const uint16_t* w, uint16_t* sum_out, float eps,
                                             hipStream_t st) {
    if (count == 0) return hipSuccess;
    if (count > (size_t)SK_MAXB || kind < 1 || kind > 3) return hipErrorInvalidValue;
    if (kind == 1 && w == nullptr) return hipErrorInvalidValue;
    if (kind != 1 && x2 == nullptr) return hipErrorInvalidValue;
    SkinnyBatch bt;
    memset(&bt, 0, sizeof bt);
    bt.count = (int)count;
    bt.trace_id = -1;
#ifdef FQ_TRACE
    bt.trace_id = g_fq_trace_next++;
#endif
    bt.pro.x2 = x2; bt.pro.w = w; bt.pro.sum_out = sum_out; bt.pro.eps = eps;
    size_t order[SK_MAXB];
    for (size_t i = 0; i < count; i++) order[i] = i;
    size_t blocks = 0, lds_sub = 0;
    for (size_t i = 0; i < count; i++) {
        const SkinnyProblem& q = pr[order[i]];
        if (q.X != pr[0].X || q.M != pr[0].M || q.K != pr[0].K) return hipErrorInvalidValue;   // one activation matrix
        if (!skinny_fused_quant_ok(int4, q.M, q.K)) return hipErrorInvalidValue;
        if (((q.N + 15) & ~size_t(15)) * (int4 ? q.K / 2 : q.K) > (size_t(1) << 31)) return hipErrorInvalidValue;
        if (q.packed) bt.pk_mask |= 1u << i;
        const int wt = sub_waves(int4, q.K, 0);
        size_t l = fq_lds_bytes(int4, wt, 1, q.M, q.K) + 16;            // + the rows' 1 / rms
        if (kind == 1) {
            l += 2 * q.M * q.K;                                         // + the summed rows (binary16)
            const size_t with_chains = (l + 2 * q.M * q.K + 256 * q.M + 15) & ~(size_t)15;
            if (i == 0) bt.pro.chains = with_chains * (4 / wt) <= 64 * 1024;    // + the same rows chain by chain, if they fit
            if (bt.pro.chains) l = with_chains;
        }
        l = (l + 15) & ~(size_t)15;
        bt.wt[i] = wt;
        bt.lds_floats[i] = (int)(l / sizeof(float));
        lds_sub = std::max(lds_sub, l * (4 / wt));
        const size_t ptiles = (q.N + SK_TN - 1) / SK_TN, per = 4 / wt;
        blocks += (ptiles + per - 1) / per;
        if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
        bt.tile_end[i] = (int)blocks;
        bt.X[i] = q.X; bt.B[i] = q.B; bt.Bs[i] = q.Bs; bt.C[i] = q.C;
        bt.M[i] = (int)q.M; bt.K[i] = (int)q.K; bt.N[i] = (int)q.N;
    }
    if (blocks == 0) return hipSuccess;
    if (lds_sub > 64 * 1024) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(256);
#define FQP_CASE(KIND)                                                                                        \
    case KIND:                                                                                                \
        if (int4) skinny_batch_sub_kernel<true, SK_BATCH_P4, KIND><<<grid, block, lds_sub, st>>>(bt);         \
        else skinny_batch_sub_kernel<false, SK_BATCH_P, KIND><<<grid, block, lds_sub, st>>>(bt);              \
        break;
    switch (kind) { FQP_CASE(1) FQP_CASE(2) FQP_CASE(3) }
#undef FQP_CASE
    return hipGetLastError();
}

bool skinny_fq_pro_ok(bool int4, int kind, size_t M, size_t K) {
    if (!skinny_fused_quant_ok(int4, M, K)) return false;
    const int wt = sub_waves(int4, K, 0);
    size_t l = fq_lds_bytes(int4, wt, 1, M, K) + 16 + (kind == 1 ? 2 * M * K : 0);
    l = (l + 15) & ~(size_t)15;
    return l * (4 / wt) <= 64 * 1024;
}

hipError_t launch_matmul_skinny(bool int4, const int8_t* A, const float* A_scales, const void* B,
                                const uint16_t* B_scales, uint16_t* C, int32_t* gsums, size_t M,
                                size_t K, size_t N, hipStream_t st, bool packed) {
    // Measured on MI355X (scripts/m16_probe.py, M = 16, tile-major twins): ring depth 2, a further wave per 4 units of
    // a row (short rows get few waves per tile: more, smaller workgroups in flight; long rows 8).  Against depth 4 and
    // 8 units per wave: INT4 8192^2 16.3 -> 13.8 us, 151936 x 1024 41.1 -> 34.9, 1024^2 5.8 -> 4.6; INT8 the same
    // within 2 % (1024^2 4.2 -> 3.7).
#ifndef SK_DEF_UPW
#define SK_DEF_UPW 4
#endif
#ifndef SK_DEF_P
#define SK_DEF_P 2
#endif
    const size_t nu = K / (int4 ? 256 : 128);
    int W = 1;
    while (W < 8 && (size_t)W * SK_DEF_UPW <= nu) W *= 2;
    // activations through LDS whenever activations + group sums + scales fit 64 KB
    const size_t G = K / 128;
    const size_t lds_alds = (M * 16 * (G + 8 + 4) + 16 * (G + 8)) * sizeof(float) + M * K;
    const bool alds = M <= 4 && lds_alds <= 60 * 1024;
    return launch_matmul_skinny_variant((alds ? 10000 : 0) + W * 100 + SK_DEF_P + (packed ? 40000 : 0), int4, A, A_scales, B,
                                        B_scales, C, gsums, M, K, N, st);
}

}  // namespace cactus_hip
