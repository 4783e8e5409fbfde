// FP16 matmul C = A * B^T (a half[M,K], b_transposed half[N,K]) on the FP16 matrix cores with
// fp32 accumulation.  Replaces cactus_matmul_f16 (kernel_gemm.cpp:102-126), which accumulates
// in fp16 lanes; results therefore agree to a tolerance, with this kernel the closer of the two
// to the exact product.
//
// 64x64 output tile per 256-thread workgroup (4 waves, each 32x32 = 2x2 MFMA tiles of
// v_mfma_f32_16x16x32_f16), K walked in steps of 32 through LDS rows padded to 80 bytes.
// Any M, N, K: out-of-range rows and the K tail are zero-filled while staging.
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

constexpr int F16_LDS_STRIDE = 40;  // halfs per LDS row (32 + 8 pad, keeps 16-byte alignment)

__global__ __launch_bounds__(256) void matmul_f16_kernel(const uint16_t* __restrict__ A,
                                                        const uint16_t* __restrict__ BT,
                                                        uint16_t* __restrict__ C, int M, int K, int N,
                                                        int vec_ok) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[64 * F16_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) uint16_t sB[64 * F16_LDS_STRIDE];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1, r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;

    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    // staging: thread -> (row = t / 4, 8-half chunk = t % 4) of each 64x32 tile
    const int srow = t >> 2, sch = t & 3;
    for (int k0 = 0; k0 < K; k0 += 32) {
        uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
        const int kk = k0 + sch * 8;
        const int gm = m0 + srow, gn = n0 + srow;
        if (vec_ok && kk + 8 <= K) {
            if (gm < M) va = *reinterpret_cast<const uint4*>(A + (size_t)gm * K + kk);
            if (gn < N) vb = *reinterpret_cast<const uint4*>(BT + (size_t)gn * K + kk);
        } else {
            uint16_t ta[8], tb[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                ta[e] = (gm < M && kk + e < K) ? A[(size_t)gm * K + kk + e] : (uint16_t)0;
                tb[e] = (gn < N && kk + e < K) ? BT[(size_t)gn * K + kk + e] : (uint16_t)0;
            }
            va = make_uint4(ta[0] | (uint32_t)ta[1] << 16, ta[2] | (uint32_t)ta[3] << 16,
                            ta[4] | (uint32_t)ta[5] << 16, ta[6] | (uint32_t)ta[7] << 16);
            vb = make_uint4(tb[0] | (uint32_t)tb[1] << 16, tb[2] | (uint32_t)tb[3] << 16,
                            tb[4] | (uint32_t)tb[5] << 16, tb[6] | (uint32_t)tb[7] << 16);
        }
        __syncthreads();
        *reinterpret_cast<uint4*>(sA + srow * F16_LDS_STRIDE + sch * 8) = va;
        *reinterpret_cast<uint4*>(sB + srow * F16_LDS_STRIDE + sch * 8) = vb;
        __syncthreads();
        v8h fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; i++)
            fa[i] = *reinterpret_cast<const v8h*>(sA + (wm * 32 + i * 16 + r) * F16_LDS_STRIDE + q * 8);
#pragma unroll
        for (int j = 0; j < 2; j++)
            fb[j] = *reinterpret_cast<const v8h*>(sB + (wn * 32 + j * 16 + r) * F16_LDS_STRIDE + q * 8);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int m = m0 + wm * 32 + i * 16 + q * 4 + e, n = n0 + wn * 32 + j * 16 + r;
                if (m < M && n < N) C[(size_t)m * N + n] = f2h(acc[i][j][e]);
            }
}

// dst[c][r] = src[r][c] for a row-major [R, C] half matrix (32x32 tiles through LDS).
__global__ __launch_bounds__(256) void transpose_f16_kernel(const uint16_t* __restrict__ src,
                                                           uint16_t* __restrict__ dst, int R, int C) {
    __shared__ uint16_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(size_t)(r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) dst[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

hipError_t launch_transpose_f16(const uint16_t* src, uint16_t* dst, size_t R, size_t C, hipStream_t st) {
    if (R == 0 || C == 0) return hipSuccess;
    const dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32));
    transpose_f16_kernel<<<grid, dim3(256), 0, st>>>(src, dst, (int)R, (int)C);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Streaming FP16 GEMV: cactus_matmul_f16 with up to 16 activation rows (decode over FP16-stored weights, the
// M = N = 1 dot products of the vector index, engine_index.cpp:18).  Roofline: HBM, 2 * N * K weight bytes.
// The weight stream is the INT8 streaming kernel's: a workgroup of W waves owns 16 weight rows, a unit is 16 rows x
// 128 bytes (64 halves per row), wave w takes units w, w + W, ... with non-temporal 16-byte buffer loads through a
// ring of P units and counted waits; a lane's 16 bytes are row r's halves [64 u + 32 h + 8 q, + 8), which is its
// share of the B operand of v_mfma_f32_16x16x32_f16, and the activation rows (staged once in LDS, rows past M
// zero) are the A operand: the matrix core is a 64-lane dot engine with fp32 accumulation over all of K.  The
// waves' partial tiles meet in LDS and are added in wave order.  pk: the tile-major twin (see matmul_skinny.hip).
template <int W, int P>
__global__ __launch_bounds__(W * 64) void gemv_f16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ BT,
                                                         uint16_t* __restrict__ C, int M, int K, int N, int pk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gv_lds[];
    uint16_t* lds_a = reinterpret_cast<uint16_t*>(gv_lds);                                   // [M][K]
    float* part = reinterpret_cast<float*>(gv_lds + (((size_t)M * K * 2 + 15) & ~(size_t)15));  // [W][16 m][16 n]
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const unsigned row_bytes = (unsigned)K * 2u;
    const unsigned nrows = pk ? (unsigned)((N + 15) & ~15) : (unsigned)N;
    const __amdgpu_buffer_rsrc_t brsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(BT), 0, (int)(nrows * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A), 0, M * K * 2, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const unsigned ustride = pk ? 2048u : 128u, half = pk ? 1024u : 64u;
    const unsigned boff = pk ? ((unsigned)n0 < nrows ? (unsigned)n0 * row_bytes + lane * 16 : OOB)
                             : ((n0 + r < N) ? (unsigned)(n0 + r) * row_bytes + q * 16 : OOB);
    const int nunits = K / 64;
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    auto bload = [&](unsigned voff, unsigned soff) -> v8h {
        const v4u_ v = __builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)voff, (int)soff, 2);
        return __builtin_bit_cast(v8h, v);
    };
    struct Unit { v8h b[2]; };
    Unit ring[P];
    auto load_unit = [&](int u, Unit& R) {
        const unsigned uo = (u < nunits) ? (unsigned)u * ustride : OOB;
        R.b[0] = bload(boff + uo, 0);
        R.b[1] = bload(boff + uo, half);
    };
#pragma unroll
    for (int p = 0; p < P; p++) load_unit(w + p * W, ring[p]);
    // activation rows -> LDS (16-byte pieces; pieces past the end read as zeros)
    const int npieces = M * K / 8;
    for (int i = t; i < npieces; i += W * 64) {
        const v4u_ v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, i * 16, 0, 0);
        *reinterpret_cast<v4u_*>(lds_a + (size_t)i * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool a_live = r < M;
    const uint16_t* aptr = lds_a + (size_t)min(r, M - 1) * K + q * 8;
    const v8h zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    auto compute_unit = [&](int u, const Unit& R) {
        const uint16_t* ap = aptr + (size_t)min(u, nunits - 1) * 64;
        v8h a0 = *reinterpret_cast<const v8h*>(ap), a1 = *reinterpret_cast<const v8h*>(ap + 32);
        a0 = a_live ? a0 : zero8;
        a1 = a_live ? a1 : zero8;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, R.b[0], acc, 0, 0, 0);   // a unit past the row holds zeros
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, R.b[1], acc, 0, 0, 0);
    };
    const int rounds = (nunits + W * P - 1) / (W * P);
    for (int rd = 0; rd + 1 < rounds; rd++) {      // no memory operation under a branch: counted waits
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int ul = w + (rd * P + p) * W;
            compute_unit(ul, ring[p]);
            load_unit(ul + P * W, ring[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++) compute_unit(w + ((rounds - 1) * P + p) * W, ring[p]);
    // lane holds column n = r, rows m = 4 q + i
#pragma unroll
    for (int i = 0; i < 4; i++) part[(w * 16 + q * 4 + i) * 16 + r] = acc[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int e = t; e < M * 16; e += W * 64) {
        const int m = e >> 4, n = e & 15;
        float sum = part[m * 16 + n];
#pragma unroll
        for (int ww = 1; ww < W; ww++) sum += part[(ww * 16 + m) * 16 + n];
        if (n0 + n < N) C[(size_t)m * N + n0 + n] = f2h(sum);
    }
}

bool gemv_f16_ok(const uint16_t* a, const uint16_t* bT, size_t M, size_t K, size_t N) {
    return M >= 1 && M <= 16 && K >= 64 && K % 64 == 0 && M * K * 2 <= 48 * 1024 &&
           ((N + 15) & ~size_t(15)) * K * 2 < (size_t(1) << 31) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0) &&
           ((reinterpret_cast<uintptr_t>(bT) & 15) == 0);
}

hipError_t launch_gemv_f16(const uint16_t* a, const uint16_t* bT, uint16_t* c, size_t M, size_t K, size_t N,
                           hipStream_t st, bool packed) {
    const size_t nunits = K / 64;
    int W = 1;
    while (W < 8 && (size_t)W * 8 <= nunits) W *= 2;      // about 4 units per wave and ring round, as the INT8 stream
    const size_t tiles = (N + 15) / 16;
    while (W > 1 && tiles * W > 16384) W /= 2;
    const size_t lds = ((M * K * 2 + 15) & ~size_t(15)) + (size_t)W * 256 * sizeof(float);
    const dim3 grid((unsigned)tiles);
    switch (W) {
        case 1: gemv_f16_kernel<1, 4><<<grid, 64, lds, st>>>(a, bT, c, (int)M, (int)K, (int)N, (int)packed); break;
        case 2: gemv_f16_kernel<2, 4><<<grid, 128, lds, st>>>(a, bT, c, (int)M, (int)K, (int)N, (int)packed); break;
        case 4: gemv_f16_kernel<4, 4><<<grid, 256, lds, st>>>(a, bT, c, (int)M, (int)K, (int)N, (int)packed); break;
        default: gemv_f16_kernel<8, 4><<<grid, 512, lds, st>>>(a, bT, c, (int)M, (int)K, (int)N, (int)packed); break;
    }
    return hipGetLastError();
}

hipError_t launch_matmul_f16(const uint16_t* a, const uint16_t* bT, uint16_t* c, size_t M, size_t K,
                             size_t N, hipStream_t st) {
    // the pipelined tile kernel (matmul_gemm.hip) when the rows are whole 128-byte groups; else the plain kernel below
    hipError_t te = hipSuccess;
    if (launch_matmul_f16_tiles(a, bT, c, M, K, N, st, &te)) return te;
    const int vec_ok = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(bT) & 15) == 0);
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), block(256);
    matmul_f16_kernel<<<grid, block, 0, st>>>(a, bT, c, (int)M, (int)K, (int)N, vec_ok);
    return hipGetLastError();
}

}  // namespace cactus_hip
