// Causal depthwise convolution over [N, L, C] rows and the LFM2 short-convolution block built on it.
//
//   cactus_conv1d_causal_depthwise_f16   cactus/kernel/kernel_conv.cpp:15-110
//   LFM2Model::build_conv1d              cactus/models/model_lfm2.cpp:123-207
//   ConvCache                            cactus/engine/engine_cache.cpp:494-597
//
// Roofline: HBM.  Algorithmic bytes per element of the plain convolution 2 in + 2 out (the K taps
// of a row are re-read from cache); per token and channel of the LFM2 block 6 in (B, C gate, X of
// the in-projection) + 2 out, plus 4 * W bytes of state per sequence and channel and call.
//
// Evaluation order kept from the reference: taps k = 0 .. K-1 read x[t - k * dilation] against
// w[c][K-1-k]; whole groups of 8 taps go through four fp32 lanes (lane j takes taps k = j mod 4,
// fused multiply-add, in k order) that are summed (l0+l1)+(l2+l3); the remaining K % 8 taps are
// fused into that sum one by one, taps before the start of the sequence skipped; one rounding to
// binary16 at the end.  The reference computes two rows per pass; each row's arithmetic is
// independent of the other's.
#include "common.hip.h"
#include "kernels.h"
#include "block_math.hip.h"

namespace cactus_hip {

namespace {

// half(acc) of an fp32 sum that ends in a fused multiply-add.  Without the empty asm hipcc folds the
// last fma and the conversion into v_fma_mixlo_f16, which rounds the exact sum once and flips
// fp32 -> fp16 ties.
__device__ __forceinline__ uint16_t f2h_sum(float acc) {
    asm volatile("" : "+v"(acc));
    return f2h(acc);
}

// `xat(k, &x)` gives tap k's input (returns false when it lies before the sequence),
// `wrev(k)` the weight it meets.
template <typename XAt, typename WRev>
__device__ __forceinline__ uint16_t conv_taps(size_t K, XAt xat, WRev wrev) {
    float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    size_t k = 0;
    for (; k + 8 <= K; k += 8) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
            const size_t kb = k + 4 * h;
            xat(kb + 0, x0); xat(kb + 1, x1); xat(kb + 2, x2); xat(kb + 3, x3);
            l0 = fmaf(x0, wrev(kb + 0), l0);
            l1 = fmaf(x1, wrev(kb + 1), l1);
            l2 = fmaf(x2, wrev(kb + 2), l2);
            l3 = fmaf(x3, wrev(kb + 3), l3);
        }
    }
    float acc = (l0 + l1) + (l2 + l3);
    for (; k < K; k++) {
        float x;
        if (xat(k, x)) acc = fmaf(wrev(k), x, acc);
    }
    return f2h_sum(acc);
}

// One thread per output element, channel fastest.
__global__ __launch_bounds__(256) void conv1d_causal_dw_kernel(const uint16_t* __restrict__ x,
                                                               const uint16_t* __restrict__ w,
                                                               uint16_t* __restrict__ y, size_t total, size_t L,
                                                               size_t C, size_t K, size_t dil) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t c = idx % C, t = (idx / C) % L, n = idx / (C * L);
    const uint16_t* xb = x + n * L * C + c;
    const uint16_t* wc = w + c * K;
    y[idx] = conv_taps(
        K,
        [&](size_t k, float& out) {
            const size_t back = k * dil;
            if (back > t) return false;
            out = h2f(xb[(t - back) * C]);
            return true;
        },
        [&](size_t k) { return h2f(wc[K - 1 - k]); });
}

// LFM2 short convolution, one launch from the in-projection's rows to the gated rows that enter the
// out-projection:  Bx = B * X (fp16 product), y = causal depthwise convolution (dilation 1) over the
// sequence's last W rows of Bx from earlier calls followed by this call's rows, gated = Cg * y (fp16
// product).  in_proj is [nseq, L, 3, C] with B, Cg, X in that order (model_lfm2.cpp:136-146).
// `state` [nseq, W, C] holds a sequence's last W rows of Bx in temporal order and starts as zeros.
// The reference keeps them in a circular buffer with a row count and convolves over only the rows
// it holds; a zero row adds nothing to the fp32 sums, so rows of zeros in place of rows not yet
// held give the same bits.
// mode 0: gated rows only; 1: gated rows, and the thread of the sequence's first row chunk then
// writes the channel's new state (it is the only one that read the old rows: the launcher gives every
// chunk at least K - 1 rows, so later chunks reach back only into this call's rows); 2: state only.
// KT: the tap count when it is 2, 3 or 4 (weights and the last KT - 1 rows stay in registers, a row
// of the in-projection is read once per chunk), 0 for any other K.
// QUANT (VEC 8, KT > 0, one workgroup of C / 8 threads -- whole waves -- per row chunk): every gated row is
// also quantized for the out-projection the way quantize_activations_fp16_to_int8 does it
// (graph_ops_nn.cpp:35-66: scale = max(absmax / 127, 1e-10), q = sat8(rne(clamp(x * (1 / scale))))), into
// q_out int8 [rows, C] and scales_out float [rows]; gated may then be null.
template <int VEC, int KT, bool QUANT = false>
__global__ __launch_bounds__(256) void lfm2_conv_gate_kernel(const uint16_t* __restrict__ in_proj,
                                                             const uint16_t* __restrict__ w,
                                                             uint16_t* __restrict__ state,
                                                             uint16_t* __restrict__ gated, size_t L, size_t C,
                                                             size_t K, size_t W, size_t rows_per_chunk, int mode,
                                                             int8_t* __restrict__ q_out = nullptr,
                                                             float* __restrict__ scales_out = nullptr) {
    __shared__ float q_red[2][4];
    const size_t cv = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cv * VEC >= C) return;
    const size_t c0 = cv * VEC, n = blockIdx.z;
    const uint16_t* ip = in_proj + n * L * 3 * C + c0;
    uint16_t* st = state + n * W * C + c0;

    // row i of (state rows ++ Bx rows) for the thread's channels
    auto xrow = [&](size_t i, uint16_t (&out)[VEC]) {
        if (i < W) {
            if (VEC == 8) {
                *reinterpret_cast<v4i*>(out) = *reinterpret_cast<const v4i*>(st + i * C);
            } else {
                out[0] = st[i * C];
            }
        } else {
            const uint16_t* r = ip + (i - W) * 3 * C;
            alignas(16) uint16_t b[VEC], xx[VEC];
            if (VEC == 8) {
                *reinterpret_cast<v4i*>(b) = *reinterpret_cast<const v4i*>(r);
                *reinterpret_cast<v4i*>(xx) = *reinterpret_cast<const v4i*>(r + 2 * C);
            } else {
                b[0] = r[0];
                xx[0] = r[2 * C];
            }
#pragma unroll
            for (int e = 0; e < VEC; e++) out[e] = h2bits(bits2h(b[e]) * bits2h(xx[e]));
        }
    };

    if (mode != 2 && KT > 0) {
        const size_t t_begin = (size_t)blockIdx.y * rows_per_chunk;
        const size_t t_end = t_begin + rows_per_chunk < L ? t_begin + rows_per_chunk : L;
        float wr[KT > 0 ? KT : 1][VEC];
#pragma unroll
        for (int k = 0; k < KT; k++)
#pragma unroll
            for (int e = 0; e < VEC; e++) wr[k][e] = h2f(w[(c0 + e) * KT + KT - 1 - k]);
        // hist[j]: row pos - (KT - 1) + j of the concatenation; rows before it count as zeros, which
        // leave the sums as the skipped taps of the reference do
        alignas(16) uint16_t hist[KT > 1 ? KT - 1 : 1][VEC];
#pragma unroll
        for (int j = 0; j < KT - 1; j++) {
            const size_t need = (size_t)(KT - 1 - j);
            if (t_begin + W >= need) xrow(t_begin + W - need, hist[j]);
            else {
#pragma unroll
                for (int e = 0; e < VEC; e++) hist[j][e] = 0;
            }
        }
        for (size_t t = t_begin; t < t_end; t++) {
            alignas(16) uint16_t cg[VEC], o[VEC], cur[VEC];
            if (VEC == 8) *reinterpret_cast<v4i*>(cg) = *reinterpret_cast<const v4i*>(ip + t * 3 * C + C);
            else cg[0] = ip[t * 3 * C + C];
            xrow(t + W, cur);
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                float acc = fmaf(wr[0][e], h2f(cur[e]), 0.f);
#pragma unroll
                for (int k = 1; k < KT; k++) acc = fmaf(wr[k][e], h2f(hist[KT - 1 - k][e]), acc);
                o[e] = h2bits(bits2h(cg[e]) * bits2h(f2h_sum(acc)));
            }
#pragma unroll
            for (int j = 0; j + 1 < KT - 1; j++)
#pragma unroll
                for (int e = 0; e < VEC; e++) hist[j][e] = hist[j + 1][e];
            if (KT > 1) {
#pragma unroll
                for (int e = 0; e < VEC; e++) hist[KT - 2][e] = cur[e];
            }
            uint16_t* g = gated + (n * L + t) * C + c0;
            if (!QUANT || gated != nullptr) {
                if (VEC == 8) *reinterpret_cast<v4i*>(g) = *reinterpret_cast<const v4i*>(o);
                else g[0] = o[0];
            }
            if constexpr (QUANT) {
                // the block holds the whole row: absmax over its waves (two LDS rows alternate, one barrier per row)
                float m = 0.0f;
#pragma unroll
                for (int e = 0; e < VEC; e++) m = fmaxf(m, fabsf(h2f(o[e])));
                m = wave_max(m);
                float* red = q_red[t & 1];
                if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
                __syncthreads();
                const int nw = (int)(blockDim.x >> 6);
                m = red[0];
                for (int i = 1; i < nw; i++) m = fmaxf(m, red[i]);
                float scale = m / 127.0f;
                if (scale < 1e-10f) scale = 1e-10f;
                const float inv = 1.0f / scale;
                uint32_t packed[2] = {0, 0};
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    float q = h2f(o[e]) * inv;
                    q = fmaxf(fminf(q, 127.0f), -128.0f);
                    packed[(e >> 2) & 1] |= (uint32_t)(clamp_i8(rne_i32(q)) & 0xFF) << (8 * (e & 3));
                }
                *reinterpret_cast<uint2*>(q_out + (n * L + t) * C + c0) = make_uint2(packed[0], packed[1]);
                if (threadIdx.x == 0) scales_out[n * L + t] = scale;
            }
        }
    }
    if (mode != 2 && KT == 0) {
        const size_t t_begin = (size_t)blockIdx.y * rows_per_chunk;
        const size_t t_end = t_begin + rows_per_chunk < L ? t_begin + rows_per_chunk : L;
        for (size_t t = t_begin; t < t_end; t++) {
            const size_t pos = t + W;   // row of the concatenation this output belongs to
            alignas(16) uint16_t cg[VEC], o[VEC];
            if (VEC == 8) *reinterpret_cast<v4i*>(cg) = *reinterpret_cast<const v4i*>(ip + t * 3 * C + C);
            else cg[0] = ip[t * 3 * C + C];
            if (K < 8) {
                // no whole group of 8 taps: the sum is the plain chain over the taps
                float acc[VEC];
#pragma unroll
                for (int e = 0; e < VEC; e++) acc[e] = 0.f;
                for (size_t k = 0; k < K && k <= pos; k++) {
                    alignas(16) uint16_t xr[VEC];
                    xrow(pos - k, xr);
#pragma unroll
                    for (int e = 0; e < VEC; e++)
                        acc[e] = fmaf(h2f(w[(c0 + e) * K + K - 1 - k]), h2f(xr[e]), acc[e]);
                }
#pragma unroll
                for (int e = 0; e < VEC; e++) o[e] = h2bits(bits2h(cg[e]) * bits2h(f2h_sum(acc[e])));
            } else {
#pragma unroll 1
                for (int e = 0; e < VEC; e++) {
                    const uint16_t* wc = w + (c0 + e) * K;
                    const uint16_t yv = conv_taps(
                        K,
                        [&](size_t k, float& out) {
                            if (k > pos) return false;
                            const size_t i = pos - k;
                            if (i < W) out = h2f(st[i * C + e]);
                            else {
                                const uint16_t* r = ip + (i - W) * 3 * C + e;
                                out = h2f(h2bits(bits2h(r[0]) * bits2h(r[2 * C])));
                            }
                            return true;
                        },
                        [&](size_t k) { return h2f(wc[K - 1 - k]); });
                    o[e] = h2bits(bits2h(cg[e]) * bits2h(yv));
                }
            }
            uint16_t* g = gated + (n * L + t) * C + c0;
            if (VEC == 8) *reinterpret_cast<v4i*>(g) = *reinterpret_cast<const v4i*>(o);
            else g[0] = o[0];
        }
    }
    if (mode == 2 || (mode == 1 && blockIdx.y == 0)) {
        // new state row j = row L + j of the concatenation; ascending j reads a state row before the
        // iteration that overwrites it
        for (size_t j = 0; j < W; j++) {
            alignas(16) uint16_t r[VEC];
            xrow(L + j, r);
            if (VEC == 8) *reinterpret_cast<v4i*>(st + j * C) = *reinterpret_cast<const v4i*>(r);
            else st[j * C] = r[0];
        }
    }
}

}  // namespace

hipError_t launch_conv1d_causal_depthwise(const uint16_t* x, const uint16_t* w, uint16_t* y, size_t N, size_t L,
                                          size_t C, size_t K, size_t dilation, hipStream_t st) {
    const size_t total = N * L * C;
    if (total == 0) return hipSuccess;
    const size_t blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    conv1d_causal_dw_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(x, w, y, total, L, C, K, dilation);
    return hipGetLastError();
}

bool lfm2_conv_gate_quant_ok(size_t C, size_t K) { return C % 512 == 0 && C <= 2048 && K >= 2 && K <= 4; }

hipError_t launch_lfm2_conv_gate_quant(const uint16_t* in_proj, const uint16_t* w, uint16_t* state, uint16_t* gated,
                                       int8_t* q_out, float* scales_out, size_t nseq, size_t L, size_t C, size_t K,
                                       size_t W, hipStream_t st) {
    if (nseq == 0 || L == 0) return hipSuccess;
    if (!lfm2_conv_gate_quant_ok(C, K) || nseq > 65535) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(in_proj) | reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(gated) |
         reinterpret_cast<uintptr_t>(q_out)) & 15)
        return hipErrorInvalidValue;
    const size_t threads = C / 8;
    size_t rpc = L;
    while (rpc > 4 && nseq * threads * ((L + rpc - 1) / rpc) < 131072) rpc = (rpc + 1) / 2;
    if (rpc + 1 < K) rpc = K - 1 < L ? K - 1 : L;
    const size_t chunks = (L + rpc - 1) / rpc;
    if (chunks > 65535) return hipErrorInvalidValue;
    const dim3 grid(1, (unsigned)chunks, (unsigned)nseq), block((unsigned)threads);
    const int mode = W > 0 ? 1 : 0;
#define CACTUS_CONV_GOQ(T) lfm2_conv_gate_kernel<8, T, true><<<grid, block, 0, st>>>(in_proj, w, state, gated, L, C, K, W, rpc, mode, q_out, scales_out)
    if (K == 2) CACTUS_CONV_GOQ(2);
    else if (K == 3) CACTUS_CONV_GOQ(3);
    else CACTUS_CONV_GOQ(4);
#undef CACTUS_CONV_GOQ
    return hipGetLastError();
}

hipError_t launch_lfm2_conv_gate(const uint16_t* in_proj, const uint16_t* w, uint16_t* state, uint16_t* gated,
                                 size_t nseq, size_t L, size_t C, size_t K, size_t W, hipStream_t st) {
    if (nseq == 0 || L == 0 || C == 0) return hipSuccess;
    if (nseq > 65535) return hipErrorInvalidValue;
    const bool vec = (C % 8 == 0) && (((reinterpret_cast<uintptr_t>(in_proj) | reinterpret_cast<uintptr_t>(state) |
                                        reinterpret_cast<uintptr_t>(gated)) & 15) == 0);
    const size_t threads = vec ? C / 8 : C;
    const unsigned gx = (unsigned)((threads + 255) / 256);
    // Rows per thread: enough chunks for a few waves per SIMD, at least 4 rows (and K - 1, see the
    // kernel) so the rows re-read at the start of a chunk stay a small share.
    size_t rpc = L;
    while (rpc > 4 && nseq * threads * ((L + rpc - 1) / rpc) < 131072) rpc = (rpc + 1) / 2;
    if (rpc + 1 < K) rpc = K - 1 < L ? K - 1 : L;
    const size_t chunks = (L + rpc - 1) / rpc;
    if (chunks > 65535) return hipErrorInvalidValue;
    const bool one_launch = chunks == 1 || rpc + 1 >= K;
    auto go = [&](unsigned gy, int mode) {
        const dim3 grid(gx, gy, (unsigned)nseq), block(256);
#define CACTUS_CONV_GO(V, T) lfm2_conv_gate_kernel<V, T><<<grid, block, 0, st>>>(in_proj, w, state, gated, L, C, K, W, rpc, mode)
        if (vec) {
            if (K == 2) CACTUS_CONV_GO(8, 2);
            else if (K == 3) CACTUS_CONV_GO(8, 3);
            else if (K == 4) CACTUS_CONV_GO(8, 4);
            else CACTUS_CONV_GO(8, 0);
        } else {
            if (K == 2) CACTUS_CONV_GO(1, 2);
            else if (K == 3) CACTUS_CONV_GO(1, 3);
            else if (K == 4) CACTUS_CONV_GO(1, 4);
            else CACTUS_CONV_GO(1, 0);
        }
#undef CACTUS_CONV_GO
        return hipGetLastError();
    };
    if (one_launch) return go((unsigned)chunks, W > 0 ? 1 : 0);
    hipError_t e = go((unsigned)chunks, 0);
    if (e != hipSuccess || W == 0) return e;
    return go(1, 2);
}

}  // namespace cactus_hip
