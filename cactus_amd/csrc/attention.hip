// Attention over [INT8 cached K/V + fp32 group scales] ++ [FP16 new K/V]  (hybrid mode,
// cactus_attention_hybrid_int8_fp16, kernel_attention.cpp:257-482) and over plain FP16 K/V
// (f16 mode, cactus_attention_f16, kernel_attention.cpp:10-255).
//
// Roofline: HBM for decode.  Algorithmic bytes per call =
//   2*cache_len*Hkv*(D + 4*ceil(D/128)) + 2*new_len*Hkv*D*2 + 2*Hq*D*2   (per sequence).
//
// Decomposition.  One 256-thread workgroup handles one (sequence, kv head, query position,
// KV split); the NQ query heads that share the kv head (GQA) are processed together so each
// K/V byte is loaded once.  Inside a wave, lane = (token slot t = lane/8, dim chunk c = lane%8):
// a wave instruction loads 8 tokens x (D/8 dims), i.e. whole 128-byte head rows of the INT8
// cache straight into VGPRs.  A wave walks 32-key blocks (4 passes of 8 tokens) -- the
// reference's block size, which matters in f16 mode where the weights inside a block come
// from a 2nd-order polynomial 2^x relative to that block's own maximum
// (kernel_attention.cpp:174-195) while blocks are combined with expf.  Waves take blocks round
// robin; wave states (m, l, O) are merged through LDS, KV splits through a float workspace
// and a second tiny kernel.  Dequantization is int8 -> fp32 in registers; the per-(token,
// head, group) scale multiplies the reduced dot product (K) or the softmax weight (V).
#include "common.hip.h"
#include "kernels.h"

#include <math.h>

#include <type_traits>

namespace cactus_hip {

constexpr int AT_WAVES = 4;
constexpr int AT_BLOCK = 32;

struct AttnParams {
    AttnArgs a;
    int nsplit;
    int gq;       // Hq / Hkv
    int qsubs;    // gq / NQ
    int G;        // scale groups per head
    float* ws;
};

// 8 int8 -> 8 floats
__device__ __forceinline__ void cvt8_i8(uint2 v, float* out) {
    const uint32_t w[2] = {v.x, v.y};
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = (float)(int8_t)((w[i >> 2] >> (8 * (i & 3))) & 0xFF);
}
// 8 halfs -> 8 floats
__device__ __forceinline__ void cvt8_f16(uint4 v, float* out) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[2 * i] = h2f((uint16_t)(w[i] & 0xFFFFu));
        out[2 * i + 1] = h2f((uint16_t)(w[i] >> 16));
    }
}

template <int DPL>
__device__ __forceinline__ void load_row_i8(const int8_t* p, int nd, float* out) {
    if (nd == DPL) {
        if constexpr (DPL >= 16) {
#pragma unroll
            for (int u = 0; u < DPL / 16; u++) {
                const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * u);
                cvt8_i8(make_uint2(v.x, v.y), out + 16 * u);
                cvt8_i8(make_uint2(v.z, v.w), out + 16 * u + 8);
            }
        } else {
            cvt8_i8(*reinterpret_cast<const uint2*>(p), out);
        }
    } else {
#pragma unroll
        for (int u = 0; u < DPL / 8; u++) {
            if (8 * u < nd) {
                cvt8_i8(*reinterpret_cast<const uint2*>(p + 8 * u), out + 8 * u);
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) out[8 * u + e] = 0.0f;
            }
        }
    }
}

template <int DPL>
__device__ __forceinline__ void load_row_f16(const uint16_t* p, int nd, float* out) {
#pragma unroll
    for (int u = 0; u < DPL / 8; u++) {
        if (8 * u < nd) {
            cvt8_f16(*reinterpret_cast<const uint4*>(p + 8 * u), out + 8 * u);
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) out[8 * u + e] = 0.0f;
        }
    }
}

// Weight of one key inside its 32-key block in f16 mode: the reference's 4-wide polynomial
// 2^x (kernel_attention.cpp:178-191) for in-block index < (block_size/4)*4, expf for the tail
// (:197-204).  Integer steps are restated literally, wrapping shift included.
__device__ __forceinline__ float f16mode_weight(float s, float bmax, bool poly) {
    if (s == -INFINITY) return 0.0f;
    if (!poly) return __expf(s - bmax);
    const float x = (s - bmax) * 1.442695f;
    const int xi = (int)x;  // v_cvt_i32_f32: toward zero, saturating
    const float xf = x - (float)xi;
    float y = fmaf(xf, 0.6931472f, 1.0f);
    y = fmaf(xf * xf, 0.2402265f, y);
    const uint32_t e = ((uint32_t)xi + 127u) << 23;
    return y * __uint_as_float(e);
}

// NB bytes (8, 16, 32 or 64) of one K/V row slice into dwords
template <int NB>
__device__ __forceinline__ void load_raw(const void* p, uint32_t* out) {
    if constexpr (NB == 8) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        out[0] = v.x; out[1] = v.y;
    } else {
#pragma unroll
        for (int u = 0; u < NB / 16; u++) {
            const uint4 v = *(reinterpret_cast<const uint4*>(p) + u);
            out[4 * u] = v.x; out[4 * u + 1] = v.y; out[4 * u + 2] = v.z; out[4 * u + 3] = v.w;
        }
    }
}
// first nd (a multiple of 8, < DPL) elements of a slice, the rest zero
template <int DPL, bool I8>
__device__ __forceinline__ void load_raw_partial(const void* p, int nd, uint32_t* out) {
    constexpr int DW8 = I8 ? 2 : 4;   // dwords per 8 elements
#pragma unroll
    for (int u = 0; u < DPL / 8; u++) {
        if (8 * u < nd) {
            load_raw<DW8 * 4>(reinterpret_cast<const char*>(p) + u * DW8 * 4, out + u * DW8);
        } else {
#pragma unroll
            for (int e = 0; e < DW8; e++) out[u * DW8 + e] = 0;
        }
    }
}
template <int DPL, bool I8>
__device__ __forceinline__ void raw_to_float(const uint32_t* raw, float* out) {
    if constexpr (I8) {
#pragma unroll
        for (int i = 0; i < DPL; i++) out[i] = (float)(int8_t)((raw[i >> 2] >> (8 * (i & 3))) & 0xFF);
    } else {
#pragma unroll
        for (int i = 0; i < DPL / 2; i++) {
            out[2 * i] = h2f((uint16_t)(raw[i] & 0xFFFFu));
            out[2 * i + 1] = h2f((uint16_t)(raw[i] >> 16));
        }
    }
}
// sum over the 8 lanes that share a token slot (DPP: quad swaps, then half-row mirror)
__device__ __forceinline__ float sum8(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}

template <int DPL, int NQ, bool F16MODE>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    __shared__ float s_m[AT_WAVES][NQ];
    __shared__ float s_l[AT_WAVES][NQ][4];
    __shared__ float s_o[AT_WAVES][NQ][4][8 * DPL];   // [wave][head][16-lane row][dim]

    const AttnArgs& a = p.a;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = lane >> 3, c = lane & 7;
    const int D = (int)a.D, Hkv = (int)a.Hkv, Hq = (int)a.Hq, S = (int)a.S;
    const int split = blockIdx.y;
    // query-head groups that read the same kv head are the slowest index: with a multiple of 8
    // (sequence, kv head, position) triples they land on the same XCD and share its L2
    int item = blockIdx.x;
    const int qp = item % S; item /= S;
    const int kvh = item % Hkv; item /= Hkv;
    const int b = item % (int)a.B;
    const int qsub = item / (int)a.B;
    const int h0 = kvh * p.gq + qsub * NQ;

    const int cl = F16MODE ? 0 : (a.cache_lens ? min(max(a.cache_lens[b], 0), (int)a.cache_len) : (int)a.cache_len);
    const int new_len = (int)a.new_len;
    const int KV = cl + new_len;
    const long long qabs = (long long)(a.pos_offsets ? (size_t)a.pos_offsets[b] : a.position_offset) + qp;

    long long kv_start = 0, kv_end = KV;
    if (F16MODE) {
        if (a.window_size > 0 && a.window_size < (size_t)KV) {  // kernel_attention.cpp:81-87
            if (qabs > (long long)a.window_size) kv_start = qabs - (long long)a.window_size;
            if (a.is_causal) kv_end = min(kv_end, qabs + 1);
        } else if (a.is_causal) {
            kv_end = min(kv_end, qabs + 1);
        }
    } else {
        if (a.is_causal) kv_end = min(kv_end, qabs + 1);      // :331-332
    }
    // Blocks of 32 keys.  f16 mode: the reference's blocks, from kv_start, the last one clamped to
    // KV rather than kv_end (:93).  Hybrid mode: the cached tokens [0, min(cl, kv_end)) in blocks
    // from 0, then the FP16 tokens [cl, kv_end) in blocks of their own -- the online softmax does
    // not depend on where block borders fall beyond rounding.
    const int c_cnt = F16MODE ? 0 : (int)min((long long)cl, kv_end);
    const int nbc = (c_cnt + AT_BLOCK - 1) / AT_BLOCK;
    const long long n_first = F16MODE ? kv_start : cl;
    const long long n_lim = F16MODE ? (long long)KV : kv_end;
    const int nbn = kv_end > n_first ? (int)((kv_end - n_first + AT_BLOCK - 1) / AT_BLOCK) : 0;
    const int nblk = nbc + nbn;
    const int bps = (nblk + p.nsplit - 1) / p.nsplit;
    const int blk_first = split * bps;
    const int blk_last = min(nblk, blk_first + bps);

    const int d0 = c * DPL;
    const int nd = max(0, min(DPL, D - d0));
    float scale = a.scale;
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);       // :27-29, :278-280

    float qf[NQ][DPL];
#pragma unroll
    for (int i = 0; i < NQ; i++)
        load_row_f16<DPL>(a.q + (((size_t)b * S + qp) * Hq + h0 + i) * D + d0, nd, qf[i]);

    float m_run[NQ], l_run[NQ], o[NQ][DPL];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        m_run[i] = -INFINITY;
        l_run[i] = 0.0f;
#pragma unroll
        for (int d = 0; d < DPL; d++) o[i][d] = 0.0f;
    }

    const int8_t* Kc = a.k_cached + (size_t)b * a.cache_batch_stride;
    const int8_t* Vc = a.v_cached + (size_t)b * a.cache_batch_stride;
    const float* Ks = a.k_scales + (size_t)b * a.scale_batch_stride;
    const float* Vs = a.v_scales + (size_t)b * a.scale_batch_stride;
    const uint16_t* Kn = a.k_new + (size_t)b * new_len * Hkv * D;
    const uint16_t* Vn = a.v_new + (size_t)b * new_len * Hkv * D;
    const uint16_t* Mk = (F16MODE && a.mask) ? a.mask + ((size_t)b * S + qp) * KV : nullptr;
    const int sgrp = F16MODE ? 0 : d0 / (int)a.qgs;

    // One block: kb = first key (absolute position), ke = one past its last key.  Every K and V
    // row slice and scale of the block is requested before anything is consumed, so one HBM
    // round trip covers the block; keys past ke re-read the last valid row and are masked.
    struct CachedBlock {   // raw INT8 row slices and scales of one 32-key block of the cache
        uint32_t rk[4][DPL / 4], rv[4][DPL / 4];
        float ksc[4], vsc[4];
    };
    auto fetch_cached = [&](long long kb, int ke, CachedBlock& cb) {
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const int kp = min((int)kb + ps * 8 + t, ke - 1);
            const size_t row = (size_t)kp * Hkv + kvh;
            cb.ksc[ps] = 1.0f;
            cb.vsc[ps] = 1.0f;
            if (nd == DPL) {
                load_raw<DPL>(Kc + row * D + d0, cb.rk[ps]);
                load_raw<DPL>(Vc + row * D + d0, cb.rv[ps]);
            } else {
                load_raw_partial<DPL, true>(Kc + row * D + d0, nd, cb.rk[ps]);
                load_raw_partial<DPL, true>(Vc + row * D + d0, nd, cb.rv[ps]);
            }
            if (nd > 0) {
                cb.ksc[ps] = Ks[row * p.G + sgrp];
                cb.vsc[ps] = Vs[row * p.G + sgrp];
            }
        }
    };
    auto do_block = [&](auto i8_tag, long long kb, int ke, const CachedBlock& cb) {
        constexpr bool I8 = decltype(i8_tag)::value;
        constexpr int RW = DPL / 2;   // dwords per FP16 row slice
        const int bsz = ke - (int)kb;
        const int vec = (bsz / 4) * 4;
        // FP16 rows (the few new tokens of a decode step, or f16 mode) are fetched pass by pass:
        // holding a whole FP16 block would double the register footprint
        uint32_t fk[RW], fv[RW];
        auto fetch_f16 = [&](const uint16_t* base, int ps, uint32_t* dst) {
            const int kp = min((int)kb + ps * 8 + t, ke - 1);
            const size_t row = (size_t)(kp - cl) * Hkv + kvh;
            if (nd == DPL) load_raw<2 * DPL>(base + row * D + d0, dst);
            else load_raw_partial<DPL, false>(base + row * D + d0, nd, dst);
        };

        float sc[NQ][4];
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const int kp = (int)kb + ps * 8 + t;
            bool masked = kp >= ke;
            if (F16MODE && !masked) {
                if (a.is_causal && kp > qabs) masked = true;                                   // :146
                else if (a.window_size > 0 && kp < qabs && (qabs - kp) > (long long)a.window_size) masked = true;
                else if (Mk && h2f(Mk[kp]) == 0.0f) masked = true;                             // :152
            }
            float kf[DPL];
            if constexpr (I8) {
                raw_to_float<DPL, true>(cb.rk[ps], kf);
            } else {
                fetch_f16(Kn, ps, fk);
                raw_to_float<DPL, false>(fk, kf);
            }
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                float part = 0.0f;
#pragma unroll
                for (int d = 0; d < DPL; d++) part = fmaf(qf[i][d], kf[d], part);
                part = sum8(I8 ? part * cb.ksc[ps] : part);
                sc[i][ps] = masked ? -INFINITY : part * scale;
            }
        }

        float wgt[NQ][4];
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const float bm = wave_max(fmaxf(fmaxf(sc[i][0], sc[i][1]), fmaxf(sc[i][2], sc[i][3])));
            float bsum = 0.0f;
            if (F16MODE) {
                if (bm > -INFINITY) {                                   // :160-169 (max is replaced)
                    const float corr = __expf(m_run[i] - bm);
                    l_run[i] *= corr;
#pragma unroll
                    for (int d = 0; d < DPL; d++) o[i][d] *= corr;
                    m_run[i] = bm;
                }
#pragma unroll
                for (int ps = 0; ps < 4; ps++) {
                    wgt[i][ps] = f16mode_weight(sc[i][ps], bm, ps * 8 + t < vec);
                    bsum += wgt[i][ps];
                }
            } else {
                const float mn = fmaxf(m_run[i], bm);
                if (mn > -INFINITY) {
                    const float corr = __expf(m_run[i] - mn);          // m_run = -inf -> 0
                    l_run[i] *= corr;
#pragma unroll
                    for (int d = 0; d < DPL; d++) o[i][d] *= corr;
                    m_run[i] = mn;
                }
#pragma unroll
                for (int ps = 0; ps < 4; ps++) {
                    wgt[i][ps] = (sc[i][ps] == -INFINITY) ? 0.0f : __expf(sc[i][ps] - mn);
                    bsum += wgt[i][ps];
                }
            }
            l_run[i] += bsum;   // per token-slot partial; lanes of one slot hold identical values
        }

#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            float vf[DPL];
            if constexpr (I8) {
                raw_to_float<DPL, true>(cb.rv[ps], vf);
            } else {
                if (ps * 8 >= bsz) continue;
                fetch_f16(Vn, ps, fv);
                raw_to_float<DPL, false>(fv, vf);
            }
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const float ww = I8 ? wgt[i][ps] * cb.vsc[ps] : wgt[i][ps];   // masked keys carry weight 0
#pragma unroll
                for (int d = 0; d < DPL; d++) o[i][d] = fmaf(vf[d], ww, o[i][d]);
            }
        }
    };

    // cached blocks of this wave, the next one requested before the current one is consumed.
    // (Fetching each block right before its use instead saves ~50 registers but stays at two
    // waves per SIMD and measured 3-6% slower at every decode shape tried.)
    int blk = blk_first + w;
    if constexpr (!F16MODE) {
        const int c_last = min(blk_last, nbc);
        CachedBlock cur, nxt;
        auto range_end = [&](int bk) { return (int)min((long long)c_cnt, (long long)bk * AT_BLOCK + AT_BLOCK); };
        if (blk < c_last) fetch_cached((long long)blk * AT_BLOCK, range_end(blk), cur);
        for (; blk < c_last; blk += AT_WAVES) {
            const bool more = blk + AT_WAVES < c_last;
            if (more) fetch_cached((long long)(blk + AT_WAVES) * AT_BLOCK, range_end(blk + AT_WAVES), nxt);
            do_block(std::true_type{}, (long long)blk * AT_BLOCK, range_end(blk), cur);
            if (more) cur = nxt;
        }
    }
    for (; blk < blk_last; blk += AT_WAVES) {
        const long long kb = n_first + (long long)(blk - nbc) * AT_BLOCK;
        CachedBlock unused;
        do_block(std::false_type{}, kb, (int)min(n_lim, kb + AT_BLOCK), unused);
    }

    // fold the 8 token slots of the wave: slots t and t^1 by DPP (rotate by 8 inside a 16-lane
    // row), the four rows through LDS together with the other waves' states
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        l_run[i] += dpp_mov<0x128>(l_run[i]);
#pragma unroll
        for (int d = 0; d < DPL; d++) o[i][d] += dpp_mov<0x128>(o[i][d]);
    }
    if ((t & 1) == 0) {
        const int row = t >> 1;
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            if (c == 0) {
                s_l[w][i][row] = l_run[i];
                if (row == 0) s_m[w][i] = m_run[i];
            }
#pragma unroll
            for (int d = 0; d < DPL; d++) s_o[w][i][row][d0 + d] = o[i][d];
        }
    }
    __syncthreads();

    for (int idx = threadIdx.x; idx < NQ * D; idx += 256) {
        const int i = idx / D, d = idx % D;
        float M = -INFINITY;
#pragma unroll
        for (int ww = 0; ww < AT_WAVES; ww++) M = fmaxf(M, s_m[ww][i]);
        float L = 0.0f, O = 0.0f;
        if (M > -INFINITY) {
#pragma unroll
            for (int ww = 0; ww < AT_WAVES; ww++) {
                const float f = __expf(s_m[ww][i] - M);
                L += ((s_l[ww][i][0] + s_l[ww][i][1]) + (s_l[ww][i][2] + s_l[ww][i][3])) * f;
                O += ((s_o[ww][i][0][d] + s_o[ww][i][1][d]) + (s_o[ww][i][2][d] + s_o[ww][i][3][d])) * f;
            }
        }
        const size_t orow = ((size_t)b * S + qp) * Hq + h0 + i;
        if (p.nsplit == 1) {
            a.out[orow * D + d] = (L > 0.0f) ? f2h_mul(O, 1.0f / L) : (uint16_t)0;   // :229-252, :460-479
        } else {
            float* rec = p.ws + (orow * p.nsplit + split) * (size_t)(D + 2);
            rec[2 + d] = O;
            if (d == 0) { rec[0] = M; rec[1] = L; }
        }
    }
}

// Any head_dim / any KV group size: one wave per (sequence, query head, query position), lanes
// 0..31 each own one key of the 32-key block and walk its D dims serially; the output row
// lives in LDS.  Same block structure and weights as attn_kernel; used for shapes the tiled
// kernel does not take (head_dim not a multiple of 8 or above 256, odd group sizes), which
// are test-sized in practice.  The reference works in 8-wide vectors and treats the
// head_dim % 8 tail dims unevenly; that is kept: f16 mode adds them to the scores with a scalar
// loop but writes zeros to those output dims (kernel_attention.cpp:138-140, 245-247), hybrid
// mode ignores them in the scores and leaves those output dims unwritten (:355-390, :464-474).
template <bool F16MODE>
__global__ __launch_bounds__(64) void attn_generic_kernel(const AttnParams p) {
    extern __shared__ float g_lds[];
    const AttnArgs& a = p.a;
    const int D = (int)a.D, Hkv = (int)a.Hkv, Hq = (int)a.Hq, S = (int)a.S;
    float* qs = g_lds;            // [D]
    float* os = g_lds + D;        // [D]
    float* wl = g_lds + 2 * D;    // [32] weights, [32] key positions as int bits
    const int lane = threadIdx.x, j = lane & 31;
    size_t item = blockIdx.x;
    const int qp = (int)(item % S); item /= S;
    const int h = (int)(item % Hq);
    const int b = (int)(item / Hq);
    const int kvh = h / p.gq;

    const int cl = F16MODE ? 0 : (a.cache_lens ? min(max(a.cache_lens[b], 0), (int)a.cache_len) : (int)a.cache_len);
    const int new_len = (int)a.new_len;
    const int KV = cl + new_len;
    const long long qabs = (long long)(a.pos_offsets ? (size_t)a.pos_offsets[b] : a.position_offset) + qp;
    long long kv_start = 0, kv_end = KV;
    if (F16MODE) {
        if (a.window_size > 0 && a.window_size < (size_t)KV) {
            if (qabs > (long long)a.window_size) kv_start = qabs - (long long)a.window_size;
            if (a.is_causal) kv_end = min(kv_end, qabs + 1);
        } else if (a.is_causal) {
            kv_end = min(kv_end, qabs + 1);
        }
    } else {
        if (a.is_causal) kv_end = min(kv_end, qabs + 1);
    }
    const int kv_lim = F16MODE ? KV : (int)kv_end;
    const int nblk = kv_end > kv_start ? (int)((kv_end - kv_start + AT_BLOCK - 1) / AT_BLOCK) : 0;
    float scale = a.scale;
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);
    const int qgs = F16MODE ? D : (int)a.qgs;
    const int Da = (D / 8) * 8;

    const size_t orow = ((size_t)b * S + qp) * Hq + h;
    for (int d = lane; d < D; d += 64) {
        qs[d] = h2f(a.q[orow * D + d]);
        os[d] = 0.0f;
    }
    __syncthreads();

    const int8_t* Kc = a.k_cached + (size_t)b * a.cache_batch_stride;
    const int8_t* Vc = a.v_cached + (size_t)b * a.cache_batch_stride;
    const float* Ks = a.k_scales + (size_t)b * a.scale_batch_stride;
    const float* Vs = a.v_scales + (size_t)b * a.scale_batch_stride;
    const uint16_t* Kn = a.k_new + (size_t)b * new_len * Hkv * D;
    const uint16_t* Vn = a.v_new + (size_t)b * new_len * Hkv * D;
    const uint16_t* Mk = (F16MODE && a.mask) ? a.mask + ((size_t)b * S + qp) * KV : nullptr;

    float m_run = -INFINITY, l_run = 0.0f;
    for (int blk = 0; blk < nblk; blk++) {
        const long long kb = kv_start + (long long)blk * AT_BLOCK;
        const int ke = (int)min((long long)kv_lim, kb + AT_BLOCK);
        const int bsz = ke - (int)kb;
        const int vec = (bsz / 4) * 4;
        const int kp = (int)kb + j;
        const bool valid = lane < 32 && kp < ke;
        float s = -INFINITY;
        if (valid) {
            float dot = 0.0f;
            if (!F16MODE && kp < cl) {
                const int8_t* kr = Kc + ((size_t)kp * Hkv + kvh) * D;
                const float* sr = Ks + ((size_t)kp * Hkv + kvh) * p.G;
                for (int g0 = 0, g = 0; g0 < Da; g0 += qgs, g++) {
                    float part = 0.0f;
                    const int ge = min(Da, g0 + qgs);
                    for (int d = g0; d < ge; d++) part = fmaf(qs[d], (float)kr[d], part);
                    dot = fmaf(part, sr[g], dot);
                }
            } else {
                const uint16_t* kr = Kn + ((size_t)(kp - cl) * Hkv + kvh) * D;
                const int de = F16MODE ? D : Da;
                for (int d = 0; d < de; d++) dot = fmaf(qs[d], h2f(kr[d]), dot);
            }
            bool masked = false;
            if (F16MODE) {
                if (a.is_causal && kp > qabs) masked = true;
                else if (a.window_size > 0 && kp < qabs && (qabs - kp) > (long long)a.window_size) masked = true;
                else if (Mk && h2f(Mk[kp]) == 0.0f) masked = true;
            }
            if (!masked) s = dot * scale;
        }
        const float bm = wave_max(s);
        float wgt;
        if (F16MODE) {
            if (bm > -INFINITY) {
                const float corr = __expf(m_run - bm);
                l_run *= corr;
                for (int d = lane; d < D; d += 64) os[d] *= corr;
                m_run = bm;
            }
            wgt = f16mode_weight(s, bm, j < vec);
        } else {
            const float mn = fmaxf(m_run, bm);
            if (mn > -INFINITY) {
                const float corr = __expf(m_run - mn);
                l_run *= corr;
                for (int d = lane; d < D; d += 64) os[d] *= corr;
                m_run = mn;
            }
            wgt = (s == -INFINITY) ? 0.0f : __expf(s - mn);
        }
        if (lane >= 32) wgt = 0.0f;
        l_run += wave_sum(wgt);
        __syncthreads();
        if (lane < 32) wl[lane] = wgt;
        __syncthreads();
        for (int d = lane; d < Da; d += 64) {
            float acc = os[d];
            for (int jj = 0; jj < bsz; jj++) {
                const int kq = (int)kb + jj;
                float v;
                if (!F16MODE && kq < cl)
                    v = (float)Vc[((size_t)kq * Hkv + kvh) * D + d] * Vs[((size_t)kq * Hkv + kvh) * p.G + d / qgs];
                else
                    v = h2f(Vn[((size_t)(kq - cl) * Hkv + kvh) * D + d]);
                acc = fmaf(v, wl[jj], acc);
            }
            os[d] = acc;
        }
    }
    for (int d = lane; d < D; d += 64) {
        if (l_run > 0.0f) {
            if (d < Da) a.out[orow * D + d] = f2h_mul(os[d], 1.0f / l_run);
            else if (F16MODE) a.out[orow * D + d] = 0;
        } else {
            a.out[orow * D + d] = 0;
        }
    }
}

// (Merging inside the attention launch -- every split fences, takes a ticket, the last one merges --
// was measured and dropped: the device-scope fences write back and invalidate the XCD's L2 under the
// other workgroups' K/V streams, 2-3.5x slower at every split shape.)
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ ws,
                                                        uint16_t* __restrict__ out, int nsplit, int D) {
    // one workgroup per output row; the per-split factors exp(m_s - M) go through LDS once, then the
    // splits are summed by 256 / (D / 2) thread groups at a time (a thread holds two dims of every
    // G-th split, independent loads), and the groups' partial sums meet in LDS
    __shared__ float s_f[256];
    __shared__ float s_red[8];
    __shared__ float s_acc[512];
    const size_t row = blockIdx.x;
    const float* base = ws + row * nsplit * (size_t)(D + 2);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float ms = tid < nsplit ? base[(size_t)tid * (D + 2)] : -INFINITY;
    const float ls = tid < nsplit ? base[(size_t)tid * (D + 2) + 1] : 0.0f;
    float M = wave_max(ms);
    if (lane == 0) s_red[wv] = M;
    __syncthreads();
    M = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float f = (M > -INFINITY && tid < nsplit) ? __expf(ms - M) : 0.0f;
    s_f[tid] = f;
    float L = wave_sum(ls * f);
    if (lane == 0) s_red[4 + wv] = L;
    __syncthreads();
    L = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    const int tpr = D / 2;              // threads per record
    const int G = 256 / tpr;            // groups of splits summed side by side
    const int g = tid / tpr, dd = (tid - g * tpr) * 2;
    if (g < G) {
        float ox = 0.0f, oy = 0.0f;
#pragma unroll 8
        for (int s = g; s < nsplit; s += G) {
            const float2 v = *reinterpret_cast<const float2*>(base + (size_t)s * (D + 2) + 2 + dd);
            ox = fmaf(v.x, s_f[s], ox);
            oy = fmaf(v.y, s_f[s], oy);
        }
        s_acc[g * D + dd] = ox;
        s_acc[g * D + dd + 1] = oy;
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        float O = 0.0f;
        for (int gg = 0; gg < G; gg++) O += s_acc[gg * D + d];
        out[row * D + d] = (L > 0.0f) ? f2h_mul(O, 1.0f / L) : (uint16_t)0;
    }
}

namespace {

struct Plan {
    int dpl, nq, nsplit;
    size_t items;
    bool mfma = false;   // the matrix-core decode kernel (attention_decode.hip)
};

bool tiled_ok(const AttnArgs& a) {
    return a.D % 8 == 0 && a.D <= 256 && (a.f16_mode || (a.qgs != 0 && a.qgs % 32 == 0));
}

Plan make_plan(const AttnArgs& a) {
    Plan pl;
    if (!tiled_ok(a)) {
        pl.dpl = 0; pl.nq = 1; pl.nsplit = 1;
        pl.items = a.B * a.Hq * a.S;
        return pl;
    }
    pl.dpl = a.D <= 64 ? 8 : (a.D <= 128 ? 16 : 32);
    const int gq = (int)(a.Hq / a.Hkv);
    const int maxnq = pl.dpl == 32 ? 1 : (pl.dpl == 16 ? 2 : 4);   // register budget: NQ * DPL <= 32
    pl.nq = 1;
    for (int cand = maxnq; cand >= 1; cand >>= 1)
        if (gq % cand == 0) { pl.nq = cand; break; }
    pl.items = a.B * a.Hkv * (size_t)(gq / pl.nq) * a.S;
    if (attn_decode_mfma_ok(a)) {   // every query head of a kv head in one wave
        pl.mfma = true;
        pl.nq = gq;
        pl.items = a.B * a.Hkv;
    }
    const size_t kvmax = (a.f16_mode ? 0 : a.cache_len) + a.new_len;
    const size_t maxblk = (kvmax + AT_BLOCK - 1) / AT_BLOCK;
    // enough workgroups to fill the chip (about 2048 of 4 waves), but at least one 32-key block per wave;
    // at most 256 splits (the merge kernel keeps one factor per split in LDS).  The matrix-core kernel's
    // units are single waves: about 1024 of them (one per SIMD) stream fastest, each with two blocks or more.
#ifndef AT_MFMA_TARGET
#define AT_MFMA_TARGET 1024
#endif
    const size_t target = pl.mfma ? AT_MFMA_TARGET : 2048;
    size_t ns = pl.items ? (target + pl.items - 1) / pl.items : 1;
    const size_t per_unit = pl.mfma ? 2 : AT_WAVES;     // blocks a unit should at least have
    const size_t cap = maxblk / per_unit ? maxblk / per_unit : 1;
    if (ns > cap) ns = cap;
    if (ns > 256) ns = 256;
    if (ns < 1) ns = 1;
    pl.nsplit = (int)ns;
    return pl;
}

template <int DPL, int NQ>
void launch_t(const AttnParams& p, dim3 grid, hipStream_t st) {
    if (p.a.f16_mode)
        attn_kernel<DPL, NQ, true><<<grid, dim3(256), 0, st>>>(p);
    else
        attn_kernel<DPL, NQ, false><<<grid, dim3(256), 0, st>>>(p);
}

}  // namespace

size_t attn_workspace_bytes(const AttnArgs& a) {
    if (attn_prefill_ok(a)) return 0;
    if (attn_prefill_hybrid_ok(a)) {
        const int ns = attn_prefill_hybrid_splits(a);
        return ns <= 1 ? 0 : a.B * a.Hq * a.S * (size_t)ns * (a.D + 2) * sizeof(float);
    }
    const Plan pl = make_plan(a);
    if (pl.nsplit <= 1) return 0;
    return a.B * a.Hq * a.S * (size_t)pl.nsplit * (a.D + 2) * sizeof(float);
}

hipError_t launch_attention_decode_fused(const AttnArgs& a, const AttnFusedIn& f, void* workspace, hipStream_t st) {
    if (!attn_decode_mfma_ok(a) || a.new_len != 1) return hipErrorInvalidValue;
    const Plan pl = make_plan(a);
    if (!pl.mfma) return hipErrorInvalidValue;
    float* ws = reinterpret_cast<float*>(workspace);
    // quantized output rows: inside the launch when a workgroup can hold a row, else by the row kernel behind it
    AttnFusedIn fl = f;
    const bool quant_behind = f.q_out != nullptr && !attn_decode_quant_ok(a, pl.nsplit);
    if (quant_behind) fl.q_out = nullptr;
    hipError_t e = launch_attention_decode_mfma(a, ws, pl.nsplit, st, &fl);
    if (e == hipSuccess && pl.nsplit > 1 && !attn_decode_merge_waves(a, pl.nsplit)) {
        attn_merge_kernel<<<dim3((unsigned)(a.B * a.Hq * a.S)), dim3(256), 0, st>>>(ws, a.out, pl.nsplit, (int)a.D);
        e = hipGetLastError();
    }
    if (e == hipSuccess && quant_behind) e = launch_act_quant(a.out, f.q_out, f.q_scales_out, a.B, a.Hq * a.D, st);
    return e;
}

// The split records of a prefill chunk (attn_prefill_hybrid_kernel) merged per TOKEN: a workgroup owns the Hq * D
// outputs of one query token, so the activation quantization
// that the output projection needs (graph_ops_nn.cpp:35-66: absmax / 127 over the row, round to nearest even) can
// follow in the same launch on the rounded fp16 values -- the same numbers as act_quant_rowwise_kernel gives on
// `out`.  Up to MERGE_TOKEN_PASSES * 2048 outputs per token (two consecutive dims per thread and pass).
constexpr int MERGE_TOKEN_PASSES = 4;
constexpr int MERGE_TOKEN_THREADS = 1024;
__global__ __launch_bounds__(MERGE_TOKEN_THREADS) void attn_merge_token_kernel(
    const float* __restrict__ ws, uint16_t* __restrict__ out, int8_t* __restrict__ q_out, float* __restrict__ q_scales,
    int nsplit, int Hq, int D) {
    __shared__ float red[MERGE_TOKEN_THREADS / 64];
    const size_t tok = blockIdx.x;
    const int QD = Hq * D, tid = threadIdx.x;
    const int rec = D + 2;
    uint32_t packed[MERGE_TOKEN_PASSES];   // two outputs per thread and pass
    float amax = 0.0f;
#pragma unroll
    for (int p = 0; p < MERGE_TOKEN_PASSES; p++) {
        const int e0 = (p * MERGE_TOKEN_THREADS + tid) * 2;
        if (e0 >= QD) break;
        const int h = e0 / D, d0 = e0 - h * D;
        const float* base = ws + (tok * Hq + h) * (size_t)nsplit * rec;
        // loads in batches of 8 splits, all issued before the first use (a record comes from another XCD's L2 or
        // from memory: a microsecond each if waited for one by one)
        float M = -INFINITY;
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            float m[8];
#pragma unroll
            for (int u = 0; u < 8; u++) m[u] = base[(size_t)min(s0 + u, nsplit - 1) * rec];
#pragma unroll
            for (int u = 0; u < 8; u++) M = fmaxf(M, m[u]);
        }
        float L = 0.0f, ox = 0.0f, oy = 0.0f;
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            float2 ml[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float* r = base + (size_t)min(s0 + u, nsplit - 1) * rec;
                ml[u] = *reinterpret_cast<const float2*>(r);
                v[u] = *reinterpret_cast<const float2*>(r + 2 + d0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float f = (M > -INFINITY && s0 + u < nsplit) ? __expf(ml[u].x - M) : 0.0f;
                L = fmaf(ml[u].y, f, L);
                ox = fmaf(v[u].x, f, ox);
                oy = fmaf(v[u].y, f, oy);
            }
        }
        const float inv = (L > 0.0f) ? 1.0f / L : 0.0f;
        const uint16_t a = (L > 0.0f) ? f2h_mul(ox, inv) : (uint16_t)0;
        const uint16_t b = (L > 0.0f) ? f2h_mul(oy, inv) : (uint16_t)0;
        packed[p] = (uint32_t)a | ((uint32_t)b << 16);
        amax = fmaxf(amax, fmaxf(fabsf(h2f(a)), fabsf(h2f(b))));
        *reinterpret_cast<uint32_t*>(out + tok * QD + e0) = packed[p];
    }
    if (q_out == nullptr) return;
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = 0.0f;
#pragma unroll
    for (int i = 0; i < MERGE_TOKEN_THREADS / 64; i++) amax = fmaxf(amax, red[i]);
    float scale = amax / 127.0f;
    if (scale < 1e-10f) scale = 1e-10f;
    if (tid == 0) q_scales[tok] = scale;
    const float qinv = 1.0f / scale;
#pragma unroll
    for (int p = 0; p < MERGE_TOKEN_PASSES; p++) {
        const int e0 = (p * MERGE_TOKEN_THREADS + tid) * 2;
        if (e0 >= QD) break;
        uint32_t w2 = 0;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            float q = h2f((uint16_t)(j ? (packed[p] >> 16) : (packed[p] & 0xFFFFu))) * qinv;
            q = fmaxf(fminf(q, 127.0f), -128.0f);
            w2 |= (uint32_t)(clamp_i8(rne_i32(q)) & 0xFF) << (8 * j);
        }
        *reinterpret_cast<uint16_t*>(q_out + tok * QD + e0) = (uint16_t)w2;
    }
}

// merge launch behind attn_prefill_hybrid_kernel: per token when a token's outputs fit the kernel above and are
// 16-byte pieces, else per (token, head) row
static bool merge_token_ok(const AttnArgs& a, const int8_t* q_out) {
    const size_t QD = a.Hq * a.D;
    return a.D % 2 == 0 && QD <= (size_t)MERGE_TOKEN_PASSES * MERGE_TOKEN_THREADS * 2 && a.B * a.S <= 0x7FFFFFFFull &&
           (reinterpret_cast<uintptr_t>(a.out) & 3) == 0 && (reinterpret_cast<uintptr_t>(q_out) & 1) == 0;
}

hipError_t launch_attention(const AttnArgs& a, void* workspace, hipStream_t st) {
    if (attn_prefill_ok(a)) return launch_attention_prefill(a, st);
    if (attn_prefill_win_ok(a)) return launch_attention_prefill_win(a, st);
    if (attn_prefill_hybrid_ok(a)) {
        hipError_t e = launch_attention_prefill_hybrid(a, reinterpret_cast<float*>(workspace), st);
        const int ns = attn_prefill_hybrid_splits(a);
        if (e == hipSuccess && ns > 1) {
            if (merge_token_ok(a, nullptr))
                attn_merge_token_kernel<<<dim3((unsigned)(a.B * a.S)), dim3(MERGE_TOKEN_THREADS), 0, st>>>(
                    reinterpret_cast<const float*>(workspace), a.out, nullptr, nullptr, ns, (int)a.Hq, (int)a.D);
            else
                attn_merge_kernel<<<dim3((unsigned)(a.B * a.Hq * a.S)), dim3(256), 0, st>>>(
                    reinterpret_cast<const float*>(workspace), a.out, ns, (int)a.D);
            e = hipGetLastError();
        }
        return e;
    }
    const Plan pl = make_plan(a);
    AttnParams p;
    p.a = a;
    p.nsplit = pl.nsplit;
    p.gq = (int)(a.Hq / a.Hkv);
    p.qsubs = p.gq / pl.nq;
    p.G = (int)((a.D + a.qgs - 1) / a.qgs);
    p.ws = reinterpret_cast<float*>(workspace);
    if (pl.items > 0x7FFFFFFFull / 2) return hipErrorInvalidValue;
    if (pl.dpl == 0) {
        const size_t lds = (2 * a.D + 32) * sizeof(float);
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        if (a.f16_mode)
            attn_generic_kernel<true><<<dim3((unsigned)pl.items), dim3(64), lds, st>>>(p);
        else
            attn_generic_kernel<false><<<dim3((unsigned)pl.items), dim3(64), lds, st>>>(p);
        return hipGetLastError();
    }
    if (pl.mfma) {
        hipError_t em = launch_attention_decode_mfma(a, p.ws, pl.nsplit, st);
        if (em == hipSuccess && pl.nsplit > 1 && !attn_decode_merge_waves(a, pl.nsplit)) {
            attn_merge_kernel<<<dim3((unsigned)(a.B * a.Hq * a.S)), dim3(256), 0, st>>>(p.ws, a.out, pl.nsplit, (int)a.D);
            em = hipGetLastError();
        }
        return em;
    }
    const dim3 grid((unsigned)pl.items, (unsigned)pl.nsplit);
    switch (pl.dpl * 10 + pl.nq) {
        case 81: launch_t<8, 1>(p, grid, st); break;
        case 82: launch_t<8, 2>(p, grid, st); break;
        case 84: launch_t<8, 4>(p, grid, st); break;
        case 161: launch_t<16, 1>(p, grid, st); break;
        case 162: launch_t<16, 2>(p, grid, st); break;
        case 321: launch_t<32, 1>(p, grid, st); break;
        default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (pl.nsplit > 1) {
        attn_merge_kernel<<<dim3((unsigned)(a.B * a.Hq * a.S)), dim3(256), 0, st>>>(p.ws, a.out, pl.nsplit,
                                                                                  (int)a.D);
        e = hipGetLastError();
    }
    return e;
}

// attention followed by the row quantization of its output (rows = tokens, Hq * D values each) for the output
// projection: one merge + quantization launch behind a chunk attention that leaves split records, else the
// attention's own launches and the row quantization kernel
hipError_t launch_attention_quant(const AttnArgs& a, void* workspace, int8_t* q_out, float* q_scales, hipStream_t st) {
    if (attn_prefill_hybrid_ok(a) && !attn_prefill_ok(a) && !attn_prefill_win_ok(a) && attn_prefill_hybrid_splits(a) > 1 &&
        merge_token_ok(a, q_out)) {
        hipError_t e = launch_attention_prefill_hybrid(a, reinterpret_cast<float*>(workspace), st);
        if (e != hipSuccess) return e;
        attn_merge_token_kernel<<<dim3((unsigned)(a.B * a.S)), dim3(MERGE_TOKEN_THREADS), 0, st>>>(
            reinterpret_cast<const float*>(workspace), a.out, q_out, q_scales, attn_prefill_hybrid_splits(a), (int)a.Hq, (int)a.D);
        return hipGetLastError();
    }
    hipError_t e = launch_attention(a, workspace, st);
    if (e != hipSuccess) return e;
    return launch_act_quant(a.out, q_out, q_scales, a.B * a.S, a.Hq * a.D, st);
}

}  // namespace cactus_hip
