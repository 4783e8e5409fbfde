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

template <int DPL, int NQ, bool F16MODE>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    __shared__ float s_m[AT_WAVES][NQ];
    __shared__ float s_l[AT_WAVES][NQ];
    __shared__ float s_o[AT_WAVES][NQ][8 * DPL];

    const AttnArgs& a = p.a;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = lane >> 3, c = lane & 7;
    const int D = (int)a.D, Hkv = (int)a.Hkv, Hq = (int)a.Hq, S = (int)a.S;
    const int split = blockIdx.y;
    int item = blockIdx.x;
    const int qp = item % S; item /= S;
    const int qsub = item % p.qsubs; item /= p.qsubs;
    const int kvh = item % Hkv;
    const int b = item / Hkv;
    const int h0 = kvh * p.gq + qsub * NQ;

    const int cl = F16MODE ? 0 : (a.cache_lens ? a.cache_lens[b] : (int)a.cache_len);
    const int new_len = (int)a.new_len;
    const int KV = cl + new_len;
    const long long qabs = (long long)(a.pos_offsets ? (size_t)a.pos_offsets[b] : a.position_offset) + qp;

    long long kv_start = 0, kv_end = KV;
    if (F16MODE) {
        if (a.window_size > 0 && a.window_size < (size_t)KV) {  // :81-87
            if (qabs > (long long)a.window_size) kv_start = qabs - (long long)a.window_size;
            if (a.is_causal) kv_end = min(kv_end, qabs + 1);
        } else if (a.is_causal) {
            kv_end = min(kv_end, qabs + 1);
        }
    } else {
        if (a.is_causal) kv_end = min(kv_end, qabs + 1);      // :331-332
    }
    const int kv_lim = F16MODE ? KV : (int)kv_end;             // :93 vs :335
    const int nblk = kv_end > kv_start ? (int)((kv_end - kv_start + AT_BLOCK - 1) / AT_BLOCK) : 0;
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
    const int sgrp = d0 / (int)a.qgs;

    for (int blk = blk_first + w; blk < blk_last; blk += AT_WAVES) {
        const long long kb = kv_start + (long long)blk * AT_BLOCK;
        const int ke = (int)min((long long)kv_lim, kb + AT_BLOCK);
        const int bsz = ke - (int)kb;
        const int vec = (bsz / 4) * 4;

        float sc[NQ][4];
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const int kp = (int)kb + ps * 8 + t;
            const bool valid = kp < ke;
            float kf[DPL];
            float ksc = 1.0f;
            if (valid && nd > 0) {
                if (!F16MODE && kp < cl) {
                    load_row_i8<DPL>(Kc + ((size_t)kp * Hkv + kvh) * D + d0, nd, kf);
                    ksc = Ks[((size_t)kp * Hkv + kvh) * p.G + sgrp];
                } else {
                    load_row_f16<DPL>(Kn + ((size_t)(kp - cl) * Hkv + kvh) * D + d0, nd, kf);
                }
            } else {
#pragma unroll
                for (int d = 0; d < DPL; d++) kf[d] = 0.0f;
            }
            bool masked = !valid;
            if (F16MODE && valid) {
                if (a.is_causal && kp > qabs) masked = true;                                   // :146
                else if (a.window_size > 0 && kp < qabs && (qabs - kp) > (long long)a.window_size) masked = true;
                else if (Mk && h2f(Mk[kp]) == 0.0f) masked = true;                             // :152
            }
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                float part = 0.0f;
#pragma unroll
                for (int d = 0; d < DPL; d++) part = fmaf(qf[i][d], kf[d], part);
                part *= ksc;
                part += __shfl_xor(part, 1, 64);
                part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 4, 64);
                sc[i][ps] = masked ? -INFINITY : part * scale;
            }
        }

        float wgt[NQ][4];
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            float bm = fmaxf(fmaxf(sc[i][0], sc[i][1]), fmaxf(sc[i][2], sc[i][3]));
            bm = fmaxf(bm, __shfl_xor(bm, 8, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
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
            if (ps * 8 >= bsz) continue;
            const int kp = (int)kb + ps * 8 + t;
            const bool valid = kp < ke;
            float vf[DPL];
            float vsc = 1.0f;
            if (valid && nd > 0) {
                if (!F16MODE && kp < cl) {
                    load_row_i8<DPL>(Vc + ((size_t)kp * Hkv + kvh) * D + d0, nd, vf);
                    vsc = Vs[((size_t)kp * Hkv + kvh) * p.G + sgrp];
                } else {
                    load_row_f16<DPL>(Vn + ((size_t)(kp - cl) * Hkv + kvh) * D + d0, nd, vf);
                }
            } else {
#pragma unroll
                for (int d = 0; d < DPL; d++) vf[d] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const float ww = wgt[i][ps] * vsc;
#pragma unroll
                for (int d = 0; d < DPL; d++) o[i][d] = fmaf(vf[d], ww, o[i][d]);
            }
        }
    }

    // fold the 8 token slots of the wave (lanes with equal c)
#pragma unroll
    for (int i = 0; i < NQ; i++) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            l_run[i] += __shfl_xor(l_run[i], off, 64);
#pragma unroll
            for (int d = 0; d < DPL; d++) o[i][d] += __shfl_xor(o[i][d], off, 64);
        }
    }
    if (t == 0) {
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            if (c == 0) { s_m[w][i] = m_run[i]; s_l[w][i] = l_run[i]; }
#pragma unroll
            for (int d = 0; d < DPL; d++) s_o[w][i][d0 + d] = o[i][d];
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
                L += s_l[ww][i] * f;
                O += s_o[ww][i][d] * f;
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

    const int cl = F16MODE ? 0 : (a.cache_lens ? a.cache_lens[b] : (int)a.cache_len);
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

__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ ws,
                                                        uint16_t* __restrict__ out, int nsplit, int D) {
    const size_t row = blockIdx.x;
    const float* base = ws + row * nsplit * (size_t)(D + 2);
    float M = -INFINITY;
    for (int s = 0; s < nsplit; s++) M = fmaxf(M, base[(size_t)s * (D + 2)]);
    for (int d = threadIdx.x; d < D; d += 256) {
        float L = 0.0f, O = 0.0f;
        if (M > -INFINITY) {
            for (int s = 0; s < nsplit; s++) {
                const float* rec = base + (size_t)s * (D + 2);
                const float f = __expf(rec[0] - M);
                L += rec[1] * f;
                O += rec[2 + d] * f;
            }
        }
        out[row * D + d] = (L > 0.0f) ? f2h_mul(O, 1.0f / L) : (uint16_t)0;
    }
}

namespace {

struct Plan {
    int dpl, nq, nsplit;
    size_t items;
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
    const int maxnq = pl.dpl == 32 ? 2 : 4;
    pl.nq = 1;
    for (int cand = maxnq; cand >= 1; cand >>= 1)
        if (gq % cand == 0) { pl.nq = cand; break; }
    pl.items = a.B * a.Hkv * (size_t)(gq / pl.nq) * a.S;
    const size_t kvmax = (a.f16_mode ? 0 : a.cache_len) + a.new_len;
    const size_t maxblk = (kvmax + AT_BLOCK - 1) / AT_BLOCK;
    size_t ns = pl.items ? 1024 / pl.items : 1;
    const size_t cap = maxblk / 8 ? maxblk / 8 : 1;
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
    const Plan pl = make_plan(a);
    if (pl.nsplit <= 1) return 0;
    return a.B * a.Hq * a.S * (size_t)pl.nsplit * (a.D + 2) * sizeof(float);
}

hipError_t launch_attention(const AttnArgs& a, void* workspace, hipStream_t st) {
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
    const dim3 grid((unsigned)pl.items, (unsigned)pl.nsplit);
    switch (pl.dpl * 10 + pl.nq) {
        case 81: launch_t<8, 1>(p, grid, st); break;
        case 82: launch_t<8, 2>(p, grid, st); break;
        case 84: launch_t<8, 4>(p, grid, st); break;
        case 161: launch_t<16, 1>(p, grid, st); break;
        case 162: launch_t<16, 2>(p, grid, st); break;
        case 164: launch_t<16, 4>(p, grid, st); break;
        case 321: launch_t<32, 1>(p, grid, st); break;
        case 322: launch_t<32, 2>(p, grid, st); break;
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

}  // namespace cactus_hip
