// Prefill-shaped cactus_attention_f16 (cactus/kernel/kernel_attention.cpp:10-255) on the FP16 matrix
// cores: many queries against FP16 K/V, causal or not, head_dim 64 or 128, no window, no mask
// tensor (those go to the VALU kernel in attention.hip).
//
// Roofline: MFMA (FP16) / VALU for the per-block softmax; algorithmic flops 4*S*KV*Hq*D (half of
// that when causal).
//
// The reference's structure is kept: keys in blocks of 32 from key 0, per block the maximum bm of
// the unmasked scores, weights 2^((s - bm) * log2e) by the reference's second-order polynomial
// (:174-195; expf for the last KV % 4 keys, :197-204), running state rescaled by exp(m_run - bm)
// with m_run REPLACED by bm (:160-169), output O / L (:229-252).  What differs is the order of the
// fp32 sums (matrix-core accumulation), held to the tolerance written in the tests.
//
// One 256-thread workgroup = 128 consecutive queries of one (sequence, query head); wave w owns 32
// of them.  Everything is computed transposed so that a LANE owns a QUERY:
//   S^T[key][query] = K . Q^T   v_mfma_f32_32x32x16_f16, A = K rows from LDS, B = the wave's Q rows
//                               (registers for the whole kernel); lane (query = lane % 32,
//                               g = lane / 32) receives 16 of the block's 32 keys, lane ^ 32 the rest
//   softmax per lane            in-lane reductions over its 16 keys, one exchange with lane ^ 32;
//                               the running max / sum / rescale factor are per-lane scalars
//   O^T[d][query] += V^T . P^T  A = V^T from LDS (the block is stored transposed there), B = the
//                               lane's own 16 weights, split into two binary16 parts (hi + lo) so
//                               the product keeps ~22 bits of the fp32 weight
// Any permutation of the MFMA's k index is harmless as long as both operands use it, so the key
// order of the P^T operand is simply the order in which the S^T accumulator holds the keys.
// The next block's K/V rows are loaded into registers while the current block is computed.
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

namespace {

typedef _Float16 v8hf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int PF_BLOCK = 32;     // keys per block (the reference's block size)
constexpr int PF_QTILE = 128;    // queries per workgroup

// weight of one key in its block: kernel_attention.cpp:178-191 (polynomial) / :197-204 (tail)
__device__ __forceinline__ float pf_weight(float s, float bmax, bool poly) {
    if (s == -INFINITY) return 0.0f;
    if (!poly) return __expf(s - bmax);
    const float x = (s - bmax) * 1.442695f;
    const int xi = (int)x;
    const float xf = x - (float)xi;
    float y = fmaf(xf, 0.6931472f, 1.0f);
    y = fmaf(xf * xf, 0.2402265f, y);
    const uint32_t e = ((uint32_t)xi + 127u) << 23;
    return y * __uint_as_float(e);
}

#ifndef PF_WAVES_PER_EU
#define PF_WAVES_PER_EU 2
#endif
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PF_WAVES_PER_EU, PF_WAVES_PER_EU)))
void attn_prefill_kernel(const AttnArgs a) {
    constexpr int KROW = D + 8;            // halves per K row in LDS (16-byte aligned, banks shifted)
    constexpr int VROW = PF_BLOCK + 4;     // halves per V^T row in LDS (8-byte aligned)
    constexpr int NDT = D / 32;            // 32-wide d tiles of the output
    constexpr int NST = D / 16;            // 16-deep d steps of the score MFMAs
    constexpr int PPT = (PF_BLOCK * D / 8) / 256;   // 16-byte pieces per thread per operand (1 or 2)
    __shared__ __attribute__((aligned(16))) uint16_t sK[PF_BLOCK * KROW];
    __shared__ __attribute__((aligned(16))) uint16_t sVt[D * VROW];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jq = lane & 31, g = lane >> 5;
    const int S = (int)a.S, KV = (int)a.new_len, Hq = (int)a.Hq, Hkv = (int)a.Hkv;
    // Work items ranked by weight (a causal tile's work grows with its position): rank k -> tile
    // nt-1 - k / (Hq*B).  The first half of the launch takes the heavy half in descending order, the
    // second half the light half in ascending order, so the two workgroups that share a CU (ids i
    // and i + total/2 when the chip holds the whole launch) add up to the same work everywhere
    // (S = 4096, 16 heads: 341 -> 276 us).  Two LDS stages with one barrier per block measured
    // slower (306 us) than this single stage with two.
    const int nt = (S + PF_QTILE - 1) / PF_QTILE, hb = Hq * (int)a.B, total = nt * hb;
    const int id = (int)blockIdx.x, half = (total + 1) / 2;
    const int rank = id < half ? id : total - 1 - (id - half);
    const int qt = nt - 1 - rank / hb;
    const int h = (rank % hb) % Hq, b = (rank % hb) / Hq;
    const int kvh = h / (Hq / Hkv);
    const int q_first = qt * PF_QTILE + w * 32;         // first query of the wave
    const int qi = q_first + jq;                        // this lane's query
    const bool q_live = qi < S;
    const long long off = (long long)a.position_offset;
    const long long qabs = off + qi;
    float scale = a.scale;
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);

    // keys any query of the workgroup / of the wave can see
    const int q_last_wg = min(S, (qt + 1) * PF_QTILE) - 1, q_last_wave = min(S - 1, q_first + 31);
    const int kv_end_wg = a.is_causal ? (int)min((long long)KV, off + q_last_wg + 1) : KV;
    const int kv_end_wave = a.is_causal ? (int)min((long long)KV, off + q_last_wave + 1) : KV;
    const int nblk = (kv_end_wg + PF_BLOCK - 1) / PF_BLOCK;

    // Q rows of the wave: the B operand of every score MFMA
    v8hf qf[NST];
    {
        const uint16_t* qrow = a.q + (((size_t)b * S + min(qi, S - 1)) * Hq + h) * D;
#pragma unroll
        for (int st = 0; st < NST; st++) qf[st] = *reinterpret_cast<const v8hf*>(qrow + st * 16 + g * 8);
    }

    const uint16_t* Kb = a.k_new + (size_t)b * KV * Hkv * D + (size_t)kvh * D;
    const uint16_t* Vb = a.v_new + (size_t)b * KV * Hkv * D + (size_t)kvh * D;
    const size_t kv_stride = (size_t)Hkv * D;   // halves between consecutive keys

    // staging: a piece = 8 halves (key, d8) of the block.  K pieces follow the rows (coalesced
    // loads, 16-byte LDS stores).  V pieces are taken key-major -- lane = (key = lane % 32, two
    // d8 columns per wave) -- so that the 2-byte stores of the transposition land on distinct
    // banks: row d of V^T is 18 dwords, the 32 keys of a lane group fill 16 consecutive dwords,
    // the wave's second d8 column sits 8 rows = 144 dwords (16 banks) further.
    v4i kst[PPT], vst[PPT];
    const int vkey = lane & 31;
    auto fetch = [&](int blk) {
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int idx = p * 256 + tid, key = idx / (D / 8), d8 = idx % (D / 8);
            const int kp = min(blk * PF_BLOCK + key, KV - 1);
            kst[p] = *reinterpret_cast<const v4i*>(Kb + (size_t)kp * kv_stride + d8 * 8);
            const int vd8 = (p * 4 + w) * 2 + g;
            const int vp = min(blk * PF_BLOCK + vkey, KV - 1);
            vst[p] = *reinterpret_cast<const v4i*>(Vb + (size_t)vp * kv_stride + vd8 * 8);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int idx = p * 256 + tid, key = idx / (D / 8), d8 = idx % (D / 8);
            *reinterpret_cast<v4i*>(sK + key * KROW + d8 * 8) = kst[p];
            const int vd8 = (p * 4 + w) * 2 + g;
#pragma unroll
            for (int e = 0; e < 8; e++)
                sVt[(vd8 * 8 + e) * VROW + vkey] = (uint16_t)(((uint32_t)vst[p][e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
        }
    };

    float m_run = -INFINITY, l_run = 0.0f;
    v16f o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[dt][r] = 0.0f;

    if (nblk > 0) fetch(0);
    for (int blk = 0; blk < nblk; blk++) {
        __syncthreads();                 // the previous block's fragments have been read
        stash();
        __syncthreads();                 // this block is in LDS
        if (blk + 1 < nblk) fetch(blk + 1);

        const int kb = blk * PF_BLOCK;
        if (kb >= kv_end_wave) continue;   // wave-uniform: nothing this wave's queries can see
        const int ke = min(KV, kb + PF_BLOCK);
        const int vec = ((ke - kb) / 4) * 4;

        // scores of 32 keys x 32 queries
        v16f sacc;
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[r] = 0.0f;
#pragma unroll
        for (int st = 0; st < NST; st++) {
            const v8hf kf = *reinterpret_cast<const v8hf*>(sK + jq * KROW + st * 16 + g * 8);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], sacc, 0, 0, 0);
        }

        // This lane's 16 keys: in-block index ib = 8*(r/4) + 4*g + r%4.  A key is visible when
        // ib < lim (the block's keys below KV and, if causal, not after the query); most blocks
        // are visible to the whole wave and whole (32 keys, all on the polynomial) -- that path
        // has no per-element selects.
        const int lim_blk = ke - kb;
        const int lim = a.is_causal ? (int)max(-1LL, min((long long)lim_blk, qabs - kb + 1)) : lim_blk;
        const bool full = lim_blk == PF_BLOCK && (!a.is_causal || off + q_first - kb + 1 >= PF_BLOCK) && q_first + 31 < S;
        const int limg = lim - 4 * g;
        float sc[16], wgt[16];
        float bm = -INFINITY, bsum = 0.0f;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                sc[r] = sacc[r] * scale;
                bm = fmaxf(bm, sc[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                sc[r] = (8 * (r >> 2) + (r & 3) < limg) ? sacc[r] * scale : -INFINITY;
                bm = fmaxf(bm, sc[r]);
            }
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        if (bm > -INFINITY) {            // :160-169, the running maximum is replaced
            const float corr = __expf(m_run - bm);
            l_run *= corr;
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f c2 = {corr, corr};
#pragma unroll
            for (int dt = 0; dt < NDT; dt++)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const v2f t = v2f{o[dt][r], o[dt][r + 1]} * c2;
                    o[dt][r] = t[0];
                    o[dt][r + 1] = t[1];
                }
            m_run = bm;
        }
        // weights: the polynomial of :178-191 evaluated without branches (a masked key's -inf gives
        // garbage that the select discards); the expf tail of :197-204 only exists in a block that
        // ends at KV with KV % 4 != 0
        const int vecg = vec - 4 * g;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float x = (sc[r] - bm) * 1.442695f;
            const int xi = (int)x;
            const float xf = x - (float)xi;
            float y = fmaf(xf, 0.6931472f, 1.0f);
            y = fmaf(xf * xf, 0.2402265f, y);
            float wv = y * __uint_as_float(((uint32_t)xi + 127u) << 23);
            if (!full) {
                if (vec != PF_BLOCK && !(8 * (r >> 2) + (r & 3) < vecg)) wv = __expf(sc[r] - bm);
                wv = (sc[r] == -INFINITY) ? 0.0f : wv;
            }
            wgt[r] = wv;
            bsum += wv;
        }
        bsum += __shfl_xor(bsum, 32);
        l_run += bsum;

        // P^T operands: the lane's weights r = 8*ks .. 8*ks+7, as hi + lo binary16 parts
        v8hf phi[2], plo[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float wv = wgt[8 * ks + e];
                const _Float16 hi = (_Float16)wv;
                phi[ks][e] = hi;
                plo[ks][e] = (_Float16)(wv - (float)hi);
            }
        // V^T operand of (d tile, k step): keys {16ks + 4g + 0..3} and {16ks + 8 + 4g + 0..3} of row d
#pragma unroll
        for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint16_t* vr = sVt + (dt * 32 + jq) * VROW + 16 * ks + 4 * g;
                const uint2 v0 = *reinterpret_cast<const uint2*>(vr);
                const uint2 v1 = *reinterpret_cast<const uint2*>(vr + 8);
                const v4i packed = {(int)v0.x, (int)v0.y, (int)v1.x, (int)v1.y};
                const v8hf vf = __builtin_bit_cast(v8hf, packed);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, phi[ks], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, plo[ks], o[dt], 0, 0, 0);
            }
        }
    }

    if (!q_live) return;
    // O^T accumulators: lane (query, g) holds d = 32*dt + 8*(r/4) + 4*g + r%4
    uint16_t* orow = a.out + (((size_t)b * S + qi) * Hq + h) * D;
    const float inv = (l_run > 0.0f) ? 1.0f / l_run : 0.0f;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            uint32_t lo = 0, hi = 0;
            if (l_run > 0.0f) {
                lo = (uint32_t)f2h_mul(o[dt][4 * r4], inv) | ((uint32_t)f2h_mul(o[dt][4 * r4 + 1], inv) << 16);
                hi = (uint32_t)f2h_mul(o[dt][4 * r4 + 2], inv) | ((uint32_t)f2h_mul(o[dt][4 * r4 + 3], inv) << 16);
            }
            *reinterpret_cast<uint2*>(orow + dt * 32 + 8 * r4 + 4 * g) = make_uint2(lo, hi);
        }
}

}  // namespace

bool attn_prefill_ok(const AttnArgs& a) {
    return a.f16_mode && a.mask == nullptr && a.window_size == 0 && (a.D == 64 || a.D == 128) && a.S >= 32 &&
           a.new_len >= 1 && a.new_len < (size_t(1) << 30) && a.S < (size_t(1) << 30) && a.Hkv > 0 &&
           a.Hq % a.Hkv == 0 && a.Hq <= 65535 && a.B <= 65535 && a.Hq * a.B <= 1048576 &&
           ((reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.k_new) |
             reinterpret_cast<uintptr_t>(a.v_new) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0;
}

hipError_t launch_attention_prefill(const AttnArgs& a, hipStream_t st) {
    if (!attn_prefill_ok(a)) return hipErrorInvalidValue;
    const size_t total = ((a.S + PF_QTILE - 1) / PF_QTILE) * a.Hq * a.B;
    if (total > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)total), block(256);
    if (a.D == 128) attn_prefill_kernel<128><<<grid, block, 0, st>>>(a);
    else attn_prefill_kernel<64><<<grid, block, 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace cactus_hip
