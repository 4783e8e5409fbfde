// Prefill-shaped cactus_attention_f16 (cactus/kernel/kernel_attention.cpp:10-255) on the FP16 matrix
// cores: many queries against FP16 K/V, causal or not, head_dim 64, 128 or 256, no mask tensor and no window
// that cuts anything (those go to the VALU kernel in attention.hip).
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

#include <type_traits>

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
// head_dim 256: the Q rows and the output accumulators of a lane alone are 192 registers, so that variant runs one
// wave per SIMD with the whole register file
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(D == 256 ? 1 : PF_WAVES_PER_EU, D == 256 ? 1 : PF_WAVES_PER_EU)))
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

        // One block's arithmetic.  FULL: every one of the block's 32 keys is visible to every query of the wave
        // (all but the blocks on the causal diagonal and the last one): no masks, no selects, no expf tail.
        auto block_math = [&](auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
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
            const int limg = lim - 4 * g;
            float sc[16], wgt[16];
            float bm = -INFINITY, bsum = 0.0f;
            if constexpr (FULL) {
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
            if (FULL || bm > -INFINITY) {    // :160-169, the running maximum is replaced
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
                if constexpr (!FULL) {
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
        };
        const int lim_blk_w = ke - kb;
        if (lim_blk_w == PF_BLOCK && (!a.is_causal || off + q_first - kb + 1 >= PF_BLOCK) && q_first + 31 < S) block_math(std::true_type{});
        else block_math(std::false_type{});
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

// ---------------------------------------------------------------------------------------
// cactus_attention_hybrid_int8_fp16 (kernel_attention.cpp:257-482) for a chunk of many queries:
// the same transposed MFMA scheme over the INT8 cache followed by the chunk's own FP16 keys.
//  - A cached block is converted to binary16 on its way into LDS (an int8 is exact there); the
//    per-key scales (one group per head: head_dim <= group size; two for head_dim 256) multiply the score
//    column and the weight of the key, as in attn_kernel: s = (q . k8) * k_scale * scale, O += (w * v_scale) * v8.
//  - Standard online softmax with expf (:395-414), so partial results over disjoint key ranges
//    combine exactly like the decode kernel's: the key blocks of a query tile are cut into
//    `nsplit` ranges, each a workgroup of its own writing (max, sum, unnormalised O) records that
//    attn_merge_kernel reduces.  A chunk has few query tiles and a long cache: without the split the
//    launch is a handful of workgroups each walking the whole cache.
// ---------------------------------------------------------------------------------------
struct HybridPlan { int nsplit; };

// 4 int8 -> 4 binary16 (exact): bytes ^ 0x80 are u = x + 128; 0x6400 | u is 1024 + u; minus 1152
__device__ __forceinline__ uint2 i8x4_to_f16x4(uint32_t x) {
    typedef _Float16 v2hf __attribute__((ext_vector_type(2)));
    const uint32_t y = x ^ 0x80808080u;
    const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, y, 0x04010400u);
    const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, y, 0x04030402u);
    const v2hf bias = {(_Float16)1152.0f, (_Float16)1152.0f};
    const v2hf a = __builtin_bit_cast(v2hf, lo) - bias, b = __builtin_bit_cast(v2hf, hi) - bias;
    return make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
}

// head_dim 256 (Gemma): two scale groups of 128 dims per cached row (kernel.h:18) - each group's half of the
// score is scaled and added, and each half of the output dims takes weights times its own V scale; one wave per
// SIMD with the whole register file, as in attn_prefill_kernel<256>.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(D == 256 ? 1 : PF_WAVES_PER_EU, D == 256 ? 1 : PF_WAVES_PER_EU)))
void attn_prefill_hybrid_kernel(const AttnArgs a, float* __restrict__ ws, int nsplit) {
    constexpr int KROW = D + 8, VROW = PF_BLOCK + 4, NDT = D / 32, NST = D / 16;
    constexpr int D16 = D / 16;            // 16-element pieces per row: a thread stages (key, d16)
    constexpr int NP = (D16 + 7) / 8;      // staging passes: 8 pieces per key and pass
    constexpr int NG = D == 256 ? 2 : 1;   // scale groups per cached row
    __shared__ __attribute__((aligned(16))) uint16_t sK[PF_BLOCK * KROW];
    __shared__ __attribute__((aligned(16))) uint16_t sVt[D * VROW];
    __shared__ __attribute__((aligned(16))) float sKs[NG][PF_BLOCK], sVs[NG][PF_BLOCK];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jq = lane & 31, g = lane >> 5;
    const int S = (int)a.S, Hq = (int)a.Hq, Hkv = (int)a.Hkv, new_len = (int)a.new_len;
    const int nt = (S + PF_QTILE - 1) / PF_QTILE;
    int item = (int)blockIdx.x;
    const int split = item % nsplit; item /= nsplit;
    const int qt = nt - 1 - item % nt; item /= nt;
    const int h = item % Hq, b = item / Hq;
    const int kvh = h / (Hq / Hkv);
    const int q_first = qt * PF_QTILE + w * 32;
    const int qi = q_first + jq;
    const bool q_live = qi < S;
    const int cl = a.cache_lens ? a.cache_lens[b] : (int)a.cache_len;
    const int KV = cl + new_len;
    const long long off = (long long)(a.pos_offsets ? (size_t)a.pos_offsets[b] : a.position_offset);
    float scale = a.scale;
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);

    // keys visible to a query: [0, kv_end); cached blocks cover [0, min(cl, kv_end)), the chunk's own
    // FP16 keys [cl, kv_end) in blocks of their own (:331-334, :351-388)
    auto kv_end_of = [&](int q) { return a.is_causal ? (int)max(0LL, min((long long)KV, off + q + 1)) : KV; };
    const int kv_end_wg = kv_end_of(min(S, (qt + 1) * PF_QTILE) - 1);
    const int kv_end_wave_hi = kv_end_of(min(S - 1, q_first + 31));   // the wave's last query sees the most
    const int kv_end_wave_lo = kv_end_of(q_first);
    const int kv_end = kv_end_of(qi);
    const int nbc = (min(cl, kv_end_wg) + PF_BLOCK - 1) / PF_BLOCK;
    const int nbn = kv_end_wg > cl ? (kv_end_wg - cl + PF_BLOCK - 1) / PF_BLOCK : 0;
    const int nblk = nbc + nbn;
    const int bps = (nblk + nsplit - 1) / nsplit;
    const int blk_first = split * bps, blk_last = min(nblk, blk_first + bps);

    v8hf qf[NST];
    {
        const uint16_t* qrow = a.q + (((size_t)b * S + min(qi, S - 1)) * Hq + h) * D;
#pragma unroll
        for (int st = 0; st < NST; st++) qf[st] = *reinterpret_cast<const v8hf*>(qrow + st * 16 + g * 8);
    }
    const int8_t* Kc = a.k_cached + (size_t)b * a.cache_batch_stride + (size_t)kvh * D;
    const int8_t* Vc = a.v_cached + (size_t)b * a.cache_batch_stride + (size_t)kvh * D;
    const float* Ks = a.k_scales + (size_t)b * a.scale_batch_stride + kvh * NG;
    const float* Vs = a.v_scales + (size_t)b * a.scale_batch_stride + kvh * NG;
    const uint16_t* Kn = a.k_new + (size_t)b * new_len * Hkv * D + (size_t)kvh * D;
    const uint16_t* Vn = a.v_new + (size_t)b * new_len * Hkv * D + (size_t)kvh * D;
    const size_t row_stride = (size_t)Hkv * D;   // elements between consecutive tokens

    // staging: thread = (key = lane % 32, d16 = 8 * pass + 2 * wave + lane / 32): 16 elements of K and of V
    const int skey = lane & 31, sd16 = w * 2 + g;
    const bool stager = sd16 < D16;
    v4i kst[NP][2], vst[NP][2];  // cached: [0] holds 16 int8; FP16: two pieces of 8 halves
    float ssv = 1.0f;            // threads 0 .. 64 NG - 1: one scale of the block (K / V alternate by wave half)
    const int s_grp = tid >> 6, s_isv = (tid >> 5) & 1;
    auto fetch = [&](int blk) {
        if (blk < nbc) {
            const int kp = min(blk * PF_BLOCK + skey, cl - 1);
            if (stager) {
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    kst[p][0] = *reinterpret_cast<const v4i*>(Kc + (size_t)kp * row_stride + (p * 8 + sd16) * 16);
                    vst[p][0] = *reinterpret_cast<const v4i*>(Vc + (size_t)kp * row_stride + (p * 8 + sd16) * 16);
                }
            }
            if (tid < 64 * NG) ssv = (s_isv ? Vs : Ks)[(size_t)kp * Hkv * NG + s_grp];
        } else {
            const int kp = min((blk - nbc) * PF_BLOCK + skey, new_len - 1);
            if (stager) {
#pragma unroll
                for (int p = 0; p < NP; p++)
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        kst[p][u] = *reinterpret_cast<const v4i*>(Kn + (size_t)kp * row_stride + (p * 8 + sd16) * 16 + u * 8);
                        vst[p][u] = *reinterpret_cast<const v4i*>(Vn + (size_t)kp * row_stride + (p * 8 + sd16) * 16 + u * 8);
                    }
            }
            ssv = 1.0f;
        }
    };
    auto stash = [&](int blk) {
        if (stager) {
#pragma unroll
            for (int p = 0; p < NP; p++) {
                v4i kh[2], vh[2];
                if (blk < nbc) {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint2 k0 = i8x4_to_f16x4((uint32_t)kst[p][0][2 * u]), k1 = i8x4_to_f16x4((uint32_t)kst[p][0][2 * u + 1]);
                        const uint2 v0 = i8x4_to_f16x4((uint32_t)vst[p][0][2 * u]), v1 = i8x4_to_f16x4((uint32_t)vst[p][0][2 * u + 1]);
                        kh[u] = v4i{(int)k0.x, (int)k0.y, (int)k1.x, (int)k1.y};
                        vh[u] = v4i{(int)v0.x, (int)v0.y, (int)v1.x, (int)v1.y};
                    }
                } else {
                    kh[0] = kst[p][0]; kh[1] = kst[p][1]; vh[0] = vst[p][0]; vh[1] = vst[p][1];
                }
                const int d0 = (p * 8 + sd16) * 16;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    *reinterpret_cast<v4i*>(sK + skey * KROW + d0 + u * 8) = kh[u];
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        sVt[(d0 + u * 8 + e) * VROW + skey] = (uint16_t)(((uint32_t)vh[u][e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
                }
            }
        }
        if (tid < 64 * NG) (s_isv ? sVs : sKs)[s_grp][skey] = ssv;
    };

    float m_run = -INFINITY, l_run = 0.0f;
    v16f o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[dt][r] = 0.0f;

    if (blk_first < blk_last) fetch(blk_first);
    for (int blk = blk_first; blk < blk_last; blk++) {
        __syncthreads();
        stash(blk);
        __syncthreads();
        if (blk + 1 < blk_last) fetch(blk + 1);

        const bool cached = blk < nbc;
        const int kb = cached ? blk * PF_BLOCK : cl + (blk - nbc) * PF_BLOCK;    // first key (absolute)
        const int seg_end_hi = cached ? min(cl, kv_end_wave_hi) : kv_end_wave_hi;
        if (kb >= seg_end_hi) continue;    // wave-uniform: nothing this wave's queries can see
        const int lim = (cached ? min(cl, kv_end) : kv_end) - kb;               // this query: ib < lim visible
        const bool full = (cached ? min(cl, kv_end_wave_lo) : kv_end_wave_lo) - kb >= PF_BLOCK && q_first + 31 < S;

        // One block's arithmetic.  FULL: all 32 keys are visible to every query of the wave: no masks.
        auto block_math = [&](auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            v16f sacc[NG];
#pragma unroll
            for (int n = 0; n < NG; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) sacc[n][r] = 0.0f;
#pragma unroll
            for (int st = 0; st < NST; st++) {
                const v8hf kf = *reinterpret_cast<const v8hf*>(sK + jq * KROW + st * 16 + g * 8);
                sacc[st * NG / NST] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], sacc[st * NG / NST], 0, 0, 0);
            }
            // per-key scales of this lane's 16 keys (in-block index 8*(r/4) + 4*g + r%4)
            float ksc[NG][16], vsc[NG][16];
#pragma unroll
            for (int n = 0; n < NG; n++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const v4f kq = *reinterpret_cast<const v4f*>(&sKs[n][8 * r4 + 4 * g]);
                    const v4f vq = *reinterpret_cast<const v4f*>(&sVs[n][8 * r4 + 4 * g]);
#pragma unroll
                    for (int e = 0; e < 4; e++) { ksc[n][4 * r4 + e] = kq[e]; vsc[n][4 * r4 + e] = vq[e]; }
                }
            const int limg = lim - 4 * g;
            float sc[16], wgt[16];
            float bm = -INFINITY, bsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float sv = sacc[0][r] * ksc[0][r];
                if constexpr (NG == 2) sv += sacc[1][r] * ksc[1][r];
                sv *= scale;
                sc[r] = (FULL || 8 * (r >> 2) + (r & 3) < limg) ? sv : -INFINITY;
                bm = fmaxf(bm, sc[r]);
            }
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float mn = fmaxf(m_run, bm);
            // :395-414.  Once the running maximum has settled no lane's changes, the factor is exactly 1 in every
            // lane and the accumulators are left alone (a wave-uniform test).
            if (__builtin_amdgcn_ballot_w64(mn > m_run) != 0) {
                const float corr = mn > m_run ? __expf(m_run - mn) : 1.0f;   // a lane whose maximum stays (or that has seen no key yet)
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
                m_run = mn;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                wgt[r] = (!FULL && sc[r] == -INFINITY) ? 0.0f : __expf(sc[r] - mn);
                bsum += wgt[r];
            }
            bsum += __shfl_xor(bsum, 32);
            l_run += bsum;

            v8hf phi[NG][2], plo[NG][2];
#pragma unroll
            for (int n = 0; n < NG; n++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float wv = wgt[8 * ks + e] * vsc[n][8 * ks + e];
                        const _Float16 hi = (_Float16)wv;
                        phi[n][ks][e] = hi;
                        plo[n][ks][e] = (_Float16)(wv - (float)hi);
                    }
#pragma unroll
            for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const uint16_t* vr = sVt + (dt * 32 + jq) * VROW + 16 * ks + 4 * g;
                    const uint2 v0 = *reinterpret_cast<const uint2*>(vr);
                    const uint2 v1 = *reinterpret_cast<const uint2*>(vr + 8);
                    const v4i packed = {(int)v0.x, (int)v0.y, (int)v1.x, (int)v1.y};
                    const v8hf vf = __builtin_bit_cast(v8hf, packed);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, phi[dt * NG / NDT][ks], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, plo[dt * NG / NDT][ks], o[dt], 0, 0, 0);
                }
            }
        };
        if (full) block_math(std::true_type{});
        else block_math(std::false_type{});
    }

    if (!q_live) return;
    const size_t orow = ((size_t)b * S + qi) * Hq + h;
    if (nsplit == 1) {
        uint16_t* op = a.out + orow * D;
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
                *reinterpret_cast<uint2*>(op + dt * 32 + 8 * r4 + 4 * g) = make_uint2(lo, hi);
            }
    } else {      // (max, sum, unnormalised output) of this key range: attn_merge_kernel's record
        float* rec = ws + (orow * nsplit + split) * (size_t)(D + 2);
        if (g == 0) { rec[0] = m_run; rec[1] = l_run; }
#pragma unroll
        for (int dt = 0; dt < NDT; dt++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                float* dst = rec + 2 + dt * 32 + 8 * r4 + 4 * g;
                *reinterpret_cast<float2*>(dst) = make_float2(o[dt][4 * r4], o[dt][4 * r4 + 1]);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(o[dt][4 * r4 + 2], o[dt][4 * r4 + 3]);
            }
    }
}

}  // namespace

bool attn_prefill_hybrid_ok(const AttnArgs& a) {
    return !a.f16_mode && a.S >= 32 && (((a.D == 64 || a.D == 128) && a.qgs >= a.D) || (a.D == 256 && a.qgs == 128)) &&
           a.Hkv > 0 && a.Hq % a.Hkv == 0 &&
           a.S < (size_t(1) << 30) && a.cache_len + a.new_len < (size_t(1) << 30) && a.cache_len + a.new_len >= 1 &&
           a.Hq * a.B <= 1048576 &&
           (a.cache_len == 0 || (a.k_cached && a.v_cached && a.k_scales && a.v_scales &&
                                 ((reinterpret_cast<uintptr_t>(a.k_cached) | reinterpret_cast<uintptr_t>(a.v_cached)) & 15) == 0 &&
                                 a.cache_batch_stride % 16 == 0)) &&
           (a.new_len == 0 || (a.k_new && a.v_new &&
                               ((reinterpret_cast<uintptr_t>(a.k_new) | reinterpret_cast<uintptr_t>(a.v_new)) & 15) == 0)) &&
           ((reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0;
}

// key splits: about 512 workgroups (1024 and 2048 measured slower: more records to write and merge;
// 256 slower on the longest caches), at least two blocks per split, at
// most 256 (the merge kernel keeps one factor per split in LDS)
int attn_prefill_hybrid_splits(const AttnArgs& a) {
    const size_t tiles = ((a.S + PF_QTILE - 1) / PF_QTILE) * a.Hq * a.B;
    const size_t blocks = (a.cache_len + PF_BLOCK - 1) / PF_BLOCK + (a.new_len + PF_BLOCK - 1) / PF_BLOCK;
#ifndef PF_SPLIT_TARGET
#define PF_SPLIT_TARGET 512
#endif
#ifndef PF_SPLIT_MINBLK
#define PF_SPLIT_MINBLK 2
#endif
    size_t ns = tiles ? (PF_SPLIT_TARGET + tiles - 1) / tiles : 1;
    const size_t cap = blocks / PF_SPLIT_MINBLK ? blocks / PF_SPLIT_MINBLK : 1;
    if (ns > cap) ns = cap;
    if (ns > 256) ns = 256;
    if (ns < 1) ns = 1;
    return (int)ns;
}

hipError_t launch_attention_prefill_hybrid(const AttnArgs& a, float* ws, hipStream_t st) {
    if (!attn_prefill_hybrid_ok(a)) return hipErrorInvalidValue;
    const int nsplit = attn_prefill_hybrid_splits(a);
    if (nsplit > 1 && ws == nullptr) return hipErrorInvalidValue;
    const size_t total = ((a.S + PF_QTILE - 1) / PF_QTILE) * a.Hq * a.B * (size_t)nsplit;
    if (total > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)total), block(256);
    if (a.D == 256) attn_prefill_hybrid_kernel<256><<<grid, block, 0, st>>>(a, ws, nsplit);
    else if (a.D == 128) attn_prefill_hybrid_kernel<128><<<grid, block, 0, st>>>(a, ws, nsplit);
    else attn_prefill_hybrid_kernel<64><<<grid, block, 0, st>>>(a, ws, nsplit);
    return hipGetLastError();
}

bool attn_prefill_ok(const AttnArgs& a) {
    // a window that no (query, key) pair of the call can exceed does nothing (kernel_attention.cpp:78-90, 149-151:
    // it moves kv_start only when window < KV and masks a key only when more than `window` behind its query)
    const bool no_window = a.window_size == 0 ||
                           (a.window_size >= a.new_len && a.position_offset + a.S <= a.window_size + 1);
    return a.f16_mode && a.mask == nullptr && no_window && (a.D == 64 || a.D == 128 || a.D == 256) && a.S >= 32 &&
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
    if (a.D == 256) attn_prefill_kernel<256><<<grid, block, 0, st>>>(a);
    else if (a.D == 128) attn_prefill_kernel<128><<<grid, block, 0, st>>>(a);
    else attn_prefill_kernel<64><<<grid, block, 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace cactus_hip
