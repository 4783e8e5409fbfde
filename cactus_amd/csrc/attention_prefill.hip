// Prefill-shaped cactus_attention_f16 (cactus/kernel/kernel_attention.cpp:10-255) on the FP16 matrix
// cores: many queries against FP16 K/V, causal or not, head_dim 64, 128 or 256, no mask tensor (that goes to the
// VALU kernel in attention.hip, as does a window on non-causal attention); a causal window that cuts visible keys
// has its own kernel below (attn_prefill_win_kernel).
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

// Lanes l and l ^ 32 combine a value without a trip through LDS: v_permlane32_swap exchanges the upper 32 lanes of its
// first operand with the lower 32 of its second, so from two copies of x one result holds the lower half's values in
// every lane and the other the upper half's.
__device__ __forceinline__ float halves_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float halves_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifndef PF_WAVES_PER_EU
#define PF_WAVES_PER_EU 2
#endif
// Timing experiments only (wrong results): 1 no staging and no barriers, 2 no weight polynomial, 4 no P.V MFMAs,
// 8 no score MFMAs, 16 no low-part MFMAs, 32 no rescale of the running output
#ifndef PF_TIMING
#define PF_TIMING 0
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
        if (!(PF_TIMING & 1)) {
        __syncthreads();                 // the previous block's fragments have been read
        stash();
        __syncthreads();                 // this block is in LDS
        if (blk + 1 < nblk) fetch(blk + 1);
        }

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
                if (!(PF_TIMING & 8)) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], sacc, 0, 0, 0);
                else sacc[st] += (float)kf[0];
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
            bm = halves_max(bm);
            if (FULL || bm > -INFINITY) {    // :160-169, the running maximum is replaced
                const float corr = __expf(m_run - bm);
                l_run *= corr;
                typedef float v2f __attribute__((ext_vector_type(2)));
                const v2f c2 = {corr, corr};
                if (!(PF_TIMING & 32))
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
                if (PF_TIMING & 2) wv = sc[r];
                if constexpr (!FULL) {
                    if (vec != PF_BLOCK && !(8 * (r >> 2) + (r & 3) < vecg)) wv = __expf(sc[r] - bm);
                    wv = (sc[r] == -INFINITY) ? 0.0f : wv;
                }
                wgt[r] = wv;
                bsum += wv;
            }
            bsum = halves_sum(bsum);
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
                    if (PF_TIMING & 4) { o[dt][ks] += (float)vf[0] * (float)phi[ks][dt] + (float)plo[ks][dt]; continue; }
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, phi[ks], o[dt], 0, 0, 0);
                    if (!(PF_TIMING & 16)) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, plo[ks], o[dt], 0, 0, 0);
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
// A sliding window that cuts (Gemma's local layers once the prompt is longer than the window; kernel_attention.cpp:
// 78-90, 149-151), causal.  The reference starts a query's 32-key blocks at ITS first visible key,
// o(q) = q_abs - window, so every query of a wave has its own block boundaries, and the polynomial weights depend on
// them (each block has its own maximum).  The matrix cores still work on tiles that all queries share: tiles of 32
// keys from the origin of the workgroup's first query, B0.  A lane's origin sits s = (o(q) - B0) mod 32 keys into a
// tile, so its block j is the last 32 - s keys of one tile followed by the first s keys of the next, and the loop runs
// one tile behind its scores: with the scores of tile t in hand, a lane closes the block made of tile t-1's keys
// ib >= s and tile t's keys ib < s (maximum, rescale of the running state, weights), which completes the weights of
// tile t-1 -- its keys below s were weighted one step earlier as the tail of the previous block and are brought to the
// new maximum by the same factor as the running state -- and P.V of tile t-1 follows.  LDS therefore holds K of tile
// t beside V of tile t-1.  Keys before a lane's origin, after its query or past KV are masked; a block with no
// visible key (lanes whose origin lies in a later tile) leaves the state alone, as the reference's blocks do.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(D == 256 ? 1 : 2, D == 256 ? 1 : 2)))
void attn_prefill_win_kernel(const AttnArgs a) {
    constexpr int KROW = D + 8;
    constexpr int VROW = PF_BLOCK + 4;
    constexpr int NDT = D / 32;
    constexpr int NST = D / 16;
    constexpr int PPT = (PF_BLOCK * D / 8) / 256;
    __shared__ __attribute__((aligned(16))) uint16_t sK[PF_BLOCK * KROW];
    __shared__ __attribute__((aligned(16))) uint16_t sVt[D * VROW];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jq = lane & 31, g = lane >> 5;
    const int S = (int)a.S, KV = (int)a.new_len, Hq = (int)a.Hq, Hkv = (int)a.Hkv;
    const int nt = (S + PF_QTILE - 1) / PF_QTILE, hb = Hq * (int)a.B;
    const int id = (int)blockIdx.x;
    const int qt = id / hb;
    const int h = (id % hb) % Hq, b = (id % hb) / Hq;
    (void)nt;
    const int kvh = h / (Hq / Hkv);
    const int q_first = qt * PF_QTILE + w * 32;
    const int qi = q_first + jq;
    const bool q_live = qi < S;
    const long long off = (long long)a.position_offset, win = (long long)a.window_size;
    const long long qabs = off + min(qi, S - 1);
    float scale = a.scale;
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);

    const bool cut = a.window_size < a.new_len;   // :81 -- only then does the window move a query's first block
    auto origin = [&](long long qa) -> long long { return (cut && qa > win) ? qa - win : 0; };
    const long long B0 = origin(off + (long long)qt * PF_QTILE);
    const long long o_q = origin(qabs);
    const int s = (int)((o_q - B0) & 31);
    // visible keys of this lane: lo_q <= key <= hi_q
    const long long lo_q = max(o_q, qabs - win);
    const long long hi_q = q_live ? min(qabs, (long long)KV - 1) : -1;

    const int q_last_wg = min(S, (qt + 1) * PF_QTILE) - 1, q_last_wave = min(S - 1, q_first + 31);
    const long long kv_end_wg = min((long long)KV, off + q_last_wg + 1);
    const long long kv_end_wave = min((long long)KV, off + q_last_wave + 1);
    const long long lo_wave = origin(off + min(q_first, S - 1));
    const bool wave_live = q_first < S;
    const int nT = kv_end_wg > B0 ? (int)((kv_end_wg - B0 + PF_BLOCK - 1) / PF_BLOCK) : 0;   // tiles; one more step closes the last block

    v8hf qf[NST];
    {
        const uint16_t* qrow = a.q + (((size_t)b * S + min(qi, S - 1)) * Hq + h) * D;
#pragma unroll
        for (int st = 0; st < NST; st++) qf[st] = *reinterpret_cast<const v8hf*>(qrow + st * 16 + g * 8);
    }
    const uint16_t* Kb = a.k_new + (size_t)b * KV * Hkv * D + (size_t)kvh * D;
    const uint16_t* Vb = a.v_new + (size_t)b * KV * Hkv * D + (size_t)kvh * D;
    const size_t kv_stride = (size_t)Hkv * D;

    // staging as in attn_prefill_kernel, K of tile t and V of tile t-1 (rows clamped into the matrix: their weights are 0)
    v4i kst[PPT], vst[PPT];
    const int vkey = lane & 31;
    auto row_of = [&](long long key) -> size_t { return (size_t)max(0LL, min(key, (long long)KV - 1)); };
    auto fetch = [&](int t) {
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int idx = p * 256 + tid, key = idx / (D / 8), d8 = idx % (D / 8);
            kst[p] = *reinterpret_cast<const v4i*>(Kb + row_of(B0 + (long long)t * PF_BLOCK + key) * kv_stride + d8 * 8);
            const int vd8 = (p * 4 + w) * 2 + g;
            vst[p] = *reinterpret_cast<const v4i*>(Vb + row_of(B0 + (long long)(t - 1) * PF_BLOCK + vkey) * kv_stride + vd8 * 8);
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
    float prev[16], carry[16];   // masked scores of tile t-1; weights of its keys ib < s (the tail of the block before)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        prev[r] = -INFINITY;
        carry[r] = 0.0f;
    }

    if (nT > 0) fetch(0);
    for (int t = 0; t <= nT && nT > 0; t++) {
        __syncthreads();
        stash();
        __syncthreads();
        if (t + 1 <= nT) fetch(t + 1);

        const long long kb = B0 + (long long)t * PF_BLOCK;   // first key of tile t
        // wave-uniform: neither tile t-1 nor tile t holds a key that a query of this wave can see
        if (!wave_live || kb + PF_BLOCK <= lo_wave || kb - PF_BLOCK >= kv_end_wave) continue;

        v16f sacc;
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[r] = 0.0f;
#pragma unroll
        for (int st = 0; st < NST; st++) {
            const v8hf kf = *reinterpret_cast<const v8hf*>(sK + jq * KROW + st * 16 + g * 8);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], sacc, 0, 0, 0);
        }
        // this lane's 16 keys of a tile: ib = 8*(r/4) + 4*g + r%4
        const int lo_i = (int)max(-64LL, min(64LL, lo_q - kb)) - 4 * g, hi_i = (int)max(-64LL, min(64LL, hi_q - kb)) - 4 * g;
        const int sg = s - 4 * g;
        float z[16];
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ibg = 8 * (r >> 2) + (r & 3);
            const float cur = (ibg >= lo_i && ibg <= hi_i) ? sacc[r] * scale : -INFINITY;
            z[r] = (ibg >= sg) ? prev[r] : cur;   // the key of the block being closed that this register pair holds
            prev[r] = cur;
            bm = fmaxf(bm, z[r]);
        }
        bm = halves_max(bm);
        if (bm > -INFINITY) {    // :160-169, the running maximum is replaced
            const float corr = __expf(m_run - bm);
            l_run *= corr;
#pragma unroll
            for (int dt = 0; dt < NDT; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[dt][r] *= corr;
#pragma unroll
            for (int r = 0; r < 16; r++) carry[r] *= corr;
            m_run = bm;
        }
        // the block ends at KV when fewer than 32 keys remain from its start: its last size % 4 keys use expf (:172, 197-204)
        const long long bstart = kb - PF_BLOCK + s;
        const int bsz = (int)min((long long)PF_BLOCK, (long long)KV - bstart);
        const int vecn = bsz & ~3;
        const bool near_end = kb + PF_BLOCK - 1 > (long long)KV;   // wave-uniform: some lane's block may be cut by KV
        float wgt[16];
        float bsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ibg = 8 * (r >> 2) + (r & 3);
            const float x = (z[r] - bm) * 1.442695f;
            const int xi = (int)x;
            const float xf = x - (float)xi;
            float y = fmaf(xf, 0.6931472f, 1.0f);
            y = fmaf(xf * xf, 0.2402265f, y);
            float wv = y * __uint_as_float(((uint32_t)xi + 127u) << 23);
            if (near_end && ((ibg - sg) & 31) >= vecn) wv = __expf(z[r] - bm);
            wv = (z[r] == -INFINITY) ? 0.0f : wv;
            bsum += wv;
            const bool head = ibg >= sg;           // a key of tile t-1; else of tile t: kept for the next step
            wgt[r] = head ? wv : carry[r];
            carry[r] = head ? 0.0f : wv;
        }
        bsum = halves_sum(bsum);
        l_run += bsum;

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
        if (t == 0) continue;   // there is no tile before the first
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
// The same arithmetic for head_dim 64 and 128, arranged so that the matrix cores and the vector ALU of a SIMD work at
// the same time.  A 512-thread workgroup = 128 queries x two GROUPS of four waves; a group takes one half of the
// tile's key blocks and keeps its own (m, l, O^T) state and its own K/V block in LDS.  Waves w and w + 4 - the same
// 32 queries in the two groups - share a SIMD, and the groups run half a period apart:
//     group 0:  M(0) | V(0) | M(1) | V(1) | ...          M(i): P(i-1).V(i-1) and K(i).Q^T on the matrix cores
//     group 1:       | M(0) | V(0) | M(1) | ...          V(i): softmax of block i on the vector ALU
// with a workgroup barrier at every '|', so one wave of each SIMD is in its MFMA phase while the other is in its VALU
// phase (left to themselves two waves of a SIMD drift into the SAME phase and each unit idles half of the time).
// The barriers are also what orders the staging: block i+1 of K and block i of V are loaded into registers at the
// start of M(i) and written to LDS at the start of V(i), after every wave of the group has finished M(i) (the
// registers are free during the softmax, where the kernel needs the most).
// V rows stay row-major in LDS; the V^T operand is read with the transposing ds_read_b64_tr_b16.
// The two partial states of a query combine exactly as the reference's chain of rescales does across the cut:
// O = O0 * exp(m0 - m1) + O1 with m0, m1 the block maxima at which the halves ended (:160-169).
// Benchmarks only (-DPF_TRACE): per-phase shader-clock totals of the waves of workgroups 0 and 1.
#ifdef PF_TRACE
__device__ unsigned long long g_pf_trace[16 * 8];
#define PF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define PF_T(var)
#endif

// NWQ: waves (32 queries each) per group.  4: the 512-thread workgroup above.  2: 256 threads and 64 queries, for
// launches that would not fill the chip with the larger workgroups - a tile's chain of key blocks is as long, but the
// four waves then have a SIMD each when the launch is at most one workgroup per CU.  (Register budget as for
// NWQ = 4: with the whole file the compiler parks the output accumulators in AGPRs and moves them around every rescale.)
template <int D, int NWQ>
__global__ __launch_bounds__(2 * NWQ * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_prefill_pp_kernel(const AttnArgs a) {
    constexpr int GT = NWQ * 64;     // threads of a group
    constexpr int QT = NWQ * 32;     // queries of a workgroup
    constexpr int KROW = D + 8;      // halves per K row: banks shift by 4 per row, the 16-byte row reads are conflict-free
    constexpr int VROW = D + 32;     // halves per V row: banks shift by 16 per row, the 4-row transposed reads are conflict-free
    constexpr int NDT = D / 32, NST = D / 16;
    constexpr int PPT = (PF_BLOCK * D / 8) / GT;             // 16-byte pieces per thread of a group per operand
    constexpr int GRP_BYTES = PF_BLOCK * (KROW + VROW) * 2;  // one group's K and V block
    constexpr int NVAL = NDT * 16 + 2;                        // merged values per lane: O^T, m, l
    extern __shared__ __attribute__((aligned(16))) uint8_t pp_lds[];
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lds_v4s;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w / NWQ, wq = w % NWQ, gtid = tid % GT;
    const int jq = lane & 31, g = lane >> 5;
    const int S = (int)a.S, KV = (int)a.new_len, Hq = (int)a.Hq, Hkv = (int)a.Hkv;
    // Heaviest tiles first (one workgroup fills a CU; the dispatcher hands the next one to whichever CU finishes).
    // Workgroup ids go round the 8 XCDs: when the heads divide by 8 each XCD keeps its own heads, so a kv head's
    // rows stay in one L2.
    const int nt = (S + QT - 1) / QT, hb = Hq * (int)a.B;
    const int id = (int)blockIdx.x;
    int qt, hbi;
    if (hb % 8 == 0) {
        const int hpc = hb / 8, x = id & 7, y = id >> 3;
        qt = nt - 1 - y / hpc;
        hbi = x * hpc + y % hpc;
    } else {
        qt = nt - 1 - id / hb;
        hbi = id % hb;
    }
    const int h = hbi % Hq, b = hbi / Hq;
    const int kvh = h / (Hq / Hkv);
    const int q_first = qt * QT + wq * 32;
    const int qi = q_first + jq;
    const bool q_live = qi < S;
    const long long off = (long long)a.position_offset;
    const long long qabs = off + qi;
    float scale = a.scale;
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);

    const int q_last_wg = min(S, (qt + 1) * QT) - 1, q_last_wave = min(S - 1, q_first + 31);
    const int kv_end_wg = a.is_causal ? (int)min((long long)KV, off + q_last_wg + 1) : KV;
    const int kv_end_wave = a.is_causal ? (int)min((long long)KV, off + q_last_wave + 1) : KV;
    const int nblk = (kv_end_wg + PF_BLOCK - 1) / PF_BLOCK;
    const int nb0 = (nblk + 1) / 2;                    // group 0: blocks [0, nb0), group 1: [nb0, nblk)
    const int iters = nb0;
    const int blk0 = grp ? nb0 : 0, blk_end = grp ? nblk : nb0;

    uint16_t* sK = reinterpret_cast<uint16_t*>(pp_lds + grp * GRP_BYTES);
    uint16_t* sV = sK + PF_BLOCK * KROW;
    float* mrg = reinterpret_cast<float*>(pp_lds + 2 * GRP_BYTES);

    v8hf qf[NST];
    {
        const uint16_t* qrow = a.q + (((size_t)b * S + min(qi, S - 1)) * Hq + h) * D;
#pragma unroll
        for (int st = 0; st < NST; st++) qf[st] = *reinterpret_cast<const v8hf*>(qrow + st * 16 + g * 8);
    }
    const uint16_t* Kb = a.k_new + (size_t)b * KV * Hkv * D + (size_t)kvh * D;
    const uint16_t* Vb = a.v_new + (size_t)b * KV * Hkv * D + (size_t)kvh * D;
    const size_t kv_stride = (size_t)Hkv * D;

    // staging by the group's 256 threads: piece = (key, 8 halves); a block index past the group's range reads a clamped row
    // (a whole block inside the rows: a wave-uniform base and the thread's constant 32-bit offsets, no address arithmetic)
    v4i kst[PPT], vst[PPT];
    unsigned piece_off[PPT];
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        const int idx = p * GT + gtid, key = idx / (D / 8), d8 = idx % (D / 8);
        piece_off[p] = (unsigned)(key * (int)kv_stride + d8 * 8) * 2u;
    }
    auto fetch_rows = [&](const uint16_t* base, int blk, v4i (&r)[PPT]) {
        if ((blk + 1) * PF_BLOCK <= KV) {
            const uint8_t* bp = reinterpret_cast<const uint8_t*>(base + (size_t)blk * PF_BLOCK * kv_stride);
#pragma unroll
            for (int p = 0; p < PPT; p++) r[p] = *reinterpret_cast<const v4i*>(bp + piece_off[p]);
        } else {
#pragma unroll
            for (int p = 0; p < PPT; p++) {
                const int idx = p * GT + gtid, key = idx / (D / 8), d8 = idx % (D / 8);
                const int kp = max(0, min(blk * PF_BLOCK + key, KV - 1));
                r[p] = *reinterpret_cast<const v4i*>(base + (size_t)kp * kv_stride + d8 * 8);
            }
        }
    };
    auto stash_rows = [&](uint16_t* dst, int row_halves, const v4i (&r)[PPT]) {
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int idx = p * GT + gtid, key = idx / (D / 8), d8 = idx % (D / 8);
            *reinterpret_cast<v4i*>(dst + key * row_halves + d8 * 8) = r[p];
        }
    };
    auto barrier = [&]() {   // LDS writes of this wave are complete; loads into registers stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // transposed V reads: lane (16-lane group gi, q = row of the 4-key block, p = 4-column piece) supplies the
    // address of key 4g + q, halves 16*(gi & 1) + 4p ..+3 and receives d = 16*(gi & 1) + lane % 16 of 4 keys
    const int gi = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const uint16_t* vtr = sV + (4 * g + tq) * VROW + 16 * (gi & 1) + 4 * tp;

    float m_run = -INFINITY, l_run = 0.0f;
    v16f o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[dt][r] = 0.0f;
    v8hf phi[2], plo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) { phi[ks][e] = (_Float16)0.0f; plo[ks][e] = (_Float16)0.0f; }
    v16f sacc;
    bool pv_pending = false;

    auto scores = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[r] = 0.0f;
#pragma unroll
        for (int st = 0; st < NST; st++) {
            const v8hf kf = *reinterpret_cast<const v8hf*>(sK + jq * KROW + st * 16 + g * 8);
            if (PF_TIMING & 8) { sacc[st] += (float)kf[0]; continue; }
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], sacc, 0, 0, 0);
        }
    };
    auto weighted_values = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const v4s v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(vtr + (16 * ks) * VROW + 32 * dt));
                const v4s v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(vtr + (16 * ks + 8) * VROW + 32 * dt));
                typedef short v8s __attribute__((ext_vector_type(8)));
                const v8s both = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const v8hf vf = __builtin_bit_cast(v8hf, both);
                if (PF_TIMING & 4) { o[dt][ks] += (float)vf[0]; continue; }
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, phi[ks], o[dt], 0, 0, 0);
                if (!(PF_TIMING & 16)) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, plo[ks], o[dt], 0, 0, 0);
            }
        }
    };
    // softmax of one block from sacc (kernel_attention.cpp:160-204), leaving the weights as hi + lo binary16 parts.
    // FULL: all 32 keys visible to all 32 queries of the wave.
    // The reference rescales (l, O) by exp(m_prev - bm) at every block, bm the block's own maximum; the output O / l does
    // not depend on the scale the two are held on, so here they stay on the scale of m_run, the maximum of the block at
    // which they were last rescaled, and a block's weights are multiplied by f = exp(bm - m_run) instead (16 products
    // in place of head_dim / 2); (l, O) move to a new scale only when f would pass e^8 (binary16 weights, fp32 sums).
    // The polynomial's argument is still relative to the block's own maximum, as the reference's is.
    // x = (s - bm) * scale * log2e <= 0 is formed from the difference of the raw scores (exactly 0 for the maximum);
    // with z = -x: trunc(x) = -floor(z) and x - trunc(x) = -fract(z).
    // Returns the factor (l, O) still have to be multiplied by (1: none); the caller applies it to O outside the
    // branches that select this lambda's variants, in place.
    float minus_one = -1.0f;
    asm volatile("" : "+v"(minus_one));
    auto softmax = [&](auto full_tag, int kb, int ke) __attribute__((always_inline)) -> float {
        constexpr bool FULL = decltype(full_tag)::value;
        const int vec = ((ke - kb) / 4) * 4;
        const int lim_blk = ke - kb;
        const int lim = a.is_causal ? (int)max(-1LL, min((long long)lim_blk, qabs - kb + 1)) : lim_blk;
        const int limg = lim - 4 * g;
        float bs = -INFINITY;
        if constexpr (!FULL) {
#pragma unroll
            for (int r = 0; r < 16; r++) sacc[r] = (8 * (r >> 2) + (r & 3) < limg) ? sacc[r] : -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) bs = fmaxf(bs, sacc[r]);
        bs = halves_max(bs);
        if (!FULL && bs == -INFINITY) {       // no key of the block is visible to this query
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int e = 0; e < 8; e++) { phi[ks][e] = (_Float16)0.0f; plo[ks][e] = (_Float16)0.0f; }
            return 1.0f;
        }
        const float bm = bs * scale;
        float dm = bm - m_run;                // +inf at the first block
        float corr = 1.0f;
        if (dm > 8.0f) {
            corr = __expf(m_run - bm);
            l_run *= corr;
            m_run = bm;
            dm = 0.0f;
        }
        const float f = __expf(dm);
        // (written with packed fp32 operations, 63 of 190 instructions instead of 250 single ones, the block takes as long)
        // z = (bs - s) * c1 as |bs * c1 - s * c1|, one fused operation (the absolute value is an operand modifier of the
        // instructions that read z): a rounding that leaves the maximum's own z a hair under 0 must not reach fract()
        const float c1 = scale * 1.442695f, bsc = bs * c1;
        const float fk1 = f * -0.6931472f, fk2 = f * 0.2402265f;     // the weights leave the polynomial already times f
        const int vecg = vec - 4 * g;
        float wsum0 = 0.0f, wsum1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            bool vis = true;
            float raw = sacc[r];
            if constexpr (!FULL) {
                vis = raw > -INFINITY;
                raw = vis ? raw : bs;
            }
            const float z = fabsf(fmaf(raw, -c1, bsc));
            const float zf = __builtin_amdgcn_fractf(z);
            float y = fmaf(zf, fk1, f);
            y = fmaf(zf * zf, fk2, y);
            float wv = __builtin_amdgcn_ldexpf(y, (int)(-z));
            if constexpr (!FULL) {
                if (vec != PF_BLOCK && !(8 * (r >> 2) + (r & 3) < vecg)) wv = __expf(-((bs - raw) * scale)) * f;
                wv = vis ? wv : 0.0f;
            }
            if (r & 1) wsum1 += wv; else wsum0 += wv;
            // hi + lo: the low part straight from the binary16 high part in one mixed-precision fma (minus_one is opaque
            // to the compiler, which would otherwise turn the fma into a conversion and a subtraction)
            const _Float16 hi = (_Float16)wv;
            phi[r >> 3][r & 7] = hi;
            plo[r >> 3][r & 7] = (_Float16)fmaf((float)hi, minus_one, wv);
        }
        l_run += halves_sum(wsum0 + wsum1);
        return corr;
    };

    // prologue: K of the group's first block into LDS
    fetch_rows(Kb, blk0, kst);
    stash_rows(sK, KROW, kst);
    barrier();
#ifndef PF_LAG
#define PF_LAG 1
#endif
    constexpr bool LAG = NWQ == 4 && PF_LAG;   // waves that share SIMDs take turns; with a SIMD each the groups run in step
    if (LAG && grp == 1) barrier();     // group 1 runs half a period behind
#ifdef PF_TRACE
    unsigned long long tr[5] = {0, 0, 0, 0, 0};
    const unsigned long long tr_c0 = clock64(), tr_r0 = wall_clock64();
#endif
    for (int it = 0; it < iters; it++) {
        const int blk = blk0 + it, kb = blk * PF_BLOCK;
        const bool live = blk < blk_end && kb < kv_end_wave;   // wave-uniform: this wave's queries see the block
        // ---- MFMA phase
        PF_T(t0);
        fetch_rows(Kb, blk + 1, kst);    // written to LDS at the start of this iteration's VALU phase
        fetch_rows(Vb, blk, vst);
        if (pv_pending) weighted_values();
        if (live) scores();
#ifdef PF_TRACE
        asm volatile("s_nop 0" :: "v"(sacc[0]), "v"(o[0][0]));
#endif
        PF_T(t1);
        barrier();
        PF_T(t2);
        // ---- VALU phase
        stash_rows(sK, KROW, kst);       // K of block it + 1
        stash_rows(sV, VROW, vst);       // V of block it
        PF_T(t3);
        float corr = 1.0f;
        if (live) {
            const int ke = min(KV, kb + PF_BLOCK);
            if (ke - kb == PF_BLOCK && (!a.is_causal || off + q_first - kb + 1 >= PF_BLOCK) && q_first + 31 < S)
                corr = softmax(std::true_type{}, kb, ke);
            else
                corr = softmax(std::false_type{}, kb, ke);
        }
        if (corr != 1.0f) {
#pragma unroll
            for (int dt = 0; dt < NDT; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[dt][r] *= corr;
        }
        pv_pending = live;
#ifdef PF_TRACE
        asm volatile("s_nop 0" :: "v"(phi[0][0]), "v"(plo[1][7]), "v"(o[0][0]));
#endif
        PF_T(t4);
        if (!(LAG && grp == 1 && it == iters - 1)) barrier();
#ifdef PF_TRACE
        const unsigned long long t5 = __builtin_readcyclecounter();
        tr[0] += t1 - t0; tr[1] += t2 - t1; tr[2] += t3 - t2; tr[3] += t4 - t3; tr[4] += t5 - t4;
#endif
    }
#ifdef PF_TRACE
    if (lane == 0 && blockIdx.x < 2) {
        const int slot = blockIdx.x * 8 + w;
        g_pf_trace[slot * 8 + 0] = clock64() - tr_c0;
        g_pf_trace[slot * 8 + 1] = wall_clock64() - tr_r0;
        for (int i = 0; i < 5; i++) g_pf_trace[slot * 8 + 2 + i] = tr[i];
        g_pf_trace[slot * 8 + 7] = iters;
    }
#endif
    if (grp == 0) {
        if (pv_pending) weighted_values();
        barrier();
        barrier();                       // group 1's state is in LDS
        const float m1 = mrg[((NVAL - 2) * NWQ + wq) * 64 + lane], l1 = mrg[((NVAL - 1) * NWQ + wq) * 64 + lane];
        if (m1 > -INFINITY) {            // the second half saw a key of this query: the chain of rescales went on
            const float f = __expf(m_run - m1);
            l_run = l_run * f + l1;
#pragma unroll
            for (int dt = 0; dt < NDT; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[dt][r] = o[dt][r] * f + mrg[((dt * 16 + r) * NWQ + wq) * 64 + lane];
        }
    } else {
        barrier();                       // V of the last block is visible to the group
        if (pv_pending) weighted_values();
#pragma unroll
        for (int dt = 0; dt < NDT; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) mrg[((dt * 16 + r) * NWQ + wq) * 64 + lane] = o[dt][r];
        mrg[((NVAL - 2) * NWQ + wq) * 64 + lane] = m_run;
        mrg[((NVAL - 1) * NWQ + wq) * 64 + lane] = l_run;
        barrier();
        return;
    }

    if (!q_live) return;
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

template <int D, int NWQ>
constexpr size_t attn_prefill_pp_lds() {
    return 2 * (size_t)PF_BLOCK * (D + 8 + D + 32) * 2 + (size_t)(D / 32 * 16 + 2) * (NWQ * 64) * sizeof(float);
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
    constexpr int KROW = D + 8, VROW = D + 32, NDT = D / 32, NST = D / 16;   // V rows: banks shift by 16 per row
    constexpr int D16 = D / 16;            // 16-element pieces per row: a thread stages (key, d16)
    constexpr int NP = (D16 + 7) / 8;      // staging passes: 8 pieces per key and pass
    constexpr int NG = D == 256 ? 2 : 1;   // scale groups per cached row
    __shared__ __attribute__((aligned(16))) uint16_t sK[PF_BLOCK * KROW];
    __shared__ __attribute__((aligned(16))) uint16_t sV[PF_BLOCK * VROW];   // row-major; read transposed (ds_read_b64_tr_b16)
    __shared__ __attribute__((aligned(16))) float sKs[NG][PF_BLOCK], sVs[NG][PF_BLOCK];
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lds_v4s;

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
    const int cl = a.cache_lens ? min(max(a.cache_lens[b], 0), (int)a.cache_len) : (int)a.cache_len;
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
                    *reinterpret_cast<v4i*>(sV + skey * VROW + d0 + u * 8) = vh[u];
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
            bm = halves_max(bm);
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
            bsum = halves_sum(bsum);
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
                    // lane (16-lane group, row q of a 4-key block, 4-column piece p) supplies the address of key
                    // 16 ks (+ 8) + 4 g + q, halves 32 dt + 16 (group & 1) + 4 p and receives its d of those 4 keys
                    const uint16_t* vr = sV + (16 * ks + 4 * g + ((lane >> 2) & 3)) * VROW + dt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                    const v4s v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(vr));
                    const v4s v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(vr + 8 * VROW));
                    typedef short v8s __attribute__((ext_vector_type(8)));
                    const v8s both = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    const v8hf vf = __builtin_bit_cast(v8hf, both);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, phi[dt * NG / NDT][ks], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, plo[dt * NG / NDT][ks], o[dt], 0, 0, 0);
                }
            }
        };
        if (full) block_math(std::true_type{});
        else block_math(std::false_type{});
    }

    if constexpr (D <= 128) {
        if (nsplit > 1) {
            // (max, sum, unnormalised output) of this key range: attn_merge_kernel's record of D + 2 floats per
            // query.  A lane holds a query's values - written from there every store would touch 64 records - so
            // the wave turns 64 dims of its 32 records around in its own piece of LDS and writes two records' worth
            // of 256-byte runs per store.
            constexpr int RP = 64 + 4;
            __shared__ __attribute__((aligned(16))) float sRec[4 * 32 * RP];
            float* mine = sRec + w * (32 * RP);
            const int rows = min(32, S - q_first);            // wave-uniform; <= 0: nothing of this wave's is live
#pragma unroll
            for (int half = 0; half < D / 64; half++) {
#pragma unroll
                for (int dd = 0; dd < 2; dd++)
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const int dt = half * 2 + dd;
                        *reinterpret_cast<v4f*>(mine + jq * RP + dd * 32 + 8 * r4 + 4 * g) =
                            v4f{o[dt][4 * r4], o[dt][4 * r4 + 1], o[dt][4 * r4 + 2], o[dt][4 * r4 + 3]};
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int r0 = 0; r0 < rows; r0 += 2) {
                    const int r = r0 + g;
                    if (r < rows) {
                        float* rec = ws + ((((size_t)b * S + q_first + r) * Hq + h) * nsplit + split) * (size_t)(D + 2);
                        *reinterpret_cast<float2*>(rec + 2 + half * 64 + jq * 2) =
                            *reinterpret_cast<const float2*>(mine + r * RP + jq * 2);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read back before the next half overwrites
            }
            if (q_live && g == 0) {
                float* rec = ws + ((((size_t)b * S + qi) * Hq + h) * nsplit + split) * (size_t)(D + 2);
                rec[0] = m_run;
                rec[1] = l_run;
            }
            return;
        }
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

#ifdef PF_TRACE
extern "C" int cactus_hip_debug_pf_trace(unsigned long long* out128) {
    return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_pf_trace), sizeof(unsigned long long) * 128);
}
#endif

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

static bool attn_prefill_common_ok(const AttnArgs& a) {
    return a.f16_mode && a.mask == nullptr && (a.D == 64 || a.D == 128 || a.D == 256) && a.S >= 32 &&
           a.new_len >= 1 && a.new_len < (size_t(1) << 30) && a.S < (size_t(1) << 30) && a.Hkv > 0 &&
           a.Hq % a.Hkv == 0 && a.Hq <= 65535 && a.B <= 65535 && a.Hq * a.B <= 1048576 &&
           ((reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.k_new) |
             reinterpret_cast<uintptr_t>(a.v_new) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0;
}

// causal attention under a window that cuts: attn_prefill_win_kernel
bool attn_prefill_win_ok(const AttnArgs& a) {
    return a.window_size > 0 && a.is_causal && !attn_prefill_ok(a) && attn_prefill_common_ok(a) &&
           a.position_offset < (size_t(1) << 40) && a.window_size < (size_t(1) << 40);
}

hipError_t launch_attention_prefill_win(const AttnArgs& a, hipStream_t st) {
    if (!attn_prefill_win_ok(a)) return hipErrorInvalidValue;
    const size_t total = ((a.S + PF_QTILE - 1) / PF_QTILE) * a.Hq * a.B;
    if (total > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)total), block(256);
    if (a.D == 256) attn_prefill_win_kernel<256><<<grid, block, 0, st>>>(a);
    else if (a.D == 128) attn_prefill_win_kernel<128><<<grid, block, 0, st>>>(a);
    else attn_prefill_win_kernel<64><<<grid, block, 0, st>>>(a);
    return hipGetLastError();
}

bool attn_prefill_ok(const AttnArgs& a) {
    // a window that no (query, key) pair of the call can exceed does nothing (kernel_attention.cpp:78-90, 149-151:
    // it moves kv_start only when window < KV and masks a key only when more than `window` behind its query)
    const bool no_window = a.window_size == 0 ||
                           (a.window_size >= a.new_len && a.position_offset + a.S <= a.window_size + 1);
    return no_window && attn_prefill_common_ok(a);
}

hipError_t launch_attention_prefill(const AttnArgs& a, hipStream_t st) {
    if (!attn_prefill_ok(a)) return hipErrorInvalidValue;
    const size_t total = ((a.S + PF_QTILE - 1) / PF_QTILE) * a.Hq * a.B;
    if (total > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)total), block(256);
    if (a.D == 256) {
        attn_prefill_kernel<256><<<grid, block, 0, st>>>(a);
    } else {
        // more dynamic LDS than the default limit: allowed once per device
        static bool allowed[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!allowed[dev]) {
            const void* fns[4] = {reinterpret_cast<const void*>(attn_prefill_pp_kernel<128, 4>),
                                  reinterpret_cast<const void*>(attn_prefill_pp_kernel<64, 4>),
                                  reinterpret_cast<const void*>(attn_prefill_pp_kernel<128, 2>),
                                  reinterpret_cast<const void*>(attn_prefill_pp_kernel<64, 2>)};
            const size_t lds[4] = {attn_prefill_pp_lds<128, 4>(), attn_prefill_pp_lds<64, 4>(), attn_prefill_pp_lds<128, 2>(),
                                   attn_prefill_pp_lds<64, 2>()};
            for (int i = 0; i < 4; i++) {
                e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds[i]);
                if (e != hipSuccess) return e;
            }
            allowed[dev] = true;
        }
        // 128-query workgroups when they fill the chip well; else 64-query workgroups, twice as many (S = 2048, 16
        // heads = 256 of the large ones: 78 us against 66; S = 4096 = 512: 183 against 196)
#ifndef PF_SMALL_TILE_BELOW
#define PF_SMALL_TILE_BELOW 384
#endif
        if (total >= PF_SMALL_TILE_BELOW) {
            if (a.D == 128) attn_prefill_pp_kernel<128, 4><<<grid, dim3(512), attn_prefill_pp_lds<128, 4>(), st>>>(a);
            else attn_prefill_pp_kernel<64, 4><<<grid, dim3(512), attn_prefill_pp_lds<64, 4>(), st>>>(a);
        } else {
            const size_t total64 = ((a.S + 63) / 64) * a.Hq * a.B;
            const dim3 grid64((unsigned)total64);
            if (a.D == 128) attn_prefill_pp_kernel<128, 2><<<grid64, dim3(256), attn_prefill_pp_lds<128, 2>(), st>>>(a);
            else attn_prefill_pp_kernel<64, 2><<<grid64, dim3(256), attn_prefill_pp_lds<64, 2>(), st>>>(a);
        }
    }
    return hipGetLastError();
}

}  // namespace cactus_hip
