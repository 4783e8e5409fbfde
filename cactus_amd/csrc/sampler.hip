// Stochastic token sampler and logits entropy on the device: the decode tail after lm_head.
//
// Roofline: launch latency / one pass over a logits row (2 B per element, read three to five times
// from L2); the sequential parts (softmax sums in the reference's element order) run on one lane
// over the few survivors of top-k.
//
// sample_kernel follows cactus_sample_f16 (cactus/kernel/kernel_nn.cpp:468-651) step by step, one
// 1024-thread workgroup per logits row:
//   1. scaled = half(logit * half(1/temperature))                     (:482-496; the bias written
//      at :473-480 is overwritten by both branches and never has an effect)
//   2. repetition penalty 1.1 for every occurrence of a token in the history (:502-513)
//   3. top-k: values below the k-th largest drop out (:515-535) -- the k-th largest of 16-bit
//      values comes from two 256-bin histograms of an order-preserving key (exact, no sort; the
//      first histogram is filled by the pass that applies the temperature)
//   4. the survivors are gathered in index order (at most SAMPLE_CAP of them)
//   5. top-p over the survivors in descending order (:537-588): fp32 exp / sum / cumulative sum in
//      that order; entries from the first position > 0 whose cumulative mass exceeds top_p are cut.
//      Equal logits keep ascending index order (std::sort leaves their order unspecified).
//   6. softmax over what is left in index order, inverse-CDF draw against `draws[row]`, the two
//      fallbacks, and the history update (:590-650)
// exp is evaluated in double and rounded to float (the reference calls its platform's expf).
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

constexpr int SAMPLE_CAP = 4096;        // survivors of top-k held in LDS
constexpr int SAMPLE_HISTORY = 128;     // kernel_nn.cpp:499
constexpr size_t SAMPLE_WIDE_GROUPS = 32;   // workgroups (histograms) of the kernel that finishes rows with more survivors
constexpr uint16_t H_NEG_INF = 0xFC00;

// larger value <=> larger key; -0 and +0 share a key
__device__ __forceinline__ uint32_t order_key(uint16_t b) {
    if (b == 0x8000) b = 0;
    return (b & 0x8000) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000);
}
__device__ __forceinline__ uint16_t hmul(uint16_t a, uint16_t b) { return f2h(h2f(a) * h2f(b)); }
__device__ __forceinline__ float exp_f(float x) { return (float)exp((double)x); }

__global__ __launch_bounds__(1024) void sample_kernel(
    const uint16_t* __restrict__ logits, uint16_t* __restrict__ work, uint32_t* __restrict__ out,
    int* __restrict__ status, int vocab, int use_temp, uint16_t inv_temp, float top_p, int top_k,
    const float* __restrict__ draws, uint32_t* __restrict__ history, uint32_t* __restrict__ hist_counts,
    uint32_t* __restrict__ wide_kth) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wcount[16];
    __shared__ uint16_t sv_bits[SAMPLE_CAP];
    __shared__ uint32_t sv_idx[SAMPLE_CAP];
    __shared__ uint16_t order[SAMPLE_CAP];
    __shared__ float pr[SAMPLE_CAP];
    __shared__ float wmax[16];
    __shared__ int s_hb, s_k2, s_kth, s_cut, s_token, s_push;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const size_t row = blockIdx.x;
    const uint16_t* L = logits + row * (size_t)vocab;
    uint16_t* F = work + row * (size_t)vocab;
    uint32_t* H = history ? history + row * SAMPLE_HISTORY : nullptr;
    const int hc = H ? min((int)hist_counts[row], SAMPLE_HISTORY) : 0;

    // The row is walked in chunks of 8 elements (one 16-byte access when the row allows it).
    const bool vec = (vocab & 7) == 0 && ((reinterpret_cast<uintptr_t>(L) | reinterpret_cast<uintptr_t>(F)) & 15) == 0;
    const int nchunk = (vocab + 7) / 8;
    auto load8 = [&](const uint16_t* src, int c, uint16_t (&v)[8]) {   // elements past the row read as -inf
        if (vec) {
            const v4i x = *reinterpret_cast<const v4i*>(src + (size_t)c * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (uint16_t)(((uint32_t)x[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (c * 8 + e < vocab) ? src[(size_t)c * 8 + e] : H_NEG_INF;
        }
    };
    const bool use_topk = top_k > 0 && top_k < vocab;

    // 1. temperature, and the histogram of the high key byte for the top-k selection
    if (t < 256) hist[t] = 0;
    __syncthreads();
    for (int c = t; c < nchunk; c += 1024) {
        uint16_t v[8];
        load8(L, c, v);
        if (use_temp) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = hmul(v[e], inv_temp);
        }
        if (vec) {
            v4i x;
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] = (int)((uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16));
            *reinterpret_cast<v4i*>(F + (size_t)c * 8) = x;
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (c * 8 + e < vocab) F[(size_t)c * 8 + e] = v[e];
        }
        if (use_topk) {
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (c * 8 + e < vocab) atomicAdd(&hist[order_key(v[e]) >> 8], 1u);
        }
    }
    __syncthreads();

    // 2. repetition penalty: the first occurrence of a token applies it once per occurrence (and
    // moves the token between histogram bins)
    if (t < hc) {
        const uint32_t tok = H[t];
        bool first = true;
        int cnt = 0;
        for (int j = 0; j < hc; j++) {
            const bool same = H[j] == tok;
            if (same && j < t) first = false;
            cnt += same;
        }
        if (first && tok < (uint32_t)vocab) {
            const uint16_t pinv = f2h(1.0f / 1.1f), pen = f2h(1.1f);
            const uint16_t v0 = F[tok];
            uint16_t v = v0;
            for (int c = 0; c < cnt; c++) v = (h2f(v) > 0.0f) ? hmul(v, pinv) : hmul(v, pen);
            F[tok] = v;
            if (use_topk) {
                atomicSub(&hist[order_key(v0) >> 8], 1u);
                atomicAdd(&hist[order_key(v) >> 8], 1u);
            }
        }
    }
    __syncthreads();

    // 3. top-k threshold: the key of the k-th largest value (0: everything passes)
    uint32_t kth = 0;
    if (use_topk) {
        if (t == 0) {
            uint32_t cum = 0;
            int b = 255;
            for (; b > 0; b--) {
                if (cum + hist[b] >= (uint32_t)top_k) break;
                cum += hist[b];
            }
            s_hb = b;
            s_k2 = top_k - (int)cum;     // rank wanted inside bin b (>= 1)
        }
        __syncthreads();
        const int hb = s_hb;
        if (t < 256) hist[t] = 0;
        __syncthreads();
        for (int c = t; c < nchunk; c += 1024) {
            uint16_t v[8];
            load8(F, c, v);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t k = order_key(v[e]);
                if (c * 8 + e < vocab && (int)(k >> 8) == hb) atomicAdd(&hist[k & 255], 1u);
            }
        }
        __syncthreads();
        if (t == 0) {
            uint32_t cum = 0;
            int b = 255;
            for (; b > 0; b--) {
                if (cum + hist[b] >= (uint32_t)s_k2) break;
                cum += hist[b];
            }
            s_kth = (hb << 8) | b;
        }
        __syncthreads();
        kth = (uint32_t)s_kth;
    }

    // 4. survivors (not below the k-th value, not -inf) in index order: wave w owns one contiguous
    // range of chunks, a lane 8 consecutive elements of each 512-element step
    auto alive_mask = [&](int c) -> uint32_t {
        if (c >= nchunk) return 0u;
        uint16_t v[8];
        load8(F, c, v);
        uint32_t m = 0;
#pragma unroll
        for (int e = 0; e < 8; e++)
            m |= (uint32_t)(c * 8 + e < vocab && v[e] != H_NEG_INF && order_key(v[e]) >= kth) << e;
        return m;
    };
    const int cper = (((nchunk + 15) / 16) + 63) & ~63;
    const int clo = w * cper, chi = min(nchunk, clo + cper);
    {
        uint32_t c = 0;
        for (int c0 = clo; c0 < chi; c0 += 64) c += (uint32_t)__popc(alive_mask(c0 + lane < chi ? c0 + lane : nchunk));
        c = (uint32_t)wave_sum((float)c);      // < 2^24: exact
        if (lane == 0) wcount[w] = c;
    }
    __syncthreads();
    uint32_t base = 0, n_all = 0;
    for (int j = 0; j < 16; j++) {
        if (j < w) base += wcount[j];
        n_all += wcount[j];
    }
    if (n_all > (uint32_t)SAMPLE_CAP) {      // uniform over the workgroup: the row goes to sample_wide_kernel
        if (t == 0) {
            out[row] = 0xFFFFFFFFu;
            if (wide_kth) wide_kth[row] = kth;
            else if (status) atomicExch(status, 1);
        }
        return;
    }
    const int n = (int)n_all;
    for (int c0 = clo; c0 < chi; c0 += 64) {
        const int c = c0 + lane < chi ? c0 + lane : nchunk;
        const uint32_t m = alive_mask(c);
        const uint32_t cnt = (uint32_t)__popc(m);
        // exclusive prefix of cnt over the lanes of the wave
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        if (m) {
            uint16_t v[8];
            load8(F, c, v);
            uint32_t pos = base + incl - cnt;
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (m & (1u << e)) {
                    sv_bits[pos] = v[e];
                    sv_idx[pos] = (uint32_t)(c * 8 + e);
                    pos++;
                }
        }
        base += total;
    }
    __syncthreads();

    // 5. top-p
    if (top_p > 0.0f && top_p < 1.0f && n > 0) {
        for (int s = t; s < n; s += 1024) {
            const uint32_t ks = order_key(sv_bits[s]);
            int rank = 0;
            for (int j = 0; j < n; j++) {
                const uint32_t kj = order_key(sv_bits[j]);
                rank += (kj > ks) || (kj == ks && j < s);
            }
            order[rank] = (uint16_t)s;
        }
        __syncthreads();
        const float mx = h2f(sv_bits[order[0]]);
        for (int r = t; r < n; r += 1024) pr[r] = exp_f(h2f(sv_bits[order[r]]) - mx);
        __syncthreads();
        if (t == 0) {
            float sum = 0.0f;
            for (int r = 0; r < n; r++) sum += pr[r];
            float cum = 0.0f;
            int cut = n;
            for (int r = 0; r < n; r++) {
                cum += pr[r] / sum;
                if (cum > top_p && r > 0) { cut = r; break; }
            }
            s_cut = cut;
        }
        __syncthreads();
        const int cut = s_cut;
        for (int r = cut + t; r < n; r += 1024) sv_bits[order[r]] = H_NEG_INF;
        __syncthreads();
    }

    // 6. softmax in index order and the draw
    float m = -INFINITY;
    for (int s = t; s < n; s += 1024)
        if (sv_bits[s] != H_NEG_INF) m = fmaxf(m, h2f(sv_bits[s]));
    m = wave_max(m);
    if (lane == 0) wmax[w] = m;
    __syncthreads();
    m = wmax[0];
    for (int j = 1; j < 16; j++) m = fmaxf(m, wmax[j]);
    for (int s = t; s < n; s += 1024)
        pr[s] = (sv_bits[s] == H_NEG_INF) ? 0.0f : exp_f(h2f(sv_bits[s]) - m);
    __syncthreads();
    if (t == 0) {
        int token = 0, push = 0;
        if (n > 0 && m != -INFINITY) {
            float sum = 0.0f;
            for (int s = 0; s < n; s++) sum += pr[s];
            if (sum != 0.0f) {
                const float sample = draws[row];
                push = 1;
                int found = -1;
                if (sample <= 0.0f) {
                    found = 0;               // the running sum is >= 0 at element 0 already
                } else {
                    float cum = 0.0f;
                    for (int s = 0; s < n; s++) {
                        cum += pr[s] / sum;
                        if (cum >= sample) { found = (int)sv_idx[s]; break; }
                    }
                    if (found < 0) {
                        for (int s = n - 1; s >= 0; s--)
                            if (pr[s] / sum > 0.0f) { found = (int)sv_idx[s]; break; }
                    }
                    if (found < 0) found = 0;
                }
                token = found;
            }
        }
        out[row] = (uint32_t)token;
        s_token = token;
        s_push = push;
    }
    __syncthreads();
    if (H && s_push) {     // kernel_nn.cpp:627-630: append, dropping the oldest beyond 128
        const uint32_t tok = (uint32_t)s_token;
        if (hc < SAMPLE_HISTORY) {
            if (t == 0) { H[hc] = tok; hist_counts[row] = (uint32_t)(hc + 1); }
        } else {
            uint32_t nxt = 0;
            if (t < SAMPLE_HISTORY - 1) nxt = H[t + 1];
            __syncthreads();
            if (t < SAMPLE_HISTORY - 1) H[t] = nxt;
            if (t == SAMPLE_HISTORY - 1) H[t] = tok;
        }
    }
}

// Rows with more survivors than LDS holds (top_k = 0 on a whole vocabulary, or a huge top_k): the same steps
// 5 and 6 over the row in global memory.  `work` holds the row after temperature and penalty, wide_kth[row]
// the top-k threshold key; rows whose token is not the marker 0xFFFFFFFF are skipped.  Workgroup g takes rows
// g, g + gridDim.x, ... and owns one 65536-bin histogram (bins[g]) and one list of (value, count) pairs.
//   top-p in descending order without a sort: there are only 65536 different binary16 values, equal values
//   have equal probabilities, and equal logits rank by ascending index, so the descending walk is a walk over
//   the non-empty bins from the top, each adding its probability `count` times (fp32, sequentially, as the
//   reference does); the cut falls after `cut_within` elements (in index order) of the bin `cut_key`.
//   The index-order sums of step 6 run over the row in chunks: all threads fill a chunk's probabilities in
//   LDS, one lane adds them in order.
constexpr int WIDE_CHUNK = 4096;
__device__ __forceinline__ uint16_t key_value(uint32_t k) {      // inverse of order_key
    return (k & 0x8000u) ? (uint16_t)(k & 0x7FFFu) : (uint16_t)(~k & 0xFFFFu);
}
__global__ __launch_bounds__(1024) void sample_wide_kernel(
    const uint16_t* __restrict__ work, uint32_t* __restrict__ out, int rows, int vocab, float top_p,
    const float* __restrict__ draws, uint32_t* __restrict__ history, uint32_t* __restrict__ hist_counts,
    const uint32_t* __restrict__ wide_kth, uint32_t* __restrict__ bins_all, uint2* __restrict__ list_all) {
    __shared__ float pr[WIDE_CHUNK];
    __shared__ uint32_t cnt[WIDE_CHUNK];       // list chunks: counts; row chunks: 0 dead, 1 alive, 2 the cut bin
    __shared__ uint32_t key[WIDE_CHUNK];       // list chunks: the bins' keys
    __shared__ uint32_t wsum[16];
    __shared__ int s_cutkey, s_cutwithin, s_token, s_push, s_done, s_rank, s_seen, s_last;
    __shared__ float s_sum, s_cum;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t* bins = bins_all + (size_t)blockIdx.x * 65536;
    uint2* list = list_all + (size_t)blockIdx.x * 65536;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        if (out[row] != 0xFFFFFFFFu) continue;           // uniform: every thread reads the same word
        __syncthreads();
        const uint16_t* F = work + (size_t)row * vocab;
        uint32_t* H = history ? history + (size_t)row * SAMPLE_HISTORY : nullptr;
        const int hc = H ? min((int)hist_counts[row], SAMPLE_HISTORY) : 0;
        const uint32_t kth = wide_kth[row];
        auto alive = [&](uint16_t v) { return v != H_NEG_INF && order_key(v) >= kth; };
        // histogram of the survivors' values
        for (int i = t; i < 65536; i += 1024) bins[i] = 0;
        __threadfence();
        __syncthreads();
        for (int i = t; i < vocab; i += 1024) {
            const uint16_t v = F[i];
            if (alive(v)) atomicAdd(&bins[order_key(v)], 1u);
        }
        __threadfence();
        __syncthreads();
        // non-empty bins in ascending key order -> list (thread t owns keys 64 t .. 64 t + 63)
        uint32_t mine = 0, tot = 0;
        for (int j = 0; j < 64; j++) {
            const uint32_t c = __hip_atomic_load(&bins[t * 64 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mine += c != 0;
            tot += c;
        }
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[w] = incl;
        const uint32_t ntot = (uint32_t)wave_sum((float)tot);       // < 2^24: exact
        __syncthreads();
        uint32_t base = 0, nlist = 0;
        for (int j = 0; j < 16; j++) {
            if (j < w) base += wsum[j];
            nlist += wsum[j];
        }
        __syncthreads();
        if (lane == 0) wsum[w] = ntot;
        {
            uint32_t pos = base + incl - mine;
            for (int j = 0; j < 64; j++) {
                const uint32_t c = __hip_atomic_load(&bins[t * 64 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c) list[pos++] = make_uint2((uint32_t)(t * 64 + j), c);
            }
        }
        __threadfence();
        __syncthreads();
        int n = 0;
        for (int j = 0; j < 16; j++) n += (int)wsum[j];
        const uint32_t maxkey = nlist ? __hip_atomic_load(&list[nlist - 1].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const float mx = h2f(key_value(maxkey));
        __syncthreads();

        // 5. top-p over the bins from the top (kernel_nn.cpp:537-588)
        if (t == 0) { s_cutkey = -1; s_cutwithin = 0; s_sum = 0.0f; s_cum = 0.0f; s_rank = 0; s_done = 0; }
        __syncthreads();
        if (top_p > 0.0f && top_p < 1.0f && n > 0) {
            for (int pass = 0; pass < 2; pass++) {          // pass 0: the sum; pass 1: the cumulative walk
                for (int hi = (int)nlist; hi > 0; hi -= WIDE_CHUNK) {
                    const int lo = max(0, hi - WIDE_CHUNK), len = hi - lo;
                    for (int j = t; j < len; j += 1024) {           // slot j: list entry hi - 1 - j (descending)
                        const uint2* e = &list[hi - 1 - j];
                        const uint32_t k = __hip_atomic_load(&e->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        cnt[j] = __hip_atomic_load(&e->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        key[j] = k;
                        pr[j] = exp_f(h2f(key_value(k)) - mx);
                    }
                    __syncthreads();
                    if (t == 0) {
                        if (pass == 0) {
                            float sum = s_sum;
                            for (int j = 0; j < len; j++) {
                                const float pj = pr[j];
                                for (uint32_t i = 0; i < cnt[j]; i++) sum += pj;
                            }
                            s_sum = sum;
                        } else {
                            const float sum = s_sum;
                            float cum = s_cum;
                            int r = s_rank;
                            bool done = false;
                            for (int j = 0; j < len && !done; j++) {
                                const float qj = pr[j] / sum;
                                for (uint32_t i = 0; i < cnt[j]; i++) {
                                    cum += qj;
                                    if (cum > top_p && r > 0) {      // ranks r.. are cut: i elements of this bin stay
                                        s_cutkey = (int)key[j];
                                        s_cutwithin = (int)i;
                                        done = true;
                                        break;
                                    }
                                    r++;
                                }
                            }
                            s_cum = cum;
                            s_rank = r;
                            s_done = done;
                        }
                    }
                    __syncthreads();
                    if (s_done) break;
                }
            }
        }
        __syncthreads();
        // what is left: bins above cutkey, and the first cutwithin elements (index order) of bin cutkey
        const int cutkey = s_done ? s_cutkey : -1, cutwithin = s_done ? s_cutwithin : 0;
        __syncthreads();

        // 6. softmax in index order and the draw (:590-650); the maximum of what is left is the top bin's value
        auto code_of = [&](uint16_t v) -> uint32_t {
            if (!alive(v)) return 0u;
            const int k = (int)order_key(v);
            if (cutkey < 0) return 1u;
            return k > cutkey ? 1u : (k == cutkey ? 2u : 0u);
        };
        if (t == 0) { s_sum = 0.0f; s_cum = 0.0f; s_seen = 0; s_last = -1; s_done = 0; s_token = 0; s_push = 0; }
        __syncthreads();
        for (int pass = 0; pass < 2 && n > 0; pass++) {     // pass 0: the sum; pass 1: the inverse-CDF walk
            if (pass == 1) {
                if (s_sum == 0.0f) break;                   // :607-611: token 0, nothing recorded
                if (t == 0) {
                    s_push = 1;
                    s_seen = 0;
                    if (draws[row] <= 0.0f) s_done = 1;     // the running sum is >= 0 at element 0 already: token 0
                    else s_token = -1;
                }
                __syncthreads();
                if (s_done) break;
            }
            for (int c0 = 0; c0 < vocab; c0 += WIDE_CHUNK) {
                const int len = min(WIDE_CHUNK, vocab - c0);
                for (int j = t; j < len; j += 1024) {
                    const uint16_t v = F[c0 + j];
                    const uint32_t cd = code_of(v);
                    cnt[j] = cd;
                    pr[j] = cd ? exp_f(h2f(v) - mx) : 0.0f;
                }
                __syncthreads();
                if (t == 0) {
                    int seen = s_seen;
                    if (pass == 0) {
                        float sum = s_sum;
                        for (int j = 0; j < len; j++) {
                            float pj = pr[j];
                            if (cnt[j] == 2u && seen++ >= cutwithin) pj = 0.0f;
                            sum += pj;
                        }
                        s_sum = sum;
                    } else {
                        const float sum = s_sum, sample = draws[row];
                        float cum = s_cum;
                        int last = s_last;
                        for (int j = 0; j < len; j++) {
                            float pj = pr[j];
                            if (cnt[j] == 2u && seen++ >= cutwithin) pj = 0.0f;
                            const float qj = pj / sum;
                            cum += qj;
                            if (qj > 0.0f) last = c0 + j;
                            if (cum >= sample) { s_token = c0 + j; s_done = 1; break; }
                        }
                        s_cum = cum;
                        s_last = last;
                    }
                    s_seen = seen;
                }
                __syncthreads();
                if (s_done) break;
            }
            __syncthreads();
        }
        __syncthreads();
        if (t == 0) {
            int token = s_token;
            if (!s_push) token = 0;
            else if (token < 0) token = s_last >= 0 ? s_last : 0;     // :640-648: the last positive share, else 0
            out[row] = (uint32_t)token;
            s_token = token;
        }
        __syncthreads();
        if (H && s_push) {
            const uint32_t tok = (uint32_t)s_token;
            if (hc < SAMPLE_HISTORY) {
                if (t == 0) { H[hc] = tok; hist_counts[row] = (uint32_t)(hc + 1); }
            } else {
                uint32_t nxt = 0;
                if (t < SAMPLE_HISTORY - 1) nxt = H[t + 1];
                __syncthreads();
                if (t < SAMPLE_HISTORY - 1) H[t] = nxt;
                if (t == SAMPLE_HISTORY - 1) H[t] = tok;
            }
        }
        __syncthreads();
    }
}

// engine_model.cpp:255-290: entropy of softmax(logits) in double, divided by log(vocab).  The
// reference adds in element order; here partial sums are combined as a tree (the float result
// differs by at most an ulp).
__global__ __launch_bounds__(1024) void entropy_kernel(const uint16_t* __restrict__ logits, float* __restrict__ out,
                                                       int vocab) {
    __shared__ double red[16];
    __shared__ float wmax[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint16_t* L = logits + (size_t)blockIdx.x * vocab;
    float m = -INFINITY;
    for (int i = t; i < vocab; i += 1024) m = fmaxf(m, h2f(L[i]));
    m = wave_max(m);
    if (lane == 0) wmax[w] = m;
    __syncthreads();
    m = wmax[0];
    for (int j = 1; j < 16; j++) m = fmaxf(m, wmax[j]);
    auto block_sum = [&](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if (lane == 0) red[w] = v;
        __syncthreads();
        double s = 0.0;
        for (int j = 0; j < 16; j++) s += red[j];
        return s;
    };
    double se = 0.0;
    for (int i = t; i < vocab; i += 1024) se += exp((double)(h2f(L[i]) - m));
    se = block_sum(se);
    const double lse = (double)m + log(se);
    double ent = 0.0;
    for (int i = t; i < vocab; i += 1024) {
        const double lp = (double)h2f(L[i]) - lse;
        const double p = exp(lp);
        if (p > 1e-10) ent -= p * lp;
    }
    ent = block_sum(ent);
    if (t == 0) out[blockIdx.x] = (float)(ent / log((double)vocab));
}

size_t sample_wide_scratch_bytes(size_t rows) {
    const size_t groups = rows < SAMPLE_WIDE_GROUPS ? rows : SAMPLE_WIDE_GROUPS;
    return ((rows * sizeof(uint32_t) + 255) & ~size_t(255)) + groups * 65536 * (sizeof(uint32_t) + sizeof(uint2));
}

hipError_t launch_sample(const uint16_t* logits, uint16_t* work, uint32_t* out, int* status, size_t rows,
                         size_t vocab, float temperature, float top_p, size_t top_k, const float* draws,
                         uint32_t* history, uint32_t* hist_counts, void* wide_scratch, hipStream_t st) {
    if (rows == 0) return hipSuccess;
    if (vocab == 0 || vocab > 0x7FFFFFFFull || rows > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const int use_temp = temperature > 0.0f;
    // half(1 / temperature) as the reference's static_cast<__fp16> gives it (round to nearest even)
    const _Float16 it = use_temp ? (_Float16)(1.0f / temperature) : (_Float16)0.0f;
    uint16_t itb;
    __builtin_memcpy(&itb, &it, 2);
    const int k = top_k > vocab ? (int)vocab : (int)top_k;
    // wide_scratch (sample_wide_scratch_bytes(rows), or null): rows with more survivors than the first kernel's
    // LDS holds are finished by the second one; without it they are reported through the token and *status
    uint32_t* wide_kth = static_cast<uint32_t*>(wide_scratch);
    sample_kernel<<<dim3((unsigned)rows), dim3(1024), 0, st>>>(logits, work, out, status, (int)vocab, use_temp, itb,
                                                               top_p, k, draws, history, hist_counts, wide_kth);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || wide_scratch == nullptr) return e;
    const size_t groups = rows < SAMPLE_WIDE_GROUPS ? rows : SAMPLE_WIDE_GROUPS;
    char* base = static_cast<char*>(wide_scratch) + ((rows * sizeof(uint32_t) + 255) & ~size_t(255));
    uint32_t* bins = reinterpret_cast<uint32_t*>(base);
    uint2* list = reinterpret_cast<uint2*>(base + groups * 65536 * sizeof(uint32_t));
    sample_wide_kernel<<<dim3((unsigned)groups), dim3(1024), 0, st>>>(work, out, (int)rows, (int)vocab, top_p, draws,
                                                                      history, hist_counts, wide_kth, bins, list);
    return hipGetLastError();
}

hipError_t launch_entropy(const uint16_t* logits, float* out, size_t rows, size_t vocab, hipStream_t st) {
    if (rows == 0) return hipSuccess;
    if (vocab == 0 || vocab > 0x7FFFFFFFull || rows > 0x7FFFFFFFull) return hipErrorInvalidValue;
    entropy_kernel<<<dim3((unsigned)rows), dim3(1024), 0, st>>>(logits, out, (int)vocab);
    return hipGetLastError();
}

}  // namespace cactus_hip
