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
//   3. top-k: values below the k-th largest drop out (:515-535).  The k-th largest of 16-bit values is a
//      selection by two 256-bin histograms of an order-preserving key (exact, no sort) - but not over the row:
//      a whole row's keys crowd into a few bins and the LDS atomics serialize.  Every thread keeps the largest
//      key it saw in the temperature pass; the (k + history length)-th largest of those 1024 maxima is a lower
//      bound of the k-th largest value (at most `history length` maxima are stale, lowered by the penalty), the
//      few elements not below it are gathered, and the exact selection runs over them.  Rows where that does
//      not work (k + history > 1024, a bound at or below -inf, more candidates than LDS holds) take the
//      selection over the whole row.
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

// Benchmarks only (-DSAMPLE_TRACE): shader-clock stamps of row 0's workgroup at the numbered steps of sample_kernel
#ifdef SAMPLE_TRACE
__device__ unsigned long long g_sample_trace[16];
#define SMP_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sample_trace[i] = clock64(); } while (0)
#else
#define SMP_T(i) do { } while (0)
#endif

// larger value <=> larger key; -0 and +0 share a key
__device__ __forceinline__ uint32_t order_key(uint16_t b) {
    if (b == 0x8000) b = 0;
    return (b & 0x8000) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000);
}
__device__ __forceinline__ uint16_t hmul(uint16_t a, uint16_t b) { return f2h(h2f(a) * h2f(b)); }
// two binary16 products / two signed 16-bit maxima per register (v_pk_mul_f16, v_pk_max_i16)
__device__ __forceinline__ int pk_mul_f16(int a, int b) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const h2v r = __builtin_bit_cast(h2v, a) * __builtin_bit_cast(h2v, b);
    return __builtin_bit_cast(int, r);
}
__device__ __forceinline__ int pk_max_i16(int a, int b) {
    typedef short s2v __attribute__((ext_vector_type(2)));
    const s2v r = __builtin_elementwise_max(__builtin_bit_cast(s2v, a), __builtin_bit_cast(s2v, b));
    return __builtin_bit_cast(int, r);
}
__device__ __forceinline__ int pk_top_i16(int a) { return max((int)(short)(a & 0xFFFF), a >> 16); }   // the larger half
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
    __shared__ int s_hb, s_k2, s_kth, s_cut, s_token, s_push, s_ncand;
    __shared__ uint32_t s_hist[SAMPLE_HISTORY];

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
    auto unpack8 = [](const v4i& x, uint16_t (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (uint16_t)(((uint32_t)x[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
    };
    // every chunk of an aligned row as its 16 bytes, thread t the chunks t, t + 1024, ...; four loads in flight per
    // thread, and the next four are requested before the current four are handed out, so that what body() stores
    // is issued behind those loads and does not stand between them and their wait
    auto for_raw_chunks = [&](const uint16_t* src, auto&& body) {
        auto fetch4 = [&](int c0, v4i (&x)[4]) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                x[u] = *reinterpret_cast<const v4i*>(src + (size_t)min(c0 + u * 1024, nchunk - 1) * 8);
        };
        auto hand4 = [&](int c0, const v4i (&x)[4]) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (c0 + u * 1024 < nchunk) body(c0 + u * 1024, x[u]);
        };
        v4i xa[4], xb[4];
        if (t < nchunk) fetch4(t, xa);
        for (int c0 = t; c0 < nchunk; c0 += 8192) {
            if (c0 + 4096 < nchunk) fetch4(c0 + 4096, xb);
            hand4(c0, xa);
            if (c0 + 4096 < nchunk) {
                if (c0 + 8192 < nchunk) fetch4(c0 + 8192, xa);
                hand4(c0 + 4096, xb);
            }
        }
    };
    // ... element by element, any row
    auto for_chunks = [&](const uint16_t* src, auto&& body) {
        if (vec) {
            for_raw_chunks(src, [&](int c, const v4i& x) {
                uint16_t v[8];
                unpack8(x, v);
                body(c, v);
            });
        } else {
            for (int c = t; c < nchunk; c += 1024) {
                uint16_t v[8];
                load8(src, c, v);
                body(c, v);
            }
        }
    };
    SMP_T(0);
    // 1. temperature; every thread keeps a witness: the key of one of its elements, the largest where that is cheap.
    // Aligned rows stay packed: the product of two binary16 values rounded once is what hmul() computes
    // (v_pk_mul_f16), and the largest bit pattern as a signed integer (v_pk_max_i16) is the largest value whenever a
    // thread has a positive one - and some element of the thread in any case, which is all the bound needs.
    uint32_t tmax = 0;
    if (vec) {
        const int it2 = (int)((uint32_t)inv_temp | ((uint32_t)inv_temp << 16));
        int wit = (int)0x80008000u;
        for_raw_chunks(L, [&](int c, const v4i& xin) {
            const int x0 = use_temp ? pk_mul_f16(xin[0], it2) : xin[0], x1 = use_temp ? pk_mul_f16(xin[1], it2) : xin[1];
            const int x2 = use_temp ? pk_mul_f16(xin[2], it2) : xin[2], x3 = use_temp ? pk_mul_f16(xin[3], it2) : xin[3];
            *reinterpret_cast<v4i*>(F + (size_t)c * 8) = v4i{x0, x1, x2, x3};
            if (use_topk) wit = pk_max_i16(pk_max_i16(wit, pk_max_i16(x0, x1)), pk_max_i16(x2, x3));
        });
        if (use_topk && t < nchunk) tmax = order_key((uint16_t)pk_top_i16(wit));
    } else {
        for_chunks(L, [&](int c, uint16_t (&v)[8]) {
            if (use_temp) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = hmul(v[e], inv_temp);
            }
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (c * 8 + e < vocab) F[(size_t)c * 8 + e] = v[e];
            if (use_topk) {
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (c * 8 + e < vocab) tmax = max(tmax, order_key(v[e]));
            }
        });
    }
    __syncthreads();

    SMP_T(1);
    // 2. repetition penalty: the first occurrence of a token applies it once per occurrence (the history goes
    // through LDS: 128 dependent reads of global memory per thread would cost more than the passes over the row)
    if (t < hc) s_hist[t] = H[t];
    __syncthreads();
    if (t < hc) {
        const uint32_t tok = s_hist[t];
        bool first = true;
        int cnt = 0;
#pragma unroll 8
        for (int j = 0; j < hc; j++) {
            const bool same = s_hist[j] == tok;
            if (same && j < t) first = false;
            cnt += same;
        }
        if (first && tok < (uint32_t)vocab) {
            const uint16_t pinv = f2h(1.0f / 1.1f), pen = f2h(1.1f);
            uint16_t v = F[tok];
            for (int c = 0; c < cnt; c++) v = (h2f(v) > 0.0f) ? hmul(v, pinv) : hmul(v, pen);
            F[tok] = v;
        }
    }
    __syncthreads();

    // Walking the histogram from bin 255 down, the bin at which the count reaches `rank` (bin 0 if it never does) and
    // the count above that bin: one wave, 4 bins per lane, suffix sums over the lanes (a lane walking 256 bins alone
    // pays an LDS round trip per bin)
    auto top_bin = [&](uint32_t rank, int& above) -> int {
        uint32_t h4[4];
#pragma unroll
        for (int e = 0; e < 4; e++) h4[e] = hist[lane * 4 + e];
        const uint32_t own = h4[0] + h4[1] + h4[2] + h4[3];
        uint32_t incl = own;                       // own + every higher lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t dn = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += dn;
        }
        uint32_t sfx = incl - own;                 // count in the bins above this lane's
        int best = -1;
        uint32_t best_above = 0;
#pragma unroll
        for (int e = 3; e >= 0; e--) {
            if (best < 0 && sfx + h4[e] >= rank) { best = lane * 4 + e; best_above = sfx; }
            sfx += h4[e];
        }
        // the highest such bin over the wave
        int top = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o, 64));
        if (top < 1) {                             // never reached (or at bin 0): bin 0, everything above it
            above = (int)(__shfl(incl, 0, 64) - __shfl(h4[0], 0, 64));
            return 0;
        }
        above = (int)__shfl(best_above, top >> 2, 64);
        return top;
    };

    // the key of the rank-th largest of the keys `visit` hands out (every thread calls visit(f) and f(key) for
    // each key it owns): 256-bin histogram of the high byte, then of the low byte inside the selected bin
    auto kth_select = [&](auto&& visit, int rank) -> uint32_t {
        if (t < 256) hist[t] = 0;
        __syncthreads();
        visit([&](uint32_t k) { atomicAdd(&hist[k >> 8], 1u); });
        __syncthreads();
        if (w == 0) {
            int cum;
            const int b = top_bin((uint32_t)rank, cum);
            if (lane == 0) { s_hb = b; s_k2 = rank - cum; }   // rank wanted inside bin b (>= 1)
        }
        __syncthreads();
        const int hb = s_hb;
        if (t < 256) hist[t] = 0;
        __syncthreads();
        visit([&](uint32_t k) { if ((int)(k >> 8) == hb) atomicAdd(&hist[k & 255], 1u); });
        __syncthreads();
        if (w == 0) {
            int cum;
            const int b = top_bin((uint32_t)s_k2, cum);
            if (lane == 0) s_kth = (hb << 8) | b;
        }
        __syncthreads();
        return (uint32_t)s_kth;
    };
    auto visit_row = [&](auto&& f) {
        for_chunks(F, [&](int c, uint16_t (&v)[8]) {
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (c * 8 + e < vocab) f(order_key(v[e]));
        });
    };

    SMP_T(2);
    // 3. top-k threshold: the key of the k-th largest value (0: everything passes), or - `exact` false - a lower
    // bound of it that few elements reach
    uint32_t kth = 0;
    bool exact = true;
    if (use_topk) {
        const int m = top_k + hc;
        if (m <= 1024) {
            kth = kth_select([&](auto&& f) { f(tmax); }, m);
            exact = false;
        }
        if (exact || kth <= order_key(H_NEG_INF)) {   // -inf elements count in the ranks but are no candidates
            kth = kth_select(visit_row, top_k);
            exact = true;
        }
    }
    SMP_T(3);
    // 3b. from a bound: the candidates in one pass and any order (LDS counter), the exact selection among them, and
    // those that stay placed by the number of others before them in the row
    bool gathered = false;
    int n_fast = 0;
    if (!exact) {
        uint32_t* cidx = reinterpret_cast<uint32_t*>(pr);      // candidates: bits in order[], index in pr[] (both free until top-p)
        if (t == 0) s_ncand = 0;
        __syncthreads();
        auto take = [&](int c, uint16_t (&v)[8]) {
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (c * 8 + e < vocab && v[e] != H_NEG_INF && order_key(v[e]) >= kth) {
                    const int p = atomicAdd(&s_ncand, 1);
                    if (p < 1024) { order[p] = v[e]; cidx[p] = (uint32_t)(c * 8 + e); }
                }
        };
        if (vec && kth > 0x8000u) {
            // a bound above +0: a value reaches it exactly when its bit pattern, as a signed integer, reaches the
            // bound's - the largest pattern of a chunk (three v_pk_max_i16) rules almost every chunk out
            const int tb = (int)(kth & 0x7FFFu);
            for_raw_chunks(F, [&](int c, const v4i& x) {
                if (pk_top_i16(pk_max_i16(pk_max_i16(x[0], x[1]), pk_max_i16(x[2], x[3]))) >= tb) {
                    uint16_t v[8];
                    unpack8(x, v);
                    take(c, v);
                }
            });
        } else {
            for_chunks(F, take);
        }
        __syncthreads();
        SMP_T(9);
        const int n_cand = s_ncand;
        if (n_cand > 1024) {                 // many equal values at the bound: the selection over the row, then step 4
            kth = kth_select(visit_row, top_k);
            exact = true;
        } else {
            kth = kth_select([&](auto&& f) { if (t < n_cand) f(order_key(order[t])); }, top_k);
            SMP_T(10);
            // those that stay move to a second list (any order), then each finds its place by counting the others
            // before it in the row
            const bool keep = t < n_cand && order_key(order[t]) >= kth;
            if (t == 0) s_ncand = 0;
            __syncthreads();
            if (keep) {
                const int p = atomicAdd(&s_ncand, 1);
                order[1024 + p] = order[t];
                cidx[1024 + p] = cidx[t];
            }
            __syncthreads();
            const int n_keep = s_ncand;
            if (t < n_keep) {
                const uint16_t my_bits = order[1024 + t];
                const uint32_t my_idx = cidx[1024 + t];
                int rank = 0;
#pragma unroll 8
                for (int j = 0; j < n_keep; j++) rank += cidx[1024 + j] < my_idx;
                sv_bits[rank] = my_bits;
                sv_idx[rank] = my_idx;
            }
            __syncthreads();
            n_fast = s_ncand;
            gathered = true;
        }
    }

    SMP_T(4);
    int n_kept = n_fast;
    if (!gathered) {
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
        uint32_t base = 0, n_all = 0;
        auto count_alive = [&]() {
            uint32_t c = 0;
#pragma unroll 4
            for (int c0 = clo; c0 < chi; c0 += 64) c += (uint32_t)__popc(alive_mask(c0 + lane < chi ? c0 + lane : nchunk));
            c = (uint32_t)wave_sum((float)c);      // < 2^24: exact
            __syncthreads();                       // the previous count has been read
            if (lane == 0) wcount[w] = c;
            __syncthreads();
            base = 0; n_all = 0;
            for (int j = 0; j < 16; j++) {
                if (j < w) base += wcount[j];
                n_all += wcount[j];
            }
        };
        count_alive();
        if (n_all > (uint32_t)SAMPLE_CAP) {      // uniform over the workgroup: the row goes to sample_wide_kernel
            if (t == 0) {
                out[row] = 0xFFFFFFFFu;
                if (wide_kth) wide_kth[row] = kth;
                else if (status) atomicExch(status, 1);
            }
            return;
        }
        n_kept = (int)n_all;
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
    }
    SMP_T(5);
    const int n = n_kept;

    SMP_T(6);
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
        if (t == 0) {        // the reference's sums, in its order; no early exit so that the LDS reads run ahead
            float sum = 0.0f;
#pragma unroll 8
            for (int r = 0; r < n; r++) sum += pr[r];
            float cum = 0.0f;
            int cut = n;
#pragma unroll 8
            for (int r = 0; r < n; r++) {
                cum += pr[r] / sum;
                if (cut == n && cum > top_p && r > 0) cut = r;
            }
            s_cut = cut;
        }
        __syncthreads();
        const int cut = s_cut;
        for (int r = cut + t; r < n; r += 1024) sv_bits[order[r]] = H_NEG_INF;
        __syncthreads();
    }

    SMP_T(7);
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
#pragma unroll 8
            for (int s = 0; s < n; s++) sum += pr[s];
            if (sum != 0.0f) {
                const float sample = draws[row];
                push = 1;
                int found = -1;
                if (sample <= 0.0f) {
                    found = 0;               // the running sum is >= 0 at element 0 already
                } else {
                    float cum = 0.0f;
                    int hit = -1;
#pragma unroll 8
                    for (int s = 0; s < n; s++) {
                        cum += pr[s] / sum;
                        if (hit < 0 && cum >= sample) hit = s;
                    }
                    if (hit >= 0) found = (int)sv_idx[hit];
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
    SMP_T(8);
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
#ifdef SAMPLE_TRACE
extern "C" int cactus_hip_debug_sample_trace(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sample_trace), sizeof(unsigned long long) * 16);
}
#endif

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

// The same figure with a row spread over `parts` workgroups (one workgroup per row leaves a single CU to evaluate
// 2 x vocab double-precision exponentials): slice maxima and sums first, then - every workgroup combines the row's
// slice results itself - the entropy terms of a slice, then one thread per row adds the slices' terms.
struct EntropyPart { double se; float m; float pad; };

__device__ __forceinline__ double entropy_block_sum(double v, double* red) {   // 256 threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void entropy_slice_sums_kernel(const uint16_t* __restrict__ logits,
                                                                 EntropyPart* __restrict__ parts, int vocab, int nparts) {
    __shared__ double red[4];
    __shared__ float wmax[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int row = blockIdx.x / nparts, g = blockIdx.x % nparts;
    const int len = (vocab + nparts - 1) / nparts, lo = min(vocab, g * len), hi = min(vocab, lo + len);
    const uint16_t* L = logits + (size_t)row * vocab;
    float m = -INFINITY;
    for (int i = lo + t; i < hi; i += 256) m = fmaxf(m, h2f(L[i]));
    m = wave_max(m);
    if (lane == 0) wmax[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    double se = 0.0;
    for (int i = lo + t; i < hi; i += 256) se += exp((double)(h2f(L[i]) - m));
    se = entropy_block_sum(se, red);
    if (t == 0) parts[(size_t)row * nparts + g] = EntropyPart{lo < hi ? se : 0.0, m, 0.0f};
}

__global__ __launch_bounds__(256) void entropy_slice_terms_kernel(const uint16_t* __restrict__ logits,
                                                                  const EntropyPart* __restrict__ parts,
                                                                  double* __restrict__ terms, int vocab, int nparts) {
    __shared__ double red[4];
    __shared__ double s_lse;
    const int t = threadIdx.x;
    const int row = blockIdx.x / nparts, g = blockIdx.x % nparts;
    const int len = (vocab + nparts - 1) / nparts, lo = min(vocab, g * len), hi = min(vocab, lo + len);
    const uint16_t* L = logits + (size_t)row * vocab;
    if (t < 64) {            // the row's maximum and sum from its slices (at most 64), a slice per lane
        const EntropyPart* P = parts + (size_t)row * nparts;
        const float mj = t < nparts ? P[t].m : -INFINITY;
        const float m = wave_max(mj);
        double se = (t < nparts && mj > -INFINITY) ? P[t].se * exp((double)(mj - m)) : 0.0;
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
        if (t == 0) s_lse = (double)m + log(se);
    }
    __syncthreads();
    const double lse = s_lse;
    double ent = 0.0;
    for (int i = lo + t; i < hi; i += 256) {
        const double lp = (double)h2f(L[i]) - lse;
        const double p = exp(lp);
        if (p > 1e-10) ent -= p * lp;
    }
    ent = entropy_block_sum(ent, red);
    if (t == 0) terms[(size_t)row * nparts + g] = ent;
}

__global__ __launch_bounds__(256) void entropy_finish_kernel(const double* __restrict__ terms, float* __restrict__ out,
                                                             int rows, int vocab, int nparts) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    double ent = 0.0;
    for (int j = 0; j < nparts; j++) ent += terms[(size_t)row * nparts + j];
    out[row] = (float)(ent / log((double)vocab));
}

// slices per row: about 1024 workgroups in all, at most 64 per row, at least 1024 logits per slice
int entropy_parts(size_t rows, size_t vocab) {
    size_t g = rows ? 1024 / rows : 1;
    if (g > 64) g = 64;
    if (g > vocab / 1024) g = vocab / 1024;
    return g < 2 ? 1 : (int)g;
}
size_t entropy_scratch_bytes(size_t rows, size_t vocab) {
    const size_t g = (size_t)entropy_parts(rows, vocab);
    return g < 2 ? 0 : rows * g * (sizeof(EntropyPart) + sizeof(double));
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

// scratch: entropy_scratch_bytes(rows, vocab) bytes, or null (then one workgroup per row)
hipError_t launch_entropy(const uint16_t* logits, float* out, size_t rows, size_t vocab, void* scratch, hipStream_t st) {
    if (rows == 0) return hipSuccess;
    if (vocab == 0 || vocab > 0x7FFFFFFFull || rows > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const int g = entropy_parts(rows, vocab);
    if (g < 2 || scratch == nullptr || rows * (size_t)g > 0x7FFFFFFFull) {
        entropy_kernel<<<dim3((unsigned)rows), dim3(1024), 0, st>>>(logits, out, (int)vocab);
        return hipGetLastError();
    }
    EntropyPart* parts = static_cast<EntropyPart*>(scratch);
    double* terms = reinterpret_cast<double*>(parts + rows * g);
    entropy_slice_sums_kernel<<<dim3((unsigned)(rows * g)), dim3(256), 0, st>>>(logits, parts, (int)vocab, g);
    entropy_slice_terms_kernel<<<dim3((unsigned)(rows * g)), dim3(256), 0, st>>>(logits, parts, terms, (int)vocab, g);
    entropy_finish_kernel<<<dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st>>>(terms, out, (int)rows, (int)vocab, g);
    return hipGetLastError();
}

}  // namespace cactus_hip
