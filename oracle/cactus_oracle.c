/*
 * cactus_oracle.c -- CPU restatement of the Cactus quantized hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cactus_oracle.h): the checker the HIP kernels
 * are compared with, and the CPU baseline bench.py times.  Plain scalar C; the
 * reference's NEON lane structure is restated with explicit per-lane scalars
 * wherever it decides a floating-point summation order.
 *
 * Citations are to /root/reference (cactus v1.5.0), path:line.
 *
 * Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp (oracle/Makefile).
 * -ffp-contract=off matters: every fused multiply-add below is an explicit
 * fmaf()/fma() so that g_fused alone decides where one rounding is used.
 */
#include "cactus_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int g_fused = 1;
static int g_threads = 1;

void oracle_set_fma(int fused) { g_fused = fused ? 1 : 0; }
void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

/* a*b + c the way the reference's compiler evaluates `c += a * b` / vfmaq_f32:
 * vfmaq_* is always fused; plain C++ statements are fused only under the
 * default -ffp-contract=on of the ARM toolchains (no flag in
 * cactus/CMakeLists.txt:21,34), which g_fused models. */
static inline float mad_stmt(float a, float b, float c) {
    if (g_fused) return fmaf(a, b, c);
    float p = a * b;
    return p + c;
}

/* ------------------------------------------------------------------------ */
/* binary16 helpers                                                          */
/* ------------------------------------------------------------------------ */

float oracle_h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* double -> binary16, round-to-nearest-even, straight from the double's bits
 * (no intermediate float, so no double rounding). */
static uint16_t d2h(double d) {
    uint64_t bits;
    memcpy(&bits, &d, 8);
    uint16_t sign = (uint16_t)((bits >> 48) & 0x8000u);
    int exp = (int)((bits >> 52) & 0x7FF);
    uint64_t man = bits & 0xFFFFFFFFFFFFFull;
    if (exp == 0x7FF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
    if (exp == 0) return sign; /* double subnormal: far below half's range */
    int e = exp - 1023;        /* unbiased */
    uint64_t sig = man | (1ull << 52); /* 53-bit significand, value = sig * 2^(e-52) */
    int shift;                 /* bits dropped from sig */
    int hexp;                  /* biased half exponent of the result before carry */
    if (e >= -14) {
        shift = 42;            /* keep 11 bits */
        hexp = e + 15;
    } else {
        shift = 42 + (-14 - e); /* subnormal half: fewer bits kept */
        hexp = 0;
        if (shift > 63) return sign; /* rounds to zero (value < 2^-25 / tie impossible) */
    }
    uint64_t kept = sig >> shift;
    uint64_t rem = sig & ((1ull << shift) - 1);
    uint64_t half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (kept & 1))) kept++;
    uint32_t out;
    if (hexp == 0) {
        out = (uint32_t)kept; /* a carry into bit 10 makes it the smallest normal: correct */
    } else {
        out = ((uint32_t)hexp << 10) + (uint32_t)(kept - 0x400u); /* carry bumps exponent */
    }
    if (out >= 0x7C00u) out = 0x7C00u; /* overflow -> inf */
    return (uint16_t)(sign | out);
}

uint16_t oracle_f2h(float f) { return d2h((double)f); }

/* fp16 a*b+c with ONE rounding to binary16 (NEON vfmaq_f16 lane semantics).
 * p is exact in double; the sum is made exact-or-sticky (round to odd) before
 * the single rounding to half. */
static uint16_t h_fma(uint16_t c, uint16_t a, uint16_t b) {
    double p = (double)oracle_h2f(a) * (double)oracle_h2f(b);
    double cc = (double)oracle_h2f(c);
    double s = cc + p;
    if (isfinite(s)) {
        double bb = s - cc;
        double err = (cc - (s - bb)) + (p - bb);
        if (err != 0.0) {
            uint64_t bits;
            memcpy(&bits, &s, 8);
            if (!(bits & 1)) {
                /* move one ulp toward the true value, which makes the last bit odd */
                int toward_up = (err > 0.0) == (s > 0.0);
                if (s == 0.0) { s = err; }
                else { bits += toward_up ? 1 : (uint64_t)-1; memcpy(&s, &bits, 8); }
            }
        }
    }
    return d2h(s);
}

/* fp16 a+b with one rounding (vadd_f16); exact in double. */
static uint16_t h_add(uint16_t a, uint16_t b) {
    return d2h((double)oracle_h2f(a) + (double)oracle_h2f(b));
}

/* ------------------------------------------------------------------------ */
/* INT4 nibble decode: cactus/kernel/kernel_utils.h:165-185                  */
/* low nibble = even element, high nibble = odd element, value n>7 ? n-16 : n */
/* ------------------------------------------------------------------------ */
static inline int8_t nib_lo(uint8_t b) { int n = b & 0x0F; return (int8_t)(n > 7 ? n - 16 : n); }
static inline int8_t nib_hi(uint8_t b) { int n = b >> 4;   return (int8_t)(n > 7 ? n - 16 : n); }

/* cactus/kernel/kernel_quants.cpp:295-351 */
void oracle_unpack_int4_to_int8(const uint8_t* packed, int8_t* unpacked, size_t unpacked_count) {
    size_t packed_count = (unpacked_count + 1) / 2;
    /* The 16-byte vector body only stores a 16-element half when it fits
     * entirely (kernel_quants.cpp:325-330); the byte tail stores element-wise
     * (:343-348).  With one thread simd_end = (packed_count/16)*16. */
    const size_t simd_end = (packed_count / 16) * 16;
    for (size_t i = 0; i < packed_count; i++) {
        size_t o = 2 * i;
        if (i < simd_end) {
            size_t blk = (i / 16) * 16;      /* first packed byte of this vector */
            size_t out0 = blk * 2;           /* first output of val[0] */
            size_t half_start = (o < out0 + 16) ? out0 : out0 + 16;
            if (half_start + 16 > unpacked_count) continue; /* half not stored */
        }
        if (o < unpacked_count) unpacked[o] = nib_lo(packed[i]);
        if (o + 1 < unpacked_count) unpacked[o + 1] = nib_hi(packed[i]);
    }
}

/* ------------------------------------------------------------------------ */
/* Grouped INT8 / INT4 matmul                                                */
/* cactus/kernel/kernel_gemm.cpp:128-327 (int8), :329-529 (int4)             */
/*   num_groups = K / group_size              (:147, :348; K tail dropped)   */
/*   acc_g = sum_{k in g} A[m,k]*B[n,k]       exact INT32 (:170-309)         */
/*   running_sum += (float)acc_g * (float)Bs[n,g], g ascending (:311-316)    */
/*   C[m,n] = (__fp16)(running_sum * A_scales[m]) (:319-324)                 */
/* ------------------------------------------------------------------------ */
static void matmul_q(const int8_t* A, const float* A_scales,
                     const int8_t* B8, const uint8_t* B4, const uint16_t* B_scales,
                     uint16_t* C, size_t M, size_t K, size_t N, size_t gs,
                     int32_t* gsum_out) {
    if (M == 0 || K == 0 || N == 0) return; /* kernel_gemm.cpp:137, :338 */
    const size_t G = K / gs;
    const size_t Kp = K / 2;
    #pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
        int8_t* brow = (int8_t*)malloc(gs ? gs : 1);
        float* run = (float*)malloc(sizeof(float) * M);
        #pragma omp for schedule(static)
        for (long long nn = 0; nn < (long long)N; nn++) {
            size_t n = (size_t)nn;
            for (size_t m = 0; m < M; m++) run[m] = 0.0f;
            for (size_t g = 0; g < G; g++) {
                const int8_t* bp;
                if (B4) {
                    const uint8_t* p = B4 + n * Kp + (g * gs) / 2;
                    for (size_t k = 0; k < gs; k += 2) {
                        brow[k] = nib_lo(p[k / 2]);
                        brow[k + 1] = nib_hi(p[k / 2]);
                    }
                    bp = brow;
                } else {
                    bp = B8 + n * K + g * gs;
                }
                const float bscale = oracle_h2f(B_scales[n * G + g]);
                for (size_t m = 0; m < M; m++) {
                    const int8_t* ap = A + m * K + g * gs;
                    int32_t acc = 0;
                    for (size_t k = 0; k < gs; k++) acc += (int32_t)ap[k] * (int32_t)bp[k];
                    if (gsum_out) gsum_out[(m * N + n) * G + g] = acc;
                    run[m] = mad_stmt((float)acc, bscale, run[m]);
                }
            }
            for (size_t m = 0; m < M; m++) C[m * N + n] = oracle_f2h(run[m] * A_scales[m]);
        }
        free(brow);
        free(run);
    }
}

void oracle_matmul_int8(const int8_t* A, const float* A_scales, const int8_t* B,
                        const uint16_t* B_scales, uint16_t* C, size_t M, size_t K, size_t N,
                        size_t group_size, int32_t* group_sums_out) {
    matmul_q(A, A_scales, B, NULL, B_scales, C, M, K, N, group_size, group_sums_out);
}

void oracle_matmul_int4(const int8_t* A, const float* A_scales, const uint8_t* B_packed,
                        const uint16_t* B_scales, uint16_t* C, size_t M, size_t K, size_t N,
                        size_t group_size, int32_t* group_sums_out) {
    matmul_q(A, A_scales, NULL, B_packed, B_scales, C, M, K, N, group_size, group_sums_out);
}

/* ------------------------------------------------------------------------ */
/* FP16 matmul with fp16 accumulators: cactus/kernel/kernel_gemm.cpp:8-126   */
/* Lane j of the 8-wide accumulator takes k = 16i+j then 16i+8+j (two        */
/* vfmaq_f16 per 16-step, :57-58), then k = K16+8i+j (:76), lane 0 takes the */
/* scalar tail in float (:88), and hsum_f16x8 folds (l0+l4)+(l2+l6) with     */
/* (l1+l5)+(l3+l7) (:8-15).                                                  */
/* ------------------------------------------------------------------------ */
void oracle_matmul_f16(const uint16_t* a, const uint16_t* bT, uint16_t* c,
                       size_t M, size_t K, size_t N) {
    const size_t K16 = (K / 16) * 16, K8 = (K / 8) * 8;
    #pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(static)
    for (long long mm = 0; mm < (long long)M; mm++) {
        size_t m = (size_t)mm;
        for (size_t n = 0; n < N; n++) {
            uint16_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint16_t* ar = a + m * K;
            const uint16_t* br = bT + n * K;
            for (size_t k = 0; k < K16; k += 16) {
                for (int j = 0; j < 8; j++) acc[j] = h_fma(acc[j], ar[k + j], br[k + j]);
                for (int j = 0; j < 8; j++) acc[j] = h_fma(acc[j], ar[k + 8 + j], br[k + 8 + j]);
            }
            for (size_t k = K16; k < K8; k += 8)
                for (int j = 0; j < 8; j++) acc[j] = h_fma(acc[j], ar[k + j], br[k + j]);
            for (size_t k = K8; k < K; k++) {
                /* __fp16 operands promote to float; the product is exact in float */
                float t = oracle_h2f(acc[0]) + oracle_h2f(ar[k]) * oracle_h2f(br[k]);
                acc[0] = oracle_f2h(t);
            }
            uint16_t s4[4];
            for (int j = 0; j < 4; j++) s4[j] = h_add(acc[j], acc[j + 4]);
            uint16_t s2_0 = h_add(s4[0], s4[2]);
            uint16_t s2_1 = h_add(s4[1], s4[3]);
            c[m * N + n] = h_add(s2_0, s2_1);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Activation quantization                                                   */
/* cactus/kernel/kernel_quants.cpp:196-221 (max_abs), :158-194 (to_int8),    */
/* cactus/graph/graph_ops_nn.cpp:46-50 (scale rule).                         */
/* Restates the single-thread partition (count < 5000, the ELEMENT_WISE gate */
/* of kernel_utils.h:431): multiples of 8 round to nearest even after the    */
/* clamp; the count%8 tail rounds half away from zero.                       */
/* ------------------------------------------------------------------------ */
float oracle_fp16_max_abs(const uint16_t* src, size_t count) {
    float m = 0.0f;
    for (size_t i = 0; i < count; i++) {
        float v = fabsf(oracle_h2f(src[i]));
        if (v > m) m = v;
    }
    return m;
}

static inline int8_t sat8(long v) { return (int8_t)(v < -128 ? -128 : (v > 127 ? 127 : v)); }

void oracle_fp16_to_int8(const uint16_t* src, int8_t* dst, size_t count, float scale) {
    const float inv = 1.0f / scale;              /* kernel_quants.cpp:159 */
    const size_t simd_end = (count / 8) * 8;
    for (size_t i = 0; i < count; i++) {
        float q = oracle_h2f(src[i]) * inv;
        q = fmaxf(fminf(q, 127.0f), -128.0f);    /* :177-178 / :191 */
        if (i < simd_end) dst[i] = sat8(lrintf(q));   /* vcvtnq_s32_f32: ties to even */
        else dst[i] = (int8_t)roundf(q);              /* std::round: ties away */
    }
}

void oracle_quantize_activations(const uint16_t* src, int8_t* dst, float* scales,
                                 size_t M, size_t K) {
    for (size_t m = 0; m < M; m++) {
        float max_abs = oracle_fp16_max_abs(src + m * K, K);
        float scale = max_abs / 127.0f;
        if (scale < 1e-10f) scale = 1e-10f;
        scales[m] = scale;
        oracle_fp16_to_int8(src + m * K, dst + m * K, K, scale);
    }
}

/* ------------------------------------------------------------------------ */
/* KV quantization: cactus/kernel/kernel_quants.cpp:223-293                  */
/* No clamp before the rounding in the vector body (:248-252, saturating     */
/* narrows); the count%8 tail uses roundf then clamps (:256-261).            */
/* ------------------------------------------------------------------------ */
static float quantize_group(const uint16_t* src, int8_t* dst, size_t count) {
    float max_abs = 0.0f;
    for (size_t k = 0; k < count; k++) {
        float v = fabsf(oracle_h2f(src[k]));
        if (v > max_abs) max_abs = v;
    }
    float scale = max_abs / 127.0f;
    if (scale < 1e-10f) scale = 1e-10f;
    const float inv = 1.0f / scale;
    const size_t simd_end = (count / 8) * 8;
    for (size_t k = 0; k < count; k++) {
        float v = oracle_h2f(src[k]) * inv;
        if (k < simd_end) dst[k] = sat8(lrintf(v));
        else dst[k] = sat8((long)roundf(v));
    }
    return scale;
}

void oracle_quantize_kv_fp16_to_int8(const uint16_t* src, int8_t* dst, float* scales,
                                     size_t seq_len, size_t kv_heads, size_t head_dim,
                                     size_t group_size) {
    const size_t G = (head_dim + group_size - 1) / group_size;
    for (size_t idx = 0; idx < seq_len * kv_heads; idx++) {
        for (size_t g = 0; g < G; g++) {
            size_t start = g * group_size;
            size_t cnt = head_dim - start < group_size ? head_dim - start : group_size;
            scales[idx * G + g] = quantize_group(src + idx * head_dim + start,
                                                 dst + idx * head_dim + start, cnt);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Attention.  Shared skeleton of cactus/kernel/kernel_attention.cpp:        */
/*   8 fp32 lanes per dot product: lane j sums dims d = j (mod 8) with fmas  */
/*   (:132-133, :369-370, :385-386); score = vaddvq(low+high) = ((s0+s1)+    */
/*   (s2+s3)) with s_j = low_j + high_j (:136, :390).  Blocks of 32 keys;    */
/*   per block the running max is REPLACED by the block max and running sum  */
/*   / accumulators are rescaled by expf(running_max - block_max) (:160-169, */
/*   :395-404).  Output accumulators are fmas in key order (:221-222,        */
/*   :438-439), final = acc * (1/running_sum) rounded to fp16 (:229-243).    */
/* ------------------------------------------------------------------------ */
static inline float lanes_sum8(const float* l) {
    float s0 = l[0] + l[4], s1 = l[1] + l[5], s2 = l[2] + l[6], s3 = l[3] + l[7];
    return (s0 + s1) + (s2 + s3);
}

/* 4-wide polynomial 2^x used by cactus_attention_f16 (kernel_attention.cpp:174-195),
 * restated on one element with the same integer steps (wrapping shift included). */
static inline float poly_exp_weight(float score, float block_max) {
    if (score == -INFINITY) return 0.0f;                  /* inf_mask -> 0 (:176,191) */
    float x = (score - block_max) * 1.442695f;
    int32_t xi;
    if (isnan(x)) xi = 0;
    else if (x <= -2147483648.0f) xi = INT32_MIN;
    else if (x >= 2147483648.0f) xi = INT32_MAX;
    else xi = (int32_t)x;                                 /* vcvtq_s32_f32: toward zero */
    float xf = x - (float)xi;
    float y = fmaf(xf, 0.6931472f, 1.0f);                 /* vfmaq_n_f32 */
    y = fmaf(xf * xf, 0.2402265f, y);                     /* vfmaq_f32   */
    uint32_t e = ((uint32_t)xi + 127u) << 23;             /* vaddq_s32 + vshlq_n_s32 */
    float p2;
    memcpy(&p2, &e, 4);
    return y * p2;
}

void oracle_attention_f16(const uint16_t* queries, const uint16_t* keys, const uint16_t* values,
                          uint16_t* output, size_t B, size_t S, size_t KV,
                          size_t Hq, size_t Hkv, size_t D, float scale,
                          const uint16_t* mask, size_t position_offset, size_t window_size,
                          int is_causal) {
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);   /* :27-29 */
    const size_t BLOCK = 32, Da = (D / 8) * 8;
    const size_t gq = Hq / Hkv;
    const long long total = (long long)(B * Hq * S);
    #pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
        float* acc = (float*)malloc(sizeof(float) * (Da ? Da : 1));
        float bs[32];
        #pragma omp for schedule(static)
        for (long long w = 0; w < total; w++) {
            size_t b = (size_t)w / (Hq * S), rem = (size_t)w % (Hq * S);
            size_t h = rem / S, qp = rem % S, kvh = h / gq;
            const uint16_t* q = queries + b * S * Hq * D + qp * Hq * D + h * D;
            const uint16_t* Kb = keys + b * KV * Hkv * D;
            const uint16_t* Vb = values + b * KV * Hkv * D;
            const uint16_t* Mk = mask ? mask + b * S * KV : NULL;
            uint16_t* o = output + b * S * Hq * D + qp * Hq * D + h * D;
            float run_max = -INFINITY, run_sum = 0.0f;
            for (size_t d = 0; d < Da; d++) acc[d] = 0.0f;
            const int is_decode = (qp == S - 1) && S > 1;  /* :75 */
            const size_t qabs = position_offset + qp;
            size_t kv_start = 0, kv_end = KV;
            if (window_size > 0 && window_size < KV) {   /* :81-90 */
                if (qabs > window_size) kv_start = qabs - window_size;
                if (is_causal && qabs + 1 < kv_end) kv_end = qabs + 1;
            } else if (is_causal) {
                if (qabs + 1 < kv_end) kv_end = qabs + 1;
            }
            for (size_t kb = kv_start; kb < kv_end; kb += BLOCK) {
                size_t ke = kb + BLOCK < KV ? kb + BLOCK : KV; /* clamps to KV, not kv_end (:93) */
                size_t bsz = ke - kb;
                float bmax = -INFINITY;
                if (!is_decode && is_causal && kb > qabs) continue; /* :98-103 */
                for (size_t i = 0; i < bsz; i++) {
                    size_t kp = kb + i;
                    if (!is_decode && is_causal && kp > qabs) { bs[i] = -INFINITY; continue; }
                    const uint16_t* k = Kb + kp * Hkv * D + kvh * D;
                    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (size_t d = 0; d < Da; d += 8)
                        for (int j = 0; j < 8; j++)
                            l[j] = fmaf(oracle_h2f(q[d + j]), oracle_h2f(k[d + j]), l[j]);
                    float sc = lanes_sum8(l);
                    for (size_t d = Da; d < D; d++)          /* scalar tail (:138-140) */
                        sc = mad_stmt(oracle_h2f(q[d]), oracle_h2f(k[d]), sc);
                    sc *= scale;
                    if (is_causal && kp > qabs) sc = -INFINITY;              /* :146 */
                    else if (window_size > 0 && kp < qabs && (qabs - kp) > window_size) sc = -INFINITY;
                    else if (Mk && oracle_h2f(Mk[qp * KV + kp]) == 0.0f) sc = -INFINITY;
                    bs[i] = sc;
                    if (sc > bmax) bmax = sc;
                }
                if (bmax > -INFINITY) {
                    float corr = expf(run_max - bmax);
                    run_sum *= corr;
                    for (size_t d = 0; d < Da; d++) acc[d] *= corr;
                    run_max = bmax;
                }
                float bsum = 0.0f;
                size_t vec = (bsz / 4) * 4;
                for (size_t i = 0; i < vec; i += 4) {        /* :174-195 */
                    float y[4];
                    for (int j = 0; j < 4; j++) { y[j] = poly_exp_weight(bs[i + j], bmax); bs[i + j] = y[j]; }
                    bsum += (y[0] + y[1]) + (y[2] + y[3]);
                }
                for (size_t i = vec; i < bsz; i++) {         /* :197-204 */
                    if (bs[i] != -INFINITY) { bs[i] = expf(bs[i] - bmax); bsum += bs[i]; }
                    else bs[i] = 0.0f;
                }
                for (size_t i = 0; i < bsz; i++) {
                    float wgt = bs[i];
                    if (wgt == 0.0f) continue;
                    const uint16_t* v = Vb + (kb + i) * Hkv * D + kvh * D;
                    for (size_t d = 0; d < Da; d++) acc[d] = fmaf(oracle_h2f(v[d]), wgt, acc[d]);
                }
                run_sum += bsum;
            }
            if (run_sum > 0.0f) {
                float inv = 1.0f / run_sum;
                for (size_t d = 0; d < Da; d++) o[d] = oracle_f2h(acc[d] * inv);
                for (size_t d = Da; d < D; d++) o[d] = 0;   /* :245-247 */
            } else {
                for (size_t d = 0; d < D; d++) o[d] = 0;
            }
        }
        free(acc);
    }
}

/* cactus/kernel/kernel_attention.cpp:257-482 */
void oracle_attention_hybrid_int8_fp16(const uint16_t* queries,
                                       const int8_t* keys_cached, const int8_t* values_cached,
                                       const float* k_scales, const float* v_scales,
                                       const uint16_t* keys_new, const uint16_t* values_new,
                                       uint16_t* output,
                                       size_t B, size_t S, size_t cache_len, size_t new_len,
                                       size_t Hq, size_t Hkv, size_t D, float scale,
                                       size_t position_offset, int is_causal, size_t qgs) {
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);   /* :278-280 */
    const size_t KV = cache_len + new_len;
    const size_t BLOCK = 32, Da = (D / 8) * 8;
    const size_t gq = Hq / Hkv;
    const size_t G = (D + qgs - 1) / qgs;
    const long long total = (long long)(B * Hq * S);
    #pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
        float* acc = (float*)malloc(sizeof(float) * (Da ? Da : 1));
        float bs[32];
        #pragma omp for schedule(static)
        for (long long w = 0; w < total; w++) {
            size_t b = (size_t)w / (Hq * S), rem = (size_t)w % (Hq * S);
            size_t h = rem / S, qp = rem % S, kvh = h / gq;
            const uint16_t* q = queries + b * S * Hq * D + qp * Hq * D + h * D;
            const int8_t* Kc = keys_cached + b * cache_len * Hkv * D;
            const int8_t* Vc = values_cached + b * cache_len * Hkv * D;
            const uint16_t* Kn = keys_new + b * new_len * Hkv * D;
            const uint16_t* Vn = values_new + b * new_len * Hkv * D;
            uint16_t* o = output + b * S * Hq * D + qp * Hq * D + h * D;
            float run_max = -INFINITY, run_sum = 0.0f;
            for (size_t d = 0; d < Da; d++) acc[d] = 0.0f;
            const size_t qabs = position_offset + qp;
            size_t kv_end = KV;
            if (is_causal && qabs + 1 < KV) kv_end = qabs + 1;  /* :331-332 */
            for (size_t kb = 0; kb < kv_end; kb += BLOCK) {
                size_t ke = kb + BLOCK < kv_end ? kb + BLOCK : kv_end;
                size_t bsz = ke - kb;
                float bmax = -INFINITY;
                for (size_t i = 0; i < bsz; i++) {
                    size_t kp = kb + i;
                    if (is_causal && kp > qabs) { bs[i] = -INFINITY; continue; }
                    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (kp < cache_len) {
                        const int8_t* k = Kc + kp * Hkv * D + kvh * D;
                        const float* ks = k_scales + (kp * Hkv + kvh) * G;  /* :353 (no batch stride) */
                        for (size_t d = 0; d < Da; d += 8) {
                            float s = ks[d / qgs];
                            for (int j = 0; j < 8; j++) {
                                float kd = (float)k[d + j] * s;             /* :366-367 */
                                l[j] = fmaf(oracle_h2f(q[d + j]), kd, l[j]);
                            }
                        }
                    } else {
                        const uint16_t* k = Kn + (kp - cache_len) * Hkv * D + kvh * D;
                        for (size_t d = 0; d < Da; d += 8)
                            for (int j = 0; j < 8; j++)
                                l[j] = fmaf(oracle_h2f(q[d + j]), oracle_h2f(k[d + j]), l[j]);
                    }
                    float sc = lanes_sum8(l) * scale;                       /* :390 */
                    bs[i] = sc;
                    if (sc > bmax) bmax = sc;
                }
                if (bmax > -INFINITY) {
                    float corr = expf(run_max - bmax);
                    run_sum *= corr;
                    for (size_t d = 0; d < Da; d++) acc[d] *= corr;
                    run_max = bmax;
                }
                float bsum = 0.0f;
                for (size_t i = 0; i < bsz; i++) {                          /* :407-414 */
                    if (bs[i] != -INFINITY) { bs[i] = expf(bs[i] - bmax); bsum += bs[i]; }
                    else bs[i] = 0.0f;
                }
                for (size_t i = 0; i < bsz; i++) {
                    float wgt = bs[i];
                    if (wgt == 0.0f) continue;
                    size_t kp = kb + i;
                    if (kp < cache_len) {
                        const int8_t* v = Vc + kp * Hkv * D + kvh * D;
                        const float* vs = v_scales + (kp * Hkv + kvh) * G;
                        for (size_t d = 0; d < Da; d++) {
                            float vd = (float)v[d] * vs[d / qgs];
                            acc[d] = fmaf(vd, wgt, acc[d]);
                        }
                    } else {
                        const uint16_t* v = Vn + (kp - cache_len) * Hkv * D + kvh * D;
                        for (size_t d = 0; d < Da; d++) acc[d] = fmaf(oracle_h2f(v[d]), wgt, acc[d]);
                    }
                }
                run_sum += bsum;
            }
            if (run_sum > 0.0f) {
                float inv = 1.0f / run_sum;
                /* dims beyond the last multiple of 8 are not written (:464-474) */
                for (size_t d = 0; d < Da; d++) o[d] = oracle_f2h(acc[d] * inv);
            } else {
                for (size_t d = 0; d < D; d++) o[d] = 0;
            }
        }
        free(acc);
    }
}

/* ======================================================================== */
/* Ops between the hot kernels of a decoder block                            */
/* ======================================================================== */

/* fp16 a*b with one rounding (vmulq_f16): the product of two halves is exact
 * in double. */
static uint16_t h_mul(uint16_t a, uint16_t b) {
    return d2h((double)oracle_h2f(a) * (double)oracle_h2f(b));
}
static uint16_t h_neg(uint16_t a) { return (uint16_t)(a ^ 0x8000u); }

/* ------------------------------------------------------------------------ */
/* cactus_rms_norm_f16: cactus/kernel/kernel_attention.cpp:484-563           */
/*   Sum of squares in fp32 over 16 lane accumulators: element i < simd_end  */
/*   (= dims rounded down to 8) goes to accumulator i % 16 with vfmaq_f32    */
/*   (:508-525: four float32x4 registers over a 16-element tile, the 8-wide  */
/*   remainder into the first two); the registers are added ((v0+v1)+v2)+v3  */
/*   lane-wise (:527-530) and reduced with vaddvq_f32 = (t0+t1)+(t2+t3)      */
/*   (:531); the scalar tail adds val*val as a plain statement (:533-536).   */
/*   rms = sqrtf(ss/dims + eps), inv = 1/rms (:538-539).  The vector part    */
/*   multiplies in fp16: (x * half(inv)) * w, each product rounded (:540-557);*/
/*   the tail computes x*inv*w in fp32 and rounds once (:559-561).           */
/* ------------------------------------------------------------------------ */
void oracle_rms_norm_f16(const uint16_t* input, const uint16_t* weight, uint16_t* output,
                         size_t batch_size, size_t dims, float eps) {
    const size_t simd_end = (dims / 8) * 8;
    for (size_t b = 0; b < batch_size; b++) {
        const uint16_t* x = input + b * dims;
        uint16_t* y = output + b * dims;
        float acc[16];
        for (int l = 0; l < 16; l++) acc[l] = 0.0f;
        for (size_t i = 0; i < simd_end; i++) {
            const float v = oracle_h2f(x[i]);
            acc[i % 16] = fmaf(v, v, acc[i % 16]);
        }
        float t[4];
        for (int l = 0; l < 4; l++) t[l] = ((acc[l] + acc[4 + l]) + acc[8 + l]) + acc[12 + l];
        float ss = (t[0] + t[1]) + (t[2] + t[3]);
        for (size_t i = simd_end; i < dims; i++) {
            const float v = oracle_h2f(x[i]);
            ss = mad_stmt(v, v, ss);
        }
        const float rms = sqrtf(ss / (float)dims + eps);
        const float inv = 1.0f / rms;
        const uint16_t inv_h = oracle_f2h(inv);
        for (size_t i = 0; i < simd_end; i++) y[i] = h_mul(h_mul(x[i], inv_h), weight[i]);
        for (size_t i = simd_end; i < dims; i++)
            y[i] = oracle_f2h(oracle_h2f(x[i]) * inv * oracle_h2f(weight[i]));
    }
}

/* ------------------------------------------------------------------------ */
/* RoPE tables: cactus/kernel/kernel_attention.cpp:594-604                   */
/*   freq = 1/powf(theta, (2.0f*i)/head_dim), angle = float(pos)*freq,       */
/*   cos/sin tables hold half(cosf(angle)), half(sinf(angle)), [pos][i].     */
/* ------------------------------------------------------------------------ */
void oracle_rope_tables(uint16_t* cos_t, uint16_t* sin_t, size_t pos0, size_t npos,
                        size_t head_dim, float theta) {
    const size_t half_dim = head_dim / 2;
    for (size_t p = 0; p < npos; p++) {
        const float pos_float = (float)(pos0 + p);
        for (size_t i = 0; i < half_dim; i++) {
            const float freq = 1.0f / powf(theta, (2.0f * (float)i) / (float)head_dim);
            const float angle = pos_float * freq;
            cos_t[p * half_dim + i] = oracle_f2h(cosf(angle));
            sin_t[p * half_dim + i] = oracle_f2h(sinf(angle));
        }
    }
}

/* ------------------------------------------------------------------------ */
/* cactus_rope_f16: cactus/kernel/kernel_attention.cpp:614-676               */
/*   Half-split rotation on [batch, seq, heads, head_dim]; token seq_idx uses*/
/*   table row start_pos + seq_idx (:628-629, :642-643).  Vector part (i <   */
/*   half_dim rounded down to 8), all in fp16: first = (x1*cos) - x2*sin with*/
/*   the product rounded and the subtraction fused (vfmsq_f16, :655), second */
/*   = (x2*cos) + x1*sin likewise (vfmaq_f16, :656).  The scalar tail (:662- */
/*   672) evaluates in fp32 (products of halves are exact there) and rounds  */
/*   the result to half.                                                      */
/* ------------------------------------------------------------------------ */
void oracle_rope_f16(const uint16_t* input, uint16_t* output, size_t batch_size, size_t seq_len,
                     size_t num_heads, size_t head_dim, size_t start_pos, float theta) {
    const size_t half_dim = head_dim / 2;
    const size_t vec = (half_dim / 8) * 8;
    uint16_t* cos_t = (uint16_t*)malloc(sizeof(uint16_t) * (seq_len * half_dim + 1));
    uint16_t* sin_t = (uint16_t*)malloc(sizeof(uint16_t) * (seq_len * half_dim + 1));
    oracle_rope_tables(cos_t, sin_t, start_pos, seq_len, head_dim, theta);
    for (size_t b = 0; b < batch_size; b++)
        for (size_t s = 0; s < seq_len; s++)
            for (size_t h = 0; h < num_heads; h++) {
                const size_t off = ((b * seq_len + s) * num_heads + h) * head_dim;
                const uint16_t* x = input + off;
                uint16_t* y = output + off;
                const uint16_t* c = cos_t + s * half_dim;
                const uint16_t* sn = sin_t + s * half_dim;
                for (size_t i = 0; i < vec; i++) {
                    const uint16_t x1 = x[i], x2 = x[i + half_dim];
                    y[i] = h_fma(h_mul(x1, c[i]), h_neg(x2), sn[i]);
                    y[i + half_dim] = h_fma(h_mul(x2, c[i]), x1, sn[i]);
                }
                for (size_t i = vec; i < half_dim; i++) {
                    const float x1 = oracle_h2f(x[i]), x2 = oracle_h2f(x[i + half_dim]);
                    const float cf = oracle_h2f(c[i]), sf = oracle_h2f(sn[i]);
                    y[i] = oracle_f2h(x1 * cf - x2 * sf);
                    y[i + half_dim] = oracle_f2h(x2 * cf + x1 * sf);
                }
            }
    free(cos_t);
    free(sin_t);
}

/* fast_exp_f32x4 on one element: cactus/kernel/kernel_utils.h:108-141 */
static inline float fast_exp(float x) {
    x = fmaxf(x, -87.0f);
    x = fminf(x, 87.0f);
    const float z = x * 1.4426950408889634f;
    int32_t zi = (int32_t)z;                      /* vcvtq_s32_f32: toward zero */
    float zf = z - (float)zi;
    if (zf < 0.0f) { zi -= 1; zf += 1.0f; }
    const float zl = zf * 0.6931471805599453f;
    float p = 0.009618129842071803f;
    p = fmaf(p, zl, 0.05550410866482158f);        /* vfmaq_f32(c, p, zl) = c + p*zl */
    p = fmaf(p, zl, 0.2402265069591007f);
    p = fmaf(p, zl, 0.6931471805599453f);
    p = fmaf(p, zl, 1.0f);
    const uint32_t e = (uint32_t)(zi + 127) << 23;
    float sc;
    memcpy(&sc, &e, 4);
    return p * sc;
}

/* fast_tanh_f32x4 on one element: cactus/kernel/kernel_utils.h:143-163 */
static inline float fast_tanh(float x) {
    const float x2 = x * x;
    const float num = 27.0f + x2;
    const float den = fmaf(9.0f, x2, 27.0f);
    float r = x * (num / den);
    if (x > 4.5f) r = 1.0f;
    if (x < -4.5f) r = -1.0f;
    return r;
}

/* The reference splits an element-wise call into `ranges` contiguous pieces
 * of num_elements / ranges elements, the last taking the remainder (kernel_utils.h:533-
 * 541; the count depends on the machine's pool size and the op's threshold, :413-419);
 * inside a piece, whole groups of 8 from the piece's start take the vector
 * formula and the rest the scalar libm one. */
typedef uint16_t (*unary_fn)(uint16_t, int vector_part);

static void unary_ranges(const uint16_t* in, uint16_t* out, size_t n, size_t ranges, unary_fn f) {
    if (ranges < 1) ranges = 1;
    if (ranges > n && n > 0) ranges = n;
    const size_t per = n / ranges;
    for (size_t t = 0; t < ranges; t++) {
        const size_t s = t * per, e = (t == ranges - 1) ? n : (t + 1) * per;
        const size_t vend = s + ((e - s) / 8) * 8;
        for (size_t i = s; i < e; i++) out[i] = f(in[i], i < vend);
    }
}

/* cactus_silu_f16: cactus/kernel/kernel_nn.cpp:12-44 */
static uint16_t silu_one(uint16_t h, int vector_part) {
    const float x = oracle_h2f(h);
    const float e = vector_part ? fast_exp(-x) : expf(-x);
    const float sig = 1.0f / (1.0f + e);
    return oracle_f2h(x * sig);
}
void oracle_silu_f16(const uint16_t* in, uint16_t* out, size_t n, size_t ranges) {
    unary_ranges(in, out, n, ranges, silu_one);
}

/* cactus_gelu_f16: cactus/kernel/kernel_nn.cpp:46-91 */
static uint16_t gelu_one(uint16_t h, int vector_part) {
    const float x = oracle_h2f(h);
    if (vector_part) {
        const float x3 = (x * x) * x;
        float inner = fmaf(0.044715f, x3, x);
        inner = 0.7978845608028654f * inner;
        const float th = fast_tanh(inner);
        return oracle_f2h((0.5f * x) * (1.0f + th));
    }
    /* scalar tail (:84-89): plain statements, contraction as the compiler chooses */
    const float x3 = 0.044715f * x * x * x;
    const float inner = 0.7978845608028654f * (x + x3);
    return oracle_f2h(0.5f * x * (1.0f + tanhf(inner)));
}
void oracle_gelu_f16(const uint16_t* in, uint16_t* out, size_t n, size_t ranges) {
    unary_ranges(in, out, n, ranges, gelu_one);
}

/* cactus_add_f16 / subtract / multiply / divide: cactus/kernel/kernel_blas.cpp:187-343.
 * One fp16 operation per element (vaddq_f16 ...); the scalar tails promote to
 * fp32 and round back, which gives the same half for all four operations.
 * op: 0 add, 1 subtract, 2 multiply, 3 divide. */
void oracle_binary_f16(int op, const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const double x = oracle_h2f(a[i]), y = oracle_h2f(b[i]);
        double r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            default: r = (double)((float)x / (float)y); break;   /* fp32 quotient, then half: 24 >= 2*11+2 */
        }
        out[i] = d2h(r);
    }
}

/* cactus_add_f16_clipped: cactus/kernel/kernel_blas.cpp:224-262 (fp32 sum clamped to +-65500) */
void oracle_add_f16_clipped(const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        float r = oracle_h2f(a[i]) + oracle_h2f(b[i]);
        r = fminf(fmaxf(r, -65500.0f), 65500.0f);
        out[i] = oracle_f2h(r);
    }
}

/* compute_embedding_node: cactus/graph/graph_ops_tensor.cpp:226-320.
 * INT8 rows with per-group fp16 scales: half(float(q) * float(scale)) (:243-254);
 * FP16 rows are copied (:301-316).  Returns -1 on an index >= vocab_size (the
 * reference throws, :261-263), 0 otherwise. */
int oracle_embedding_int8(const int8_t* emb, const uint16_t* scales, const int32_t* indices,
                          uint16_t* out, size_t num_indices, size_t vocab_size, size_t hidden_dim,
                          size_t group_size) {
    const size_t num_groups = hidden_dim / group_size;
    for (size_t i = 0; i < num_indices; i++) {
        const size_t idx = (size_t)indices[i];
        if (indices[i] < 0 || idx >= vocab_size) return -1;
        for (size_t g = 0; g < num_groups; g++) {
            const float sc = oracle_h2f(scales[idx * num_groups + g]);
            for (size_t k = g * group_size; k < (g + 1) * group_size && k < hidden_dim; k++)
                out[i * hidden_dim + k] = oracle_f2h((float)emb[idx * hidden_dim + k] * sc);
        }
    }
    return 0;
}

int oracle_embedding_f16(const uint16_t* emb, const int32_t* indices, uint16_t* out,
                         size_t num_indices, size_t vocab_size, size_t hidden_dim) {
    for (size_t i = 0; i < num_indices; i++) {
        const size_t idx = (size_t)indices[i];
        if (indices[i] < 0 || idx >= vocab_size) return -1;
        memcpy(out + i * hidden_dim, emb + idx * hidden_dim, hidden_dim * sizeof(uint16_t));
    }
    return 0;
}

/* Greedy branch of cactus_sample_f16: cactus/kernel/kernel_nn.cpp:450-466 */
uint32_t oracle_sample_greedy_f16(const uint16_t* logits, size_t vocab_size) {
    if (vocab_size == 0) return 0;
    size_t best_idx = 0;
    float best_val = oracle_h2f(logits[0]);
    for (size_t i = 1; i < vocab_size; i++) {
        const float val = oracle_h2f(logits[i]);
        if (val > best_val) { best_val = val; best_idx = i; }
    }
    return (uint32_t)best_idx;
}

/* ---------------------------------------------------------------------------------------
 * Stochastic branch of cactus_sample_f16: cactus/kernel/kernel_nn.cpp:468-651.
 * PARITY UNPINNED beyond reading: the reference has no known-answer test for it.  Two points
 * are implementation-defined in the reference and fixed here: std::sort's order of equal
 * logits in the top-p pass (:546; here equal values keep ascending index order) and
 * std::exp(float) (here this host's libm expf).
 *
 * history / hist_count: the reference's process-wide `token_history` (:498-500), passed in and
 * updated in place (at most 128 entries, oldest first). */
static int sample_desc_cmp(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}
typedef struct { float v; uint32_t idx; } sample_pair;
static int sample_pair_cmp(const void* a, const void* b) {
    const sample_pair* x = (const sample_pair*)a; const sample_pair* y = (const sample_pair*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static void sample_push(uint32_t* history, size_t* hist_count, uint32_t tok) {   /* :627-630 */
    if (!history) return;
    if (*hist_count >= 128) {
        memmove(history, history + 1, 127 * sizeof(uint32_t));
        history[127] = tok;
        *hist_count = 128;
    } else {
        history[(*hist_count)++] = tok;
    }
}

uint32_t oracle_sample_f16(const uint16_t* logits, size_t vocab_size, float temperature, float top_p,
                           size_t top_k, float sample, uint32_t* history, size_t* hist_count) {
    if (temperature == 0.0f && top_p <= 0.0f && top_k == 0) return oracle_sample_greedy_f16(logits, vocab_size);
    const float NEG_INF = -INFINITY;
    float* f = (float*)malloc((vocab_size + 1) * sizeof(float));   /* fp16 values held as floats */
    /* :469-496 -- the bias written at :473-480 is overwritten by both branches of :482-496 */
    if (temperature > 0) {
        const float inv_temp = oracle_h2f(oracle_f2h(1.0f / temperature));
        for (size_t i = 0; i < vocab_size; i++) f[i] = oracle_h2f(oracle_f2h(oracle_h2f(logits[i]) * inv_temp));
    } else {
        for (size_t i = 0; i < vocab_size; i++) f[i] = oracle_h2f(logits[i]);
    }
    /* :502-513 repetition penalty, once per occurrence in the history */
    if (history && *hist_count > 0) {
        const float pinv = oracle_h2f(oracle_f2h(1.0f / 1.1f)), pen = oracle_h2f(oracle_f2h(1.1f));
        for (size_t h = 0; h < *hist_count; h++) {
            const uint32_t tok = history[h];
            if (tok < vocab_size) f[tok] = oracle_h2f(oracle_f2h(f[tok] > 0.0f ? f[tok] * pinv : f[tok] * pen));
        }
    }
    /* :515-535 top-k: everything below the k-th largest value becomes -inf */
    if (top_k > 0 && top_k < vocab_size) {
        float* s = (float*)malloc(vocab_size * sizeof(float));
        memcpy(s, f, vocab_size * sizeof(float));
        qsort(s, vocab_size, sizeof(float), sample_desc_cmp);
        const float kth = s[top_k - 1];
        free(s);
        for (size_t i = 0; i < vocab_size; i++) if (f[i] < kth) f[i] = NEG_INF;
    }
    /* :537-588 top-p over the survivors in descending order */
    if (top_p > 0.0f && top_p < 1.0f) {
        sample_pair* sp = (sample_pair*)malloc((vocab_size + 1) * sizeof(sample_pair));
        size_t n = 0;
        for (size_t i = 0; i < vocab_size; i++) if (f[i] != NEG_INF) { sp[n].v = f[i]; sp[n].idx = (uint32_t)i; n++; }
        qsort(sp, n, sizeof(sample_pair), sample_pair_cmp);
        const float mx = n ? sp[0].v : 0.0f;
        float* pr = (float*)malloc((n + 1) * sizeof(float));
        float sum = 0.0f;
        for (size_t i = 0; i < n; i++) { pr[i] = expf(sp[i].v - mx); sum += pr[i]; }
        for (size_t i = 0; i < n; i++) pr[i] /= sum;
        float cum = 0.0f;
        int reached = 0;
        for (size_t i = 0; i < n; i++) {
            cum += pr[i];
            if (cum > top_p && i > 0) reached = 1;
            if (reached) f[sp[i].idx] = NEG_INF;
        }
        free(pr);
        free(sp);
    }
    /* :590-615 softmax over what is left */
    uint32_t result = 0;
    int push = 0;
    float mx = vocab_size ? f[0] : NEG_INF;
    for (size_t i = 1; i < vocab_size; i++) if (mx < f[i]) mx = f[i];
    if (vocab_size == 0 || mx == NEG_INF) { free(f); return 0; }
    float* probs = (float*)malloc(vocab_size * sizeof(float));
    float sum = 0.0f;
    for (size_t i = 0; i < vocab_size; i++) {
        if (f[i] == NEG_INF) probs[i] = 0.0f;
        else { probs[i] = expf(f[i] - mx); sum += probs[i]; }
    }
    if (sum == 0.0f) { free(probs); free(f); return 0; }
    for (size_t i = 0; i < vocab_size; i++) probs[i] /= sum;
    /* :617-650 inverse-CDF draw, then the two fallbacks */
    float cum = 0.0f;
    size_t i = 0;
    for (; i < vocab_size; i++) {
        cum += probs[i];
        if (cum >= sample) { result = (uint32_t)i; push = 1; break; }
    }
    if (!push) {
        for (size_t j = vocab_size; j > 0; j--) if (probs[j - 1] > 0.0f) { result = (uint32_t)(j - 1); push = 1; break; }
    }
    if (!push) { result = 0; push = 1; }
    sample_push(history, hist_count, result);
    free(probs);
    free(f);
    return result;
}

/* The draw of kernel_nn.cpp:617-620 for a given nonzero seed: std::mt19937(seed), then
 * std::uniform_real_distribution<float>(0, 1) as libstdc++ implements it (one 32-bit output,
 * float(u) / 2^32, a result of 1 replaced by the float below 1). */
float oracle_sample_uniform(uint32_t seed) {
    uint32_t mt[624];
    mt[0] = seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    uint32_t y = (mt[0] & 0x80000000u) | (mt[1] & 0x7FFFFFFFu);
    y = mt[397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    float r = (float)y / 4294967296.0f;
    if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
    return r;
}

/* Normalised entropy of the last logits row: cactus/engine/engine_model.cpp:255-290 */
float oracle_logits_entropy_f16(const uint16_t* logits, size_t vocab_size) {
    float mx = oracle_h2f(logits[0]);
    for (size_t i = 1; i < vocab_size; i++) { const float v = oracle_h2f(logits[i]); if (mx < v) mx = v; }
    double sum_exp = 0.0;
    for (size_t i = 0; i < vocab_size; i++) sum_exp += exp((double)(oracle_h2f(logits[i]) - mx));
    const double lse = (double)mx + log(sum_exp);
    double entropy = 0.0;
    for (size_t i = 0; i < vocab_size; i++) {
        const double lp = (double)oracle_h2f(logits[i]) - lse;
        const double p = exp(lp);
        if (p > 1e-10) entropy -= p * lp;
    }
    return (float)(entropy / log((double)vocab_size));
}

/* ------------------------------------------------------------------------ */
/* cactus_conv1d_causal_depthwise_f16: cactus/kernel/kernel_conv.cpp:15-110  */
/*   wrev[k] = float(W[c][K-1-k]) (:28-30).  Whole groups of 8 taps: four    */
/*   fp32 lanes, lane j takes taps k+j then k+4+j with vfmaq_f32, a tap      */
/*   before the sequence enters as 0 (:39-94); lanes summed by vaddvq_f32 =  */
/*   (l0+l1)+(l2+l3) (:96-97); the last K % 8 taps are `acc += w * x`        */
/*   statements, skipped when before the sequence (:99-104); one rounding to */
/*   fp16 (:106-107).  The two rows of a pass (t0, t1) are independent.      */
/* ------------------------------------------------------------------------ */
void oracle_conv1d_causal_depthwise_f16(const uint16_t* input, const uint16_t* weight, uint16_t* output,
                                        size_t N, size_t L, size_t C, size_t K, size_t dilation) {
    float* wrev = (float*)malloc((K ? K : 1) * sizeof(float));
    for (size_t n = 0; n < N; n++) {
        const uint16_t* Xb = input + n * L * C;
        uint16_t* Yb = output + n * L * C;
        for (size_t c = 0; c < C; c++) {
            for (size_t k = 0; k < K; k++) wrev[k] = oracle_h2f(weight[c * K + (K - 1 - k)]);
            for (size_t t = 0; t < L; t++) {
                float lane[4] = {0.f, 0.f, 0.f, 0.f};
                size_t k = 0;
                for (; k + 8 <= K; k += 8) {
                    for (size_t half = 0; half < 2; half++) {
                        for (size_t j = 0; j < 4; j++) {
                            const size_t kk = k + 4 * half + j, back = kk * dilation;
                            const float x = back <= t ? oracle_h2f(Xb[(t - back) * C + c]) : 0.f;
                            lane[j] = fmaf(x, wrev[kk], lane[j]);
                        }
                    }
                }
                float acc = (lane[0] + lane[1]) + (lane[2] + lane[3]);
                for (; k < K; k++) {
                    const size_t back = k * dilation;
                    if (back <= t) acc = mad_stmt(wrev[k], oracle_h2f(Xb[(t - back) * C + c]), acc);
                }
                Yb[t * C + c] = oracle_f2h(acc);
            }
        }
    }
    free(wrev);
}

/* ------------------------------------------------------------------------ */
/* LFM2Model::build_conv1d between its two matmuls                           */
/* (cactus/models/model_lfm2.cpp:136-204) with ConvCache                     */
/* (cactus/engine/engine_cache.cpp:494-597), one sequence:                   */
/*   in_proj [L, 3, C] -> B, Cg, X (:136-145); Bx = B * X (:146);            */
/*   the cache's rows, oldest first -- the segment from head to the end,     */
/*   then the one from 0 to head once the buffer is full, rows 0..count-1    */
/*   before that (get_window, engine_cache.cpp:511-549; :167-190) -- are     */
/*   concatenated in front of Bx; causal depthwise conv with dilation 1      */
/*   over the concatenation (:196-199); the last L rows (:200-203);          */
/*   gated = Cg * y (:204).  Then ConvCache::update copies the last          */
/*   min(L, W) rows of Bx into the circular buffer (:585-596).               */
/*   state [W, C], *head, *count are the buffer (zeros, 0, 0 at the start).  */
/* ------------------------------------------------------------------------ */
void oracle_lfm2_conv_gate_f16(const uint16_t* in_proj, const uint16_t* weight, uint16_t* state,
                               size_t* head, size_t* count, uint16_t* gated,
                               size_t L, size_t C, size_t K, size_t W) {
    size_t held = 0;
    if (W > 0 && *count > 0) held = *count < W ? *count : W;
    const size_t total = held + L;
    uint16_t* cat = (uint16_t*)malloc((total ? total : 1) * C * sizeof(uint16_t));
    uint16_t* y = (uint16_t*)malloc((total ? total : 1) * C * sizeof(uint16_t));
    size_t row = 0;
    if (held > 0) {
        if (*count < W) {
            memcpy(cat, state, held * C * sizeof(uint16_t));
            row = held;
        } else {
            const size_t len2 = W - *head;
            memcpy(cat, state + *head * C, len2 * C * sizeof(uint16_t));
            memcpy(cat + len2 * C, state, *head * C * sizeof(uint16_t));
            row = W;
        }
    }
    uint16_t* bx = cat + row * C;
    for (size_t t = 0; t < L; t++)
        for (size_t c = 0; c < C; c++)
            bx[t * C + c] = h_mul(in_proj[(t * 3 + 0) * C + c], in_proj[(t * 3 + 2) * C + c]);
    oracle_conv1d_causal_depthwise_f16(cat, weight, y, 1, total, C, K, 1);
    const size_t start = total > L ? total - L : 0;
    for (size_t t = 0; t < L; t++)
        for (size_t c = 0; c < C; c++)
            gated[t * C + c] = h_mul(in_proj[(t * 3 + 1) * C + c], y[(start + t) * C + c]);
    if (W > 0) {
        const size_t copy_rows = L < W ? L : W, start_row = L > W ? L - W : 0;
        for (size_t i = 0; i < copy_rows; i++) {
            memcpy(state + *head * C, bx + (start_row + i) * C, C * sizeof(uint16_t));
            *head = (*head + 1) % W;
            if (*count < W) ++*count;
        }
    }
    free(cat);
    free(y);
}

/* cactus_scalar_op_f16, arithmetic ops: cactus/kernel/kernel_scalar.cpp:7-154.  The scalar is
 * cast to fp16 once (:8); vector body and tail both apply one fp16 operation per element.
 * op: 0 add, 1 subtract, 2 multiply, 3 divide. */
void oracle_scalar_op_f16(const uint16_t* in, uint16_t* out, size_t n, float scalar, int op) {
    const uint16_t sb = oracle_f2h(scalar);
    for (size_t i = 0; i < n; i++) oracle_binary_f16(op, in + i, &sb, out + i, 1);
}
