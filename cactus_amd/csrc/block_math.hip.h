// Device arithmetic of the decoder-block ops shared by the element-wise kernels and by the matmul
// prologues that produce their activation row on the fly: the reference's RMS-norm accumulation
// order, its fp16 output products, the polynomial exp / rational tanh of its 8-wide paths.
#pragma once

#include "common.hip.h"

namespace cactus_hip {
namespace {

typedef _Float16 h16;

__device__ __forceinline__ h16 bits2h(uint16_t b) { return __builtin_bit_cast(h16, b); }
__device__ __forceinline__ uint16_t h2bits(h16 h) { return __builtin_bit_cast(uint16_t, h); }

// 1 / rms of one row, computed by the 16 lanes [base, base + 16) of a wave; `elem(i)` returns the
// binary16 bits of element i.  Loads run 8 ahead of the fma chain.
template <typename Elem>
__device__ __forceinline__ float row_inv_rms(Elem elem, int dims, float eps, int l, int base) {
    const int simd_end = dims & ~7;
    float acc = 0.0f;
    int i = l;
    for (; i + 7 * 16 < simd_end; i += 8 * 16) {
        uint16_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = elem(i + 16 * u);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float f = h2f(v[u]);
            acc = fmaf(f, f, acc);
        }
    }
    for (; i < simd_end; i += 16) {
        const float f = h2f(elem(i));
        acc = fmaf(f, f, acc);
    }
    const float a0 = __shfl(acc, base + (l & 3)), a1 = __shfl(acc, base + 4 + (l & 3));
    const float a2 = __shfl(acc, base + 8 + (l & 3)), a3 = __shfl(acc, base + 12 + (l & 3));
    const float t = ((a0 + a1) + a2) + a3;
    const float t0 = __shfl(t, base), t1 = __shfl(t, base + 1), t2 = __shfl(t, base + 2), t3 = __shfl(t, base + 3);
    float ss = (t0 + t1) + (t2 + t3);
    for (int k = simd_end; k < dims; k++) {
        const float f = h2f(elem(k));
        ss = fmaf(f, f, ss);
    }
    const float rms = __builtin_sqrtf(ss / (float)dims + eps);
    return 1.0f / rms;
}

// The same value from a row stored chain by chain: `chain` points at this lane's elements l, l + 16, l + 32 ... of
// the row, contiguous (dims % 128 == 0, 16-byte aligned): eight of them per LDS read instead of one.
__device__ __forceinline__ float row_inv_rms_chain(const uint16_t* chain, int dims, float eps, int l, int base) {
    float acc = 0.0f;
    const v4i* cp = reinterpret_cast<const v4i*>(chain);
    for (int b = 0; b < dims / 128; b++) {
        const v4i v = cp[b];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float lo = h2f((uint16_t)((uint32_t)v[e] & 0xFFFFu)), hi = h2f((uint16_t)((uint32_t)v[e] >> 16));
            acc = fmaf(lo, lo, acc);
            acc = fmaf(hi, hi, acc);
        }
    }
    const float a0 = __shfl(acc, base + (l & 3)), a1 = __shfl(acc, base + 4 + (l & 3));
    const float a2 = __shfl(acc, base + 8 + (l & 3)), a3 = __shfl(acc, base + 12 + (l & 3));
    const float t = ((a0 + a1) + a2) + a3;
    const float t0 = __shfl(t, base), t1 = __shfl(t, base + 1), t2 = __shfl(t, base + 2), t3 = __shfl(t, base + 3);
    const float ss = (t0 + t1) + (t2 + t3);
    const float rms = __builtin_sqrtf(ss / (float)dims + eps);
    return 1.0f / rms;
}

// normalized element i: fp16 products in the vector part, one fp32 expression in the tail
__device__ __forceinline__ uint16_t rms_out(uint16_t xb, uint16_t wb, float inv, h16 inv_h, bool vector_part) {
    if (vector_part) {
        const h16 p = bits2h(xb) * inv_h;
        return h2bits(p * bits2h(wb));
    }
    return f2h_mul(h2f(xb) * inv, h2f(wb));   // all fp32, one conversion at the end (kernel_nn.cpp tail)
}

// one row of D halves in LDS -> int8 + one scale per group, by the 16 lanes of a group; lane l
// owns elements 8l..8l+7 of each 128-wide slice of a group (kernel_quants.cpp:223-293)
__device__ __forceinline__ void quant_row_groups(const uint16_t* row, int D, int gs, int8_t* dst, float* sc, int l) {
    const int G = (D + gs - 1) / gs;
    for (int g = 0; g < G; g++) {
        const int start = g * gs, cnt = min(gs, D - start), cnt8 = cnt & ~7;
        float m = 0.0f;
        for (int k0 = l * 8; k0 < cnt; k0 += 128)
            for (int k = k0; k < k0 + 8 && k < cnt; k++) m = fmaxf(m, fabsf(h2f(row[start + k])));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
        float scale = m / 127.0f;
        if (scale < 1e-10f) scale = 1e-10f;
        const float inv = 1.0f / scale;
        if (l == 0) sc[g] = scale;
        for (int k0 = l * 8; k0 < cnt; k0 += 128)
            for (int k = k0; k < k0 + 8 && k < cnt; k++) {
                const float v = h2f(row[start + k]) * inv;
                dst[start + k] = (int8_t)((k < cnt8) ? clamp_i8(rne_i32(v)) : clamp_i8((int)roundf(v)));
            }
    }
}

// RoPE of one element pair with the arithmetic of rope_kernel's 8-wide part (kernel_attention.cpp:612-676): fp16
// products, a fused subtract / add
__device__ __forceinline__ void rope_pair(uint16_t cb, uint16_t sb, uint16_t x1b, uint16_t x2b, uint16_t& o1, uint16_t& o2) {
    const h16 c = bits2h(cb), sn = bits2h(sb), x1 = bits2h(x1b), x2 = bits2h(x2b);
    const h16 p1 = x1 * c, p2 = x2 * c;
    o1 = h2bits(__builtin_fmaf16(-x2, sn, p1));
    o2 = h2bits(__builtin_fmaf16(x1, sn, p2));
}

// polynomial e^x of the vector paths (kernel_utils.h:108-141)
__device__ __forceinline__ float fast_exp(float x) {
    x = fmaxf(x, -87.0f);
    x = fminf(x, 87.0f);
    const float z = x * 1.4426950408889634f;
    int zi = (int)z;   // toward zero
    float zf = z - (float)zi;
    if (zf < 0.0f) { zi -= 1; zf += 1.0f; }
    const float zl = zf * 0.6931471805599453f;
    float p = 0.009618129842071803f;
    p = fmaf(p, zl, 0.05550410866482158f);
    p = fmaf(p, zl, 0.2402265069591007f);
    p = fmaf(p, zl, 0.6931471805599453f);
    p = fmaf(p, zl, 1.0f);
    return p * __builtin_bit_cast(float, (uint32_t)(zi + 127) << 23);
}

// rational tanh of the vector paths (kernel_utils.h:143-163)
__device__ __forceinline__ float fast_tanh(float x) {
    const float x2 = x * x;
    const float num = 27.0f + x2;
    const float den = fmaf(9.0f, x2, 27.0f);
    float r = x * (num / den);
    if (x > 4.5f) r = 1.0f;
    if (x < -4.5f) r = -1.0f;
    return r;
}

enum { OP_SILU = 0, OP_GELU = 1 };

template <int OP>
__device__ __forceinline__ uint16_t unary_one(uint16_t hb, bool vector_part) {
    const float x = h2f(hb);
    if (OP == OP_SILU) {
        const float e = vector_part ? fast_exp(-x) : expf(-x);
        const float sig = 1.0f / (1.0f + e);
        return f2h_mul(x, sig);   // fp32 product, then fp32 -> fp16: two roundings (vmulq_f32 + vcvt_f16_f32)
    } else {
        if (vector_part) {
            const float x3 = (x * x) * x;
            float inner = fmaf(0.044715f, x3, x);
            inner = 0.7978845608028654f * inner;
            const float th = fast_tanh(inner);
            const float hx = 0.5f * x, ot = 1.0f + th;
            return f2h_mul(hx, ot);
        }
        const float c1 = 0.044715f * x, c2 = c1 * x, x3 = c2 * x;
        const float sum = x + x3;
        const float inner = 0.7978845608028654f * sum;
        const float hx = 0.5f * x, ot = 1.0f + tanhf(inner);
        return f2h_mul(hx, ot);
    }
}


}  // namespace
}  // namespace cactus_hip
