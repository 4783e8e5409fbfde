// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));

// binary16 bit pattern <-> float (exact / round-to-nearest-even).
__device__ __forceinline__ float h2f(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ uint16_t f2h(float f) {
    return __half_as_ushort(__float2half_rn(f));
}

// half(a * b) with TWO roundings (fp32 product, then fp32 -> fp16), as the reference's
// `(__fp16)(x * y)` evaluates (kernel_gemm.cpp:322).  Without the empty asm hipcc folds the
// multiply into v_fma_mixlo_f16, which rounds the exact product once and flips fp32->fp16 ties.
__device__ __forceinline__ uint16_t f2h_mul(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return f2h(p);
}

// Full-wave (64 lane) butterfly reductions; every lane ends with the result.
// max is exact and order-free, so this one runs on DPP (no LDS round trips): quad swaps, half
// row mirror and row mirror leave every lane with its 16-lane row's maximum, then the four rows
// are combined through scalar reads.  Call with all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Slot of the token with absolute index `abs` in a cache of `window` slots whose first `sink` slots
// keep the first tokens (KVCache, engine_cache.cpp:268-329: first `sink` tokens + the most recent
// `window - sink`): tokens fill slots 0..window-1 in order, then each new token overwrites the oldest
// of the recent ones, which is the slot its index maps to going round the recent region.  The cache
// then holds the tokens the reference's sliding window holds, rotated.  window == 0: no window.
__host__ __device__ __forceinline__ long ring_slot(long abs, long window, long sink) {
    if (window <= 0 || abs < window) return abs;
    return sink + (abs - window) % (window - sink);
}

// Round to nearest, ties to even (NEON vcvtnq_s32_f32), as an int.
__device__ __forceinline__ int rne_i32(float v) { return (int)rintf(v); }

__device__ __forceinline__ int clamp_i8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }

// max |x| over 8 halves held in four 32-bit words
__device__ __forceinline__ float absmax8_bits(v4i v) {
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t w = (uint32_t)v[i];
        m = fmaxf(m, fabsf(h2f((uint16_t)(w & 0xFFFFu))));
        m = fmaxf(m, fabsf(h2f((uint16_t)(w >> 16))));
    }
    return m;
}

// 8 halves -> 8 int8: q = sat8(rne(clamp(x * inv, -128, 127)))  (kernel_quants.cpp:158-184)
__device__ __forceinline__ uint2 quant8_bits(v4i v, float inv) {
    uint32_t out[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t w = (uint32_t)v[j >> 1];
        const uint16_t hb = (uint16_t)((j & 1) ? (w >> 16) : (w & 0xFFFFu));
        float q = h2f(hb) * inv;
        q = fmaxf(fminf(q, 127.0f), -128.0f);
        const int qi = clamp_i8(rne_i32(q));
        out[j >> 2] |= (uint32_t)(qi & 0xFF) << (8 * (j & 3));
    }
    return make_uint2(out[0], out[1]);
}
