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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Round to nearest, ties to even (NEON vcvtnq_s32_f32), as an int.
__device__ __forceinline__ int rne_i32(float v) { return (int)rintf(v); }

__device__ __forceinline__ int clamp_i8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
