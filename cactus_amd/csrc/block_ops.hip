// The ops between the hot kernels of a decoder block: RMS norm, RoPE, SiLU, GELU, the
// element-wise fp16 binaries, clipped add and the embedding lookup.
//
// Roofline: HBM for all of them (a few bytes of traffic per element, a handful of ALU ops);
// algorithmic bytes per element: rms_norm 2 in + 2 out (+ 2 weight, cache resident), rope 4,
// unary 4, binary 6, embedding INT8 1 in + 2 out.
//
// Every kernel reproduces the reference's floating-point evaluation order, including where it
// switches between an 8-wide "vector" formula and a scalar tail:
//   rms_norm   cactus/kernel/kernel_attention.cpp:484-563
//   rope       cactus/kernel/kernel_attention.cpp:614-676 (tables :594-604, built on the host)
//   silu/gelu  cactus/kernel/kernel_nn.cpp:12-91, polynomials cactus/kernel/kernel_utils.h:108-163
//   binaries   cactus/kernel/kernel_blas.cpp:187-343
//   embedding  cactus/graph/graph_ops_tensor.cpp:226-320
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

namespace {

typedef _Float16 h16;

__device__ __forceinline__ h16 bits2h(uint16_t b) { return __builtin_bit_cast(h16, b); }
__device__ __forceinline__ uint16_t h2bits(h16 h) { return __builtin_bit_cast(uint16_t, h); }

// ---------------------------------------------------------------------------------------
// RMS norm.  16 lanes per row, 16 rows per 256-thread workgroup.  Lane l owns the reference's
// fp32 accumulator l: element i < simd_end (dims rounded down to 8) is folded into accumulator
// i % 16 with an fma, in increasing i (four float32x4 registers over a 16-element tile, the 8-wide
// remainder into the first two).  The registers are added ((v0+v1)+v2)+v3 lane-wise, the four
// lanes as (t0+t1)+(t2+t3), then the scalar tail elements are fused in.  The vector part of the
// output multiplies in fp16, (x * half(inv)) * w with both products rounded; the tail computes
// x*inv*w in fp32 and rounds once.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rms_norm_kernel(const uint16_t* __restrict__ x,
                                                       const uint16_t* __restrict__ w,
                                                       uint16_t* __restrict__ y, size_t rows, int dims,
                                                       float eps) {
    const int l = threadIdx.x & 15;
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row < rows;
    const size_t r = live ? row : rows - 1;   // dead groups recompute the last row and store nothing
    const uint16_t* xr = x + r * dims;
    const int simd_end = dims & ~7;
    float acc = 0.0f;
    for (int i = l; i < simd_end; i += 16) {
        const float v = h2f(xr[i]);
        acc = fmaf(v, v, acc);
    }
    const int base = (threadIdx.x & 63) & ~15;
    const float a0 = __shfl(acc, base + (l & 3)), a1 = __shfl(acc, base + 4 + (l & 3));
    const float a2 = __shfl(acc, base + 8 + (l & 3)), a3 = __shfl(acc, base + 12 + (l & 3));
    const float t = ((a0 + a1) + a2) + a3;
    const float t0 = __shfl(t, base), t1 = __shfl(t, base + 1), t2 = __shfl(t, base + 2), t3 = __shfl(t, base + 3);
    float ss = (t0 + t1) + (t2 + t3);
    for (int i = simd_end; i < dims; i++) {
        const float v = h2f(xr[i]);
        ss = fmaf(v, v, ss);
    }
    const float rms = __builtin_sqrtf(ss / (float)dims + eps);
    const float inv = 1.0f / rms;
    if (!live) return;
    const h16 inv_h = (h16)inv;
    uint16_t* yr = y + r * dims;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(xr) | reinterpret_cast<uintptr_t>(yr) |
                          reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    if (vec_ok) {
        for (int c = l; c < simd_end / 8; c += 16) {
            const v4i xv = *reinterpret_cast<const v4i*>(xr + c * 8);
            const v4i wv = *reinterpret_cast<const v4i*>(w + c * 8);
            v4i o;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t res = 0;
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    const h16 xe = bits2h((uint16_t)((uint32_t)xv[e] >> (16 * hh)));
                    const h16 we = bits2h((uint16_t)((uint32_t)wv[e] >> (16 * hh)));
                    const h16 p = xe * inv_h;
                    res |= (uint32_t)h2bits(p * we) << (16 * hh);
                }
                o[e] = (int)res;
            }
            *reinterpret_cast<v4i*>(yr + c * 8) = o;
        }
    } else {
        for (int i = l; i < simd_end; i += 16) {
            const h16 p = bits2h(xr[i]) * inv_h;
            yr[i] = h2bits(p * bits2h(w[i]));
        }
    }
    for (int i = simd_end + l; i < dims; i += 16) yr[i] = f2h(h2f(xr[i]) * inv * h2f(w[i]));
}

// ---------------------------------------------------------------------------------------
// RoPE, half-split rotation on [batch, seq, heads, head_dim].  One thread per (token, head, i).
// Vector part (i < half_dim rounded down to 8), all fp16: first = (x1*cos) - x2*sin with the
// product rounded and the subtraction fused, second = (x2*cos) + x1*sin likewise.  The tail
// evaluates in fp32 (products of halves are exact there) and rounds the result to half.
// cos_t / sin_t: [seq][half_dim] rows of the host-built tables, already offset by start_pos.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                   const uint16_t* __restrict__ cos_t,
                                                   const uint16_t* __restrict__ sin_t, size_t total,
                                                   int seq_len, int num_heads, int half_dim) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // over batch*seq*heads*half_dim
    if (idx >= total) return;
    const int i = (int)(idx % half_dim);
    const size_t th = idx / half_dim;                             // (batch*seq + s) * heads + h
    const int s = (int)((th / num_heads) % seq_len);
    const size_t off = th * (size_t)(2 * half_dim);
    const uint16_t cb = cos_t[(size_t)s * half_dim + i], sb = sin_t[(size_t)s * half_dim + i];
    const uint16_t x1b = in[off + i], x2b = in[off + half_dim + i];
    if (i < (half_dim & ~7)) {
        const h16 c = bits2h(cb), sn = bits2h(sb), x1 = bits2h(x1b), x2 = bits2h(x2b);
        const h16 p1 = x1 * c, p2 = x2 * c;
        out[off + i] = h2bits(__builtin_fmaf16(-x2, sn, p1));
        out[off + half_dim + i] = h2bits(__builtin_fmaf16(x1, sn, p2));
    } else {
        const float c = h2f(cb), sn = h2f(sb), x1 = h2f(x1b), x2 = h2f(x2b);
        const float a = x1 * c, b = x2 * sn, d = x2 * c, e = x1 * sn;
        out[off + i] = f2h(a - b);
        out[off + half_dim + i] = f2h(d + e);
    }
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
        return f2h(x * sig);
    } else {
        if (vector_part) {
            const float x3 = (x * x) * x;
            float inner = fmaf(0.044715f, x3, x);
            inner = 0.7978845608028654f * inner;
            const float th = fast_tanh(inner);
            const float hx = 0.5f * x, ot = 1.0f + th;
            return f2h(hx * ot);
        }
        const float c1 = 0.044715f * x, c2 = c1 * x, x3 = c2 * x;
        const float sum = x + x3;
        const float inner = 0.7978845608028654f * sum;
        const float hx = 0.5f * x, ot = 1.0f + tanhf(inner);
        return f2h(hx * ot);
    }
}

// The call is treated as one contiguous piece: whole groups of 8 from the start take the vector
// formula, the last n % 8 elements the scalar libm one (what the reference does whenever it runs
// the call on one thread).
template <int OP>
__global__ __launch_bounds__(256) void unary_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                    size_t n) {
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;   // chunk of 8 halves
    const size_t vend = n & ~(size_t)7;
    if (c * 8 >= n) return;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (c * 8 < vend && vec_ok) {
        const v4i xv = *reinterpret_cast<const v4i*>(in + c * 8);
        v4i o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t lo = unary_one<OP>((uint16_t)((uint32_t)xv[e] & 0xFFFFu), true);
            const uint32_t hi = unary_one<OP>((uint16_t)((uint32_t)xv[e] >> 16), true);
            o[e] = (int)(lo | (hi << 16));
        }
        *reinterpret_cast<v4i*>(out + c * 8) = o;
    } else {
        for (size_t i = c * 8; i < c * 8 + 8 && i < n; i++) out[i] = unary_one<OP>(in[i], i < vend);
    }
}

enum { BIN_ADD = 0, BIN_SUB = 1, BIN_MUL = 2, BIN_DIV = 3, BIN_ADD_CLIPPED = 4 };

template <int OP>
__device__ __forceinline__ uint16_t binary_one(uint16_t ab, uint16_t bb) {
    if (OP == BIN_ADD_CLIPPED) {
        float r = h2f(ab) + h2f(bb);
        r = fminf(fmaxf(r, -65500.0f), 65500.0f);
        return f2h(r);
    }
    if (OP == BIN_DIV) return f2h(h2f(ab) / h2f(bb));   // fp32 quotient then half = the fp16 quotient
    const h16 a = bits2h(ab), b = bits2h(bb);
    return h2bits(OP == BIN_ADD ? a + b : (OP == BIN_SUB ? a - b : a * b));
}

template <int OP>
__global__ __launch_bounds__(256) void binary_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                     uint16_t* __restrict__ out, size_t n) {
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c * 8 >= n) return;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                          reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (c * 8 + 8 <= n && vec_ok) {
        const v4i av = *reinterpret_cast<const v4i*>(a + c * 8);
        const v4i bv = *reinterpret_cast<const v4i*>(b + c * 8);
        v4i o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t lo = binary_one<OP>((uint16_t)((uint32_t)av[e] & 0xFFFFu), (uint16_t)((uint32_t)bv[e] & 0xFFFFu));
            const uint32_t hi = binary_one<OP>((uint16_t)((uint32_t)av[e] >> 16), (uint16_t)((uint32_t)bv[e] >> 16));
            o[e] = (int)(lo | (hi << 16));
        }
        *reinterpret_cast<v4i*>(out + c * 8) = o;
    } else {
        for (size_t i = c * 8; i < c * 8 + 8 && i < n; i++) out[i] = binary_one<OP>(a[i], b[i]);
    }
}

// Embedding rows.  INT8 with per-group fp16 scales: half(float(q) * float(scale)); FP16: copy.
// An index outside [0, vocab) writes nothing and raises *bad (checked by the caller).
__global__ __launch_bounds__(256) void embedding_int8_kernel(const int8_t* __restrict__ emb,
                                                             const uint16_t* __restrict__ scales,
                                                             const int32_t* __restrict__ indices,
                                                             uint16_t* __restrict__ out, size_t num_indices,
                                                             size_t vocab, int hidden, int group_size,
                                                             int* __restrict__ bad) {
    const size_t i = blockIdx.x;
    if (i >= num_indices) return;
    const long idx = indices[i];
    if (idx < 0 || (size_t)idx >= vocab) {
        if (threadIdx.x == 0 && bad != nullptr) atomicExch(bad, 1);
        return;
    }
    const int num_groups = hidden / group_size;
    const int8_t* row = emb + (size_t)idx * hidden;
    for (int k = threadIdx.x; k < num_groups * group_size; k += 256) {
        const float sc = h2f(scales[(size_t)idx * num_groups + k / group_size]);
        out[i * hidden + k] = f2h((float)row[k] * sc);
    }
}

__global__ __launch_bounds__(256) void embedding_f16_kernel(const uint16_t* __restrict__ emb,
                                                            const int32_t* __restrict__ indices,
                                                            uint16_t* __restrict__ out, size_t num_indices,
                                                            size_t vocab, int hidden, int* __restrict__ bad) {
    const size_t i = blockIdx.x;
    if (i >= num_indices) return;
    const long idx = indices[i];
    if (idx < 0 || (size_t)idx >= vocab) {
        if (threadIdx.x == 0 && bad != nullptr) atomicExch(bad, 1);
        return;
    }
    for (int k = threadIdx.x; k < hidden; k += 256) out[i * hidden + k] = emb[(size_t)idx * hidden + k];
}

// Embedding indices as the graph holds them (FP32 or INT8 values) -> int32; a value that is not
// a valid size_t (negative, NaN) becomes -1 so that the lookup reports it.
__global__ __launch_bounds__(256) void indices_to_i32_kernel(const void* __restrict__ src, int is_f32,
                                                             int32_t* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (is_f32) {
        const float v = static_cast<const float*>(src)[i];
        dst[i] = (v >= 0.0f && v < 2147483648.0f) ? (int32_t)v : -1;
    } else {
        dst[i] = static_cast<const int8_t*>(src)[i];   // negative stays negative = out of range
    }
}

}  // namespace

hipError_t launch_indices_to_i32(const void* src, bool is_f32, int32_t* dst, size_t n, hipStream_t st) {
    indices_to_i32_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(src, is_f32 ? 1 : 0, dst, n);
    return hipGetLastError();
}

hipError_t launch_rms_norm(const uint16_t* x, const uint16_t* w, uint16_t* y, size_t rows, size_t dims, float eps,
                           hipStream_t st) {
    rms_norm_kernel<<<dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st>>>(x, w, y, rows, (int)dims, eps);
    return hipGetLastError();
}

hipError_t launch_rope(const uint16_t* in, uint16_t* out, const uint16_t* cos_t, const uint16_t* sin_t,
                       size_t batch, size_t seq, size_t heads, size_t head_dim, hipStream_t st) {
    const size_t half_dim = head_dim / 2;
    const size_t total = batch * seq * heads * half_dim;
    rope_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(in, out, cos_t, sin_t, total, (int)seq,
                                                                           (int)heads, (int)half_dim);
    return hipGetLastError();
}

hipError_t launch_unary(int op, const uint16_t* in, uint16_t* out, size_t n, hipStream_t st) {
    const dim3 grid((unsigned)(((n + 7) / 8 + 255) / 256)), block(256);
    if (op == OP_SILU) unary_kernel<OP_SILU><<<grid, block, 0, st>>>(in, out, n);
    else if (op == OP_GELU) unary_kernel<OP_GELU><<<grid, block, 0, st>>>(in, out, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_binary(int op, const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n, hipStream_t st) {
    const dim3 grid((unsigned)(((n + 7) / 8 + 255) / 256)), block(256);
    switch (op) {
        case BIN_ADD: binary_kernel<BIN_ADD><<<grid, block, 0, st>>>(a, b, out, n); break;
        case BIN_SUB: binary_kernel<BIN_SUB><<<grid, block, 0, st>>>(a, b, out, n); break;
        case BIN_MUL: binary_kernel<BIN_MUL><<<grid, block, 0, st>>>(a, b, out, n); break;
        case BIN_DIV: binary_kernel<BIN_DIV><<<grid, block, 0, st>>>(a, b, out, n); break;
        case BIN_ADD_CLIPPED: binary_kernel<BIN_ADD_CLIPPED><<<grid, block, 0, st>>>(a, b, out, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_embedding_int8(const int8_t* emb, const uint16_t* scales, const int32_t* indices, uint16_t* out,
                                 size_t num_indices, size_t vocab, size_t hidden, size_t group_size, int* bad,
                                 hipStream_t st) {
    embedding_int8_kernel<<<dim3((unsigned)num_indices), dim3(256), 0, st>>>(emb, scales, indices, out, num_indices,
                                                                            vocab, (int)hidden, (int)group_size, bad);
    return hipGetLastError();
}

hipError_t launch_embedding_f16(const uint16_t* emb, const int32_t* indices, uint16_t* out, size_t num_indices,
                                size_t vocab, size_t hidden, int* bad, hipStream_t st) {
    embedding_f16_kernel<<<dim3((unsigned)num_indices), dim3(256), 0, st>>>(emb, indices, out, num_indices, vocab,
                                                                           (int)hidden, bad);
    return hipGetLastError();
}

}  // namespace cactus_hip
