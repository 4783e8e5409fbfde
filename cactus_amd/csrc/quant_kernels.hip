// Quantization kernels: per-row activation quantization, per-(token,head,group)
// KV quantization, INT4 unpack.  All are HBM/L2-bound byte shuffles: 16-byte
// loads, 8-byte stores, shuffle reductions, no LDS tiling needed.
//
// Rounding rules restate the reference bit for bit:
//   cactus/kernel/kernel_quants.cpp:158-194  (fp16 -> int8: x*(1/scale), clamp, ties-to-even
//                                             for multiples of 8, ties-away for the tail)
//   cactus/kernel/kernel_quants.cpp:223-264  (KV group: no clamp before rounding, saturate after)
//   cactus/graph/graph_ops_nn.cpp:46-50      (scale = max(absmax/127, 1e-10))
#include "common.hip.h"
#include "kernels.h"

namespace cactus_hip {

__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = fmaxf(r, red[i]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ float absmax8(uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m = fmaxf(m, fabsf(h2f((uint16_t)(w[i] & 0xFFFFu))));
        m = fmaxf(m, fabsf(h2f((uint16_t)(w[i] >> 16))));
    }
    return m;
}

// One 256-thread block per row.  scales_in == nullptr: derive the row scale from
// the row's absmax and store it in scales_out[row]; otherwise use *scales_in for
// every row (cactus_fp16_to_int8 with a caller-supplied scale).
__global__ __launch_bounds__(256) void act_quant_rowwise_kernel(
    const uint16_t* __restrict__ src, int8_t* __restrict__ dst, float* __restrict__ scales_out,
    float fixed_scale, int use_fixed, size_t K) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const uint16_t* s = src + row * K;
    int8_t* d = dst + row * K;
    const size_t K8 = (K / 8) * 8;
    const bool vec = ((K & 7) == 0) && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 7) == 0);

    float scale;
    if (use_fixed) {
        scale = fixed_scale;
    } else {
        float m = 0.0f;
        if (vec) {
            for (size_t i = threadIdx.x * 8; i < K; i += 256 * 8)
                m = fmaxf(m, absmax8(*reinterpret_cast<const uint4*>(s + i)));
        } else {
            for (size_t i = threadIdx.x; i < K; i += 256) m = fmaxf(m, fabsf(h2f(s[i])));
        }
        m = block_max_256(m, red);
        scale = m / 127.0f;
        if (scale < 1e-10f) scale = 1e-10f;
        if (threadIdx.x == 0) scales_out[row] = scale;
    }
    const float inv = 1.0f / scale;

    if (vec) {
        for (size_t i = threadIdx.x * 8; i < K; i += 256 * 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(s + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t out[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint16_t hb = (uint16_t)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu));
                float q = h2f(hb) * inv;
                q = fmaxf(fminf(q, 127.0f), -128.0f);
                const int qi = clamp_i8(rne_i32(q));
                out[j >> 2] |= (uint32_t)(qi & 0xFF) << (8 * (j & 3));
            }
            *reinterpret_cast<uint2*>(d + i) = make_uint2(out[0], out[1]);
        }
    } else {
        for (size_t i = threadIdx.x; i < K; i += 256) {
            float q = h2f(s[i]) * inv;
            q = fmaxf(fminf(q, 127.0f), -128.0f);
            d[i] = (int8_t)((i < K8) ? clamp_i8(rne_i32(q)) : (int)roundf(q));
        }
    }
}

// max |x| over a flat array: per-block reduce then one integer atomicMax (non-negative
// floats order like their bit patterns).  *out must be zeroed before the launch.
__global__ __launch_bounds__(256) void max_abs_kernel(const uint16_t* __restrict__ src, size_t count,
                                                     float* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(h2f(src[i])));
    m = block_max_256(m, red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}

// One 16-lane subgroup per (token*head, group) item; lane j owns elements 8j..8j+7 of each
// 128-wide slice of the group.
// blockIdx.y = 1 takes the second (src, dst, scales) set: K and V of a step in one launch.
__device__ __forceinline__ void kv_quant_rows(
    const uint16_t* __restrict__ src, int8_t* __restrict__ dst, float* __restrict__ scales,
    size_t rows, size_t D, size_t gs, size_t G, const int32_t* __restrict__ slots, size_t heads,
    size_t cache_batch_stride, size_t scale_batch_stride, size_t max_slots,
    const int32_t* __restrict__ positions, size_t seq_len, int window, int sink) {
    const size_t item = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int j = threadIdx.x & 15;
    const bool live = item < rows * G;
    const size_t row = live ? item / G : 0, g = live ? item % G : 0;
    const size_t start = g * gs;
    const size_t cnt = live ? ((D - start < gs) ? D - start : gs) : 0;
    const size_t cnt8 = (cnt / 8) * 8;
    const uint16_t* s = src + row * D + start;
    // Destination: row for row, or (append form) into the slots of each sequence's cache; a slot
    // outside [0, max_slots) stores nothing.
    size_t doff = row * D + start, soff = row * G;
    bool place = true;
    if ((slots != nullptr || positions != nullptr) && live) {
        // source row ((b, s), h): token s of sequence b goes to slot slots[b] + s, or to the ring slot
        // of its absolute position positions[b] + s
        const size_t tok = row / heads, h = row - tok * heads, b = tok / seq_len, sq = tok - b * seq_len;
        const long first = positions ? positions[b] : slots[b];
        const long sl = positions ? ring_slot(first + (long)sq, window, sink) : first + (long)sq;
        place = first >= 0 && (size_t)sl < max_slots;
        const size_t slot = place ? (size_t)sl : 0;
        doff = b * cache_batch_stride + (slot * heads + h) * D + start;
        soff = b * scale_batch_stride + (slot * heads + h) * G;
    }
    int8_t* d = dst + doff;
    const bool vec = ((D & 7) == 0) && ((gs & 7) == 0) && ((((uintptr_t)src) & 15) == 0) &&
                     ((((uintptr_t)d) & 7) == 0);

    float m = 0.0f;
    for (size_t k0 = (size_t)j * 8; k0 < cnt; k0 += 128) {
        if (vec && k0 + 8 <= cnt) {
            m = fmaxf(m, absmax8(*reinterpret_cast<const uint4*>(s + k0)));
        } else {
            for (size_t k = k0; k < k0 + 8 && k < cnt; k++) m = fmaxf(m, fabsf(h2f(s[k])));
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
    float scale = m / 127.0f;
    if (scale < 1e-10f) scale = 1e-10f;
    const float inv = 1.0f / scale;
    if (live && place && j == 0) scales[soff + g] = scale;
    if (!place) return;

    for (size_t k0 = (size_t)j * 8; k0 < cnt; k0 += 128) {
        if (vec && k0 + 8 <= cnt) {
            const uint4 v = *reinterpret_cast<const uint4*>(s + k0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t out[2] = {0, 0};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint16_t hb = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xFFFFu));
                const int qi = clamp_i8(rne_i32(h2f(hb) * inv));
                out[e >> 2] |= (uint32_t)(qi & 0xFF) << (8 * (e & 3));
            }
            *reinterpret_cast<uint2*>(d + k0) = make_uint2(out[0], out[1]);
        } else {
            for (size_t k = k0; k < k0 + 8 && k < cnt; k++) {
                const float v = h2f(s[k]) * inv;
                d[k] = (int8_t)((k < cnt8) ? clamp_i8(rne_i32(v)) : clamp_i8((int)roundf(v)));
            }
        }
    }
}

__global__ __launch_bounds__(256) void kv_quant_kernel(
    const uint16_t* __restrict__ src0, int8_t* __restrict__ dst0, float* __restrict__ scales0,
    const uint16_t* __restrict__ src1, int8_t* __restrict__ dst1, float* __restrict__ scales1,
    size_t rows, size_t D, size_t gs, size_t G, const int32_t* __restrict__ slots, size_t heads,
    size_t cache_batch_stride, size_t scale_batch_stride, size_t max_slots,
    const int32_t* __restrict__ positions = nullptr, size_t seq_len = 1, int window = 0, int sink = 0) {
    kv_quant_rows(blockIdx.y ? src1 : src0, blockIdx.y ? dst1 : dst0, blockIdx.y ? scales1 : scales0, rows, D, gs, G, slots,
                  heads, cache_batch_stride, scale_batch_stride, max_slots, positions, seq_len, window, sink);
}

// One thread per packed byte.  Low nibble -> even element, high nibble -> odd element,
// n > 7 ? n - 16 : n (kernel_utils.h:165-185).  The reference's 16-byte vector body stores a
// 16-element half only when it fits entirely (kernel_quants.cpp:325-330); restated here so
// an odd unpacked_count leaves the same elements untouched.
__global__ __launch_bounds__(256) void unpack_int4_kernel(const uint8_t* __restrict__ packed,
                                                         int8_t* __restrict__ unpacked,
                                                         size_t unpacked_count) {
    const size_t packed_count = (unpacked_count + 1) / 2;
    const size_t simd_end = (packed_count / 16) * 16;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= packed_count) return;
    const size_t o = 2 * i;
    if (i < simd_end) {
        const size_t out0 = (i / 16) * 32;
        const size_t half_start = (o < out0 + 16) ? out0 : out0 + 16;
        if (half_start + 16 > unpacked_count) return;
    }
    const uint8_t b = packed[i];
    const int lo = b & 0x0F, hi = b >> 4;
    if (o < unpacked_count) unpacked[o] = (int8_t)(lo > 7 ? lo - 16 : lo);
    if (o + 1 < unpacked_count) unpacked[o + 1] = (int8_t)(hi > 7 ? hi - 16 : hi);
}

// ------------------------------------------------------------------ launchers

hipError_t launch_act_quant(const uint16_t* src, int8_t* dst, float* scales, size_t M, size_t K,
                            hipStream_t st) {
    if (M == 0 || K == 0) return hipSuccess;
    act_quant_rowwise_kernel<<<dim3((unsigned)M), dim3(256), 0, st>>>(src, dst, scales, 0.0f, 0, K);
    return hipGetLastError();
}

hipError_t launch_fp16_to_int8(const uint16_t* src, int8_t* dst, size_t count, float scale,
                               hipStream_t st) {
    if (count == 0) return hipSuccess;
    // One row of `count` elements: the tail rule (count % 8) is per call, as in the reference.
    act_quant_rowwise_kernel<<<dim3(1), dim3(256), 0, st>>>(src, dst, nullptr, scale, 1, count);
    return hipGetLastError();
}

hipError_t launch_max_abs(const uint16_t* src, size_t count, float* out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    if (e != hipSuccess || count == 0) return e;
    size_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    max_abs_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(src, count, out);
    return hipGetLastError();
}

hipError_t launch_kv_quant(const uint16_t* src, int8_t* dst, float* scales, size_t seq, size_t heads,
                           size_t D, size_t gs, hipStream_t st) {
    const size_t rows = seq * heads;
    if (rows == 0 || D == 0) return hipSuccess;
    const size_t G = (D + gs - 1) / gs;
    const size_t items = rows * G;
    kv_quant_kernel<<<dim3((unsigned)((items + 15) / 16)), dim3(256), 0, st>>>(src, dst, scales, nullptr, nullptr,
                                                                             nullptr, rows, D, gs, G, nullptr, 1, 0, 0, 0);
    return hipGetLastError();
}

hipError_t launch_kv_quant_pair(const uint16_t* k_src, int8_t* k_dst, float* k_scales, const uint16_t* v_src,
                                int8_t* v_dst, float* v_scales, size_t seq, size_t heads, size_t D, size_t gs,
                                hipStream_t st) {
    const size_t rows = seq * heads;
    if (rows == 0 || D == 0) return hipSuccess;
    const size_t G = (D + gs - 1) / gs;
    const size_t items = rows * G;
    kv_quant_kernel<<<dim3((unsigned)((items + 15) / 16), 2), dim3(256), 0, st>>>(
        k_src, k_dst, k_scales, v_src, v_dst, v_scales, rows, D, gs, G, nullptr, 1, 0, 0, 0);
    return hipGetLastError();
}

// Up to KV_QUANT_MAX_JOBS equally shaped (src, dst, scales) sets in one launch, blockIdx.y = set: the K and V rows
// of every layer of a step (KVCache::update_from_graph, engine_cache.cpp:186-266).
struct KvQuantJobSet {
    const uint16_t* src[KV_QUANT_MAX_JOBS];
    int8_t* dst[KV_QUANT_MAX_JOBS];
    float* scales[KV_QUANT_MAX_JOBS];
};
__global__ __launch_bounds__(256) void kv_quant_jobs_kernel(const KvQuantJobSet jobs, size_t rows, size_t D, size_t gs, size_t G) {
    const int jb = blockIdx.y;
    kv_quant_rows(jobs.src[jb], jobs.dst[jb], jobs.scales[jb], rows, D, gs, G, nullptr, 1, 0, 0, 0, nullptr, 1, 0, 0);
}

hipError_t launch_kv_quant_jobs(const KvQuantJob* jobs, size_t count, size_t seq, size_t heads, size_t D, size_t gs,
                                hipStream_t st) {
    const size_t rows = seq * heads;
    if (rows == 0 || D == 0) return hipSuccess;
    const size_t G = (D + gs - 1) / gs;
    const size_t items = rows * G;
    for (size_t first = 0; first < count; first += KV_QUANT_MAX_JOBS) {
        const size_t n = count - first < (size_t)KV_QUANT_MAX_JOBS ? count - first : (size_t)KV_QUANT_MAX_JOBS;
        KvQuantJobSet set;
        for (size_t i = 0; i < (size_t)KV_QUANT_MAX_JOBS; i++) {
            const KvQuantJob& j = jobs[first + (i < n ? i : 0)];
            set.src[i] = j.src;
            set.dst[i] = j.dst;
            set.scales[i] = j.scales;
        }
        kv_quant_jobs_kernel<<<dim3((unsigned)((items + 15) / 16), (unsigned)n), dim3(256), 0, st>>>(set, rows, D, gs, G);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_kv_append(const uint16_t* k_src, int8_t* k_cache, float* k_scales, const uint16_t* v_src,
                            int8_t* v_cache, float* v_scales, size_t batch, size_t heads, size_t D, size_t gs,
                            const int32_t* slots, size_t cache_batch_stride, size_t scale_batch_stride,
                            size_t max_slots, hipStream_t st) {
    const size_t rows = batch * heads;
    if (rows == 0 || D == 0) return hipSuccess;
    const size_t G = (D + gs - 1) / gs;
    const size_t items = rows * G;
    kv_quant_kernel<<<dim3((unsigned)((items + 15) / 16), 2), dim3(256), 0, st>>>(
        k_src, k_cache, k_scales, v_src, v_cache, v_scales, rows, D, gs, G, slots, heads, cache_batch_stride,
        scale_batch_stride, max_slots);
    return hipGetLastError();
}

hipError_t launch_kv_append_windowed(const uint16_t* k_src, int8_t* k_cache, float* k_scales, const uint16_t* v_src,
                                     int8_t* v_cache, float* v_scales, const int32_t* positions, size_t batch,
                                     size_t seq_len, size_t heads, size_t D, size_t gs, size_t cache_batch_stride,
                                     size_t scale_batch_stride, size_t max_slots, size_t window, size_t sink,
                                     hipStream_t st) {
    const size_t rows = batch * seq_len * heads;
    if (rows == 0 || D == 0) return hipSuccess;
    const size_t G = (D + gs - 1) / gs;
    const size_t items = rows * G;
    kv_quant_kernel<<<dim3((unsigned)((items + 15) / 16), 2), dim3(256), 0, st>>>(
        k_src, k_cache, k_scales, v_src, v_cache, v_scales, rows, D, gs, G, nullptr, heads, cache_batch_stride,
        scale_batch_stride, max_slots, positions, seq_len, (int)window, (int)sink);
    return hipGetLastError();
}

// positions[b] += advance; cache_lens[b] = tokens the cache of sequence b holds = min(position, window)
__global__ __launch_bounds__(256) void kv_positions_advance_kernel(int32_t* __restrict__ positions,
                                                                  int32_t* __restrict__ cache_lens, size_t batch,
                                                                  int advance, int window) {
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const int p = positions[b] + advance;
    positions[b] = p;
    if (cache_lens) cache_lens[b] = (window > 0 && p > window) ? window : p;
}

hipError_t launch_kv_positions_advance(int32_t* positions, int32_t* cache_lens, size_t batch, size_t advance,
                                       size_t window, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    kv_positions_advance_kernel<<<dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st>>>(
        positions, cache_lens, batch, (int)advance, (int)window);
    return hipGetLastError();
}

hipError_t launch_unpack_int4(const uint8_t* packed, int8_t* unpacked, size_t count, hipStream_t st) {
    const size_t pc = (count + 1) / 2;
    if (pc == 0) return hipSuccess;
    unpack_int4_kernel<<<dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, st>>>(packed, unpacked, count);
    return hipGetLastError();
}

}  // namespace cactus_hip
