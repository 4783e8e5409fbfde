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
#include "block_math.hip.h"

namespace cactus_hip {

namespace {

// ---------------------------------------------------------------------------------------
// RMS norm.  16 lanes per row, 16 rows per 256-thread workgroup.  Lane l owns the reference's
// fp32 accumulator l: element i < simd_end (dims rounded down to 8) is folded into accumulator
// i % 16 with an fma, in increasing i (four float32x4 registers over a 16-element tile, the 8-wide
// remainder into the first two).  The registers are added ((v0+v1)+v2)+v3 lane-wise, the four
// lanes as (t0+t1)+(t2+t3), then the scalar tail elements are fused in.  The vector part of the
// output multiplies in fp16, (x * half(inv)) * w with both products rounded; the tail computes
// x*inv*w in fp32 and rounds once.
// ---------------------------------------------------------------------------------------

// ADD: the row normalized is a + b (one fp16 add per element, as cactus_add_f16), which is also
// written to sum_out: the residual add and the norm that follows it in a decoder block
// (model_qwen.cpp:125-126), one launch.
// A workgroup takes rpb (<= 16) rows.  All 256 threads first copy the rows (summed if ADD) into
// LDS with 16-byte loads, so that a row costs one load latency instead of one per step of the
// 16-lane accumulation chain; the chain and the output pass then read LDS.  rpb = 0: rows too
// long for LDS, 16 rows per workgroup read straight from global memory.
template <bool ADD>
__global__ __launch_bounds__(256) void rms_norm_kernel(const uint16_t* __restrict__ x,
                                                       const uint16_t* __restrict__ addend,
                                                       const uint16_t* __restrict__ w_in,
                                                       uint16_t* __restrict__ sum_out,
                                                       uint16_t* __restrict__ y, size_t rows, int dims,
                                                       float eps, int rpb) {
    const uint16_t* w = w_in;
    extern __shared__ __attribute__((aligned(16))) uint16_t rn_lds[];   // [rpb][dims]
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const bool staged = rpb > 0;
    const int per_block = staged ? rpb : 16;
    const size_t row0 = (size_t)blockIdx.x * per_block;
    if (staged) {
        const size_t nrows = min((size_t)rpb, rows - row0);
        const size_t n = nrows * dims;
        const uint16_t* xs = x + row0 * dims;
        const uint16_t* as = ADD ? addend + row0 * dims : nullptr;
        bool vec = ((reinterpret_cast<uintptr_t>(xs)) & 15) == 0;
        if (ADD) vec = vec && (reinterpret_cast<uintptr_t>(as) & 15) == 0;
        const size_t nv = vec ? n / 8 : 0;
        for (size_t c = threadIdx.x; c < nv; c += 256) {
            v4i xv = *reinterpret_cast<const v4i*>(xs + c * 8);
            if (ADD) {
                const v4i av = *reinterpret_cast<const v4i*>(as + c * 8);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t lo = h2bits(bits2h((uint16_t)((uint32_t)xv[e] & 0xFFFFu)) + bits2h((uint16_t)((uint32_t)av[e] & 0xFFFFu)));
                    const uint32_t hi = h2bits(bits2h((uint16_t)((uint32_t)xv[e] >> 16)) + bits2h((uint16_t)((uint32_t)av[e] >> 16)));
                    xv[e] = (int)(lo | (hi << 16));
                }
            }
            *reinterpret_cast<v4i*>(rn_lds + c * 8) = xv;
        }
        for (size_t i = nv * 8 + threadIdx.x; i < n; i += 256)
            rn_lds[i] = ADD ? h2bits(bits2h(xs[i]) + bits2h(as[i])) : xs[i];
        // the weight row rides along (behind the rows, 16-byte aligned), so that every global load
        // of the kernel is in flight at once
        uint16_t* wl = rn_lds + (((size_t)rpb * dims + 7) & ~(size_t)7);
        const size_t wv = ((reinterpret_cast<uintptr_t>(w) & 15) == 0) ? (size_t)dims / 8 : 0;
        for (size_t c = threadIdx.x; c < wv; c += 256)
            *reinterpret_cast<v4i*>(wl + c * 8) = *reinterpret_cast<const v4i*>(w + c * 8);
        for (size_t i = wv * 8 + threadIdx.x; i < (size_t)dims; i += 256) wl[i] = w[i];
        __syncthreads();
        w = wl;
    }
    const size_t row = row0 + grp;
    const bool live = row < rows && grp < per_block;
    const size_t r = live ? row : (staged ? row0 : rows - 1);   // idle groups redo a valid row, store nothing
    const uint16_t* xr = x + r * dims;
    const uint16_t* ar = ADD ? addend + r * dims : nullptr;
    const uint16_t* lr = rn_lds + (r - row0) * dims;
    const int simd_end = dims & ~7;
    const int base = (threadIdx.x & 63) & ~15;
    auto elem = [&](int i) -> uint16_t {
        if (staged) return lr[i];
        if (ADD) return h2bits(bits2h(xr[i]) + bits2h(ar[i]));
        return xr[i];
    };
    const float inv = row_inv_rms(elem, dims, eps, l, base);
    if (!live) return;
    const h16 inv_h = (h16)inv;
    uint16_t* yr = y + r * dims;
    uint16_t* sr = ADD ? sum_out + r * dims : nullptr;
    bool vec_ok = (reinterpret_cast<uintptr_t>(yr) & 15) == 0;
    if (!staged) vec_ok = vec_ok && ((reinterpret_cast<uintptr_t>(xr) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    if (staged) vec_ok = vec_ok && ((dims & 7) == 0);   // LDS rows stay 16-byte aligned
    if (ADD) vec_ok = vec_ok && (reinterpret_cast<uintptr_t>(sr) & 15) == 0 &&
                      (staged || (reinterpret_cast<uintptr_t>(ar) & 15) == 0);
    if (vec_ok) {
        for (int c = l; c < simd_end / 8; c += 16) {
            v4i xv = staged ? *reinterpret_cast<const v4i*>(lr + c * 8) : *reinterpret_cast<const v4i*>(xr + c * 8);
            const v4i wv = *reinterpret_cast<const v4i*>(w + c * 8);
            v4i av = {0, 0, 0, 0};
            if (ADD && !staged) av = *reinterpret_cast<const v4i*>(ar + c * 8);
            v4i o, so;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t res = 0, sres = 0;
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    uint16_t xb = (uint16_t)((uint32_t)xv[e] >> (16 * hh));
                    if (ADD && !staged) xb = h2bits(bits2h(xb) + bits2h((uint16_t)((uint32_t)av[e] >> (16 * hh))));
                    sres |= (uint32_t)xb << (16 * hh);
                    res |= (uint32_t)rms_out(xb, (uint16_t)((uint32_t)wv[e] >> (16 * hh)), inv, inv_h, true) << (16 * hh);
                }
                o[e] = (int)res;
                so[e] = (int)sres;
            }
            *reinterpret_cast<v4i*>(yr + c * 8) = o;
            if (ADD) *reinterpret_cast<v4i*>(sr + c * 8) = so;
        }
    } else {
        for (int i = l; i < simd_end; i += 16) {
            const uint16_t xb = elem(i);
            yr[i] = rms_out(xb, w[i], inv, inv_h, true);
            if (ADD) sr[i] = xb;
        }
    }
    for (int i = simd_end + l; i < dims; i += 16) {
        const uint16_t xb = elem(i);
        yr[i] = rms_out(xb, w[i], inv, inv_h, false);
        if (ADD) sr[i] = xb;
    }
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

// Per-head RMS norm of the q and k projections followed by RoPE (model_qwen.cpp:63-81), one
// launch: a 16-lane group takes one (token, head) row of head_dim elements, normalizes it into
// LDS with the arithmetic of rms_norm_kernel, then rotates it with the arithmetic of rope_kernel.
// Rows [0, tokens*Hq) are q rows, the rest k rows; cos_t / sin_t as in rope_kernel.
// With `ap.v` set the launch also appends the step's K and V to the INT8 cache
// (engine_cache.cpp:234-267): the rotated k row and the v row (rows [tokens*(Hq+Hkv),
// tokens*(Hq+2*Hkv)) of the launch) are quantized per group of `gs` dims with the arithmetic of
// kv_quant_kernel and stored at slot slots[b] + s of sequence b (slots == nullptr: row for row).
// With `ap.positions` set, sequence b's tokens have absolute positions positions[b] + s: RoPE reads
// the table row of that position (cos_t / sin_t then start at position 0 and hold rows up to
// max_position) and an appended row goes to ring_slot(position, window, sink).
struct KvAppendArgs {
    const uint16_t* v = nullptr;
    int8_t* kc = nullptr;
    int8_t* vc = nullptr;
    float* ks = nullptr;
    float* vs = nullptr;
    const int32_t* slots = nullptr;
    const int32_t* positions = nullptr;
    size_t cache_batch_stride = 0, scale_batch_stride = 0, max_slots = 0;
    int gs = 128;
    int window = 0, sink = 0, max_position = 0;
};

__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const uint16_t* __restrict__ q,
                                                           const uint16_t* __restrict__ k,
                                                           const uint16_t* __restrict__ qw,
                                                           const uint16_t* __restrict__ kw,
                                                           uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_out,
                                                           const uint16_t* __restrict__ cos_t,
                                                           const uint16_t* __restrict__ sin_t, size_t tokens,
                                                           int Hq, int Hkv, int D, int seq_len, float eps,
                                                           const KvAppendArgs ap) {
    extern __shared__ __attribute__((aligned(16))) uint16_t qk_lds[];   // [16][D]
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t nq = tokens * Hq, nqk = nq + tokens * Hkv, total = nqk + (ap.v ? tokens * Hkv : 0);
    const size_t row = (size_t)blockIdx.x * 16 + grp;
    const bool live = row < total;
    const size_t r = live ? row : total - 1;
    const bool is_q = r < nq, is_v = r >= nqk;
    const size_t hr = is_q ? r : (is_v ? r - nqk : r - nq);   // token * heads + head
    const size_t token = hr / (is_q ? Hq : Hkv);
    uint16_t* nrm = qk_lds + grp * D;
    // where a k / v row goes in the cache
    size_t doff = hr * D, soff = hr * (size_t)((D + ap.gs - 1) / ap.gs);
    bool place = live && ap.v != nullptr && !is_q;
    if (place && (ap.slots != nullptr || ap.positions != nullptr)) {
        const size_t b = token / seq_len, sq = token % seq_len, h = hr - token * Hkv;
        const long first = ap.positions ? ap.positions[b] : ap.slots[b];
        const long sl = ap.positions ? ring_slot(first + (long)sq, ap.window, ap.sink) : first + (long)sq;
        place = first >= 0 && (size_t)sl < ap.max_slots;
        const size_t slot = place ? (size_t)sl : 0;
        doff = b * ap.cache_batch_stride + (slot * Hkv + h) * D;
        soff = b * ap.scale_batch_stride + (slot * Hkv + h) * (size_t)((D + ap.gs - 1) / ap.gs);
    }
    uint16_t* orow = (is_q ? q_out : k_out) + hr * D;
    // rows move 8 elements per access when head_dim and the pointers allow it: the row is copied to
    // LDS first and the 16-lane chain reads it from there
    const bool vecD = (D & 7) == 0 &&
                      ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(ap.v) |
                        reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(k_out)) & 15) == 0;
    auto wave_sync = [] {   // the 16 lanes of a row sit in one wave: orders their LDS writes before the reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // q and k rows of up to 256 dims: the norm weights of the lane's elements and the rotation factors of its pairs
    // are requested together with the row, not behind the reduction and the barrier (three dependent round trips
    // otherwise)
    const int half_dim = D / 2;
    int s = (int)(token % seq_len);
    if (ap.positions) s = max(0, min(ap.positions[token / seq_len] + s, ap.max_position));
    const bool pre = vecD && D <= 256 && !is_v;
    uint16_t wreg[16], creg[8], sreg[8];
    if (pre) {
        const uint16_t* w = is_q ? qw : kw;
#pragma unroll
        for (int j = 0; j < 16; j++) wreg[j] = w[min(l + 16 * j, D - 1)];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            creg[j] = cos_t[(size_t)s * half_dim + min(l + 16 * j, half_dim - 1)];
            sreg[j] = sin_t[(size_t)s * half_dim + min(l + 16 * j, half_dim - 1)];
        }
    }
    if (is_v) {   // a v row goes straight to the quantizer
        const uint16_t* vr = ap.v + hr * D;
        if (vecD) {
            for (int c = l; c < D / 8; c += 16) *reinterpret_cast<v4i*>(nrm + c * 8) = *reinterpret_cast<const v4i*>(vr + c * 8);
        } else {
            for (int i = l; i < D; i += 16) nrm[i] = vr[i];
        }
    } else {
        const uint16_t* xr = (is_q ? q : k) + hr * D;
        const uint16_t* w = is_q ? qw : kw;
        const int base = (threadIdx.x & 63) & ~15;
        const int simd_end = D & ~7;
        if (vecD) {
            for (int c = l; c < D / 8; c += 16) *reinterpret_cast<v4i*>(nrm + c * 8) = *reinterpret_cast<const v4i*>(xr + c * 8);
            wave_sync();
            const float inv = row_inv_rms([&](int i) -> uint16_t { return nrm[i]; }, D, eps, l, base);
            const h16 inv_h = (h16)inv;
            if (pre) {
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (l + 16 * j < D) nrm[l + 16 * j] = rms_out(nrm[l + 16 * j], wreg[j], inv, inv_h, true);
            } else {
                for (int i = l; i < D; i += 16) nrm[i] = rms_out(nrm[i], w[i], inv, inv_h, true);   // in place: own elements
            }
        } else {
            const float inv = row_inv_rms([&](int i) -> uint16_t { return xr[i]; }, D, eps, l, base);
            const h16 inv_h = (h16)inv;
            for (int i = l; i < D; i += 16) nrm[i] = rms_out(xr[i], w[i], inv, inv_h, i < simd_end);
        }
    }
    __syncthreads();   // the one barrier of the kernel, reached by every thread
    if (!live) return;
    if (is_v) {
        if (place) quant_row_groups(nrm, D, ap.gs, ap.vc + doff, ap.vs + soff, l);
        return;
    }
    auto rotate = [&](int i, uint16_t cb, uint16_t sb) {
        const uint16_t x1b = nrm[i], x2b = nrm[half_dim + i];
        uint16_t o1, o2;
        if (i < (half_dim & ~7)) {
            const h16 c = bits2h(cb), sn = bits2h(sb), x1 = bits2h(x1b), x2 = bits2h(x2b);
            const h16 p1 = x1 * c, p2 = x2 * c;
            o1 = h2bits(__builtin_fmaf16(-x2, sn, p1));
            o2 = h2bits(__builtin_fmaf16(x1, sn, p2));
        } else {
            const float c = h2f(cb), sn = h2f(sb), x1 = h2f(x1b), x2 = h2f(x2b);
            const float pa = x1 * c, pb = x2 * sn, pd = x2 * c, pe = x1 * sn;
            o1 = f2h(pa - pb);
            o2 = f2h(pd + pe);
        }
        if (!vecD) {
            orow[i] = o1;
            orow[half_dim + i] = o2;
        }
        if (place || vecD) { nrm[i] = o1; nrm[half_dim + i] = o2; }   // the lane's own two elements: the rotated row
    };
    if (pre) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (l + 16 * j < half_dim) rotate(l + 16 * j, creg[j], sreg[j]);
    } else {
        for (int i = l; i < half_dim; i += 16) rotate(i, cos_t[(size_t)s * half_dim + i], sin_t[(size_t)s * half_dim + i]);
    }
    if (place || vecD) wave_sync();
    if (vecD) {
        for (int c = l; c < D / 8; c += 16) *reinterpret_cast<v4i*>(orow + c * 8) = *reinterpret_cast<const v4i*>(nrm + c * 8);
    }
    if (place) quant_row_groups(nrm, D, ap.gs, ap.kc + doff, ap.ks + soff, l);
}

// The call is treated as one contiguous piece: whole groups of 8 from the start take the vector
// formula, the last n % 8 elements the scalar libm one (what the reference does whenever it runs
// the call on one thread).
// MUL: the result is multiplied by mul[i] in fp16 (cactus_multiply_f16 on the rounded activation):
// silu(gate) * up of the MLP (model_qwen.cpp:114-115) in one launch.
template <int OP, bool MUL>
__global__ __launch_bounds__(256) void unary_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ mul,
                                                    uint16_t* __restrict__ out, size_t n) {
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;   // chunk of 8 halves
    const size_t vend = n & ~(size_t)7;
    if (c * 8 >= n) return;
    bool vec_ok = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (MUL) vec_ok = vec_ok && (reinterpret_cast<uintptr_t>(mul) & 15) == 0;
    auto times = [](uint16_t a, uint16_t m) -> uint16_t { return MUL ? h2bits(bits2h(a) * bits2h(m)) : a; };
    if (c * 8 < vend && vec_ok) {
        const v4i xv = *reinterpret_cast<const v4i*>(in + c * 8);
        v4i mv = {0, 0, 0, 0};
        if (MUL) mv = *reinterpret_cast<const v4i*>(mul + c * 8);
        v4i o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t lo = times(unary_one<OP>((uint16_t)((uint32_t)xv[e] & 0xFFFFu), true), (uint16_t)((uint32_t)mv[e] & 0xFFFFu));
            const uint32_t hi = times(unary_one<OP>((uint16_t)((uint32_t)xv[e] >> 16), true), (uint16_t)((uint32_t)mv[e] >> 16));
            o[e] = (int)(lo | (hi << 16));
        }
        *reinterpret_cast<v4i*>(out + c * 8) = o;
    } else {
        for (size_t i = c * 8; i < c * 8 + 8 && i < n; i++)
            out[i] = times(unary_one<OP>(in[i], i < vend), MUL ? mul[i] : (uint16_t)0);
    }
}

// ---------------------------------------------------------------------------------------
// Row producer + activation quantization in one launch: the matmul that follows a norm or an
// activation in a decoder block quantizes its FP16 input row by row (graph_ops_nn.cpp:35-66), and
// at decode sizes that quantization is a launch of its own.  One 256-thread workgroup per row:
//   MODE 0: y = rms_norm(a [+ b]) * w      (the sum is also written to sum_out when b is given)
//   MODE 1: y = silu(a) * b                MODE 2: y = gelu(a) * b
// each with the arithmetic of the kernels above, then scale = max(absmax(y) / 127, 1e-10) and
// q = sat8(round(clamp(y * (1 / scale)))) -- ties to even for the row's whole groups of 8, away
// from zero for its last dims % 8 elements, as cactus_fp16_to_int8 (kernel_quants.cpp:158-194).
// LDS: the input row (MODE 0) and the produced row, binary16.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void row_quant_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                        const uint16_t* __restrict__ w, uint16_t* __restrict__ sum_out,
                                                        uint16_t* __restrict__ y_out, int8_t* __restrict__ q_out,
                                                        float* __restrict__ scales_out, int dims, float eps,
                                                        size_t total_vend) {
    extern __shared__ __attribute__((aligned(16))) uint16_t rq_lds[];   // [2][dims]
    __shared__ float red[4];
    __shared__ float s_inv;
    uint16_t* xin = rq_lds;
    uint16_t* yb = rq_lds + dims;
    const int t = threadIdx.x;
    const size_t row = blockIdx.x, off = row * (size_t)dims;
    float m = 0.0f;
    // 8 elements per access when the row allows it
    bool vec = (dims & 7) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                                    reinterpret_cast<uintptr_t>(sum_out) | reinterpret_cast<uintptr_t>(y_out)) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(q_out) & 7) == 0;
    if (MODE == 0) vec = vec && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
    const int nch = dims / 8;
    auto half_of = [](const v4i& x, int e) -> uint16_t { return (uint16_t)(((uint32_t)x[e >> 1] >> (16 * (e & 1))) & 0xFFFFu); };
    auto pack8 = [](const uint16_t (&v)[8]) -> v4i {
        v4i x;
#pragma unroll
        for (int e = 0; e < 4; e++) x[e] = (int)((uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16));
        return x;
    };
    if (MODE == 0) {
        // the norm weights of the thread's first piece are requested with the row itself, not behind the reduction
        v4i w_first = {0, 0, 0, 0};
        if (vec && t < nch) w_first = *reinterpret_cast<const v4i*>(w + t * 8);
        if (vec) {
            for (int c = t; c < nch; c += 256) {
                v4i xv = *reinterpret_cast<const v4i*>(a + off + c * 8);
                if (b) {
                    const v4i bv = *reinterpret_cast<const v4i*>(b + off + c * 8);
                    uint16_t v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = h2bits(bits2h(half_of(xv, e)) + bits2h(half_of(bv, e)));
                    xv = pack8(v);
                }
                *reinterpret_cast<v4i*>(xin + c * 8) = xv;
                if (sum_out) *reinterpret_cast<v4i*>(sum_out + off + c * 8) = xv;
            }
        } else {
            for (int i = t; i < dims; i += 256) {
                const uint16_t v = b ? h2bits(bits2h(a[off + i]) + bits2h(b[off + i])) : a[off + i];
                xin[i] = v;
                if (sum_out) sum_out[off + i] = v;
            }
        }
        __syncthreads();
        if (t < 16) {
            const float inv = row_inv_rms([&](int i) -> uint16_t { return xin[i]; }, dims, eps, t, 0);
            if (t == 0) s_inv = inv;
        }
        __syncthreads();
        const float inv = s_inv;
        const h16 inv_h = (h16)inv;
        const int simd_end = dims & ~7;
        if (vec) {
            for (int c = t; c < nch; c += 256) {
                const v4i xv = *reinterpret_cast<const v4i*>(xin + c * 8);
                const v4i wv = c == t ? w_first : *reinterpret_cast<const v4i*>(w + c * 8);
                uint16_t y[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    y[e] = rms_out(half_of(xv, e), half_of(wv, e), inv, inv_h, true);
                    m = fmaxf(m, fabsf(h2f(y[e])));
                }
                *reinterpret_cast<v4i*>(yb + c * 8) = pack8(y);
            }
        } else {
            for (int i = t; i < dims; i += 256) {
                const uint16_t y = rms_out(xin[i], w[i], inv, inv_h, i < simd_end);
                yb[i] = y;
                m = fmaxf(m, fabsf(h2f(y)));
            }
        }
    } else if (vec) {
        for (int c = t; c < nch; c += 256) {
            const v4i av = *reinterpret_cast<const v4i*>(a + off + c * 8);
            const v4i bv = *reinterpret_cast<const v4i*>(b + off + c * 8);
            uint16_t y[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint16_t act = unary_one<MODE == 1 ? OP_SILU : OP_GELU>(half_of(av, e), off + c * 8 + e < total_vend);
                y[e] = h2bits(bits2h(act) * bits2h(half_of(bv, e)));
                m = fmaxf(m, fabsf(h2f(y[e])));
            }
            *reinterpret_cast<v4i*>(yb + c * 8) = pack8(y);
        }
    } else {
        for (int i = t; i < dims; i += 256) {
            const uint16_t act = unary_one<MODE == 1 ? OP_SILU : OP_GELU>(a[off + i], off + i < total_vend);
            const uint16_t y = h2bits(bits2h(act) * bits2h(b[off + i]));
            yb[i] = y;
            m = fmaxf(m, fabsf(h2f(y)));
        }
    }
    m = wave_max(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float scale = m / 127.0f;
    if (scale < 1e-10f) scale = 1e-10f;
    if (t == 0) scales_out[row] = scale;
    const float iq = 1.0f / scale;
    const int q8 = dims & ~7;
    if (vec) {      // each thread re-reads the chunks it produced
        for (int c = t; c < nch; c += 256) {
            const v4i yv = *reinterpret_cast<const v4i*>(yb + c * 8);
            *reinterpret_cast<uint2*>(q_out + off + c * 8) = quant8_bits(yv, iq);
            if (y_out) *reinterpret_cast<v4i*>(y_out + off + c * 8) = yv;
        }
    } else {
        for (int i = t; i < dims; i += 256) {
            const uint16_t y = yb[i];
            float q = h2f(y) * iq;
            q = fmaxf(fminf(q, 127.0f), -128.0f);
            q_out[off + i] = (int8_t)((i < q8) ? clamp_i8(rne_i32(q)) : (int)roundf(q));
            if (y_out) y_out[off + i] = y;
        }
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

// cactus_scalar_op_f16 (kernel_scalar.cpp:7-154): the scalar is rounded to binary16 once, then one
// fp16 operation per element, vector body and tail alike.
template <int OP>
__global__ __launch_bounds__(256) void scalar_kernel(const uint16_t* __restrict__ a, uint16_t scalar_bits,
                                                     uint16_t* __restrict__ out, size_t n) {
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c * 8 >= n) return;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (c * 8 + 8 <= n && vec_ok) {
        const v4i av = *reinterpret_cast<const v4i*>(a + c * 8);
        v4i o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t lo = binary_one<OP>((uint16_t)((uint32_t)av[e] & 0xFFFFu), scalar_bits);
            const uint32_t hi = binary_one<OP>((uint16_t)((uint32_t)av[e] >> 16), scalar_bits);
            o[e] = (int)(lo | (hi << 16));
        }
        *reinterpret_cast<v4i*>(out + c * 8) = o;
    } else {
        for (size_t i = c * 8; i < c * 8 + 8 && i < n; i++) out[i] = binary_one<OP>(a[i], scalar_bits);
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

// Greedy branch of cactus_sample_f16 (kernel_nn.cpp:450-466): the first index whose value is
// strictly greater than every earlier one, starting from element 0 -- so ties go to the lowest
// index, NaNs are never chosen, and a NaN in element 0 keeps index 0.  One workgroup per row.
__global__ __launch_bounds__(1024) void argmax_kernel(const uint16_t* __restrict__ logits, uint32_t* __restrict__ out,
                                                      size_t vocab) {
    __shared__ float s_val[16];
    __shared__ uint32_t s_idx[16];
    const uint16_t* row = logits + (size_t)blockIdx.x * vocab;
    const float first = h2f(row[0]);
    float best = first;
    uint32_t bi = 0;
    if (first == first) {   // not NaN
        if ((vocab & 7) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
            // eight logits per access; a thread's indices still increase, so its first maximum is kept
            for (size_t c = threadIdx.x; c < vocab / 8; c += 1024) {
                const v4i x = *reinterpret_cast<const v4i*>(row + c * 8);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float v = h2f((uint16_t)(((uint32_t)x[e >> 1] >> (16 * (e & 1))) & 0xFFFFu));
                    if (v > best) { best = v; bi = (uint32_t)(c * 8 + e); }
                }
            }
        } else {
            for (size_t i = threadIdx.x; i < vocab; i += 1024) {
                const float v = h2f(row[i]);
                if (v > best) { best = v; bi = (uint32_t)i; }   // per thread i increases: first maximum kept
            }
        }
    }
    auto better = [](float v, uint32_t i, float bv, uint32_t bidx) { return v > bv || (v == bv && i < bidx); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const uint32_t oi = __shfl_xor(bi, o);
        if (better(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++)
            if (better(s_val[w], s_idx[w], best, bi)) { best = s_val[w]; bi = s_idx[w]; }
        out[blockIdx.x] = bi;
    }
}

// The same rule with a row spread over several workgroups (long rows, few of them).  A candidate
// is packed as (order-preserving key of the value) << 32 | (0xFFFFFFFF - index), so a 64-bit max
// prefers the larger value and, between equals, the lower index.  scratch[row] = {best, arrivals},
// all zero between launches: the workgroup that arrives last decodes the winner, stores it and
// zeroes the two words again.  A NaN in element 0 makes every workgroup contribute nothing: 0.
__global__ __launch_bounds__(256) void argmax_split_kernel(const uint16_t* __restrict__ logits,
                                                           uint32_t* __restrict__ out, size_t vocab,
                                                           unsigned long long* __restrict__ scratch, int chunk) {
    __shared__ unsigned long long s_best[4];
    const size_t r = blockIdx.y;
    const uint16_t* row = logits + r * vocab;
    const float first = h2f(row[0]);
    unsigned long long best = 0;
    if (first == first) {
        const size_t lo = (size_t)blockIdx.x * chunk, hi = min(vocab, lo + (size_t)chunk);
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = h2f(row[i]);
            if (v == v) {
                const uint32_t b = __builtin_bit_cast(uint32_t, v);
                const uint32_t key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // -0 < +0 here; both are
                const uint32_t keyz = (v == 0.0f) ? 0x80000000u : key;              // "equal" to the reference
                const unsigned long long c = ((unsigned long long)keyz << 32) | (0xFFFFFFFFu - (uint32_t)i);
                best = c > best ? c : best;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ov = __shfl_xor(best, o);
        best = ov > best ? ov : best;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) best = s_best[w] > best ? s_best[w] : best;
        unsigned long long* slot = scratch + 2 * r;
        if (best) atomicMax(slot, best);
        __threadfence();
        const unsigned long long arrived = atomicAdd(slot + 1, 1ull) + 1;
        if (arrived == gridDim.x) {
            __threadfence();
            const unsigned long long win = atomicExch(slot, 0ull);
            atomicExch(slot + 1, 0ull);
            out[r] = win ? 0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFull) : 0u;
        }
    }
}

}  // namespace

// scratch: 2 zeroed 64-bit words per row (nullptr: one workgroup per row)
hipError_t launch_argmax(const uint16_t* logits, uint32_t* out, size_t rows, size_t vocab,
                         unsigned long long* scratch, hipStream_t st) {
    if (scratch != nullptr && vocab >= 16384 && rows <= 64) {
        const int chunk = 4096;
        argmax_split_kernel<<<dim3((unsigned)((vocab + chunk - 1) / chunk), (unsigned)rows), dim3(256), 0, st>>>(
            logits, out, vocab, scratch, chunk);
    } else {
        argmax_kernel<<<dim3((unsigned)rows), dim3(1024), 0, st>>>(logits, out, vocab);
    }
    return hipGetLastError();
}

hipError_t launch_indices_to_i32(const void* src, bool is_f32, int32_t* dst, size_t n, hipStream_t st) {
    indices_to_i32_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(src, is_f32 ? 1 : 0, dst, n);
    return hipGetLastError();
}

// rows staged in LDS per workgroup: as many as fit 32 KB, at most 16; 0 = too long to stage
static int rms_rows_per_block(size_t dims) {
    const size_t fit = (30 * 1024) / (dims * sizeof(uint16_t));   // rows + one weight row within 64 KB
    return (int)(fit > 16 ? 16 : fit);
}

static size_t rms_lds_bytes(int rpb, size_t dims) {
    return rpb ? ((((size_t)rpb * dims + 7) & ~(size_t)7) + dims) * sizeof(uint16_t) : 0;
}

hipError_t launch_rms_norm(const uint16_t* x, const uint16_t* w, uint16_t* y, size_t rows, size_t dims, float eps,
                           hipStream_t st) {
    const int rpb = rms_rows_per_block(dims);
    const size_t per = rpb ? rpb : 16;
    rms_norm_kernel<false><<<dim3((unsigned)((rows + per - 1) / per)), dim3(256), rms_lds_bytes(rpb, dims), st>>>(
        x, nullptr, w, nullptr, y, rows, (int)dims, eps, rpb);
    return hipGetLastError();
}

hipError_t launch_add_rms_norm(const uint16_t* a, const uint16_t* b, const uint16_t* w, uint16_t* sum_out,
                               uint16_t* y, size_t rows, size_t dims, float eps, hipStream_t st) {
    const int rpb = rms_rows_per_block(dims);
    const size_t per = rpb ? rpb : 16;
    rms_norm_kernel<true><<<dim3((unsigned)((rows + per - 1) / per)), dim3(256), rms_lds_bytes(rpb, dims), st>>>(
        a, b, w, sum_out, y, rows, (int)dims, eps, rpb);
    return hipGetLastError();
}

hipError_t launch_qk_norm_rope(const uint16_t* q, const uint16_t* k, const uint16_t* qw, const uint16_t* kw,
                               uint16_t* q_out, uint16_t* k_out, const uint16_t* cos_t, const uint16_t* sin_t,
                               size_t tokens, size_t Hq, size_t Hkv, size_t D, size_t seq_len, float eps,
                               hipStream_t st) {
    const size_t total = tokens * (Hq + Hkv);
    qk_norm_rope_kernel<<<dim3((unsigned)((total + 15) / 16)), dim3(256), 16 * D * sizeof(uint16_t), st>>>(
        q, k, qw, kw, q_out, k_out, cos_t, sin_t, tokens, (int)Hq, (int)Hkv, (int)D, (int)seq_len, eps, KvAppendArgs());
    return hipGetLastError();
}

hipError_t launch_qk_norm_rope_append(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* qw,
                                      const uint16_t* kw, uint16_t* q_out, uint16_t* k_out, const uint16_t* cos_t,
                                      const uint16_t* sin_t, int8_t* kc, int8_t* vc, float* ks, float* vs,
                                      const int32_t* slots, size_t batch, size_t seq_len, size_t Hq, size_t Hkv,
                                      size_t D, float eps, size_t gs, size_t cache_batch_stride,
                                      size_t scale_batch_stride, size_t max_slots, hipStream_t st) {
    const size_t tokens = batch * seq_len, total = tokens * (Hq + 2 * Hkv);
    KvAppendArgs ap;
    ap.v = v; ap.kc = kc; ap.vc = vc; ap.ks = ks; ap.vs = vs; ap.slots = slots;
    ap.cache_batch_stride = cache_batch_stride; ap.scale_batch_stride = scale_batch_stride; ap.max_slots = max_slots;
    ap.gs = (int)gs;
    qk_norm_rope_kernel<<<dim3((unsigned)((total + 15) / 16)), dim3(256), 16 * D * sizeof(uint16_t), st>>>(
        q, k, qw, kw, q_out, k_out, cos_t, sin_t, tokens, (int)Hq, (int)Hkv, (int)D, (int)seq_len, eps, ap);
    return hipGetLastError();
}

hipError_t launch_qk_norm_rope_seq(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* qw,
                                   const uint16_t* kw, uint16_t* q_out, uint16_t* k_out, const uint16_t* cos_t,
                                   const uint16_t* sin_t, int8_t* kc, int8_t* vc, float* ks, float* vs,
                                   const int32_t* positions, size_t max_position, size_t batch, size_t seq_len,
                                   size_t Hq, size_t Hkv, size_t D, float eps, size_t gs, size_t cache_batch_stride,
                                   size_t scale_batch_stride, size_t max_slots, size_t window, size_t sink,
                                   hipStream_t st) {
    const size_t tokens = batch * seq_len, total = tokens * (Hq + (v ? 2 : 1) * Hkv);
    KvAppendArgs ap;
    ap.v = v; ap.kc = kc; ap.vc = vc; ap.ks = ks; ap.vs = vs; ap.positions = positions;
    ap.cache_batch_stride = cache_batch_stride; ap.scale_batch_stride = scale_batch_stride; ap.max_slots = max_slots;
    ap.gs = (int)gs; ap.window = (int)window; ap.sink = (int)sink; ap.max_position = (int)max_position;
    qk_norm_rope_kernel<<<dim3((unsigned)((total + 15) / 16)), dim3(256), 16 * D * sizeof(uint16_t), st>>>(
        q, k, qw, kw, q_out, k_out, cos_t, sin_t, tokens, (int)Hq, (int)Hkv, (int)D, (int)seq_len, eps, ap);
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
    if (op == OP_SILU) unary_kernel<OP_SILU, false><<<grid, block, 0, st>>>(in, nullptr, out, n);
    else if (op == OP_GELU) unary_kernel<OP_GELU, false><<<grid, block, 0, st>>>(in, nullptr, out, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_unary_mul(int op, const uint16_t* in, const uint16_t* mul, uint16_t* out, size_t n, hipStream_t st) {
    const dim3 grid((unsigned)(((n + 7) / 8 + 255) / 256)), block(256);
    if (op == OP_SILU) unary_kernel<OP_SILU, true><<<grid, block, 0, st>>>(in, mul, out, n);
    else if (op == OP_GELU) unary_kernel<OP_GELU, true><<<grid, block, 0, st>>>(in, mul, out, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// mode 0: rms norm of a (+ b if given), 1: silu(a) * b, 2: gelu(a) * b; then row quantization
hipError_t launch_row_quant(int mode, const uint16_t* a, const uint16_t* b, const uint16_t* w, uint16_t* sum_out,
                            uint16_t* y_out, int8_t* q_out, float* scales_out, size_t rows, size_t dims, float eps,
                            hipStream_t st) {
    if (rows == 0 || dims == 0) return hipSuccess;
    if (rows > 0x7FFFFFFFull || dims > 16384) return hipErrorInvalidValue;   // two binary16 rows in LDS
    const dim3 grid((unsigned)rows), block(256);
    const size_t lds = 2 * dims * sizeof(uint16_t);
    const size_t vend = (rows * dims) & ~(size_t)7;
    if (mode == 0) row_quant_kernel<0><<<grid, block, lds, st>>>(a, b, w, sum_out, y_out, q_out, scales_out, (int)dims, eps, vend);
    else if (mode == 1) row_quant_kernel<1><<<grid, block, lds, st>>>(a, b, w, sum_out, y_out, q_out, scales_out, (int)dims, eps, vend);
    else if (mode == 2) row_quant_kernel<2><<<grid, block, lds, st>>>(a, b, w, sum_out, y_out, q_out, scales_out, (int)dims, eps, vend);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_scalar_op(int op, const uint16_t* a, float scalar, uint16_t* out, size_t n, hipStream_t st) {
    const dim3 grid((unsigned)(((n + 7) / 8 + 255) / 256)), block(256);
    const uint16_t sb = __builtin_bit_cast(uint16_t, (_Float16)scalar);   // static_cast<__fp16>(scalar_value)
    switch (op) {
        case BIN_ADD: scalar_kernel<BIN_ADD><<<grid, block, 0, st>>>(a, sb, out, n); break;
        case BIN_SUB: scalar_kernel<BIN_SUB><<<grid, block, 0, st>>>(a, sb, out, n); break;
        case BIN_MUL: scalar_kernel<BIN_MUL><<<grid, block, 0, st>>>(a, sb, out, n); break;
        case BIN_DIV: scalar_kernel<BIN_DIV><<<grid, block, 0, st>>>(a, sb, out, n); break;
        default: return hipErrorInvalidValue;
    }
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
