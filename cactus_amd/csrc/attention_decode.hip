// Decode attention over the INT8 KV cache on the matrix cores
// (cactus_attention_hybrid_int8_fp16, kernel_attention.cpp:257-482, for one query position per
// sequence: Model::decode, and the batched form of it).
//
// Roofline: HBM.  Algorithmic bytes per sequence = 2*cache_len*Hkv*(D + 4) + new K/V + q + out.
//
// One wave per (sequence, kv head, key range): it streams the range's 32-key blocks and finishes its
// output (or its split record) from registers, so a workgroup is just 4 (or 8) neighbouring items - the kv
// heads of one sequence, whose rows are adjacent in memory - with no barrier and no LDS exchange.  Every
// query head of the kv head (GQA, up to 16) rides in the same MFMAs, so a cached byte is fetched and
// converted once whatever the group size, and the per-byte VALU work does not grow with it: one v_xor
// per 4 values and one v_perm per 2.
//
// Everything is computed transposed, so that a lane owns a query head (lane % 16):
//   S^T = K * Q^T   v_mfma_f32_16x16x32_f16, A = 16 keys x 32 dims, B = Q^T (columns = heads).  A lane
//                   loads 16 contiguous bytes of one key row straight from global memory (a wave
//                   instruction covers 16 rows x 64 bytes); the MFMA's contraction order is free as
//                   long as both operands use it, so Q is laid out to match and K needs no staging.
//   O^T = V^T * P^T A = 16 dims x 32 keys needs, per lane, 8 keys of one dim: the transpose of what
//                   memory holds.  The raw INT8 rows go global -> LDS by LDS-DMA (no registers), and
//                   ds_read_b64_tr_b16 reads them back transposed, two adjacent dims at a time (a
//                   pair of bytes is its 16-bit element); B = P^T is the lane's own weights, in the
//                   key order the first MFMA left them in (no shuffle).
// INT8 -> binary16 is exact and costs no arithmetic: byte ^ 0x80 under the high byte 0x64 is the half
// 1152 + value.  The constant is not subtracted per element: sum_d q[d] (1152 + k[d]) = 1152 sum(q) +
// q.k, so the score subtracts 1152 * sum(q) once, and O subtracts 1152 * (sum of the weights fed to
// the MFMA, summed by an MFMA from the same operand) block by block.  Weights enter as hi + lo binary16
// parts (about 22 bits).
// Per-key scales multiply the score column (K) and the weight (V), as the reference's dequantization
// does per element.  With two scale groups per row (head_dim 256) each group's 128 dims have their own score
// chain, scaled and added, and their own weights (p * the group's V scale) for their half of the output dims.  Softmax is the online form on base-2 scores (s * log2 e, one v_exp_f32 per weight);
// key ranges (splits) leave (max, sum, O) records that attn_merge_kernel reduces.
// The LDS reads of the block loop are inline asm with hand-placed s_waitcnt lgkmcnt: the compiler would
// order an LDS read it can see behind every outstanding LDS-DMA request, the next block's prefetch
// included, and the loop would stop overlapping loads with arithmetic.
#include "common.hip.h"
#include "kernels.h"
#include "block_math.hip.h"

#include <math.h>

#include <type_traits>

namespace cactus_hip {

namespace {

constexpr int DC_BLOCK = 32;     // keys per block
#ifndef DC_AHEAD_N
#define DC_AHEAD_N 1
#endif
// Blocks a wave requests ahead of the one it computes, head_dim 64 / 128: K rows into DC_AHEAD + 1 register sets, V rows
// and both scales into as many LDS stages per wave.  head_dim 256 always has 1 (its stages are 8.7 KB).  Measured from
// HBM (cache sets in turn) with the block arithmetic compiled out (-DDC_ABLATE_STREAM: requests, waits and launch shape
// only): 256 sequences x 512 tokens 54.4 / 54.1 / 54.4 us with 1 / 2 / 3 blocks ahead, 64 x 4096 102.1 / 100.6 / 100.6,
// LFM2 heads 1024 x 512 116.4 / 118.3 / 117.4 - the depth of the ring is not what bounds the stream, and the arithmetic
// adds 1 to 2 us to it.  With the arithmetic, 2 ahead needs 252 registers + 36 at head_dim 128 (one wave per SIMD,
// 60.9 us) and is slower at head_dim 64 too (127.0 against 120.5), so 1 it stays.
constexpr int DC_AHEAD = DC_AHEAD_N;
constexpr int DC_MAX_NEW = 8;    // FP16 tokens of the step itself (decode: 1)

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __fp16 v2h __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v8h as_v8h(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const v4i v = {(int)a, (int)b, (int)c, (int)d};
    return __builtin_bit_cast(v8h, v);
}
// two INT8 (already ^ 0x80) -> two halves 1152 + value
__device__ __forceinline__ uint32_t pair_lo(uint32_t x) { return __builtin_amdgcn_perm(0x64646464u, x, 0x04010400u); }   // bytes 0, 1
__device__ __forceinline__ uint32_t pair_hi(uint32_t x) { return __builtin_amdgcn_perm(0x64646464u, x, 0x04030402u); }   // bytes 2, 3
__device__ __forceinline__ uint32_t pair_even(uint32_t x) { return __builtin_amdgcn_perm(0x64646464u, x, 0x04020400u); } // bytes 0, 2
__device__ __forceinline__ uint32_t pair_odd(uint32_t x) { return __builtin_amdgcn_perm(0x64646464u, x, 0x04030401u); }  // bytes 1, 3

// LDS byte address of a pointer into a __shared__ array
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// LDS reads the compiler does not see (see compute_block); the caller waits with s_waitcnt lgkmcnt
__device__ __forceinline__ v4f lds_read_b128(uint32_t addr) {
    v4f r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ uint2 lds_read_tr16(uint32_t addr) {
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}

// Lanes l, l ^ 16 (and l ^ 32) combine a value on the vector ALU: v_permlane16_swap / v_permlane32_swap exchange the
// odd 16-lane rows (upper 32 lanes) of the first operand with the even rows (lower 32) of the second, so from two
// copies of x one result holds the even rows' values in both rows of a pair and the other the odd rows' - no trip
// through the LDS crossbar as a shuffle takes.  Sum and maximum are symmetric, so both lanes get the same bits.
__device__ __forceinline__ float pair16_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair32_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair16_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pair32_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the four lanes l % 16 + {0, 16, 32, 48}
__device__ __forceinline__ float quad_rows_max(float x) { return pair32_max(pair16_max(x)); }
__device__ __forceinline__ float quad_rows_sum(float x) { return pair32_sum(pair16_sum(x)); }

struct DecodeParams {
    AttnArgs a;
    int nsplit;
    int n_items;  // batch * kv heads
    int gq;       // query heads per kv head (<= 16)
    float* ws;
    AttnFusedIn f;    // FUSED kernels: what turns the projections' rows into q and k, and the cache to append to
};

// chunk swizzle of the V image: 16-byte chunk c of key row r sits at chunk c ^ swz(r), which keeps the
// transposed reads of a 32-lane half (8 rows x 32 bytes) on 64 distinct banks
template <int D>
__device__ __forceinline__ int v_swz(int row) {
    return D == 256 ? (row & 7) << 1 : D == 128 ? ((row >> 1) & 3) << 1 : ((row >> 2) & 1) << 1;
}

// FUSED: a.q / a.k_new / a.v_new are the q / k / v projections' rows of a decode step (one new token per
// sequence).  Each wave first runs what qk_norm_rope_kernel does for its rows - per-head RMS norm and RoPE of its
// query heads and of its kv head's key, with that kernel's arithmetic (16 lanes per row, rows in LDS) - and the
// wave of key range 0 also quantizes the rotated key and the value into the cache at slot cache_lens[b]
// (engine_cache.cpp:234-267), past the rows any wave reads.  One launch less per layer and token.
// MERGE: the NW waves of a workgroup are the NW key ranges of ONE item; their (max, sum, O) records meet in LDS
// (each wave's own, by then idle, ring region) and the workgroup writes the finished rows itself.  For launches of
// few items and short caches, where a second launch for the merge costs more than the attention (one decoding
// sequence: 8 items of 16 blocks).
// QOUT (with FUSED, one key range): the NW waves of a workgroup are the NW kv heads of ONE sequence, so the workgroup
// holds a whole output row [Hq * D] and also leaves it quantized for the output projection - the row's absmax over the
// waves, then act_quant_rowwise_kernel's arithmetic on the fp16 values (graph_ops_nn.cpp:35-66) - which saves the
// activation-quantization launch in front of that projection.
template <int D, int NW, bool FUSED, bool MERGE = false, bool QOUT = false>
__global__ __launch_bounds__(NW * 64) void attn_decode_mfma_kernel(const DecodeParams p) {
    constexpr int ROWB = D;                         // bytes per key row
    constexpr int VBYTES = DC_BLOCK * ROWB;         // V image of one block
    constexpr int STAGE = VBYTES + 512;             // + 32 * NG K scales, 256 B in all with what the request's idle lanes clear, the V scales likewise
    constexpr int KL = D / 64;                      // 16-byte loads per lane and 16-key group (D = 64: 1)
    constexpr int NC = D / 32;                      // 32-dim column blocks
    constexpr int NG = D == 256 ? 2 : 1;            // scale groups per row (head_dim 256: two groups of 128 dims)
    constexpr int CG = NC / NG;                     // column blocks per scale group
    constexpr int AHEAD = D == 256 ? 1 : DC_AHEAD;
    constexpr int DC_STAGES = AHEAD + 1, DC_KSETS = AHEAD + 1;
    constexpr int RING = NW * DC_STAGES * STAGE;
    constexpr int ROWS_OWN = FUSED && D == 256 ? 18 * D * 2 : 0;   // per wave: the fused rows when a ring stage cannot hold them
    static_assert(ROWS_OWN || !FUSED || 18 * D * 2 <= STAGE, "a stage holds the fused rows");
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING + (FUSED ? NW * D * 4 : 0) + (QOUT ? NW * 4 : 0) + NW * ROWS_OWN];   // + each wave's rotated key row and value row (+ its absmax) (+ rows)

    const AttnArgs& a = p.a;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hd = lane & 15, q = lane >> 4;         // hd: query head (B / D column) or key / dim row (A operand)
    const int Hkv = (int)a.Hkv, Hq = (int)a.Hq, gq = p.gq;
    const int split = MERGE ? w : (int)blockIdx.y;
    const int nsplit = MERGE ? NW : p.nsplit;
    const int item = MERGE ? (int)blockIdx.x : (int)blockIdx.x * NW + w;   // QOUT: NW = Hkv, a workgroup is a sequence
    if (item >= p.n_items) return;                  // wave-uniform (MERGE: never taken, every wave reaches the barrier)
    const int kvh = item % Hkv, b = item / Hkv;
    const int h0 = kvh * gq;

    // a device-side length is never trusted beyond the rows the caller vouched for (a.cache_len: the slot count)
    const int cl = a.cache_lens ? min(max(a.cache_lens[b], 0), (int)a.cache_len) : (int)a.cache_len;
    const int new_len = (int)a.new_len;
    const int KV = cl + new_len;
    const long long qabs = (long long)(a.pos_offsets ? (size_t)a.pos_offsets[b] : a.position_offset);
    long long kv_end = KV;
    if (a.is_causal) kv_end = min(kv_end, qabs + 1);          // kernel_attention.cpp:331-332
    const int c_cnt = (int)min((long long)cl, kv_end);        // visible cached keys
    const int nbc = (c_cnt + DC_BLOCK - 1) / DC_BLOCK;
    const int bps = (nbc + nsplit - 1) / nsplit;
    const int blk_first = split * bps, blk_last = min(nbc, blk_first + bps);

    float scale = a.scale;
    if (scale == 0.0f) scale = 1.0f / sqrtf((float)D);        // :278-280

    const int8_t* Kc = a.k_cached + (size_t)b * a.cache_batch_stride;
    const int8_t* Vc = a.v_cached + (size_t)b * a.cache_batch_stride;
    const float* Ks = a.k_scales + (size_t)b * a.scale_batch_stride;
    const float* Vs = a.v_scales + (size_t)b * a.scale_batch_stride;
#ifdef DC_ABLATE_LINEAR
    // measurement build (with DC_ABLATE_STREAM): the same bytes read as if the cache were [kv head][slot][D]
    const size_t row_stride = (size_t)D;
    const unsigned head_off = (unsigned)kvh * (unsigned)c_cnt * D;
    const unsigned valid_bytes = (unsigned)c_cnt * (unsigned)Hkv * D;
#else
    const size_t row_stride = (size_t)Hkv * D;
    const unsigned head_off = (unsigned)kvh * D;
#endif

    uint8_t* my_lds = lds + w * (DC_STAGES * STAGE);
    uint16_t* kn_lds = reinterpret_cast<uint16_t*>(lds + RING) + (FUSED ? w * 2 * D : 0);
    uint16_t* vn_lds = kn_lds + D;       // FUSED: the step's own value row, kept for the last softmax step (no load at the tail)
    uint8_t* rows_lds = ROWS_OWN ? lds + RING + (FUSED ? NW * D * 4 : 0) + (QOUT ? NW * 4 : 0) + w * ROWS_OWN : my_lds + (DC_STAGES - 1) * STAGE;
    struct KRegs { v4i k[2][KL]; };
    // per-lane pieces of the addresses
    constexpr int ROW_CHUNKS = D / 16;                               // 16-byte chunks of a row
    const int dma_row = lane / ROW_CHUNKS;                           // row inside one DMA instruction
    const int dma_chunk = lane % ROW_CHUNKS;
    constexpr int DMA_ROWS = 64 / ROW_CHUNKS;
    constexpr int NDMA = DC_BLOCK / DMA_ROWS;

    // K, V and the scales come through buffer descriptors that end with the last visible cached row: rows
    // past it (the tail of a range's last block, and whole blocks past this wave's share, whose offsets are
    // pushed out of range) read as zeros and cost no memory traffic.  Requests are therefore made
    // unconditionally and the block loop has no branch around a memory operation (a branch there turns
    // every counted wait into a wait for everything).
#ifndef DC_ABLATE_LINEAR
    const unsigned valid_bytes = (unsigned)c_cnt * (unsigned)row_stride;
#endif
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(Kc), 0, (int)valid_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(Vc), 0, (int)valid_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ksrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ks), 0, c_cnt * Hkv * NG * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vsrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vs), 0, c_cnt * Hkv * NG * 4, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned k_lane_off[2], v_lane_off[NDMA];
#pragma unroll
    for (int g = 0; g < 2; g++) k_lane_off[g] = (unsigned)(16 * g + hd) * (unsigned)row_stride + head_off + 16 * q;
#pragma unroll
    for (int i = 0; i < NDMA; i++) {
        const int row = i * DMA_ROWS + dma_row;
        v_lane_off[i] = (unsigned)row * (unsigned)row_stride + head_off + ((dma_chunk ^ v_swz<D>(row)) << 4);
    }
    // one scale per lane and request: lanes 0..32 NG - 1 the block's 32 keys (NG = 2: key lane / 2, group lane % 2),
    // the other lanes nothing
    const unsigned s_lane_off = lane < 32 * NG ? (unsigned)(((lane / NG) * Hkv + kvh) * NG + lane % NG) * 4u : OOB;

    // K: 16 keys x 64 bytes per wave instruction, straight to registers
    auto issue_k = [&](int blk, KRegs& kr) __attribute__((always_inline)) {
        const unsigned base = blk < blk_last ? (unsigned)blk * DC_BLOCK * (unsigned)row_stride : OOB;
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int ld = 0; ld < KL; ld++)
                kr.k[g][ld] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(krsrc, (int)(base + k_lane_off[g]) + 64 * ld, 0, 0));
    };
    // V: raw rows into the LDS image (lane-linear per instruction, swizzled through the source address), and
    // the scales: each request writes its 32 scales and, from its idle lanes, 128 bytes of zeros behind them
    auto issue_v = [&](int blk, int stage) __attribute__((always_inline)) {
        const unsigned base = blk < blk_last ? (unsigned)blk * DC_BLOCK * (unsigned)row_stride : OOB;
        const unsigned sbase = blk < blk_last ? (unsigned)blk * DC_BLOCK * (unsigned)(Hkv * NG) * 4u : OOB;
        uint8_t* st = my_lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NDMA; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lptr_t)(st + i * 1024), 16, (int)(base + v_lane_off[i]), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ksrsrc, (lptr_t)(st + VBYTES), 4, (int)(sbase + s_lane_off), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrsrc, (lptr_t)(st + VBYTES + 256), 4, (int)(sbase + s_lane_off), 0, 0, 0);
    };

    KRegs kr[DC_KSETS];
    // The first block of the wave's range is requested before the wave turns to its rows (FUSED: their loads, tables
    // and arithmetic; otherwise the q loads), which would otherwise all precede the first byte of the cache stream
    // (the plain launch: 256 sequences x 512 tokens 58.8 -> 55.2 us, LFM2 heads 1024 x 512 120.2 -> 117.4;
    // -DDC_LATE_FIRST is the old order); FUSED rows then sit in the ring's
    // last stage, which later requests overwrite once the q rows are in registers
    // (head_dim 256: up to 18 rows of 512 bytes do not fit a stage and have an area of their own).  The requests are
    // unconditional like all others: a wave without blocks reads nothing.
#ifdef DC_LATE_FIRST
    constexpr bool early = FUSED;
#else
    constexpr bool early = true;
#endif
    // request order: K(first), V(first) .. K(first + AHEAD - 1), V(first + AHEAD - 1); then per block j: K(j + AHEAD), V(j + AHEAD)
    auto issue_first = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            issue_k(blk_first + u, kr[u]);
            issue_v(blk_first + u, u);
        }
    };
    if constexpr (early) issue_first();
    if constexpr (FUSED) {
        // rows of this wave, 16 lanes each, four at a time: its query heads, its key, and (key range 0) its value;
        // they sit in the still idle ring: [gq] normalized + rotated q rows, the k row, the raw v row
        const AttnFusedIn& f = p.f;
        uint16_t* rows = reinterpret_cast<uint16_t*>(rows_lds);
        const int l = lane & 15, g = lane >> 4, base = lane & ~15;
        const int nrows = gq + (split == 0 ? 2 : 0);      // only key range 0 uses the step's own key and value
        const int pos = max(0, min((int)qabs, f.max_position));
        const bool place = split == 0 && (size_t)cl < f.max_slots;
        const size_t doff = (size_t)b * a.cache_batch_stride + ((size_t)cl * Hkv + kvh) * D;
        const size_t soff = (size_t)b * a.scale_batch_stride + ((size_t)cl * Hkv + kvh) * (size_t)((D + (int)a.qgs - 1) / (int)a.qgs);
        for (int r0 = 0; r0 < nrows; r0 += 4) {
            const bool act = r0 + g < nrows;
            const int r = min(r0 + g, nrows - 1);
            const bool is_q = r < gq, is_k = r == gq, is_v = r > gq;
            const uint16_t* src = is_q ? a.q + ((size_t)b * Hq + h0 + r) * D
                                       : (is_k ? a.k_new : a.v_new) + ((size_t)b * Hkv + kvh) * D;
            uint16_t* nrm = rows + (size_t)r * D;
            // everything the row needs from memory is requested at once: its halves, the norm weights of the lane's
            // elements, the rotation factors of its pairs (one round trip instead of three dependent ones)
            constexpr int RC = (D / 8 + 15) / 16;
            v4i rowv[RC];
            uint16_t wreg[D / 16], creg[D / 32], sreg[D / 32];
            const uint16_t* wv = is_q ? f.q_weight : f.k_weight;
#pragma unroll
            for (int j = 0; j < RC; j++) rowv[j] = *reinterpret_cast<const v4i*>(src + min(l + 16 * j, D / 8 - 1) * 8);
#pragma unroll
            for (int j = 0; j < D / 16; j++) wreg[j] = wv[l + 16 * j];
#pragma unroll
            for (int j = 0; j < D / 32; j++) {
                creg[j] = f.cos_t[(size_t)pos * (D / 2) + l + 16 * j];
                sreg[j] = f.sin_t[(size_t)pos * (D / 2) + l + 16 * j];
            }
            if (act)
#pragma unroll
                for (int j = 0; j < RC; j++)
                    if (l + 16 * j < D / 8) *reinterpret_cast<v4i*>(nrm + (l + 16 * j) * 8) = rowv[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float inv = row_inv_rms([&](int i) -> uint16_t { return nrm[i]; }, D, f.eps, l, base);
            const h16 inv_h = (h16)inv;
            if (act && !is_v) {
#pragma unroll
                for (int j = 0; j < D / 16; j++) nrm[l + 16 * j] = rms_out(nrm[l + 16 * j], wreg[j], inv, inv_h, true);   // own elements
#pragma unroll
                for (int j = 0; j < D / 32; j++) {                                                                       // ... and their RoPE pairs
                    const int i = l + 16 * j;
                    uint16_t o1, o2;
                    rope_pair(creg[j], sreg[j], nrm[i], nrm[D / 2 + i], o1, o2);
                    nrm[i] = o1;
                    nrm[D / 2 + i] = o2;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (act && is_k) {
                for (int c = l; c < D / 8; c += 16) *reinterpret_cast<v4i*>(kn_lds + c * 8) = *reinterpret_cast<const v4i*>(nrm + c * 8);
                if (f.k_out)      // key range 0 only (the other ranges do not process the key): one writer per row
                    for (int c = l; c < D / 8; c += 16)
                        *reinterpret_cast<v4i*>(f.k_out + ((size_t)b * Hkv + kvh) * D + c * 8) = *reinterpret_cast<const v4i*>(nrm + c * 8);
            }
            if (act && is_v)
                for (int c = l; c < D / 8; c += 16) *reinterpret_cast<v4i*>(vn_lds + c * 8) = *reinterpret_cast<const v4i*>(nrm + c * 8);
            if (act && place && !is_q)
                quant_row_groups(nrm, D, (int)a.qgs, (is_k ? f.kc : f.vc) + doff, (is_k ? f.ks : f.vs) + soff, l);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // Q^T operand: lane (head, q) holds, for MFMA step (ld, half), Q[head][64 ld + 16 q + 8 half .. + 8)
    v8h qreg[KL * 2];
    float qsum[NG];
#pragma unroll
    for (int n = 0; n < NG; n++) qsum[n] = 0.0f;
    {
        const uint16_t* qrow = FUSED ? reinterpret_cast<const uint16_t*>(rows_lds) + (size_t)min(hd, gq - 1) * D
                                     : a.q + ((size_t)b * Hq + h0 + min(hd, gq - 1)) * D;
#pragma unroll
        for (int s = 0; s < KL * 2; s++) {
            v4i raw = *reinterpret_cast<const v4i*>(qrow + 64 * (s >> 1) + 16 * q + 8 * (s & 1));
            if (hd >= gq) raw = v4i{0, 0, 0, 0};
            qreg[s] = __builtin_bit_cast(v8h, raw);
#pragma unroll
            for (int e = 0; e < 8; e++) qsum[(s >> 1) * NG / KL] += (float)qreg[s][e];
        }
#pragma unroll
        for (int n = 0; n < NG; n++) {
            qsum[n] = quad_rows_sum(qsum[n]);
        }
    }
    float nqsum[NG];        // per scale group: the group's scores are rescaled separately
#pragma unroll
    for (int n = 0; n < NG; n++) nqsum[n] = -1152.0f * qsum[n];
    const float scale2 = scale * 1.44269504f;     // scores are kept times log2(e)
    if constexpr (FUSED) {      // the q rows are in registers: the ring may be written from here on
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // m_run is the running maximum minus 6 (scores are base-2 logarithms): weights are 2^(s - m_run) <= 64, which
    // keeps the binary16 parts of small weights out of the subnormal range; numerator and denominator carry
    // the same factor.  pend: 1152 * (sum of the last block's weights as the MFMA saw them), still inside oacc.
    float m_run = -INFINITY, l_part = 0.0f, pend[NG];
#pragma unroll
    for (int n = 0; n < NG; n++) pend[n] = 0.0f;
    v4f oacc[NC][2];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int e = 0; e < 2; e++) oacc[c][e] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    // transposed-read addresses: lane i of group q supplies row 4q + (i >> 2), columns 4 (i & 3) .. + 4 of the block
    // (rows 16 .. 31 are the same lanes' addresses plus 16 rows: the swizzle has period 16 or less)
    uint32_t tr_off[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int row = 4 * q + (hd >> 2), pp = hd & 3;
        const int chunk = (2 * c + (pp >> 1)) ^ v_swz<D>(row);
        tr_off[c] = row * ROWB + chunk * 16 + 8 * (pp & 1);
    }

    // full: all 32 keys of the block are valid (every block but a range's last), so nothing is masked
    auto compute_block = [&](auto full_tag, int blk, int stage, KRegs& kr) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const uint8_t* st = my_lds + stage * STAGE;
#ifdef DC_ABLATE_STREAM
        // measurement build: the block's requests and waits as they are, none of its arithmetic (what the loads,
        // the ring and the launch shape alone allow)
        {
            int x = 0;
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int ld = 0; ld < KL; ld++) x ^= kr.k[g][ld][0] ^ kr.k[g][ld][1] ^ kr.k[g][ld][2] ^ kr.k[g][ld][3];
            v4f t = lds_read_b128(lds_addr(st) + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t));
            oacc[0][0][0] += __int_as_float(x & 0xFF) + t[0];
            return;
        }
#endif
        const int kb = blk * DC_BLOCK;
        const int nv = min(DC_BLOCK, c_cnt - kb);            // valid keys of the block (wave-uniform)
        // Every LDS read of the loop is issued here, by hand: the compiler orders an LDS read it can see behind
        // ALL outstanding LDS-DMA requests (a wait for the next block's prefetch as well), which would take
        // the overlap of loads and arithmetic away.  The stage's requests are complete (the caller waited).
        const uint32_t sa = lds_addr(st);
        // a lane's scales: keys 16 g + 4 q .. + 4, NG groups each (key-major: NG = 2 interleaves the groups)
        v4f ksv[2][NG], vsv[2][NG];
        uint2 tv[2][NC];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int n = 0; n < NG; n++) {
                ksv[g][n] = lds_read_b128(sa + VBYTES + NG * (64 * g + 16 * q) + 16 * n);
                vsv[g][n] = lds_read_b128(sa + VBYTES + 256 + NG * (64 * g + 16 * q) + 16 * n);
            }
        // scale of key j (0..3) of the lane's four, group n
        auto pick = [&](const v4f (&sv)[NG], int j, int n) __attribute__((always_inline)) -> float {
            return NG == 1 ? sv[0][j] : sv[j >> 1][2 * (j & 1) + n];
        };
        // ---- scores: S^T[key][head]; the two key groups' accumulation chains alternate.  The chains start
        // at -1152 * sum(q), which takes the conversion's constant out again.
        v4f acc[2][NG];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int n = 0; n < NG; n++) acc[g][n] = v4f{nqsum[n], nqsum[n], nqsum[n], nqsum[n]};
#pragma unroll
        for (int ld = 0; ld < KL; ld++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const uint32_t x0 = (uint32_t)kr.k[g][ld][2 * hf] ^ 0x80808080u;
                    const uint32_t x1 = (uint32_t)kr.k[g][ld][2 * hf + 1] ^ 0x80808080u;
                    const v8h ka = as_v8h(pair_lo(x0), pair_hi(x0), pair_lo(x1), pair_hi(x1));
                    acc[g][ld * NG / KL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qreg[2 * ld + hf], acc[g][ld * NG / KL], 0, 0, 0);
                }
        if constexpr (NG == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ksv[0][0]), "+v"(ksv[0][1]), "+v"(ksv[1][0]), "+v"(ksv[1][1]),
                         "+v"(vsv[0][0]), "+v"(vsv[0][1]), "+v"(vsv[1][0]), "+v"(vsv[1][1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ksv[0][0]), "+v"(ksv[1][0]), "+v"(vsv[0][0]), "+v"(vsv[1][0]));
        // scores in the base-2 domain: s * log2(e), so that the weights are one v_exp_f32 each
        float sc[2][4];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                sc[g][j] = acc[g][0][j] * (pick(ksv[g], j, 0) * scale2);
                if constexpr (NG == 2) sc[g][j] += acc[g][1][j] * (pick(ksv[g], j, 1) * scale2);
                if (!FULL && 16 * g + 4 * q + j >= nv) sc[g][j] = -INFINITY;
            }
        // ---- online softmax, per head (lane % 16); a head's 32 keys sit in its 4 lanes (q = 0..3)
        float bm = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])),
                         fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
        bm = quad_rows_max(bm);
        // the V image's transposed reads: in flight while the weights are computed
#pragma unroll
        for (int c = 0; c < NC; c++) {
            tv[0][c] = lds_read_tr16(sa + tr_off[c]);
            tv[1][c] = lds_read_tr16(sa + tr_off[c] + 16 * ROWB);
        }
        // unconditional (a factor of 1 when the maximum stays): a branch around it makes the compiler copy
        // every accumulator on the path that skips it
        {
            const float mn = fmaxf(m_run, bm - 6.0f);          // the block holds a valid key: finite
            const float corr = __builtin_amdgcn_exp2f(m_run - mn);   // m_run = -inf -> 0
            l_part *= corr;
            // the rescale also takes out the constant the last block's V conversion added (1152 per weight):
            // the accumulators never hold more than one block's worth of it, so what the subtraction costs in
            // precision does not grow with the context
            float npc[NG];
#pragma unroll
            for (int n = 0; n < NG; n++) npc[n] = -pend[n] * corr;
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int j = 0; j < 4; j++) oacc[c][e][j] = fmaf(oacc[c][e][j], corr, npc[c / CG]);
            m_run = mn;
        }
        // weights: w = p * v_scale as hi + lo halves (hi rounded toward zero, lo the rest: 21 bits or more)
        uint32_t ph[NG][4], pl[NG][4];
        float bsum = 0.0f;
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const float p0 = __builtin_amdgcn_exp2f(sc[g][2 * jj] - m_run);      // a masked key: 2^-inf = 0
                const float p1 = __builtin_amdgcn_exp2f(sc[g][2 * jj + 1] - m_run);
                bsum += p0 + p1;
#pragma unroll
                for (int n = 0; n < NG; n++) {
                    const float w0 = p0 * pick(vsv[g], 2 * jj, n), w1 = p1 * pick(vsv[g], 2 * jj + 1, n);
                    const v2h hi = __builtin_amdgcn_cvt_pkrtz(w0, w1);
                    const v2h lo = __builtin_amdgcn_cvt_pkrtz(w0 - (float)hi[0], w1 - (float)hi[1]);
                    ph[n][2 * g + jj] = __builtin_bit_cast(uint32_t, hi);
                    pl[n][2 * g + jj] = __builtin_bit_cast(uint32_t, lo);
                }
            }
        l_part += bsum;
        v8h phi[NG], plo[NG];
#pragma unroll
        for (int n = 0; n < NG; n++) {
            phi[n] = as_v8h(ph[n][0], ph[n][1], ph[n][2], ph[n][3]);
            plo[n] = as_v8h(pl[n][0], pl[n][1], pl[n][2], pl[n][3]);
        }
        if constexpr (NC == 8)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0][0]), "+v"(tv[1][0]), "+v"(tv[0][1]), "+v"(tv[1][1]), "+v"(tv[0][2]),
                         "+v"(tv[1][2]), "+v"(tv[0][3]), "+v"(tv[1][3]), "+v"(tv[0][4]), "+v"(tv[1][4]), "+v"(tv[0][5]),
                         "+v"(tv[1][5]), "+v"(tv[0][6]), "+v"(tv[1][6]), "+v"(tv[0][7]), "+v"(tv[1][7]));
        else if constexpr (NC == 4)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0][0]), "+v"(tv[1][0]), "+v"(tv[0][1]), "+v"(tv[1][1]), "+v"(tv[0][2]),
                         "+v"(tv[1][2]), "+v"(tv[0][3]), "+v"(tv[1][3]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0][0]), "+v"(tv[1][0]), "+v"(tv[0][1]), "+v"(tv[1][1]));
        // the block's sum of weights exactly as the MFMAs below see them (a row of ones as the A operand: every
        // row of the result is the column's sum), for the constant's removal
#pragma unroll
        for (int n = 0; n < NG; n++) {
            const v8h ones = as_v8h(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
            v4f wb = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, phi[n], v4f{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
            wb = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, plo[n], wb, 0, 0, 0);
            pend[n] = 1152.0f * wb[0];
        }
        // ---- O^T[dim][head] += V^T * P^T, 32 dims per column block: even dims and odd dims alternate
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint2 ua = tv[0][c], ub = tv[1][c];
            const uint32_t x0 = ua.x ^ 0x80808080u, x1 = ua.y ^ 0x80808080u;
            const uint32_t x2 = ub.x ^ 0x80808080u, x3 = ub.y ^ 0x80808080u;
            const v8h ve = as_v8h(pair_even(x0), pair_even(x1), pair_even(x2), pair_even(x3));
            const v8h vo = as_v8h(pair_odd(x0), pair_odd(x1), pair_odd(x2), pair_odd(x3));
            oacc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ve, phi[c / CG], oacc[c][0], 0, 0, 0);
            oacc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vo, phi[c / CG], oacc[c][1], 0, 0, 0);
            oacc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ve, plo[c / CG], oacc[c][0], 0, 0, 0);
            oacc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vo, plo[c / CG], oacc[c][1], 0, 0, 0);
        }
    };

    // this wave's blocks, requested AHEAD blocks before their use: K into the register set, V into the LDS stage whose
    // block was consumed last; the counted waits leave exactly the requests issued after V(j) in flight when block j is
    // computed.  Requests are unconditional (past the wave's share they read nothing); only the arithmetic of
    // a block that does not exist is skipped.  Every lane runs every iteration: the transposed reads need
    // all lanes.  A stage is rewritten by the requests issued after its consumer, whose LDS reads have
    // completed by then (the consumer waited for them).
    if (blk_first < blk_last) {
        int blk = blk_first;
        const int mine = blk_last - blk;      // blocks of this wave
        auto compute = [&](int bk, int stage, KRegs& k) __attribute__((always_inline)) {
            if (bk * DC_BLOCK + DC_BLOCK <= c_cnt) compute_block(std::true_type{}, bk, stage, k);
            else compute_block(std::false_type{}, bk, stage, k);
        };
        constexpr int K_OPS = 2 * KL, V_OPS = NDMA + 2;    // vector-memory operations of a block's K and V requests
        // what may still be in flight when K(j) and V(j) must have landed: the requests issued after V(j)
        constexpr int LEAVE = AHEAD * (K_OPS + V_OPS);
        if constexpr (!early) issue_first();
        for (int it = 0; it < mine; it += DC_KSETS) {
#pragma unroll
            for (int u = 0; u < DC_KSETS; u++) {           // block blk sits in register set u and LDS stage u
                issue_k(blk + AHEAD, kr[(u + AHEAD) % DC_KSETS]);
                issue_v(blk + AHEAD, (u + AHEAD) % DC_STAGES);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEAVE) : "memory");
                if (it + u < mine) compute(blk, u, kr[u]);
                blk++;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last, empty requests still write this wave's LDS
    } else if constexpr (early) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the early requests of a wave without blocks
    }

    // ---- this wave's state -> LDS (the ring is idle now), then the workgroup's merge
    l_part = quad_rows_sum(l_part);
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int j = 0; j < 4; j++) oacc[c][e][j] -= pend[c / CG];
    const uint16_t* Kn = a.k_new + (size_t)b * new_len * Hkv * D;
    const uint16_t* Vn = a.v_new + (size_t)b * new_len * Hkv * D;
    const int n_new = split == 0 ? (int)max(0LL, min((long long)new_len, kv_end - cl)) : 0;   // visible ones
    // the FP16 tokens of the step itself (split 0 adds them) continue the online softmax; a lane holds its
    // head's Q dims 64 ld + 16 q + 8 half .. + 8 and output dims 32 c + 8 q .. + 8
    for (int t = 0; t < n_new; t++) {
        const uint16_t* kn = Kn + ((size_t)t * Hkv + kvh) * D;
        const uint16_t* vn = Vn + ((size_t)t * Hkv + kvh) * D;
        float part = 0.0f;
#pragma unroll
        for (int st = 0; st < KL * 2; st++) {
            const int koff = 64 * (st >> 1) + 16 * q + 8 * (st & 1);
            const v8h kv = FUSED ? *reinterpret_cast<const v8h*>(kn_lds + koff)      // the key this wave rotated itself
                                 : *reinterpret_cast<const v8h*>(kn + koff);
#pragma unroll
            for (int e = 0; e < 8; e++) part = fmaf((float)qreg[st][e], (float)kv[e], part);
        }
        part = quad_rows_sum(part);
        const float s2 = part * scale2;
        const float mn = fmaxf(m_run, s2 - 6.0f);
        const float corr = __builtin_amdgcn_exp2f(m_run - mn), pt = __builtin_amdgcn_exp2f(s2 - mn);
        l_part = l_part * corr + pt;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const v8h vv = FUSED ? *reinterpret_cast<const v8h*>(vn_lds + 32 * c + 8 * q)
                                 : *reinterpret_cast<const v8h*>(vn + 32 * c + 8 * q);
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int j = 0; j < 4; j++) oacc[c][e][j] = fmaf((float)vv[2 * j + e], pt, oacc[c][e][j] * corr);
        }
        m_run = mn;
    }
    if constexpr (MERGE) {
        // this wave's record -> its own ring region: [head][m, l, O[D]] (every request of the wave has landed:
        // rows past its range were still requested, as zeros)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* recs = reinterpret_cast<float*>(my_lds);
        static_assert(16 * (D + 2) * 4 <= DC_STAGES * STAGE, "record fits the wave's ring region");
        if (hd < gq) {
            float* rec = recs + hd * (D + 2);
            if (q == 0) {
                rec[0] = m_run;
                rec[1] = l_part;
            }
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float2*>(rec + 2 + 32 * c + 8 * q + 2 * j) = make_float2(oacc[c][0][j], oacc[c][1][j]);
        }
        __syncthreads();
        // TPH threads per head, each four dims at a time; factors 2^(m_s - M) (scores are base 2 here)
        constexpr int TPH = NW * 4;
        const int head = (int)threadIdx.x / TPH, part = (int)threadIdx.x % TPH;
        if (head < gq) {
            const float* r0 = reinterpret_cast<const float*>(lds) + head * (D + 2);
            constexpr int WSTR = DC_STAGES * STAGE / 4;     // floats between two waves' regions
            float M = -INFINITY;
#pragma unroll
            for (int sp = 0; sp < NW; sp++) M = fmaxf(M, r0[sp * WSTR]);
            float f[NW], L = 0.0f;
#pragma unroll
            for (int sp = 0; sp < NW; sp++) {
                f[sp] = M > -INFINITY ? __builtin_amdgcn_exp2f(r0[sp * WSTR] - M) : 0.0f;
                L = fmaf(r0[sp * WSTR + 1], f[sp], L);
            }
            const float inv = 1.0f / L;
            const size_t orow = (size_t)b * Hq + h0 + head;
            for (int d4 = part; d4 < D / 4; d4 += TPH) {
                float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int sp = 0; sp < NW; sp++) {
                    const float2 v0 = *reinterpret_cast<const float2*>(r0 + sp * WSTR + 2 + 4 * d4);
                    const float2 v1 = *reinterpret_cast<const float2*>(r0 + sp * WSTR + 4 + 4 * d4);
                    o[0] = fmaf(v0.x, f[sp], o[0]);
                    o[1] = fmaf(v0.y, f[sp], o[1]);
                    o[2] = fmaf(v1.x, f[sp], o[2]);
                    o[3] = fmaf(v1.y, f[sp], o[3]);
                }
                uint32_t pk[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const uint16_t lo = L > 0.0f ? f2h_mul(o[2 * j], inv) : (uint16_t)0;       // :460-479
                    const uint16_t hi = L > 0.0f ? f2h_mul(o[2 * j + 1], inv) : (uint16_t)0;
                    pk[j] = (uint32_t)lo | ((uint32_t)hi << 16);
                }
                *reinterpret_cast<uint2*>(a.out + orow * D + 4 * d4) = make_uint2(pk[0], pk[1]);
            }
        }
        return;
    }
    if constexpr (QOUT) {
        const size_t orow = (size_t)b * Hq + h0 + min(hd, gq - 1);
        const bool live = hd < gq && l_part > 0.0f;     // :460-479
        const float inv = 1.0f / l_part;
        uint16_t hv[NC][8];
        float m = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const uint16_t v = live ? f2h_mul(oacc[c][e][j], inv) : (uint16_t)0;
                    hv[c][2 * j + e] = v;
                    m = fmaxf(m, fabsf(h2f(v)));
                }
        m = wave_max(m);
        float* smax = reinterpret_cast<float*>(lds + RING + (FUSED ? NW * D * 4 : 0));
        if (lane == 0) smax[w] = m;
        __syncthreads();
        float rm = smax[0];
#pragma unroll
        for (int i = 1; i < NW; i++) rm = fmaxf(rm, smax[i]);
        float qscale = rm / 127.0f;
        if (qscale < 1e-10f) qscale = 1e-10f;
        if (threadIdx.x == 0) p.f.q_scales_out[b] = qscale;
        const float iq = 1.0f / qscale;
        if (hd < gq) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                v4i packed;
#pragma unroll
                for (int j = 0; j < 4; j++) packed[j] = (int)((uint32_t)hv[c][2 * j] | ((uint32_t)hv[c][2 * j + 1] << 16));
                *reinterpret_cast<v4i*>(a.out + orow * D + 32 * c + 8 * q) = packed;
                *reinterpret_cast<uint2*>(p.f.q_out + orow * D + 32 * c + 8 * q) = quant8_bits(packed, iq);
            }
        }
        return;
    }
    if (hd >= gq) return;
    const size_t orow = (size_t)b * Hq + h0 + hd;
    if (p.nsplit == 1) {
        const float inv = 1.0f / l_part;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            uint32_t pk[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint16_t lo = l_part > 0.0f ? f2h_mul(oacc[c][0][j], inv) : (uint16_t)0;   // :460-479
                const uint16_t hi = l_part > 0.0f ? f2h_mul(oacc[c][1][j], inv) : (uint16_t)0;
                pk[j] = (uint32_t)lo | ((uint32_t)hi << 16);
            }
            *reinterpret_cast<v4i*>(a.out + orow * D + 32 * c + 8 * q) = v4i{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
        }
    } else {
        float* rec = p.ws + (orow * p.nsplit + split) * (size_t)(D + 2);
        if (q == 0) {
            rec[0] = m_run * 0.69314718f;        // the merge works in the natural-log domain (-inf stays); every
                                              // split's reference is 6 below its maximum, which cancels
            rec[1] = l_part;
        }
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float2*>(rec + 2 + 32 * c + 8 * q + 2 * j) = make_float2(oacc[c][0][j], oacc[c][1][j]);
    }
}

}  // namespace

// One query position per sequence over an INT8 cache with head_dim 64 or 128 and one scale group per row, or
// head_dim 256 and two groups of 128 (Gemma, model_gemma.cpp:67-113; KV_QUANT_GROUP_SIZE, kernel.h:18);
// up to 16 query heads per kv head and up to 8 FP16 tokens of the step itself.
bool attn_decode_mfma_ok(const AttnArgs& a) {
    if (a.f16_mode || a.S != 1 || a.new_len > (size_t)DC_MAX_NEW) return false;
    if (a.D != 64 && a.D != 128 && a.D != 256) return false;
    if ((a.D == 256 ? a.qgs != 128 : a.qgs < a.D) || a.Hkv == 0 || a.Hq % a.Hkv != 0 || a.Hq / a.Hkv > 16) return false;
    if (a.cache_len == 0 && a.cache_lens == nullptr) return false;           // nothing cached: the plain kernel
    if (!a.k_cached || !a.v_cached || !a.k_scales || !a.v_scales || !a.q || !a.out) return false;
    if (a.new_len && (!a.k_new || !a.v_new)) return false;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.k_cached) | reinterpret_cast<uintptr_t>(a.v_cached) |
                         reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.out) |
                         reinterpret_cast<uintptr_t>(a.k_new) | reinterpret_cast<uintptr_t>(a.v_new);
    if ((al & 15) != 0 || a.cache_batch_stride % 16 != 0) return false;
    if (a.B * a.Hkv > 0x7FFFFFFFull / 2 || a.cache_len + a.new_len >= (size_t(1) << 30)) return false;
    if (a.cache_len * a.Hkv * a.D >= (size_t(1) << 31)) return false;   // one buffer descriptor per sequence, 32-bit offsets
    return true;
}

// The fused launch can also leave the output rows quantized (AttnFusedIn::q_out) when a workgroup can hold a whole
// row: 8 kv heads of head_dim 128, one key range per item.  (At head_dim 64 the 8-wave workgroups it needs cost
// more than the launch they save: LFM2-shape step 1.32 -> 1.35 ms at 128 sequences.)
bool attn_decode_quant_ok(const AttnArgs& a, int planned_splits) {
    return planned_splits == 1 && a.Hkv == 8 && a.D == 128 && a.new_len == 1;
}

// Waves per workgroup when the key ranges of an item are merged inside the workgroup (0: not for this shape).
// It replaces the merge launch whenever the plan splits at all and an eighth (head_dim 256: a half) of the cache
// is at most 8 blocks per wave: short caches of few sequences, where a launch costs more than the work.
int attn_decode_merge_waves(const AttnArgs& a, int planned_splits) {
    if (planned_splits <= 1) return 0;
    const int nw = a.D == 256 ? 2 : 8;
    const size_t blocks = (a.cache_len + DC_BLOCK - 1) / DC_BLOCK;
    return blocks <= (size_t)nw * 8 ? nw : 0;
}

hipError_t launch_attention_decode_mfma(const AttnArgs& a, float* ws, int nsplit, hipStream_t st, const AttnFusedIn* fused) {
    DecodeParams p;
    p.a = a;
    p.nsplit = nsplit;
    p.n_items = (int)(a.B * a.Hkv);
    p.gq = (int)(a.Hq / a.Hkv);
    p.ws = ws;
    if (fused) p.f = *fused;
    if (fused && (a.new_len != 1 || !a.is_causal || ((!a.cache_lens || !a.pos_offsets) && a.B != 1))) return hipErrorInvalidValue;
    if (fused && fused->q_out) {   // the caller checked attn_decode_quant_ok
        const dim3 grid((unsigned)a.B);
        attn_decode_mfma_kernel<128, 8, true, false, true><<<grid, dim3(512), 0, st>>>(p);
        return hipGetLastError();
    }
    if (const int mw = attn_decode_merge_waves(a, nsplit)) {
        p.nsplit = mw;
        const dim3 grid((unsigned)p.n_items);
#define DECODE_MERGE_LAUNCH(F)                                                                     \
        if (a.D == 256) attn_decode_mfma_kernel<256, 2, F, true><<<grid, dim3(128), 0, st>>>(p);      \
        else if (a.D == 128) attn_decode_mfma_kernel<128, 8, F, true><<<grid, dim3(512), 0, st>>>(p); \
        else attn_decode_mfma_kernel<64, 8, F, true><<<grid, dim3(512), 0, st>>>(p);
        if (fused) { DECODE_MERGE_LAUNCH(true) } else { DECODE_MERGE_LAUNCH(false) }
#undef DECODE_MERGE_LAUNCH
        return hipGetLastError();
    }
    if (nsplit > 1 && ws == nullptr) return hipErrorInvalidValue;
    // 8 waves per workgroup cover a whole 8-head row of the cache; the wider workgroup pays when there are
    // many of them and their LDS rings are small (head_dim 64)
    const int nw = a.D == 256 ? 2 : (a.D == 64 && p.n_items >= 4096) ? 8 : 4;
    const dim3 grid((unsigned)((p.n_items + nw - 1) / nw), (unsigned)nsplit);
#define DECODE_LAUNCH(F)                                                                   \
    if (a.D == 256) attn_decode_mfma_kernel<256, 2, F><<<grid, dim3(128), 0, st>>>(p);      \
    else if (a.D == 128) attn_decode_mfma_kernel<128, 4, F><<<grid, dim3(256), 0, st>>>(p); \
    else if (nw == 8) attn_decode_mfma_kernel<64, 8, F><<<grid, dim3(512), 0, st>>>(p);     \
    else attn_decode_mfma_kernel<64, 4, F><<<grid, dim3(256), 0, st>>>(p);
    if (fused) { DECODE_LAUNCH(true) } else { DECODE_LAUNCH(false) }
#undef DECODE_LAUNCH
    return hipGetLastError();
}

}  // namespace cactus_hip
