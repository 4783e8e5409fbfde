/* Whole decode / prefill steps of the decoder models on the device-resident hot path, for a C or C++ engine.
 *
 * The reference builds one graph per token for ONE sequence (models/model_qwen.cpp:51-172,
 * models/model_lfm2.cpp:123-331; Model::decode / Model::prefill, engine_model.cpp:170-297).  These entry points
 * enqueue the same op sequence for `batch` independent sequences (or one prefill chunk) on the caller's stream
 * from the launches of cactus_hip.h - the back-to-back small ops as that header's one-launch sequences - with every
 * piece of per-sequence state in device memory the caller owns (token ids, positions, INT8 KV caches, LFM2
 * convolution rows).  A step allocates nothing (scratch comes in as `workspace`; the attention's split records use the
 * library workspace: run a step once, or cactus_hip_workspace_reserve, before capturing), synchronizes nothing, and
 * can be captured into a HIP graph and replayed.  Results per sequence are those of the reference's ops in the
 * reference's order (tests/test_gpu_steps.py, tests/test_gpu_free_running.py hold them to the CPU oracle).
 *
 * Weights are described in the reference's layouts: a quantized matrix is int8[N, K] (or packed INT4 uint8[N, K/2])
 * with half scales [N, K/128] (group size 128, python/src/tensor_io.py:14); norm weights are half vectors. */
#ifndef CACTUS_HIP_STEPS_H
#define CACTUS_HIP_STEPS_H

#include "cactus_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cactus_hip_qmatrix {
    const void* B;                 /* int8[N, K], or packed uint8[N, K/2] when int4 != 0; NULL: absent */
    const cactus_half_t* scales;   /* half[N, K/128] */
    size_t N, K;
    int int4;
} cactus_hip_qmatrix;

/* One decoder block.  kind 0: attention block (Qwen3; LFM2's attention blocks): q, k, v, o, q_norm, k_norm.
 * kind 1: LFM2 short-convolution block (model_lfm2.cpp:123-207): in_proj [3 * hidden, hidden], conv_weight
 * half[hidden, taps], out_proj.  Both: in_norm, post_norm, gate, up, down (SwiGLU MLP). */
typedef struct cactus_hip_block_weights {
    int kind;
    cactus_hip_qmatrix q, k, v, o;
    cactus_hip_qmatrix in_proj, out_proj;
    const cactus_half_t* conv_weight;
    cactus_hip_qmatrix gate, up, down;
    const cactus_half_t *in_norm, *post_norm, *q_norm, *k_norm;
} cactus_hip_block_weights;

typedef struct cactus_hip_decoder_weights {
    size_t hidden, q_heads, kv_heads, head_dim, ffn, vocab, layers;
    size_t conv_taps, conv_rows;            /* LFM2: taps of the short convolution, rows of carried state (taps - 1) */
    float eps, theta;
    const cactus_hip_block_weights* blocks; /* [layers], host memory */
    cactus_hip_qmatrix embedding;           /* [vocab, hidden] */
    cactus_hip_qmatrix lm_head;             /* B == NULL: tied to the embedding (model_qwen.cpp:21-23) */
    const cactus_half_t* final_norm;
} cactus_hip_decoder_weights;

/* Per-sequence decode state, all device memory.  Caches cover the attention blocks in block order:
 * int8 [attention_blocks][batch][slots][kv_heads][head_dim], scales float [..][slots][kv_heads][ceil(head_dim / 128)].
 * window > 0: the reference's sliding window with `sink` leading tokens (engine_cache.cpp:268-329) held as a ring
 * of `window` slots; 0: slot = position.  positions: tokens seen; cache_lens = min(position, window or infinity).
 * conv_state: half [conv_blocks][batch][conv_rows][hidden] in block order (LFM2), else NULL. */
typedef struct cactus_hip_decode_state {
    size_t batch, slots, window, sink, max_position;
    int8_t *k_cache, *v_cache;
    float *k_scales, *v_scales;
    int32_t *positions, *cache_lens, *tokens;   /* tokens: in = each sequence's last token, out = the greedy choice */
    cactus_half_t* conv_state;
} cactus_hip_decode_state;

/* Bytes of `workspace` (256-byte aligned device memory) a decode step of `batch` sequences needs.  Call it once
 * outside any stream capture before capturing a step of that batch: it also creates the second stream and the events
 * of the split step below. */
size_t cactus_hip_decode_step_workspace_bytes(const cactus_hip_decoder_weights* w, size_t batch);

/* One decode step: embedding lookup of state->tokens, every block, final norm, lm_head into logits half[batch, vocab],
 * greedy sampler back into state->tokens; advance != 0 also moves positions / cache_lens on by one token.
 * Batches of at most 4 sequences on an all-attention model run every matmul as a streaming launch that produces its
 * own input rows; larger ones quantize rows in the producing launch and use the tile kernel.
 * Optional (off by default, see cactus_hip_decode_step_split): batches of at least that many sequences run the blocks
 * as two half-batches, the second half on a stream of the library's own that forks from `stream` after the embedding
 * and joins it before the lm_head (a capture of `stream` takes in both), the halves' attention launches taking turns,
 * so that one half reads its KV cache while the other runs its matmuls.  Rows are independent: every result is the
 * single stream's, bit for bit.  (That stream and its events exist once per device: split steps that overlap on one
 * device must be serialized by the caller.) */
int cactus_hip_decode_step_dev(const cactus_hip_decoder_weights* w, const cactus_hip_decode_state* state,
                               void* workspace, cactus_half_t* logits, int advance, void* stream);

/* Smallest batch that cactus_hip_decode_step_dev splits into two half-batches (0, the default: never split); returns
 * the previous value.  The environment variable CACTUS_HIP_STEP_SPLIT sets the initial value.  No counterpart in the
 * reference, which decodes one sequence.  Measured on MI355X (Qwen3-0.6B shape, 512 cached tokens): 256 sequences 3.03 ms
 * unsplit, 3.22 ms split; LFM2-1.2B shape, 1024 sequences 3.60 / 3.55 ms. */
size_t cactus_hip_decode_step_split(size_t min_batch);

/* A chunked prefill of ONE sequence, the way the engine walks a long prompt (Model::prefill, engine_model.cpp:170-219):
 * `chunk` tokens at a time, chunk 0 against nothing (FP16 attention), the others against the INT8 cache.  tokens: device
 * int32[chunk] of this chunk; cache_start = tokens before it; cache_start_dev: device int32 holding cache_start (the
 * append kernel reads it); caches cover the attention blocks in block order: int8 [attention_blocks][capacity][kv_heads]
 * [head_dim] / float [attention_blocks][capacity][kv_heads][1] with capacity >= cache_start + chunk.  conv_state: half
 * [conv_blocks][conv_rows][hidden] in block order, the rows LFM2's short convolutions carry from chunk to chunk
 * (model_lfm2.cpp:166-191); chunk 0 (cache_start == 0) clears it; NULL for a model without such blocks.  With one
 * sequence the caches and conv_state have the layout of cactus_hip_decode_state with batch 1 and slots = capacity: a
 * decode step carries on from them.  last != 0: also the final norm, the lm_head on the last token into logits
 * half[1, vocab] and the greedy token into *token_out. */
typedef struct cactus_hip_prefill_state {
    size_t capacity;
    int8_t *k_cache, *v_cache;
    float *k_scales, *v_scales;
    cactus_half_t* conv_state;
} cactus_hip_prefill_state;
size_t cactus_hip_prefill_chunk_workspace_bytes(const cactus_hip_decoder_weights* w, size_t chunk);
int cactus_hip_prefill_chunk_dev(const cactus_hip_decoder_weights* w, const cactus_hip_prefill_state* state,
                                 const int32_t* tokens, size_t chunk, size_t cache_start, const int32_t* cache_start_dev,
                                 void* workspace, int last, cactus_half_t* logits, int32_t* token_out, void* stream);

/* The whole prompt: every chunk of cactus_hip_prefill_chunk_dev in one call (Model::prefill's loop, engine_model.cpp:
 * 170-219), up to `lanes` (1 .. 8) chunks in flight at a time.  A chunk's launches are short and leave most of the chip
 * idle, and chunk c + 1 needs nothing of chunk c but its rows in a block's cache when it reaches that block's attention
 * (or the rows it left in a convolution block's state when it reaches that convolution): chunk c runs on stream
 * c % lanes (lane 0 = `stream`, the others the library's own, forked from and joined to `stream`, so a capture of
 * `stream` takes in all of them), records an event per block once its K / V rows are appended (its convolution has
 * run), and its attention (convolution) in a block waits for the previous chunk's event of that block.  The results are those of the chunks called
 * one after the other, bit for bit.  tokens: device int32[n_tokens]; the last chunk may be short; chunk_starts_dev:
 * device int32[chunks] holding c * chunk; workspace: cactus_hip_prefill_workspace_bytes(w, chunk, lanes) bytes (call it
 * outside any stream capture: it also creates the streams and events); logits half[1, vocab] and *token_out receive the
 * last token's logits and greedy choice.  The library owns one set of these streams and events per device: calls that
 * overlap on one device (from several host threads, or on several caller streams at once) would share them and must be
 * serialized by the caller; calls on different devices are independent.  No counterpart in the reference, whose chunks
 * are separate graph executions. */
size_t cactus_hip_prefill_workspace_bytes(const cactus_hip_decoder_weights* w, size_t chunk, size_t lanes);
int cactus_hip_prefill_dev(const cactus_hip_decoder_weights* w, const cactus_hip_prefill_state* state, const int32_t* tokens,
                           size_t n_tokens, size_t chunk, const int32_t* chunk_starts_dev, void* workspace, size_t lanes,
                           cactus_half_t* logits, int32_t* token_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
