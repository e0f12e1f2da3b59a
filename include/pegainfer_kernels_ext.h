/*
 * pegainfer_kernels_ext.h - entry points of libpegainfer_kernels_hip.so that have NO counterpart in
 * the reference's ffi.rs.  They exist for the MI355X host DAG (libpegainfer_qwen3.so): at ~1 ms per
 * decode step the reference's 14 launches per layer and per-request sampling sync are first-order
 * costs, so the host may call these fused / batched forms instead.  Each one is
 * REQUIRED to produce bit-identical results to the sequence of reference-named calls it replaces;
 * tests/test_gpu_fused.py checks exactly that.
 */
#ifndef PEGAINFER_KERNELS_EXT_H
#define PEGAINFER_KERNELS_EXT_H

#include "pegainfer_kernels.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Greedy token for every row of logits[rows, vocab] in one launch (replaces, per request,
 * extract_vec + flashinfer_top1_cuda + sync + D2H: executor.rs:324-328, ops/sampling.rs:122-168).
 * state_scratch: rows*16 bytes, zeroed once by the caller, left zeroed.  Ties -> lowest index. */
pegainfer_status_t pegainfer_batched_top1(const Half* logits, int32_t vocab_size, int32_t rows, int64_t row_stride, uint8_t* state_scratch, int32_t* out_tokens, pegainfer_stream_t stream);

/* Decode GEMM (T <= 64 token columns, weights streamed once) with optional prologue / epilogue fused in; same
 * accumulation cores as gemm_graphsafe_cuda (dot2 GEMV at T <= 2, skinny MFMA GEMM at 3 <= T <= 16, the tiled
 * LDS-DMA GEMM - split over K for matrices below 16384 rows - at 17 <= T <= 64), so every form below is
 * bit-identical to the unfused reference-ABI sequence:
 *   norm_weight == NULL                : Y = W.X                                  (== gemm_graphsafe_cuda)
 *   norm_weight, residual == NULL      : Y = W.rms_norm(X)                        (rms_norm_batched_cuda + gemm)
 *   norm_weight, residual, hidden_out  : hidden_out = bf16(X + residual); Y = W.rms_norm(X + residual)
 *                                        (fused_add_rms_norm_batched_cuda + gemm; hidden_out != X)
 *   silu_intermediate = I > 0          : W = [gate; up] (M == 2I), Y[T, I] = silu_mul_fused(W.x)
 *                                        (gemm + silu_mul_fused_cuda)
 * The plain form takes T <= 64; the prologue / epilogue forms T <= 16 (larger decode batches use
 * pegainfer_gemm_silu / pegainfer_gemm_add_rms_norm around the tiled kernels instead).  Returns hipErrorInvalidValue for shapes it does not
 * take (K % 8, misaligned, ...) so the caller can fall back to the unfused sequence. */
pegainfer_status_t pegainfer_gemv_fused(const Half* W, const Half* X, Half* Y, int32_t M, int32_t T, int32_t K, const Half* residual, const Half* norm_weight, Half* hidden_out, float eps, int32_t silu_intermediate, pegainfer_stream_t stream);

/* pegainfer_gemv_fused with other rounding points (bits 0 and 2: T <= 4; bit 1 alone: T <= 16): flags bit 0 = the norm weight is (1 + w)
 * (rms_norm_batched_offset_cuda), bit 1 = hidden_out = bf16(X + residual) and the norm runs over that ROUNDED sum
 * (add_cuda then rms_norm_batched_offset_cuda, batch_decode.rs:246-262) instead of FlashInfer's fused add+norm,
 * bit 2 = SwiGLU as bf16(bf16(silu(gate)) * up) (silu_mul_triton_aot_cuda, elementwise.cu:28-42).  Bit-identical to
 * those unfused sequences. */
pegainfer_status_t pegainfer_gemv_fused_ex(const Half* W, const Half* X, Half* Y, int32_t M, int32_t T, int32_t K, const Half* residual, const Half* norm_weight, Half* hidden_out, float eps, int32_t silu_intermediate, int32_t flags, pegainfer_stream_t stream);

/* Prefill q/k/v projection as ONE GEMM over the row-stacked weight W[M0 + M1 + M2, K] (q_proj; k_proj; v_proj
 * rows), writing the three contiguous buffers the reference's prefill kernels take: Y0[T, M0], Y1[T, M1],
 * Y2[T, M2].  Replaces the three gemm_cuda calls of prefill.rs:120-129 (the k/v projections alone cover a
 * quarter of the chip).  The outputs are the row ranges of gemm_cuda over the stacked matrix, bit for bit; against
 * three separate gemm_cuda calls they are bit-identical whenever the separate calls take the same K-split plan as
 * the stacked matrix (always up to 64 columns and for un-split shapes; a 1024-row k_proj alone at ~1K tokens is a
 * split-K shape, the 6144-row stacked matrix is not - there parity is the GEMM tolerance).  Falls back to three
 * calls for shapes the tiled kernel does not take. */
pegainfer_status_t pegainfer_gemm_split3(const Half* W, const Half* X, Half* Y0, int32_t M0, Half* Y1, int32_t M1, Half* Y2, int32_t M2, int32_t T, int32_t K, pegainfer_stream_t stream);

/* General form of pegainfer_gemm_split3: n_out in 2..4 outputs Y[i][T, Ms[i]] from the row-stacked weight
 * W[sum Ms, K] (Qwen3.5: in_proj_qkv | z | b | a, and gate_proj | up_proj).  Host arrays of n_out entries. */
pegainfer_status_t pegainfer_gemm_split(const Half* W, const Half* X, int32_t n_out, Half* const* Y, const int32_t* Ms, int32_t T, int32_t K, pegainfer_stream_t stream);

/* gate_up GEMM with SwiGLU in the epilogue: Y[T, I] = silu_mul_fused(W[2I, K] . X) = gemm_cuda + silu_mul_fused_cuda
 * (prefill.rs:167-175) without the [T, 2I] round trip.  Same bits as that pair.  gate_up_scratch [T, 2I] is used only
 * when the shape falls back to the pair (T <= 16, small or unaligned matrices); may be NULL otherwise. */
pegainfer_status_t pegainfer_gemm_silu(const Half* W, const Half* X, Half* Y, Half* gate_up_scratch, int32_t I, int32_t T, int32_t K, pegainfer_stream_t stream);
/* same with the Qwen3.5 activation: bf16(bf16(silu(gate)) * up) == gemm_cuda x2 + silu_mul_triton_aot_cuda */
pegainfer_status_t pegainfer_gemm_silu_rounded(const Half* W, const Half* X, Half* Y, Half* gate_up_scratch, int32_t I, int32_t T, int32_t K, pegainfer_stream_t stream);

/* o_proj / down_proj + residual add + RMSNorm: exactly gemm_cuda(W, X, y_scratch, M, T, K) followed by
 * fused_add_rms_norm_batched_cuda(hidden, y_scratch, norm_weight, normed_out, M, T, eps) (batch_decode.rs:262-270,
 * 288-296; prefill.rs:150-160).  On a split-K shape (decode batches of 17..64 columns; prefill when the tiling has
 * fewer tiles than CUs) the slice sum, the add and the norm are one launch over the fp32 partials (y_scratch [T, M]
 * is then left untouched); other shapes run the two calls.  Same bits either way. */
pegainfer_status_t pegainfer_gemm_add_rms_norm(const Half* W, const Half* X, Half* y_scratch, Half* hidden, const Half* norm_weight, Half* normed_out, int32_t M, int32_t T, int32_t K, float eps, pegainfer_stream_t stream);

/* down_proj + residual add: exactly gemm_cuda(W, X, y_scratch, M, T, K) followed by add_cuda(a, y_scratch, out, M*T)
 * (prefill.rs:176-185).  On a split-K shape the slice sum and the add are one launch over the fp32 partials
 * (y_scratch [T, M] is then left untouched).  out may alias a.  Same bits either way. */
pegainfer_status_t pegainfer_gemm_add(const Half* W, const Half* X, Half* y_scratch, const Half* a, Half* out, int32_t M, int32_t T, int32_t K, pegainfer_stream_t stream);
/* down_proj + residual add + the NEXT layer's input RMSNorm (prefill): gemm_cuda + add_cuda + rms_norm_batched_cuda in
 * at most two launches, same bits (the norm sees the bf16-rounded sum: prefill.rs:183 then prefill.rs:89). */
pegainfer_status_t pegainfer_gemm_add_then_rms_norm(const Half* W, const Half* X, Half* y_scratch, const Half* a, Half* out, const Half* norm_weight, Half* normed_out, int32_t M, int32_t T, int32_t K, float eps, pegainfer_stream_t stream);
/* prefill_qk_norm_rope_only_cuda / qk_norm_rope_batched_decode_cuda + paged_kv_scatter_cuda in ONE launch (head_dim 128):
 * q and k are normalised + rotated in place exactly as the reference kernel does, and the rotated k row and the v row
 * of every token go to cache slot page_indices[page_indptr[batch_indices[i]] + positions[i] / page_size] - the same
 * bytes the two calls leave in q, k and the cache. */
int32_t pegainfer_qk_norm_rope_scatter(Half* q, Half* k, const Half* v, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, const int32_t* positions, const int32_t* batch_indices, Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int64_t stride_page, int32_t tokens, float rms_eps, pegainfer_stream_t stream);
/* the same over the row-stacked output of ONE q|k|v GEMM (qkv [tokens][(Hq + 2 Hkv) * 128], the fused decode path's
 * layout): normalised + rotated q heads go to the dense q_out [tokens][Hq * 128] the prefill attention reads, k heads
 * (normalised + rotated) and v heads straight to their cache slots.  Same bytes in q and in the cache as the three-buffer
 * form; used by the short-prompt prefill path (<= 16 tokens: one stacked GEMV with the norm in its prologue). */
int32_t pegainfer_qkv_stacked_norm_rope_scatter(const Half* qkv, Half* q_out, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, const int32_t* positions, const int32_t* batch_indices, Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int64_t stride_page, int32_t tokens, float rms_eps, pegainfer_stream_t stream);

/* Decode attention with the per-head q/k RMSNorm + RoPE and the KV append folded in (head_dim 128):
 * reads the raw fused-QKV GEMV output qkv[bs, (Hq + 2 Hkv) * 128], writes the new K (normalised, rotated)
 * and V rows into the paged cache and the attention output [bs, Hq*128].  Bit-identical to
 * qk_norm_rope_batched_decode_cuda + paged_kv_scatter_cuda + paged_attention_decode[_split_kv]_cuda
 * (ops/attention.rs:469-511, 572-796).  use_split selects the partition-KV plan arrays.
 * slot_desc (optional, 16-byte aligned): one record of 8 int32 per slot {b, lo, hi, page_indptr[b], position,
 * kv_len, o_indptr[b], o_indptr[b+1]}, lo < 0 for padding slots - the same plan, pre-resolved on the host so a
 * workgroup needs one metadata load instead of four dependent ones.
 * merge_counters (optional, use_split only): batch_size * num_kv_heads * 32 int32 (one cache line per (request, kv head)
 * counter, the counter itself at index (b * num_kv_heads + kvh) * 32), zero before the first call.  When
 * given, the last workgroup of each (request, kv head) to finish merges that head group's partials in the
 * same launch (agent-scope release/acquire around one atomic) and re-arms the counter; no merge launch.  Output
 * rows of requests that own no slot (padding columns) are then left untouched.  Same bits either way. */
int32_t pegainfer_fused_decode_attention(const Half* qkv, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* positions, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, float rms_eps, int32_t use_split, const int32_t* split_request_indices, const int32_t* split_kv_tile_indices, const int32_t* split_kv_chunk_size_ptr, const int32_t* split_o_indptr, const uint8_t* split_block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t split_slots, int64_t stride_page, float sm_scale, const int32_t* slot_desc, int32_t* merge_counters, pegainfer_stream_t stream);
/* pegainfer_fused_decode_attention (partition form, merge_counters required) + the o_proj GEMV that consumes it, in ONE
 * launch, for steps with one request: the grid is the attention's split_slots x num_kv_heads grid of 8-wave workgroups;
 * the workgroups of the request's KV chunks (slot_desc words 6 / 7 of slot 0, read on the device: a captured launch is
 * replayed while the request grows) run the attention, the workgroups of the padding slots - at least min_padding_slots
 * of them by the caller's plan - request the o_proj rows into registers meanwhile, the merging workgroups publish the attention row
 * write-through and arrive on done_counter (num_kv_heads device ints, 32 ints = one cache line apart, all zero before the
 * launch: head group g arrives on int 32 * g when a K block of the o_proj deal is one head group, else all on int 0), then the o_proj workgroups finish
 * their dot products from registers - the bits of the two stand-alone launches.  o_proj [hidden, num_qo_heads * head_dim] row-major,
 * attn_proj_out [hidden], status (optional) receives 0x300 when the bounded wait for the attention rows expired.
 * Returns hipErrorInvalidValue (1) when the shape does not fit; the caller then issues the two launches. */
int32_t pegainfer_fused_decode_attention_oproj(const Half* qkv, Half* attn_out, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* positions, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, float rms_eps, const int32_t* split_request_indices, const int32_t* split_kv_tile_indices, const int32_t* split_kv_chunk_size_ptr, const int32_t* split_o_indptr, const uint8_t* split_block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t split_slots, int32_t min_padding_slots, int64_t stride_page, float sm_scale, const int32_t* slot_desc, int32_t* merge_counters, const Half* o_proj, Half* attn_proj_out, int32_t hidden, int32_t* done_counter, uint32_t* status, pegainfer_stream_t stream);
/* 1 when a step of batch_size (1 or 2) requests with this configuration can take pegainfer_fused_decode_attention_oproj
 * (the launcher's own shape test, without pointers): a host runtime asks once at model creation and plans its KV chunks
 * for the form from the first step on. */
int32_t pegainfer_fused_decode_attention_oproj_supported(int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t hidden, int32_t split_slots, int32_t min_padding_slots, int32_t batch_size);

/* Debug aid (not in ffi.rs): buf = device array of (workgroups per launch, <= 4096) * 8 uint64, or NULL.  Every later
 * dot2-GEMV launch stamps it with the 100 MHz wall clock per workgroup: [0] entry, [1] x staged, [2] first weight block
 * consumed, [3] K loop of the last row group done, [4] exit, [5] XCC id.  tools/gemv_probe.py prints the breakdown. */
void pegainfer_debug_gemv_trace(uint64_t* buf);

/* Debug / test hook (not in ffi.rs), no device work: force how the batched-decode GEMM at 3..16 token columns
 * (skinny_resident_kernel) combines its 8 waves' partial sums per row block: 0 = two barriers, 1 = one barrier,
 * 4 = LDS tickets without a barrier, 5 = lazy tickets (no waiting wave either); -1 = the launcher's own choice (lazy
 * tickets where a workgroup walks more than two row blocks).  The forms produce the same bits; the hook exists so that a
 * test can show it inside one process. */
void pegainfer_debug_skinny_flush(int32_t mode);

/* Debug aid (not in ffi.rs): buf = device array of slots * num_kv_heads * 8 uint64, or NULL to switch off.  Every later
 * fused decode-attention launch stamps it with the 100 MHz wall clock at its phase boundaries: [0] entry, [1] slot
 * record read, [2] q prologue done, [3] KV scan done, [4] partials published, [5] ticket drawn, [6] merge done,
 * [7] = 1 for the workgroup that merged.  tools/attn_probe.py prints the per-phase means. */
void pegainfer_debug_attn_trace(uint64_t* buf);
/* Debug / test hook, no device work: which kernel a GEMM of this shape takes.  silu_I > 0 asks
 * for the SwiGLU form.  out[0] = kind (0 GEMV / skinny family; 12 / 13 / 22 / 23 128-row LDS-DMA kernel; 256 / 257 the
 * 256 x 256 kernel / its SwiGLU form with the thin last round on the 128-row kernel; 1280 / 1281 the 128 x 256 kernel
 * plain / SwiGLU; 3000 + rt the weight-streaming kernel on rt x 16-row tiles; 1000 + tt a K-split plan on tt = 64 / 128
 * (128-row), 129 (128 x 256), 256 (256 x 256) tiles; 2000 + rt a tt = 64 plan whose GEMM half runs on the streaming
 * kernel), out[1] = K slices, out[2] = K tiles per slice (kind 257: activation-column tiles of the head).
 * T <= 16 (kind 0): out[1] = how the resident-x skinny kernel's waves meet per row block (0 two barriers, 1 one barrier,
 * 4 tickets, 5 lazy tickets; -1 = the dot2 GEMV at 1-2 columns or the tiled skinny kernel), out[2] = partial buffers * 100
 * + rows per row block. */
pegainfer_status_t pegainfer_debug_gemm_route(int32_t M, int32_t T, int32_t K, int32_t silu_I, int32_t* out);

/* Debug / test hook: 1 = prefill GEMMs whose 256 x 256 tiling leaves >= 8 % of a CU round idle take the persistent stream-K
 * launch (kind 258 of pegainfer_debug_gemm_route), 0 = never, -1 = the environment decides (PEGAINFER_STREAMK=1; default off:
 * measured slower than the data-parallel launch, profiles/r6_streamk_*).  Process-wide. */
void pegainfer_debug_streamk(int32_t on);

// (round 6) One decode step of a Qwen3.5 linear-attention layer's token mixer for ONE request in one launch: exactly
// conv1d_prefill_cuda(x_qkv, conv_weight, conv_state, tmp, C, 1, kernel_size) (ffi.rs conv1d_prefill_cuda; recurrent.rs:49-63) ->
// gated_delta_rule_decode_cuda(tmp, b_proj, a_proj, dt_bias, A_log, state, tmp2, ...) (recurrent.rs:64-79) ->
// rms_norm_gated_cuda(tmp2, norm_weight, gate, out, num_value_heads, val_dim, eps), bit for bit, without the two temporaries.
// key_dim == val_dim == 128, num_value_heads a multiple of num_key_heads, kernel_size 2..5; anything else returns
// hipErrorInvalidValue (1) and launches nothing - run the three calls.  tickets: num_key_heads int32 words, zero before the first
// call (the value heads of a key head meet there to shift the shared q / k conv windows once; the words reset themselves), may be
// shared by every layer and request of one stream.
int32_t pegainfer_linear_attn_decode_fused(const Half* x_qkv, const Half* conv_weight, Half* conv_state, const Half* b_proj, const Half* a_proj, const Half* dt_bias, const float* A_log, float* state, const float* norm_weight, const Half* gate, Half* out, int32_t num_key_heads, int32_t num_value_heads, int32_t key_dim, int32_t val_dim, int32_t kernel_size, float eps, int32_t* tickets, pegainfer_stream_t stream);

/* Partition-KV decode attention at head_dim 256 (Qwen3.5 full-attention layers).  No counterpart in ffi.rs (its
 * hd256 decode symbol is non-partition only, ffi.rs:1286-1306); arguments and scratch contract are exactly those of
 * paged_attention_decode_split_kv_cuda, plus the optional merge_counters of pegainfer_fused_decode_attention
 * (batch_size * num_kv_heads zeroed int32: the last workgroup merges in the same launch).  Equal to the non-partition
 * result within bf16 rounding of the partials. */
int32_t pegainfer_paged_attention_decode_split_kv_hd256(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const int32_t* o_indptr, const uint8_t* block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale, int32_t* merge_counters, pegainfer_stream_t stream);

/* zero n_words 32-bit device words with a kernel launch (captured as a KERNEL node: see the note in elementwise.hip on
 * hipMemsetAsync nodes inside replayed graphs) */
pegainfer_status_t pegainfer_zero_words(void* ptr, int32_t n_words, pegainfer_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
