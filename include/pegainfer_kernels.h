/*
 * pegainfer_kernels.h - C ABI of libpegainfer_kernels_hip.so (MI355X / gfx950).
 *
 * Drop-in replacement for the native library behind the reference's
 * `pegainfer-kernels` crate: every entry point below has the SAME symbol name,
 * argument order and argument meaning as the `unsafe extern "C"` declaration
 * it replaces in  pegainfer-kernels/src/ffi.rs  (line cited per function), so
 * the Rust wrappers in pegainfer-kernels/src/ops/ *.rs link against it
 * unchanged once `CUstream` is aliased to `hipStream_t`.
 * The historical `_cuda` suffix is kept on purpose: it is the ABI.
 *
 * Conventions (ffi.rs:1-5):
 *   Half      = uint16_t holding a bf16 bit pattern, device pointer
 *   sizes     = int32_t, strides/offsets = int64_t (elements)
 *   stream    = hipStream_t (last argument), all work is enqueued, never synced
 *   HiddenStates [d, T] = T contiguous vectors of length d (token-major)
 *   DeviceMatrix [rows, cols] = row-major
 * Return styles, preserved per symbol:
 *   void                - errors surface at the next sync (norms, GEMM, ...)
 *   pegainfer_status_t  - hipGetLastError() cast to int (reference: CUresult)
 *   int32_t             - 0 ok, non-zero = hipError_t (reference: cudaError_t)
 */
#ifndef PEGAINFER_KERNELS_H
#define PEGAINFER_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t Half;                /* ffi.rs:4 */
typedef void* pegainfer_stream_t;     /* hipStream_t (reference: CUstream) */
typedef int32_t pegainfer_status_t;   /* hipError_t as int (reference: CUresult) */

/* ---- library / device lifecycle (ffi.rs:159-161; csrc/linear.cu:14-42) ---- */
int32_t cuda_set_device(int32_t device_ordinal);
void cublas_init(void);      /* per-thread GEMM state: split-K workspace for gemm_cuda */
void cublas_destroy(void);

/* ---- casts used around the MP8 collectives (ffi.rs:8-20) ---- */
pegainfer_status_t deepseek_bf16_to_f32_cuda(const Half* input, float* output, int32_t n, pegainfer_stream_t stream);
pegainfer_status_t deepseek_f32_to_bf16_cuda(const float* input, Half* output, int32_t n, pegainfer_stream_t stream);

/* ---- norms (ffi.rs:22-68, 981-1011) ---- */
void rms_norm_cuda(const Half* x, const Half* weight, Half* out, int32_t n, float eps, pegainfer_stream_t stream);
void rms_norm_batched_cuda(const Half* x, const Half* weight, Half* out, int32_t hidden_dim, int32_t seq_len, float eps, pegainfer_stream_t stream);
void fused_add_rms_norm_cuda(Half* hidden, const Half* residual, const Half* weight, Half* out, int32_t n, float eps, pegainfer_stream_t stream);
void fused_add_rms_norm_batched_cuda(Half* hidden, const Half* residual, const Half* weight, Half* out, int32_t hidden_dim, int32_t batch_size, float eps, pegainfer_stream_t stream);
void rms_norm_batched_offset_cuda(const Half* x, const Half* weight, Half* out, int32_t hidden_dim, int32_t seq_len, float eps, pegainfer_stream_t stream);
void rms_norm_offset_cuda(const Half* x, const Half* weight, Half* out, int32_t n, float eps, pegainfer_stream_t stream);
void rms_norm_gated_cuda(const Half* x, const float* weight, const Half* gate, Half* out, int32_t num_heads, int32_t head_dim, float eps, pegainfer_stream_t stream);

/* ---- elementwise / embedding (ffi.rs:41-47, 70-96, 143-157) ---- */
pegainfer_status_t add_cuda(const Half* a, const Half* b, Half* out, int32_t n, pegainfer_stream_t stream);
pegainfer_status_t silu_mul_triton_aot_cuda(const Half* gate, const Half* up, Half* out, int32_t n, pegainfer_stream_t stream);
void silu_mul_fused_cuda(const Half* gate_up, Half* out, int32_t intermediate_size, int32_t bs, pegainfer_stream_t stream);
pegainfer_status_t embedding_batched_cuda(const Half* embed, const uint32_t* token_ids, Half* out, int32_t hidden_size, int32_t seq_len, pegainfer_stream_t stream);
pegainfer_status_t embedding_batched_vocab_shard_cuda(const Half* embed, const uint32_t* token_ids, Half* out, int32_t hidden_size, int32_t seq_len, uint32_t vocab_start, uint32_t part_vocab_size, pegainfer_stream_t stream);
pegainfer_status_t embedding_decode_cuda(const Half* embed, const uint32_t* token_id, Half* out, int32_t hidden_size, pegainfer_stream_t stream);

/* ---- GEMM call sites (ffi.rs:122-140; csrc/linear.cu:45-75) ----
 * Y[M,N] = W[M,K] . X[K,N]; W row-major, X/Y token-major (N vectors of K / M).
 * gemm_graphsafe_cuda never touches library-owned workspace (capture-safe). */
void gemm_cuda(const Half* W, const Half* X, Half* Y, int32_t M, int32_t N, int32_t K, pegainfer_stream_t stream);
void gemm_graphsafe_cuda(const Half* W, const Half* X, Half* Y, int32_t M, int32_t N, int32_t K, pegainfer_stream_t stream);

/* ---- per-head QK RMSNorm + RoPE (ffi.rs:164-178, 1143-1157) ---- */
void prefill_qk_norm_rope_only_cuda(Half* q_batch, Half* k_batch, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t seq_len, int32_t start_pos, float rms_eps, pegainfer_stream_t stream);
void qk_norm_rope_batched_decode_cuda(Half* q, Half* k, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, const int32_t* positions, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t batch_size, float rms_eps, pegainfer_stream_t stream);

/* ---- paged KV append (ffi.rs:1160-1179) ---- */
int32_t paged_kv_scatter_cuda(const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const Half* src_k, const Half* src_v, const int32_t* batch_indices, const int32_t* positions, int32_t nnz, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int64_t stride_page, int64_t src_stride_n, int64_t src_stride_h, pegainfer_stream_t stream);

/* ---- prefill plan helpers, host side (ffi.rs:1182-1211) ---- */
int32_t batch_prefill_paged_num_tiles(int32_t seq_len, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim);
int32_t batch_prefill_paged_num_tiles_with_cta_tile_q(int32_t seq_len, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t cta_tile_q_override);
int32_t batch_prefill_cta_tile_q(int32_t total_seq_len, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim);
int32_t batch_prefill_cta_tile_q_with_override(int32_t total_seq_len, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t cta_tile_q_override);

/* ---- paged causal prefill attention (ffi.rs:1214-1267) ---- */
int32_t batch_prefill_paged_cuda(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* q_indptr, const int32_t* request_indices, const int32_t* qo_tile_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const uint32_t* total_num_rows, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t seq_len, int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale, pegainfer_stream_t stream);
int32_t batch_prefill_paged_cuda_with_cta_tile_q(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* q_indptr, const int32_t* request_indices, const int32_t* qo_tile_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const uint32_t* total_num_rows, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t seq_len, int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale, int32_t cta_tile_q_override, pegainfer_stream_t stream);
/* contiguous-HND single-request prefill (ffi.rs:1270-1283) */
int32_t single_prefill_cuda(const Half* q, Half* output, const Half* k_cache, const Half* v_cache, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t seq_len, int32_t kv_len, int32_t max_seq_len, float sm_scale, pegainfer_stream_t stream);

/* ---- paged decode attention (ffi.rs:1337-1385) ---- */
int32_t paged_attention_decode_cuda(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int64_t stride_page, float sm_scale, pegainfer_stream_t stream);
int32_t paged_attention_decode_split_kv_cuda(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const int32_t* o_indptr, const uint8_t* block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale, pegainfer_stream_t stream);

/* ---- Qwen3.5 hybrid extras (ffi.rs:181-226, 1014-1039, 1286-1334) ---- */
void prefill_attention_hd256_prep_cuda(const Half* q_full_batch, const Half* k_batch, const Half* v_batch, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, Half* q_batch_out, Half* k_cache, Half* v_cache, int32_t num_q_heads, int32_t num_kv_heads, int32_t seq_len, const int32_t* start_pos_ptr, int32_t rotary_dim, float rms_eps, int32_t max_seq_len, pegainfer_stream_t stream);
void attention_gate_batch_hd256_cuda(const Half* q_full_batch, Half* attn_out, int32_t num_q_heads, int32_t seq_len, pegainfer_stream_t stream);
void qk_norm_partial_rope_batched_decode_hd256_cuda(const Half* q_full_batch, Half* k_batch, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache, const int32_t* positions, Half* q_batch_out, int32_t num_q_heads, int32_t num_kv_heads, int32_t batch_size, int32_t rotary_dim, float rms_eps, pegainfer_stream_t stream);
void gated_delta_rule_decode_cuda(const Half* qkv, const Half* b_proj, const Half* a_proj, const Half* dt_bias, const float* A_log, float* state, Half* output, int32_t num_key_heads, int32_t num_value_heads, int32_t key_dim, int32_t val_dim, pegainfer_stream_t stream);
void conv1d_prefill_cuda(const Half* x_seq, const Half* conv_weight, Half* conv_state, Half* out_seq, int32_t num_channels, int32_t seq_len, int32_t kernel_size, pegainfer_stream_t stream);
int32_t paged_attention_decode_cuda_hd256(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int64_t stride_page, float sm_scale, pegainfer_stream_t stream);
int32_t batch_prefill_paged_cuda_hd256(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d, const int32_t* q_indptr, const int32_t* request_indices, const int32_t* qo_tile_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const uint32_t* total_num_rows, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t seq_len, int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale, pegainfer_stream_t stream);

/* ---- Qwen3.5 gated delta rule, chunk-wise prefill (ffi.rs:1041-1137; Triton-AOT in the reference, source
 * tools/triton/gated_delta_rule_chunkwise_kernels.py).  Fixed chunk 64, key_dim = value_dim = 128.  Token-major:
 * q,k,w [T,H,128] bf16; v,u,v_new,output [T,H,128] bf16; g,beta [T,H] f32; a_tril [T,H,64] f32; a_inv [T,H,64]
 * bf16; state [H,K,V] f32 (V contiguous); chunk_state [ceil(T/64),H,K,V] f32.  Operator order:
 * recurrent.rs:368-470.  CUresult-style status. ---- */
pegainfer_status_t gated_delta_rule_prefill_chunk_prepare_cuda(const Half* qkv, const Half* b_proj, const Half* a_proj, const Half* dt_bias, const float* a_log, Half* q_out, Half* k_out, Half* v_out, float* g_out, float* beta_out, int32_t num_key_heads, int32_t num_value_heads, int32_t qkv_dim, int32_t seq_len, pegainfer_stream_t stream);
pegainfer_status_t gated_delta_rule_prefill_chunk_cumsum_cuda(const float* g_in, float* g_out, int32_t seq_len, int32_t num_value_heads, pegainfer_stream_t stream);
pegainfer_status_t gated_delta_rule_prefill_chunk_a_cuda(const Half* k, const float* g_cumsum, const float* beta, float* a_tril, int32_t seq_len, int32_t num_value_heads, pegainfer_stream_t stream);
pegainfer_status_t gated_delta_rule_prefill_chunk_solve_cuda(const float* a_tril, Half* a_inv, int32_t seq_len, int32_t num_value_heads, pegainfer_stream_t stream);
pegainfer_status_t gated_delta_rule_prefill_chunk_recompute_cuda(const Half* k, const Half* v, const float* beta, Half* w, Half* u, const Half* a_inv, const float* g_cumsum, int32_t seq_len, int32_t num_value_heads, pegainfer_stream_t stream);
pegainfer_status_t gated_delta_rule_prefill_chunk_state_cuda(const Half* k, const Half* w, const Half* u, const float* g_cumsum, const float* initial_state, float* chunk_state, Half* v_new, float* final_state, int32_t seq_len, int32_t num_value_heads, pegainfer_stream_t stream);
pegainfer_status_t gated_delta_rule_prefill_chunk_o_cuda(const Half* q, const Half* k, const Half* v_new, const float* chunk_state, const float* g_cumsum, Half* output, int32_t seq_len, int32_t num_value_heads, float scale, pegainfer_stream_t stream);

/* ---- sampling (ffi.rs:98-120) ---- */
void argmax_cuda(const Half* x, int32_t* out, int32_t n, pegainfer_stream_t stream);
void flashinfer_top1_cuda(const Half* logits, Half* top1_value_scratch, uint8_t* row_states_scratch, int32_t* output, int32_t vocab_size, pegainfer_stream_t stream);
void gpu_sample_flashinfer_cuda(const Half* logits, float* probs_scratch, uint8_t* valid_scratch, int32_t* output, int32_t vocab_size, float inv_temperature, int32_t top_k, float top_p, uint64_t seed, pegainfer_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PEGAINFER_KERNELS_H */
