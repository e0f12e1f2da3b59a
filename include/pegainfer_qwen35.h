/*
 * pegainfer_qwen35.h - C ABI of the Qwen3.5 hybrid (linear + full attention) host runtime inside
 * libpegainfer_qwen3.so.  Mirrors, for the forward pass only:
 *
 *   pegainfer-qwen35-4b/src/config.rs            -> the create() arguments
 *   pegainfer-qwen35-4b/src/weights.rs:102-296   -> tensor names (prefix model.language_model), f32 A_log / norm
 *   pegainfer-qwen35-4b/src/recurrent_state.rs   -> per-request conv_state (bf16) + GDR state (f32) per linear layer
 *   pegainfer-qwen35-4b/src/prefill.rs:21-449    -> prefill (one request per call, chunk-wise GDR, HD256 paged attn)
 *   pegainfer-qwen35-4b/src/batch_decode.rs      -> batched decode with per-slot recurrent updates
 *   pegainfer-qwen35-4b/src/batch_decode_graph.rs-> hipGraph replay of the decode step
 *
 * Scheduler / sampling params / tokenizer are out of scope.  0 on success, negative on error.
 */
#ifndef PEGAINFER_QWEN35_H
#define PEGAINFER_QWEN35_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pegainfer_qwen35_t;

/* layer_is_full[num_layers]: 1 = full attention layer, 0 = linear attention (config.rs layer_types).
 * linear key_dim = value_dim = 128 (the chunk-wise kernels' fixed shape).  enable_graph: capture the decode step
 * (re-captured when the set of request ids in the batch changes - the recurrent state addresses are per request).
 * split_policy: 0 = the reference's call (non-partition HD256 decode attention), 1 = partition the KV scan of the
 * full-attention layers towards >= 256 workgroups (pegainfer_paged_attention_decode_split_kv_hd256). */
pegainfer_qwen35_t pegainfer_qwen35_create(int32_t device_ordinal, int32_t hidden_size, int32_t intermediate_size,
                                           int32_t num_layers, int32_t vocab_size, int32_t num_attention_heads,
                                           int32_t num_kv_heads, int32_t head_dim, int32_t linear_num_key_heads,
                                           int32_t linear_num_value_heads, int32_t linear_conv_kernel_dim,
                                           float rms_norm_eps, float rope_theta, int32_t rotary_dim,
                                           const int32_t* layer_is_full, int32_t max_position_embeddings,
                                           int32_t num_kv_pages, int32_t max_batch_size, int32_t enable_graph,
                                           int32_t split_policy);
void pegainfer_qwen35_destroy(pegainfer_qwen35_t m);
const char* pegainfer_qwen35_last_error(pegainfer_qwen35_t m);

/* is_f32 = 1 for ...linear_attn.A_log and ...linear_attn.norm.weight (host float32), else host bf16 bits */
int32_t pegainfer_qwen35_load_tensor(pegainfer_qwen35_t m, const char* name, const void* host, int64_t numel,
                                     int32_t is_f32);
/* the stored image of one tensor (bf16 bits, or f32 for A_log / linear_attn.norm.weight) back on the host */
int32_t pegainfer_qwen35_export_tensor(pegainfer_qwen35_t m, const char* name, void* host, int64_t numel, int32_t is_f32);
int32_t pegainfer_qwen35_fill_synthetic(pegainfer_qwen35_t m, uint64_t seed, float std);
int32_t pegainfer_qwen35_finalize(pegainfer_qwen35_t m);
/* native mmap load of a .safetensors file / HF directory (tensors under model.language_model.); finalises */
int32_t pegainfer_qwen35_load_safetensors(pegainfer_qwen35_t m, const char* path);

int32_t pegainfer_qwen35_new_request(pegainfer_qwen35_t m);
int32_t pegainfer_qwen35_drop_request(pegainfer_qwen35_t m, int32_t request_id);
int32_t pegainfer_qwen35_request_seq_len(pegainfer_qwen35_t m, int32_t request_id);
int32_t pegainfer_qwen35_available_pages(pegainfer_qwen35_t m);
int32_t pegainfer_qwen35_capacity_pages(pegainfer_qwen35_t m);
/* the max_batch_size the model was created with (rows of the decode buffers) */
int32_t pegainfer_qwen35_max_batch_size(pegainfer_qwen35_t m);
int32_t pegainfer_qwen35_vocab_size(pegainfer_qwen35_t m);

/* prefill_forward (prefill.rs:21-120): appends n_tokens to the request (recurrent + conv state carried over),
 * returns the greedy token of the last position and optionally its logits (bf16 bits [vocab]). */
int32_t pegainfer_qwen35_prefill(pegainfer_qwen35_t m, int32_t request_id, int32_t n_tokens, const uint32_t* tokens,
                                 int32_t* out_token, void* out_logits_host);
/* batch_decode_graph (batch_decode.rs:113-196): one token per request; greedy tokens + optional logits [n, vocab] */
int32_t pegainfer_qwen35_decode(pegainfer_qwen35_t m, int32_t n_requests, const int32_t* request_ids,
                                const uint32_t* token_ids, int32_t* out_tokens, void* out_logits_host);
/* Sample row `column` of the LAST prefill (one row) / decode step's logits with the reference's gpu_sample rule
 * (ops/sampling.rs:109-170), and its TokenLogprob (compute_logprobs_from_cpu, executor.rs:400-434): as
 * pegainfer_qwen3_sample / pegainfer_qwen3_logprobs. */
int32_t pegainfer_qwen35_sample(pegainfer_qwen35_t m, int32_t column, float temperature, int32_t top_k, float top_p,
                                float random_val, int32_t* out_token);
int32_t pegainfer_qwen35_logprobs(pegainfer_qwen35_t m, int32_t column, uint32_t token, int32_t top_k, float* out_logprob,
                                  uint32_t* out_top_ids, float* out_top_logprobs);
/* n_steps GREEDY decode steps enqueued back to back with one host synchronisation - the twin of
 * pegainfer_qwen3_decode_greedy_chain (include/pegainfer_qwen3.h): same graph, same kernels, same bits as n_steps calls of
 * pegainfer_qwen35_decode; out_tokens [n_steps][n_requests] */
int32_t pegainfer_qwen35_decode_greedy_chain(pegainfer_qwen35_t m, int32_t n_requests, const int32_t* request_ids,
                                             const uint32_t* first_token_ids, int32_t n_steps, int32_t* out_tokens);
float pegainfer_qwen35_last_step_ms(pegainfer_qwen35_t m);
/* average ms per launch of one GEMV call site over the layers' real weights, hipEvents on the model stream (bench.py
 * roofline): which 0 = gate|up with the residual add + (1 + w) RMSNorm prologue and SwiGLU epilogue (the dominant kernel of
 * the fused bs = 1 step), 1 = down_proj */
float pegainfer_qwen35_bench_gemv(pegainfer_qwen35_t m, int32_t which, int32_t iters);
/* Per-layer hidden-state tap, the twin of pegainfer_qwen3_debug_hidden (accuracy-parity-playbook.md:15-24): while
 * enabled steps run eagerly and the residual stream leaving every layer (decode: all columns; prefill: the last prompt
 * position) stays on the device; debug_hidden copies one layer's rows, bf16 bits [rows, hidden], returns the row count */
int32_t pegainfer_qwen35_debug_hidden_enable(pegainfer_qwen35_t m, int32_t enable);
int32_t pegainfer_qwen35_debug_hidden(pegainfer_qwen35_t m, int32_t layer, void* out_host_bf16, int32_t max_rows);
int64_t pegainfer_qwen35_weight_bytes(pegainfer_qwen35_t m);

#ifdef __cplusplus
}
#endif
#endif /* PEGAINFER_QWEN35_H */
