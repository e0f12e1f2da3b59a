/*
 * pegainfer_qwen3.h - C ABI of libpegainfer_qwen3.so: the C++ host runtime that sits directly
 * above the kernel ABI and mirrors, for the forward-pass hot path only, what the reference
 * keeps in Rust:
 *
 *   pegainfer-core/src/page_pool.rs, kv_pool.rs      -> PagePool / KvPool / KvState
 *   pegainfer-core/src/cuda_graph.rs:25-57           -> run_or_capture (hipGraph per bucket x path)
 *   pegainfer-core/src/weight_loader.rs:210-244      -> RoPE tables
 *   pegainfer-qwen3-4b/src/weights.rs:83-334         -> weight upload, q/k/v + gate/up vstack
 *   pegainfer-qwen3-4b/src/batch_decode_buffers.rs   -> fixed decode buffers + split-KV plan
 *   pegainfer-qwen3-4b/src/batch_decode.rs:17-295    -> batch_decode DAG
 *   pegainfer-qwen3-4b/src/prefill.rs:73-285         -> batch_prefill DAG
 *   pegainfer-qwen3-4b/src/executor.rs:541-640       -> execute_prefill / execute_decode / drop_request
 *
 * The reference's Rust crates cannot be built here (no cargo); this library is the stand-in
 * host that drives the kernels the same way, so tests/bench exercise the real call sequence.
 * Scheduler, HTTP frontend and tokenizer are out of scope.
 * All functions return 0 on success or a negative error (see pegainfer_qwen3_last_error).
 */
#ifndef PEGAINFER_QWEN3_H
#define PEGAINFER_QWEN3_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pegainfer_qwen3_t;

/* decode_mode: 0 = replay the reference's op sequence 1:1 through the reference-named symbols
 *              1 = MI355X fused decode kernels (bit-identical results, fewer launches)
 * split_policy: 0 = reference gate (bs<=2 && L>=1024, chunk max(256, ceil(L/64)))
 *               1 = MI355X policy (fill >=256 workgroups) */
pegainfer_qwen3_t pegainfer_qwen3_create(int32_t device_ordinal, int32_t hidden_size, int32_t num_layers,
                                         int32_t num_attention_heads, int32_t num_kv_heads, int32_t head_dim,
                                         int32_t intermediate_size, int32_t vocab_size, float rms_norm_eps,
                                         float rope_theta, int32_t tie_word_embeddings,
                                         int32_t max_position_embeddings, int32_t num_kv_pages,
                                         int32_t max_batch_size, int32_t enable_graph, int32_t decode_mode,
                                         int32_t split_policy);
void pegainfer_qwen3_destroy(pegainfer_qwen3_t m);
const char* pegainfer_qwen3_last_error(pegainfer_qwen3_t m);

/* weights: HF tensor names (weights.rs:102-296), host bf16 bits, uploaded verbatim */
int32_t pegainfer_qwen3_load_tensor(pegainfer_qwen3_t m, const char* name, const void* host_bf16, int64_t numel);
/* seeded N(mean, std) bf16 checkpoint generated on the device (BASELINE.md §3 synthetic weights) */
int32_t pegainfer_qwen3_fill_synthetic(pegainfer_qwen3_t m, uint64_t seed, float std);
int32_t pegainfer_qwen3_finalize(pegainfer_qwen3_t m);
/* the bf16 bits of one tensor (HF name, full logical shape; q/k/v and gate/up are rows of the stacked device matrices)
 * copied back to the host: hands the checkpoint the engine computes with - loaded or synthetic - to a checker */
int32_t pegainfer_qwen3_export_tensor(pegainfer_qwen3_t m, const char* name, void* host_bf16, int64_t numel);
/* Native checkpoint load (weight_loader.rs:18-206 + weights.rs:121-291): `path` is a
 * .safetensors file or an HF model directory (model.safetensors, or model.safetensors.index.json + shards), mmap'ed;
 * with tp_world > 1 the rank's slices are cut on the fly (q/k/v/gate/up rows, o/down columns) into a model created
 * with the LOCAL dims.  Finalises the model. */
int32_t pegainfer_qwen3_load_safetensors(pegainfer_qwen3_t m, const char* path, int32_t tp_rank, int32_t tp_world);
/* config.json -> create (local dims for the TP rank) -> load_safetensors: a model ready for prefill, no Python */
pegainfer_qwen3_t pegainfer_qwen3_from_pretrained(const char* model_dir, int32_t device_ordinal, int32_t tp_rank,
                                                  int32_t tp_world, int32_t num_kv_pages, int32_t max_batch_size,
                                                  int32_t enable_graph, int32_t decode_mode, int32_t split_policy);

/* Tensor parallel (the reference's Qwen3 TP, weights.rs:121-291,396-405, executor.rs:580-588 - there one
 * thread per rank with cudarc NCCL; here one PROCESS per GPU with RCCL over xGMI).  Create the model with the
 * LOCAL head counts / intermediate size, load the rank's weight shards (q/k/v/gate/up row-sharded, o/down
 * column-sharded, embeddings / norms / lm_head replicated), then attach: every decode/prefill step all-reduces
 * (sum, bf16, in place, on the model stream) after O-proj and after down-proj.  world == 1 is a no-op. */
int32_t pegainfer_qwen3_rccl_unique_id(void* out_128_bytes);
int32_t pegainfer_qwen3_attach_tp(pegainfer_qwen3_t m, int32_t rank, int32_t world, const void* unique_id_128_bytes);
/* Same, over a communicator the caller built (pegainfer_comm_t of include/pegainfer_comm.h; the caller keeps ownership
 * and keeps it alive until the model is destroyed).  An RCCL communicator behaves like attach_tp.  A PEER-ONLY one
 * (pegainfer_comm_create_peer_only: no RCCL, slabs mapped over hipIpc) carries every all-reduce on the one-shot
 * peer-access kernel - prefill-sized payloads in 64 KB pieces - so the sharded runtime also runs with several ranks on
 * ONE device, which is how a single-GPU box executes the TP data path (tests/test_gpu_tp_one_gpu.py).
 * A step whose one-shot all-reduce hit its bounded wait (a peer late or gone) fails with -5; nothing is returned, and the
 * failure is PERMANENT for this model: the status block is sticky, the ranks' epochs may have diverged and the failing
 * step had already advanced seq_len / appended KV of its requests.  Every later prefill / decode returns -5 at once; the
 * host drops the touched requests and rebuilds model + communicator (what a dead NCCL rank costs the reference too). */
int32_t pegainfer_qwen3_attach_comm(pegainfer_qwen3_t m, void* comm);
/* 1 when the attached communicator's <= 64 KB all-reduces take the one-shot peer-access kernel */
int32_t pegainfer_qwen3_tp_oneshot_active(pegainfer_qwen3_t m);

/* requests = KvState handles (kv_pool.rs:147-260) */
int32_t pegainfer_qwen3_new_request(pegainfer_qwen3_t m);
int32_t pegainfer_qwen3_drop_request(pegainfer_qwen3_t m, int32_t request_id);
int32_t pegainfer_qwen3_request_seq_len(pegainfer_qwen3_t m, int32_t request_id);
int32_t pegainfer_qwen3_available_pages(pegainfer_qwen3_t m);
/* KvPool::capacity_pages (kv_pool.rs): pages in the pool including the reserved padding page */
int32_t pegainfer_qwen3_capacity_pages(pegainfer_qwen3_t m);
/* the max_batch_size the model was created with (rows of the decode buffers) */
int32_t pegainfer_qwen3_max_batch_size(pegainfer_qwen3_t m);
int32_t pegainfer_qwen3_vocab_size(pegainfer_qwen3_t m);

/* batch_prefill (prefill.rs:220-285): greedy first token per request; optional last-position
 * logits copied to host as bf16 bits [n_requests, vocab]. */
int32_t pegainfer_qwen3_prefill(pegainfer_qwen3_t m, int32_t n_requests, const int32_t* request_ids,
                                const int32_t* prompt_lens, const uint32_t* tokens_concat, int32_t* out_tokens,
                                void* out_logits_host);
/* batch_prefill with echo = true (prefill.rs:196-261, compute_all_position_logits): additionally the final-norm +
 * lm_head logits of EVERY prompt position, bf16 bits [total_tokens, vocab] on the host (prompt log-probabilities). */
int32_t pegainfer_qwen3_prefill_echo(pegainfer_qwen3_t m, int32_t n_requests, const int32_t* request_ids,
                                     const int32_t* prompt_lens, const uint32_t* tokens_concat, int32_t* out_tokens,
                                     void* out_logits_host, void* out_all_logits_host);
/* unified_step (unified_forward.rs:78-198): new prompts arrive while decodes are active.  ids / lens / tokens
 * list the n_prefill prompt requests first, then n_decode requests with exactly one token each (lens == 1).
 * All token columns share the GEMMs / norms / RoPE / KV append; attention is one batch-prefill call over the
 * prompt columns plus one batch-decode call over the trailing decode columns.  Returns one greedy token (and
 * optionally one logits row) per request, prompts first. */
int32_t pegainfer_qwen3_unified_step(pegainfer_qwen3_t m, int32_t n_prefill, int32_t n_decode,
                                     const int32_t* request_ids, const int32_t* lens, const uint32_t* tokens_concat,
                                     int32_t* out_tokens, void* out_logits_host);
/* batch_decode (batch_decode.rs:17-80) + greedy token per request; optional logits [n, vocab]. */
int32_t pegainfer_qwen3_decode(pegainfer_qwen3_t m, int32_t n_requests, const int32_t* request_ids,
                               const uint32_t* token_ids, int32_t* out_tokens, void* out_logits_host);
/* n_steps GREEDY decode steps of one batch enqueued back to back with ONE host synchronisation at the end: the token of
 * step s never leaves the device (the batched top-1 inside the captured graph writes it into the token_ids slot of the
 * device metadata block, which step s + 1's upload leaves alone), the per-step metadata blocks go through a ring of pinned buffers.  Same graphs and kernels, same bits
 * as n_steps calls of pegainfer_qwen3_decode (the reference's loop, executor.rs:541-640, synchronises and samples on the
 * host every step: 30-65 us of an MI355X's 2 ms step).  out_tokens [n_steps][n_requests].  Greedy only - temperature
 * sampling, stop tokens and logprobs need the host between steps.  The chain is validated as a whole (positions, pages)
 * before any request advances. */
int32_t pegainfer_qwen3_decode_greedy_chain(pegainfer_qwen3_t m, int32_t n_requests, const int32_t* request_ids,
                                            const uint32_t* first_token_ids, int32_t n_steps, int32_t* out_tokens);
/* Sample request `column` of the LAST prefill/decode step's logits with the reference's
 * gpu_sample rule (ops/sampling.rs:109-170). */
int32_t pegainfer_qwen3_sample(pegainfer_qwen3_t m, int32_t column, float temperature, int32_t top_k, float top_p,
                               float random_val, int32_t* out_token);

/* Per-token log-probabilities (executor.rs:400-434 compute_logprobs_from_cpu, :807-816 extract_logprobs): the logits
 * row of request `column` of the LAST prefill / decode step is copied to the host as f32; out_logprob = logit[token] -
 * log_sum_exp, and the top_k largest entries (value descending; the reference's ordered insertion, literally: among equal
 * values the later index comes first, and a value equal to the last entry of a full list does not enter) go to
 * out_top_ids / out_top_logprobs.  Returns the number of top entries (min(top_k, vocab)) or < 0. */
int32_t pegainfer_qwen3_logprobs(pegainfer_qwen3_t m, int32_t column, uint32_t token, int32_t top_k, float* out_logprob,
                                 uint32_t* out_top_ids, float* out_top_logprobs);
/* the same arithmetic on a host f32 row (pure host: CPU-testable against oracle/ops.py:compute_logprobs) */
int32_t pegainfer_logprobs_from_logits(const float* logits_f32, int32_t n, uint32_t token, int32_t top_k,
                                       float* out_logprob, uint32_t* out_top_ids, float* out_top_logprobs);

/* Per-layer hidden-state tap - the reference's own debugging method (docs/playbooks/accuracy-parity-playbook.md:15-24:
 * find the first-diff token, then compare LAYERS).  While enabled, prefill / decode steps run eagerly (same kernels,
 * same bits as the graph replay) and the residual stream leaving every layer - one row per request: all columns of a
 * decode step, each request's LAST prompt position of a prefill - is kept on the device.
 * pegainfer_qwen3_debug_hidden copies layer `layer`'s rows of the last step, bf16 bits [rows, hidden], to the host and
 * returns the row count (<= max_rows), or < 0.  Test / diagnosis only; costs one D2D copy per layer while on. */
int32_t pegainfer_qwen3_debug_hidden_enable(pegainfer_qwen3_t m, int32_t enable);
int32_t pegainfer_qwen3_debug_hidden(pegainfer_qwen3_t m, int32_t layer, void* out_host_bf16, int32_t max_rows);

/* device-side timing of the last decode step's graph (hipEvent pair on the model stream), ms */
float pegainfer_qwen3_last_step_ms(pegainfer_qwen3_t m);
int32_t pegainfer_qwen3_last_attention_path(pegainfer_qwen3_t m);
/* average ms per launch of one GEMM call site over the layers' real weights, hipEvents on the model
 * stream (bench.py roofline).  which: 0 fused qkv, 1 o, 2 gate_up, 3 down, 4 lm_head (plain GEMM call sites);
 * 5 gate_up with add+RMSNorm prologue and SwiGLU epilogue, 6 qkv with add+RMSNorm prologue, 7 lm_head with
 * add+RMSNorm prologue (the fused decode step's kernels); +10 = same layer every launch (cache-warm probe);
 * bs = token columns */
float pegainfer_qwen3_bench_gemv(pegainfer_qwen3_t m, int32_t which, int32_t iters, int32_t bs);
int64_t pegainfer_qwen3_weight_bytes(pegainfer_qwen3_t m);
void* pegainfer_qwen3_stream(pegainfer_qwen3_t m);

/* ---- pure-host pieces, exported for CPU unit tests (no device needed) ---- */
/* native safetensors reader: shape (up to 4 dims), dtype string, weighted byte checksum of tensor `name`; path is a
 * file or an HF directory (index.json + shards supported).  0 ok, -1 unreadable checkpoint, -2 no such tensor */
int32_t pegainfer_safetensors_probe(const char* path, const char* name, int64_t* shape4, int32_t* ndim, char* dtype8,
                                    uint64_t* byte_sum, int32_t* num_tensors);
void* pegainfer_pagepool_create(int32_t capacity_pages);
void pegainfer_pagepool_destroy(void* pool);
int32_t pegainfer_pagepool_available(void* pool);
/* pops n pages into out (page ids) or returns -1 leaving the pool unchanged (page_pool.rs:46-70) */
int32_t pegainfer_pagepool_acquire(void* pool, int32_t n, int32_t* out_pages);
/* returns pages in reverse so the next acquire hands them out in the original order (:118-127) */
void pegainfer_pagepool_release(void* pool, const int32_t* pages, int32_t n);
/* BatchDecodeBuffers::sync_split_kv_meta / attention_path (batch_decode_buffers.rs:229-287) for
 * the given policy.  Outputs sized padded_bs*64 (+1 for o_indptr: padded_bs+1).
 * Returns the number of split slots; *use_split = 1 when the split path is chosen. */
int32_t pegainfer_split_kv_plan(int32_t policy, int32_t n_requests, const int32_t* seq_lens, int32_t padded_bs,
                                int32_t num_kv_heads, int32_t* request_indices, int32_t* kv_tile_indices,
                                int32_t* o_indptr, uint8_t* block_valid_mask, int32_t* kv_chunk_size,
                                int32_t* use_split);
int32_t pegainfer_bucket_for(int32_t bs);

#ifdef __cplusplus
}
#endif
#endif /* PEGAINFER_QWEN3_H */
