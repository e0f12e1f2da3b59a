/*
 * pegainfer_scheduler.h - C ABI of the continuous-batching scheduler in libpegainfer_qwen3.so (SURVEY.md §8 (f)
 * rank 2: "the immediate caller of the path").  Mirrors pegainfer-qwen3-4b/src/scheduler.rs:97-327 with
 * scheduler/{plan,resolve,effects}.rs: admission by KV-page budget (a request is admitted only when its MAXIMUM
 * context fits next to the future pages of every active request; requests that can never fit are rejected), one
 * plan per iteration (Prefill / Decode / Unified), stop-token / length resolution, TokenEvent stream, failure of a
 * step -> Error to every touched request + drop, dropped receiver -> request retired.
 *
 * The reference runs the loop on a dedicated thread fed by channels; here ONE call of pegainfer_sched_step() is one
 * iteration of scheduler_loop, submissions go straight into the deferred queue and events are polled - the host
 * owns the thread.  Per-token logprobs / top_logprobs and prompt echo (executor.rs:211-284,400-434; resolve.rs:31-132;
 * effects.rs:67-90) are carried since round 4: pegainfer_sched_submit_ex, the trailing executor callbacks, the logprob
 * fields of the event and PEGAINFER_EVENT_PROMPT_TOKEN.
 */
#ifndef PEGAINFER_SCHEDULER_H
#define PEGAINFER_SCHEDULER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pegainfer_sched_t;

/* ModelExecutor (executor.rs:502-512) as a table of callbacks.  execute(): n_prefill prompt requests first, then
 * n_decode single-token requests; ids / lens (decode lens == 1) / per-request sampling params / random_vals have
 * n_prefill + n_decode entries, tokens is the concatenation; writes one token per request to out_tokens; returns 0
 * or an error (last_error() then describes it).  n_prefill == 0 -> execute_decode, n_decode == 0 -> execute_prefill,
 * both > 0 -> execute_unified. */
typedef struct {
  /* sizeof(pegainfer_executor_vtbl) as the CALLER's build sees it.  pegainfer_sched_create copies that many bytes
   * and treats every later field as NULL, so the table can grow at the end (as it did with max_batch_size) without
   * the library reading past an older caller's struct; a size that does not reach `last_error` is rejected. */
  size_t struct_size;
  void* user;
  int32_t (*page_size)(void* user);
  int32_t (*max_request_pages)(void* user);
  int32_t (*available_pages)(void* user);
  int32_t (*is_stop_token)(void* user, uint32_t token);
  int32_t (*drop_request)(void* user, uint64_t request_id);
  int32_t (*execute)(void* user, int32_t n_prefill, int32_t n_decode, const uint64_t* request_ids, const int32_t* lens,
                     const uint32_t* tokens, const float* temperature, const int32_t* top_k, const float* top_p,
                     const float* random_vals, uint32_t* out_tokens);
  const char* (*last_error)(void* user);
  /* optional (may be NULL = unlimited): the largest n_prefill + n_decode one execute() call accepts - the model's
   * decode-buffer batch (the reference lane always allocates its 64 bucket, batch_decode_buffers.rs:12-46; here it is
   * a constructor argument).  Admission leaves requests beyond it in the deferred queue instead of failing the step. */
  int32_t (*max_batch_size)(void* user);
  /* ---- optional, round 4: logprobs / echo (all may be NULL: requests then get no logprob, like a None in the reference) ----
   * logprobs: TokenLogprob of row `row` of the LAST execute() / execute_echo() call (rows = requests, prompts first) for
   *   the token that was emitted: *out_logprob, and up to top_k (id, logprob) pairs, value descending
   *   (compute_logprobs_from_cpu, executor.rs:400-434).  Returns the number of pairs written, < 0 on error.
   * execute_echo: execute() for a PREFILL plan in which at least one request asked for echo (plan.rs:62-66 any_echo): the
   *   executor additionally keeps the logits of EVERY prompt position (prefill.rs:196-212) for prompt_logprobs.
   * prompt_logprobs: TokenLogprob of `target_token` under the logits of concatenated prompt position `position` of the
   *   last execute_echo() (extract_prompt_logprobs, executor.rs:818-831). */
  int32_t (*logprobs)(void* user, int32_t row, uint32_t token, int32_t top_k, float* out_logprob, uint32_t* out_top_ids,
                      float* out_top_logprobs);
  int32_t (*execute_echo)(void* user, int32_t n_prefill, const uint64_t* request_ids, const int32_t* lens,
                          const uint32_t* tokens, const float* temperature, const int32_t* top_k, const float* top_p,
                          const float* random_vals, uint32_t* out_tokens);
  int32_t (*prompt_logprobs)(void* user, int32_t position, uint32_t target_token, int32_t top_k, float* out_logprob,
                             uint32_t* out_top_ids, float* out_top_logprobs);
} pegainfer_executor_vtbl;

enum { PEGAINFER_EVENT_TOKEN = 1, PEGAINFER_EVENT_FINISHED = 2, PEGAINFER_EVENT_ERROR = 3, PEGAINFER_EVENT_REJECTED = 4,
       /* one element of TokenEvent::PromptTokens { ids, logprobs } (engine.rs:68-71; echo requests): token = the prompt
        * token, prompt_tokens = its index, completion_tokens = the prompt length; a request's elements are consecutive */
       PEGAINFER_EVENT_PROMPT_TOKEN = 5 };
enum { PEGAINFER_FINISH_STOP = 0, PEGAINFER_FINISH_LENGTH = 1 };
enum { PEGAINFER_PLAN_NONE = 0, PEGAINFER_PLAN_PREFILL = 1, PEGAINFER_PLAN_DECODE = 2, PEGAINFER_PLAN_UNIFIED = 3 };

/* TokenEvent (pegainfer-engine/src/engine.rs:58-87) */
typedef struct {
  uint64_t request_id;
  int32_t kind;
  uint32_t token;          /* TOKEN */
  int32_t finish_reason;   /* FINISHED */
  int32_t prompt_tokens;   /* FINISHED / ERROR / REJECTED */
  int32_t completion_tokens;
  /* TOKEN / PROMPT_TOKEN: Option<TokenLogprob> (engine.rs:34-38).  The n_top (id, logprob) pairs of the event sit at
   * [top_index, top_index + n_top) of the arrays pegainfer_sched_poll_tops returns for the SAME poll call. */
  int32_t has_logprob;
  float logprob;
  int32_t n_top;
  int32_t top_index;
} pegainfer_token_event;

pegainfer_sched_t pegainfer_sched_create(const pegainfer_executor_vtbl* executor, uint64_t seed);
/* scheduler over a Qwen3 host model (pegainfer_qwen3.h): prefill -> pegainfer_qwen3_prefill, decode ->
 * pegainfer_qwen3_decode, unified -> pegainfer_qwen3_unified_step, sampled requests -> pegainfer_qwen3_sample.
 * `model` is a pegainfer_qwen3_t and must outlive the scheduler. */
pegainfer_sched_t pegainfer_sched_create_qwen3(void* model, uint64_t seed, const uint32_t* stop_tokens, int32_t n_stop);
/* scheduler over the Qwen3.5 hybrid runtime (pegainfer_qwen35.h): one prefill call per admitted prompt, one batched
 * decode per iteration; sampled requests draw through pegainfer_qwen35_sample, logprobs through pegainfer_qwen35_logprobs
 * (round 4; echo logits are not computed for this model).  `model` is a pegainfer_qwen35_t. */
pegainfer_sched_t pegainfer_sched_create_qwen35(void* model, uint64_t seed, const uint32_t* stop_tokens, int32_t n_stop);
void pegainfer_sched_destroy(pegainfer_sched_t s);

/* EngineHandle::submit: returns the RequestId the events carry (ids count up from 0 in submission order) */
uint64_t pegainfer_sched_submit(pegainfer_sched_t s, const uint32_t* prompt_tokens, int32_t n_tokens, int32_t max_tokens,
                                float temperature, int32_t top_k, float top_p, int32_t ignore_eos);
/* GenerateRequest with logprobs / echo (engine.rs:46-56): logprobs = how many top_logprobs every emitted token carries
 * (0 = no logprob at all), echo != 0 = the prompt comes back as PROMPT_TOKEN events (with logprobs when the request was
 * prefilled alone or with other prompts; a prompt admitted into a Unified step gets none, executor.rs:352-358) */
uint64_t pegainfer_sched_submit_ex(pegainfer_sched_t s, const uint32_t* prompt_tokens, int32_t n_tokens, int32_t max_tokens,
                                   float temperature, int32_t top_k, float top_p, int32_t ignore_eos, int32_t logprobs,
                                   int32_t echo);
/* the receiver of this request went away: no further events; the request is retired at its next token */
int32_t pegainfer_sched_cancel(pegainfer_sched_t s, uint64_t request_id);
/* one scheduler_loop iteration; returns the plan kind that ran, PLAN_NONE when idle, -1 when the step failed
 * (every touched request got an ERROR event and was dropped, like scheduler.rs:307-327) */
int32_t pegainfer_sched_step(pegainfer_sched_t s);
int32_t pegainfer_sched_poll(pegainfer_sched_t s, pegainfer_token_event* out, int32_t max_events);
/* the top_logprobs pairs of the events handed out by the most recent pegainfer_sched_poll; returns how many exist
 * (copies min(that, max_pairs)) */
int32_t pegainfer_sched_poll_tops(pegainfer_sched_t s, uint32_t* out_ids, float* out_logprobs, int32_t max_pairs);
int32_t pegainfer_sched_num_active(pegainfer_sched_t s);
int32_t pegainfer_sched_num_deferred(pegainfer_sched_t s);
/* message of the most recent ERROR / REJECTED event */
const char* pegainfer_sched_last_message(pegainfer_sched_t s);

/* The random_val stream the scheduler hands to its executor: the reference's StdRng::seed_from_u64(seed) +
 * `rng.random::<f32>()` (scheduler.rs:104, plan.rs:46-70), restated from the published algorithms of the crates pinned in
 * Cargo.lock (rand 0.10.1, chacha20 0.10.0, rand_core 0.10.1): ChaCha12 keyed by a PCG32 expansion of the seed, f32 = top
 * 24 bits of a word * 2^-24.  pegainfer_std_rng_stream writes the first n draws (as f32 and / or the raw words; either
 * pointer may be NULL); pegainfer_chacha_block is the block function alone (rounds = 8 / 12 / 20) so it can be checked
 * against the published keystreams.  The seed expansion has no offline vector: "parity unpinned". */
void pegainfer_chacha_block(const uint32_t* key8, uint64_t counter, int32_t rounds, uint32_t* out16);
void pegainfer_std_rng_stream(uint64_t seed, int32_t n, float* out_f32, uint32_t* out_u32);

#ifdef __cplusplus
}
#endif
#endif /* PEGAINFER_SCHEDULER_H */
