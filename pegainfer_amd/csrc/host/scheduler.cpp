// Continuous-batching scheduler (include/pegainfer_scheduler.h): the reference's scheduler_loop, one iteration
// per step() call, over an executor given as a callback table or bound to the Qwen3 host runtime.
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <deque>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "pegainfer_qwen3.h"
#include "pegainfer_qwen35.h"
#include "pegainfer_scheduler.h"

namespace psched {

struct Sampling { float temperature; int32_t top_k; float top_p; bool ignore_eos; };

struct Active {   // ActiveRequestState (scheduler.rs:31-41)
  uint64_t id; uint32_t last_token; int generated, max_tokens, prompt_len; Sampling p; int logprobs = 0;
};
struct Pending {  // PendingRequest (scheduler.rs:43-51)
  uint64_t id; std::vector<uint32_t> prompt; Sampling p; int max_tokens; int logprobs = 0; bool echo = false;
};
struct TokenLogprob {   // Option<TokenLogprob> (engine.rs:34-38)
  bool some = false; float logprob = 0.f; std::vector<uint32_t> ids; std::vector<float> vals;
};
struct Event { pegainfer_token_event e; std::vector<uint32_t> ids; std::vector<float> vals; };

static inline int pages_needed(long tokens, int page_size) { return (int)((tokens + page_size - 1) / page_size); }

// The reference's random_val stream: rand::rngs::StdRng::seed_from_u64(seed), one f32 per sampled request
// (scheduler.rs:104, plan.rs:46-70).  The generator is third-party (rand 0.10.1 / chacha20 0.10.0 / rand_core 0.10.1 in
// Cargo.lock, sources not vendored), restated from the published algorithms: ChaCha with 12 rounds, key = 32 seed bytes
// from a PCG32 expansion of the u64, 64-bit block counter from 0, stream id 0, words handed out in order; f32 = the top
// 24 bits of a word * 2^-24.  The block function is pinned by the published zero-key keystreams
// (tests/test_std_rng.py, against oracle/std_rng.py and this code through pegainfer_chacha_block / pegainfer_std_rng_stream); the seed
// expansion has no offline vector ("parity unpinned").  Greedy decoding never reads these values.
struct Rng {
  uint32_t key[8] = {0};
  uint64_t counter = 0;
  uint32_t buf[16] = {0};
  int idx = 16;

  void seed(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; ++i) {
      state = state * MUL + INC;
      const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
      const uint32_t rot = (uint32_t)(state >> 59);
      key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    counter = 0;
    idx = 16;
  }
  static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  static inline void quarter(uint32_t* s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
  }
  static void block(const uint32_t* key8, uint64_t ctr, int rounds, uint32_t* out16) {
    uint32_t init[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u};
    for (int i = 0; i < 8; ++i) init[4 + i] = key8[i];
    init[12] = (uint32_t)ctr; init[13] = (uint32_t)(ctr >> 32); init[14] = 0; init[15] = 0;
    uint32_t s[16];
    for (int i = 0; i < 16; ++i) s[i] = init[i];
    for (int r = 0; r < rounds / 2; ++r) {   // one column + one diagonal round per iteration
      quarter(s, 0, 4, 8, 12); quarter(s, 1, 5, 9, 13); quarter(s, 2, 6, 10, 14); quarter(s, 3, 7, 11, 15);
      quarter(s, 0, 5, 10, 15); quarter(s, 1, 6, 11, 12); quarter(s, 2, 7, 8, 13); quarter(s, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out16[i] = s[i] + init[i];
  }
  void refill() {
    block(key, counter, 12, buf);
    ++counter;
    idx = 0;
  }
  uint32_t next_u32() {
    if (idx >= 16) refill();
    return buf[idx++];
  }
  float next_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

// executor bound to the Qwen3 host runtime (executor.rs:541-760 for one rank)
struct Qwen3Exec {
  pegainfer_qwen3_t model;
  std::unordered_set<uint32_t> stop;
  std::unordered_map<uint64_t, int32_t> slot;  // RequestId -> model request id
  std::string err;
  // echo = true prefill (prefill.rs:196-212): logits of every prompt position, bf16 bits [total_tokens, vocab], on the host
  bool echo_request = false;
  std::vector<uint16_t> echo_logits;
  std::vector<float> row_f32;
  int64_t echo_vocab = 0;
  static int32_t page_size(void*) { return 16; }
  static int32_t max_request_pages(void* u) { return pegainfer_qwen3_capacity_pages(((Qwen3Exec*)u)->model) - 1; }
  static int32_t available_pages(void* u) { return pegainfer_qwen3_available_pages(((Qwen3Exec*)u)->model); }
  static int32_t max_batch_size(void* u) { return pegainfer_qwen3_max_batch_size(((Qwen3Exec*)u)->model); }
  static int32_t is_stop_token(void* u, uint32_t t) { return ((Qwen3Exec*)u)->stop.count(t) ? 1 : 0; }
  static int32_t drop_request(void* u, uint64_t id) {
    auto* e = (Qwen3Exec*)u;
    auto it = e->slot.find(id);
    if (it == e->slot.end()) return 0;
    const int32_t rc = pegainfer_qwen3_drop_request(e->model, it->second);
    e->slot.erase(it);
    return rc;
  }
  static const char* last_error(void* u) { return ((Qwen3Exec*)u)->err.c_str(); }
  static int32_t execute(void* u, int32_t n_pf, int32_t n_dec, const uint64_t* ids, const int32_t* lens,
                         const uint32_t* tokens, const float* temp, const int32_t* top_k, const float* top_p,
                         const float* rv, uint32_t* out) {
    auto* e = (Qwen3Exec*)u;
    const int n = n_pf + n_dec;
    std::vector<int32_t> mids(n), out_i(n);
    for (int i = 0; i < n; ++i) {
      if (i < n_pf) {
        const int32_t r = pegainfer_qwen3_new_request(e->model);
        if (r < 0) { e->err = "pegainfer_qwen3_new_request failed"; return -1; }
        e->slot[ids[i]] = r;
        mids[i] = r;
      } else {
        auto it = e->slot.find(ids[i]);
        if (it == e->slot.end()) { e->err = "decode request without model state"; return -1; }
        mids[i] = it->second;
      }
    }
    int32_t rc;
    if (n_dec == 0 && e->echo_request) {
      int64_t T = 0;
      for (int i = 0; i < n_pf; ++i) T += lens[i];
      e->echo_vocab = pegainfer_qwen3_vocab_size(e->model);
      e->echo_logits.resize((size_t)T * (size_t)e->echo_vocab);
      rc = pegainfer_qwen3_prefill_echo(e->model, n_pf, mids.data(), lens, tokens, out_i.data(), nullptr, e->echo_logits.data());
    } else if (n_dec == 0) rc = pegainfer_qwen3_prefill(e->model, n_pf, mids.data(), lens, tokens, out_i.data(), nullptr);
    else if (n_pf == 0) rc = pegainfer_qwen3_decode(e->model, n_dec, mids.data(), tokens, out_i.data(), nullptr);
    else rc = pegainfer_qwen3_unified_step(e->model, n_pf, n_dec, mids.data(), lens, tokens, out_i.data(), nullptr);
    if (rc) { const char* m = pegainfer_qwen3_last_error(e->model); e->err = m ? m : "model step failed"; return rc; }
    for (int i = 0; i < n; ++i) {
      const bool greedy = (temp[i] <= 0.0f || top_k[i] == 1) && top_p[i] >= 1.0f;  // ops/sampling.rs:122
      if (!greedy) {
        int32_t t = 0;
        rc = pegainfer_qwen3_sample(e->model, i, temp[i], top_k[i], top_p[i], rv[i], &t);
        if (rc) { const char* m = pegainfer_qwen3_last_error(e->model); e->err = m ? m : "sample failed"; return rc; }
        out_i[i] = t;
      }
      out[i] = (uint32_t)out_i[i];
    }
    return 0;
  }
};

// logprobs / echo callbacks of the Qwen3 executor (executor.rs:211-284, 807-831)
struct Qwen3ExecLp {
  static int32_t logprobs(void* u, int32_t row, uint32_t token, int32_t top_k, float* lp, uint32_t* ids, float* vals) {
    auto* e = (Qwen3Exec*)u;
    const int32_t rc = pegainfer_qwen3_logprobs(e->model, row, token, top_k, lp, ids, vals);
    if (rc < 0) { const char* m = pegainfer_qwen3_last_error(e->model); e->err = m ? m : "logprobs failed"; }
    return rc;
  }
  static int32_t execute_echo(void* u, int32_t n_pf, const uint64_t* ids, const int32_t* lens, const uint32_t* tokens,
                              const float* temp, const int32_t* top_k, const float* top_p, const float* rv, uint32_t* out) {
    auto* e = (Qwen3Exec*)u;
    e->echo_request = true;
    const int32_t rc = Qwen3Exec::execute(u, n_pf, 0, ids, lens, tokens, temp, top_k, top_p, rv, out);
    e->echo_request = false;
    return rc;
  }
  static int32_t prompt_logprobs(void* u, int32_t pos, uint32_t target, int32_t top_k, float* lp, uint32_t* ids, float* vals) {
    auto* e = (Qwen3Exec*)u;
    const int64_t V = e->echo_vocab;
    if (V <= 0 || pos < 0 || (int64_t)(pos + 1) * V > (int64_t)e->echo_logits.size()) { e->err = "no echo logits for that position"; return -1; }
    e->row_f32.resize((size_t)V);
    const uint16_t* src = e->echo_logits.data() + (size_t)pos * V;
    for (int64_t i = 0; i < V; ++i) { const uint32_t w = (uint32_t)src[i] << 16; std::memcpy(&e->row_f32[i], &w, 4); }
    return pegainfer_logprobs_from_logits(e->row_f32.data(), (int32_t)V, target, top_k, lp, ids, vals);
  }
};

// executor bound to the Qwen3.5 hybrid runtime: the reference prefills one request per call (prefill.rs:21) and
// decodes the active set as one batch (batch_decode.rs:113); a Unified plan is its prefills followed by one batched
// decode (the recurrent state makes a fused mixed step a different kernel set, unified_forward.rs - not built).
// Sampled requests draw their token right after their own prefill / the batched decode (gpu_sample over that row);
// logprobs of prompt rows come from a host copy of the row taken at the same moment (the next prefill reuses the buffer).
struct Qwen35Exec {
  pegainfer_qwen35_t model;
  std::unordered_set<uint32_t> stop;
  std::unordered_map<uint64_t, int32_t> slot;
  std::string err;
  int n_pf_last = 0;                              // rows [0, n_pf_last) of the last execute() were prefills
  std::vector<std::vector<float>> pf_rows;        // f32 logits rows of the prompts that asked for logprobs (empty otherwise)
  std::vector<uint8_t> want_lp;                   // per prompt row of the NEXT execute(): the request carries logprobs > 0
                                                  // (set by the owning scheduler; empty = nobody asked)
  std::vector<uint16_t> row_bits;
  static int32_t page_size(void*) { return 16; }
  static int32_t max_request_pages(void* u) { return pegainfer_qwen35_capacity_pages(((Qwen35Exec*)u)->model) - 1; }
  static int32_t available_pages(void* u) { return pegainfer_qwen35_available_pages(((Qwen35Exec*)u)->model); }
  static int32_t max_batch_size(void* u) { return pegainfer_qwen35_max_batch_size(((Qwen35Exec*)u)->model); }
  static int32_t is_stop_token(void* u, uint32_t t) { return ((Qwen35Exec*)u)->stop.count(t) ? 1 : 0; }
  static int32_t drop_request(void* u, uint64_t id) {
    auto* e = (Qwen35Exec*)u;
    auto it = e->slot.find(id);
    if (it == e->slot.end()) return 0;
    const int32_t rc = pegainfer_qwen35_drop_request(e->model, it->second);
    e->slot.erase(it);
    return rc;
  }
  static const char* last_error(void* u) { return ((Qwen35Exec*)u)->err.c_str(); }
  static bool greedy(float t, int32_t k, float p) { return (t <= 0.0f || k == 1) && p >= 1.0f; }   // ops/sampling.rs:122
  static int32_t execute(void* u, int32_t n_pf, int32_t n_dec, const uint64_t* ids, const int32_t* lens,
                         const uint32_t* tokens, const float* temp, const int32_t* top_k, const float* top_p,
                         const float* rv, uint32_t* out) {
    auto* e = (Qwen35Exec*)u;
    auto fail = [&](const char* what) {
      const char* m = pegainfer_qwen35_last_error(e->model);
      e->err = std::string(what) + ": " + (m ? m : "");
      return -1;
    };
    e->n_pf_last = n_pf;
    e->pf_rows.assign((size_t)n_pf, {});
    const int32_t V = pegainfer_qwen35_vocab_size(e->model);
    size_t off = 0;
    for (int i = 0; i < n_pf; ++i) {
      const int32_t r = pegainfer_qwen35_new_request(e->model);
      if (r < 0) return fail("new_request");
      e->slot[ids[i]] = r;
      int32_t tok = 0;
      // the 300-500 KB logits row comes back (and is widened to f32) only for a prompt whose request asked for
      // logprobs: the next prefill reuses the device buffer, so the copy has to be taken now or never
      const bool keep_row = (size_t)i < e->want_lp.size() && e->want_lp[(size_t)i];
      if (keep_row) e->row_bits.resize((size_t)V);
      if (pegainfer_qwen35_prefill(e->model, r, lens[i], tokens + off, &tok, keep_row ? e->row_bits.data() : nullptr)) return fail("prefill");
      if (!greedy(temp[i], top_k[i], top_p[i]) &&
          pegainfer_qwen35_sample(e->model, 0, temp[i], top_k[i], top_p[i], rv[i], &tok))
        return fail("sample");
      if (keep_row) {
        e->pf_rows[i].resize((size_t)V);
        for (int32_t j = 0; j < V; ++j) { const uint32_t w = (uint32_t)e->row_bits[j] << 16; std::memcpy(&e->pf_rows[i][j], &w, 4); }
      }
      out[i] = (uint32_t)tok;
      off += (size_t)lens[i];
    }
    if (n_dec > 0) {
      std::vector<int32_t> mids(n_dec), toks(n_dec);
      for (int j = 0; j < n_dec; ++j) {
        auto it = e->slot.find(ids[n_pf + j]);
        if (it == e->slot.end()) { e->err = "decode request without model state"; return -1; }
        mids[j] = it->second;
      }
      if (pegainfer_qwen35_decode(e->model, n_dec, mids.data(), tokens + off, toks.data(), nullptr)) return fail("decode");
      for (int j = 0; j < n_dec; ++j) {
        const int i = n_pf + j;
        if (!greedy(temp[i], top_k[i], top_p[i]) &&
            pegainfer_qwen35_sample(e->model, j, temp[i], top_k[i], top_p[i], rv[i], &toks[j]))
          return fail("sample");
        out[i] = (uint32_t)toks[j];
      }
    }
    return 0;
  }
  static int32_t logprobs(void* u, int32_t row, uint32_t token, int32_t top_k, float* lp, uint32_t* ids, float* vals) {
    auto* e = (Qwen35Exec*)u;
    if (row < e->n_pf_last) {
      const std::vector<float>& r = e->pf_rows[(size_t)row];
      if (r.empty()) { e->err = "logprobs of a prompt row that was not flagged in want_lp"; return -1; }
      return pegainfer_logprobs_from_logits(r.data(), (int32_t)r.size(), token, top_k, lp, ids, vals);
    }
    const int32_t rc = pegainfer_qwen35_logprobs(e->model, row - e->n_pf_last, token, top_k, lp, ids, vals);
    if (rc < 0) { const char* m = pegainfer_qwen35_last_error(e->model); e->err = m ? m : "logprobs failed"; }
    return rc;
  }
};

struct Scheduler {
  pegainfer_executor_vtbl ex;
  Qwen3Exec* owned = nullptr;
  Qwen35Exec* owned35 = nullptr;
  Rng rng;
  std::vector<Active> active;
  std::vector<Pending> deferred;
  std::unordered_set<uint64_t> closed;
  std::deque<Event> events;
  std::vector<uint32_t> poll_ids;    // top_logprobs of the events handed out by the last poll
  std::vector<float> poll_vals;
  uint64_t next_id = 0;
  std::string last_message;

  bool send(uint64_t id, int kind, uint32_t token, int reason, int prompt_tokens, int completion_tokens,
            const TokenLogprob* lp = nullptr) {
    if (closed.count(id)) return false;
    Event ev{pegainfer_token_event{id, kind, token, reason, prompt_tokens, completion_tokens, 0, 0.f, 0, 0}, {}, {}};
    if (lp && lp->some) {
      ev.e.has_logprob = 1; ev.e.logprob = lp->logprob; ev.e.n_top = (int32_t)lp->ids.size();
      ev.ids = lp->ids; ev.vals = lp->vals;
    }
    events.push_back(std::move(ev));
    return true;
  }
  // Some(extract_logprobs(..)) when the request asked for logprobs (executor.rs:222-226,273-277) and the executor can
  TokenLogprob row_logprob(int row, uint32_t token, int top_k, bool* failed) {
    TokenLogprob lp;
    if (top_k <= 0 || !ex.logprobs) return lp;
    lp.ids.resize((size_t)top_k); lp.vals.resize((size_t)top_k);
    const int32_t n = ex.logprobs(ex.user, row, token, top_k, &lp.logprob, lp.ids.data(), lp.vals.data());
    if (n < 0) { *failed = true; return TokenLogprob(); }
    lp.ids.resize((size_t)n); lp.vals.resize((size_t)n);
    lp.some = true;
    return lp;
  }
  static int max_tokens_of(int prompt_len, int max_tokens) { return prompt_len + std::max(max_tokens - 1, 0); }

  int step() {
    if (active.empty() && deferred.empty()) return PEGAINFER_PLAN_NONE;
    // ---- admit_deferred_requests (scheduler.rs:222-261) ----
    const int ps = ex.page_size(ex.user);
    long future = 0;
    for (const Active& a : active)
      future += std::max(0, pages_needed(max_tokens_of(a.prompt_len, a.max_tokens), ps) -
                                pages_needed(a.prompt_len + std::max(a.generated - 1, 0), ps));
    long budget = std::max<long>(0, (long)ex.available_pages(ex.user) - future);
    const int max_req = ex.max_request_pages(ex.user);
    // rows one execute() call may carry: active requests always decode, so only the remainder can be admitted
    long rows_left = ex.max_batch_size ? std::max<long>(0, (long)ex.max_batch_size(ex.user) - (long)active.size()) : -1;
    std::vector<Pending> pending, still;
    for (Pending& r : deferred) {
      const int need = pages_needed(max_tokens_of((int)r.prompt.size(), r.max_tokens), ps);
      if (need > max_req) {
        last_message = "request requires more KV pages than this model instance can provide: prompt_tokens=" +
                       std::to_string(r.prompt.size()) + ", max_context_tokens=" +
                       std::to_string(max_tokens_of((int)r.prompt.size(), r.max_tokens));
        send(r.id, PEGAINFER_EVENT_REJECTED, 0, 0, (int)r.prompt.size(), 0);
      } else if (need <= budget && rows_left != 0) {
        budget -= need;
        if (rows_left > 0) --rows_left;
        pending.push_back(std::move(r));
      } else {
        still.push_back(std::move(r));
      }
    }
    deferred = std::move(still);
    // ---- build_next_plan (plan.rs:31-45) ----
    const bool have_active = !active.empty();
    int plan;
    if (!pending.empty() && have_active) plan = PEGAINFER_PLAN_UNIFIED;
    else if (!pending.empty()) plan = PEGAINFER_PLAN_PREFILL;
    else if (have_active) plan = PEGAINFER_PLAN_DECODE;
    else return PEGAINFER_PLAN_NONE;
    // ---- execute_plan (plan.rs:47-117): prompts first, one random_val per request ----
    const int n_pf = plan == PEGAINFER_PLAN_DECODE ? 0 : (int)pending.size();
    const int n_dec = plan == PEGAINFER_PLAN_PREFILL ? 0 : (int)active.size();
    const int n = n_pf + n_dec;
    std::vector<uint64_t> ids(n);
    std::vector<int32_t> lens(n), top_k(n);
    std::vector<float> temp(n), top_p(n), rv(n);
    std::vector<uint32_t> tokens, out(n);
    for (int i = 0; i < n_pf; ++i) {
      const Pending& r = pending[i];
      ids[i] = r.id; lens[i] = (int32_t)r.prompt.size();
      temp[i] = r.p.temperature; top_k[i] = r.p.top_k; top_p[i] = r.p.top_p; rv[i] = rng.next_f32();
      tokens.insert(tokens.end(), r.prompt.begin(), r.prompt.end());
    }
    for (int j = 0; j < n_dec; ++j) {
      const Active& a = active[j];
      const int i = n_pf + j;
      ids[i] = a.id; lens[i] = 1;
      temp[i] = a.p.temperature; top_k[i] = a.p.top_k; top_p[i] = a.p.top_p; rv[i] = rng.next_f32();
      tokens.push_back(a.last_token);
    }
    bool any_echo = false;   // plan.rs:62-66: only a pure Prefill plan computes the all-position logits
    if (plan == PEGAINFER_PLAN_PREFILL)
      for (const Pending& r : pending) any_echo = any_echo || r.echo;
    const bool echo_step = any_echo && ex.execute_echo && ex.prompt_logprobs;
    if (owned35) {   // its prefill keeps a host copy of a prompt's logits row only when that request wants logprobs
      owned35->want_lp.assign((size_t)n_pf, 0);
      for (int i = 0; i < n_pf; ++i) owned35->want_lp[(size_t)i] = pending[i].logprobs > 0;
    }
    int rc = echo_step ? ex.execute_echo(ex.user, n_pf, ids.data(), lens.data(), tokens.data(), temp.data(), top_k.data(),
                                         top_p.data(), rv.data(), out.data())
                       : ex.execute(ex.user, n_pf, n_dec, ids.data(), lens.data(), tokens.data(), temp.data(), top_k.data(),
                                    top_p.data(), rv.data(), out.data());
    // logprobs of the emitted tokens + prompt logprobs: part of the executor's result (build_*_request_results,
    // executor.rs:211-284) - a failure here fails the step like any other executor error
    std::vector<TokenLogprob> lps(n);
    std::vector<std::vector<TokenLogprob>> echo_lps(n_pf);
    if (!rc) {
      bool failed = false;
      size_t off = 0;
      for (int i = 0; i < n_pf && !failed; ++i) {
        const Pending& r = pending[i];
        lps[i] = row_logprob(i, out[i], r.logprobs, &failed);
        if (r.echo) {
          echo_lps[i].resize(r.prompt.size());          // [None, lp(1), ..., lp(n - 1)] or all None (executor.rs:227-252)
          if (echo_step)
            for (size_t j = 1; j < r.prompt.size() && !failed; ++j) {
              TokenLogprob& lp = echo_lps[i][j];
              const int k = std::max(r.logprobs, 0);
              lp.ids.resize((size_t)k); lp.vals.resize((size_t)k);
              const int32_t nt = ex.prompt_logprobs(ex.user, (int32_t)(off + j - 1), r.prompt[j], k, &lp.logprob,
                                                    lp.ids.data(), lp.vals.data());
              if (nt < 0) { lp = TokenLogprob(); continue; }   // extract_prompt_logprobs -> None on failure (.ok())
              lp.ids.resize((size_t)nt); lp.vals.resize((size_t)nt);
              lp.some = true;
            }
        }
        off += r.prompt.size();
      }
      for (int j = 0; j < n_dec && !failed; ++j) lps[n_pf + j] = row_logprob(n_pf + j, out[n_pf + j], active[j].logprobs, &failed);
      if (failed) rc = -1;
    }
    if (rc) {  // fail_touched_requests (scheduler.rs:307-327): active targets first, then the pending ones
      const char* m = ex.last_error ? ex.last_error(ex.user) : nullptr;
      last_message = m ? m : "execution step failed";
      for (int j = 0; j < n_dec; ++j) {
        send(active[j].id, PEGAINFER_EVENT_ERROR, 0, 0, active[j].prompt_len, active[j].generated);
        ex.drop_request(ex.user, active[j].id);
      }
      for (int i = 0; i < n_pf; ++i) {
        send(pending[i].id, PEGAINFER_EVENT_ERROR, 0, 0, (int)pending[i].prompt.size(), 0);
        ex.drop_request(ex.user, pending[i].id);
      }
      active.clear();
      return -1;
    }
    // ---- prompt echoes first (resolve.rs:40-49, effects.rs:75-82) ----
    for (int i = 0; i < n_pf; ++i) {
      const Pending& r = pending[i];
      if (!r.echo) continue;
      for (size_t j = 0; j < r.prompt.size(); ++j)
        send(r.id, PEGAINFER_EVENT_PROMPT_TOKEN, r.prompt[j], 0, (int)j, (int)r.prompt.size(), &echo_lps[i][j]);
    }
    // ---- decode results: resolve.rs:96-132 + effects.rs:84-159 ----
    std::vector<size_t> retire;
    for (int j = 0; j < n_dec; ++j) {
      const uint64_t id = ids[n_pf + j];
      const uint32_t tok = out[n_pf + j];
      size_t idx = active.size();
      for (size_t k = 0; k < active.size(); ++k)
        if (active[k].id == id) { idx = k; break; }
      if (idx == active.size()) continue;
      Active& a = active[idx];
      const int completion = a.generated + 1;
      if (!a.p.ignore_eos && ex.is_stop_token(ex.user, tok)) {
        send(id, PEGAINFER_EVENT_FINISHED, 0, PEGAINFER_FINISH_STOP, a.prompt_len, completion);
        ex.drop_request(ex.user, id);
        retire.push_back(idx);
      } else if (completion >= a.max_tokens) {
        if (send(id, PEGAINFER_EVENT_TOKEN, tok, 0, 0, 0, &lps[n_pf + j]))
          send(id, PEGAINFER_EVENT_FINISHED, 0, PEGAINFER_FINISH_LENGTH, a.prompt_len, completion);
        ex.drop_request(ex.user, id);
        retire.push_back(idx);
      } else if (!send(id, PEGAINFER_EVENT_TOKEN, tok, 0, 0, 0, &lps[n_pf + j])) {
        ex.drop_request(ex.user, id);
        retire.push_back(idx);
      } else {
        a.last_token = tok;
        a.generated = completion;
      }
    }
    for (size_t r = retire.size(); r-- > 0;) {  // Vec::swap_remove in reverse
      active[retire[r]] = active.back();
      active.pop_back();
    }
    // ---- prefill results: resolve.rs:31-94 + effects.rs:164-216 ----
    for (int i = 0; i < n_pf; ++i) {
      const Pending& r = pending[i];
      const uint32_t tok = out[i];
      const int plen = (int)r.prompt.size();
      if (!r.p.ignore_eos && ex.is_stop_token(ex.user, tok)) {
        send(r.id, PEGAINFER_EVENT_FINISHED, 0, PEGAINFER_FINISH_STOP, plen, 0);
        ex.drop_request(ex.user, r.id);
      } else if (r.max_tokens <= 1) {
        if (send(r.id, PEGAINFER_EVENT_TOKEN, tok, 0, 0, 0, &lps[i]))
          send(r.id, PEGAINFER_EVENT_FINISHED, 0, PEGAINFER_FINISH_LENGTH, plen, 1);
        ex.drop_request(ex.user, r.id);
      } else if (send(r.id, PEGAINFER_EVENT_TOKEN, tok, 0, 0, 0, &lps[i])) {
        active.push_back(Active{r.id, tok, 1, r.max_tokens, plen, r.p, r.logprobs});
      } else {
        ex.drop_request(ex.user, r.id);
      }
    }
    return plan;
  }
};

}  // namespace psched

using psched::Scheduler;
static Scheduler* SC(pegainfer_sched_t s) { return static_cast<Scheduler*>(s); }

extern "C" {

pegainfer_sched_t pegainfer_sched_create(const pegainfer_executor_vtbl* executor, uint64_t seed) {
  if (!executor) return nullptr;
  // versioned copy: only the bytes the caller's build knows about; optional trailing callbacks default to NULL
  const size_t mandatory = offsetof(pegainfer_executor_vtbl, last_error) + sizeof(executor->last_error);
  if (executor->struct_size < mandatory) return nullptr;
  pegainfer_executor_vtbl v;
  std::memset(&v, 0, sizeof(v));
  std::memcpy(&v, executor, std::min(executor->struct_size, sizeof(v)));
  if (!v.page_size || !v.max_request_pages || !v.available_pages || !v.is_stop_token || !v.drop_request || !v.execute)
    return nullptr;
  Scheduler* s = new Scheduler();
  s->ex = v;
  s->rng.seed(seed);
  return s;
}
pegainfer_sched_t pegainfer_sched_create_qwen3(void* model, uint64_t seed, const uint32_t* stop_tokens, int32_t n_stop) {
  if (!model) return nullptr;
  auto* e = new psched::Qwen3Exec();
  e->model = model;
  for (int i = 0; i < n_stop; ++i) e->stop.insert(stop_tokens[i]);
  pegainfer_executor_vtbl v{sizeof(pegainfer_executor_vtbl), e, &psched::Qwen3Exec::page_size, &psched::Qwen3Exec::max_request_pages,
                            &psched::Qwen3Exec::available_pages, &psched::Qwen3Exec::is_stop_token,
                            &psched::Qwen3Exec::drop_request, &psched::Qwen3Exec::execute, &psched::Qwen3Exec::last_error,
                            &psched::Qwen3Exec::max_batch_size, &psched::Qwen3ExecLp::logprobs,
                            &psched::Qwen3ExecLp::execute_echo, &psched::Qwen3ExecLp::prompt_logprobs};
  Scheduler* s = static_cast<Scheduler*>(pegainfer_sched_create(&v, seed));
  s->owned = e;
  return s;
}
pegainfer_sched_t pegainfer_sched_create_qwen35(void* model, uint64_t seed, const uint32_t* stop_tokens, int32_t n_stop) {
  if (!model) return nullptr;
  auto* e = new psched::Qwen35Exec();
  e->model = model;
  for (int i = 0; i < n_stop; ++i) e->stop.insert(stop_tokens[i]);
  pegainfer_executor_vtbl v{sizeof(pegainfer_executor_vtbl), e, &psched::Qwen35Exec::page_size, &psched::Qwen35Exec::max_request_pages,
                            &psched::Qwen35Exec::available_pages, &psched::Qwen35Exec::is_stop_token,
                            &psched::Qwen35Exec::drop_request, &psched::Qwen35Exec::execute, &psched::Qwen35Exec::last_error,
                            &psched::Qwen35Exec::max_batch_size, &psched::Qwen35Exec::logprobs, nullptr, nullptr};
  Scheduler* s = static_cast<Scheduler*>(pegainfer_sched_create(&v, seed));
  s->owned35 = e;
  return s;
}
void pegainfer_sched_destroy(pegainfer_sched_t s) {
  if (!s) return;
  delete SC(s)->owned;
  delete SC(s)->owned35;
  delete SC(s);
}
uint64_t pegainfer_sched_submit_ex(pegainfer_sched_t s, const uint32_t* prompt_tokens, int32_t n_tokens, int32_t max_tokens,
                                   float temperature, int32_t top_k, float top_p, int32_t ignore_eos, int32_t logprobs,
                                   int32_t echo) {
  Scheduler* sc = SC(s);
  psched::Pending p;
  p.id = sc->next_id++;
  p.prompt.assign(prompt_tokens, prompt_tokens + (n_tokens > 0 ? n_tokens : 0));
  p.p = psched::Sampling{temperature, top_k, top_p, ignore_eos != 0};
  p.max_tokens = max_tokens;
  p.logprobs = logprobs > 0 ? logprobs : 0;
  p.echo = echo != 0;
  sc->deferred.push_back(std::move(p));
  return sc->deferred.back().id;
}
uint64_t pegainfer_sched_submit(pegainfer_sched_t s, const uint32_t* prompt_tokens, int32_t n_tokens, int32_t max_tokens,
                                float temperature, int32_t top_k, float top_p, int32_t ignore_eos) {
  return pegainfer_sched_submit_ex(s, prompt_tokens, n_tokens, max_tokens, temperature, top_k, top_p, ignore_eos, 0, 0);
}
int32_t pegainfer_sched_cancel(pegainfer_sched_t s, uint64_t request_id) {
  SC(s)->closed.insert(request_id);
  return 0;
}
int32_t pegainfer_sched_step(pegainfer_sched_t s) { return SC(s)->step(); }
int32_t pegainfer_sched_poll(pegainfer_sched_t s, pegainfer_token_event* out, int32_t max_events) {
  Scheduler* sc = SC(s);
  int n = 0;
  sc->poll_ids.clear();
  sc->poll_vals.clear();
  while (n < max_events && !sc->events.empty()) {
    psched::Event& ev = sc->events.front();
    ev.e.top_index = (int32_t)sc->poll_ids.size();
    sc->poll_ids.insert(sc->poll_ids.end(), ev.ids.begin(), ev.ids.end());
    sc->poll_vals.insert(sc->poll_vals.end(), ev.vals.begin(), ev.vals.end());
    out[n++] = ev.e;
    sc->events.pop_front();
  }
  return n;
}
int32_t pegainfer_sched_poll_tops(pegainfer_sched_t s, uint32_t* out_ids, float* out_logprobs, int32_t max_pairs) {
  Scheduler* sc = SC(s);
  const int32_t n = (int32_t)sc->poll_ids.size();
  const int32_t m = std::min(n, std::max(max_pairs, 0));
  if (out_ids && m > 0) std::memcpy(out_ids, sc->poll_ids.data(), (size_t)m * 4);
  if (out_logprobs && m > 0) std::memcpy(out_logprobs, sc->poll_vals.data(), (size_t)m * 4);
  return n;
}
int32_t pegainfer_sched_num_active(pegainfer_sched_t s) { return (int32_t)SC(s)->active.size(); }
int32_t pegainfer_sched_num_deferred(pegainfer_sched_t s) { return (int32_t)SC(s)->deferred.size(); }
const char* pegainfer_sched_last_message(pegainfer_sched_t s) { return SC(s)->last_message.c_str(); }

void pegainfer_chacha_block(const uint32_t* key8, uint64_t counter, int32_t rounds, uint32_t* out16) {
  psched::Rng::block(key8, counter, rounds, out16);
}

void pegainfer_std_rng_stream(uint64_t seed, int32_t n, float* out_f32, uint32_t* out_u32) {
  psched::Rng r;
  r.seed(seed);
  for (int32_t i = 0; i < n; ++i) {
    const uint32_t w = r.next_u32();
    if (out_u32) out_u32[i] = w;
    if (out_f32) out_f32[i] = (float)(w >> 8) * (1.0f / 16777216.0f);
  }
}

}  // extern "C"
