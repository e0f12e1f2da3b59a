// C++ host runtime for the Qwen3 forward-pass hot path on MI355X (see include/pegainfer_qwen3.h for
// the reference files each piece mirrors).  One model instance = one device + one in-order HIP stream
// (reference: DeviceContext, pegainfer-kernels/src/tensor.rs:12-67); every op is enqueued on it through
// the C ABI of libpegainfer_kernels_hip.so; the decode step is captured once per (batch bucket,
// attention path) into a hipGraph and replayed (core/src/cuda_graph.rs:25-57), with all per-step
// metadata packed into ONE pinned block and uploaded by ONE hipMemcpyAsync before the launch
// (the reference issues 13 small H2D copies, batch_decode.rs:51-59).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "kv_pool.h"
#include "safetensors_loader.h"
#include "pegainfer_kernels.h"
#include "pegainfer_kernels_ext.h"
#include "pegainfer_comm.h"
#include "pegainfer_qwen3.h"

namespace pq {

#define PQ_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                          \
      return -1;                                                                             \
    }                                                                                        \
  } while (0)

static inline uint16_t host_f2bf(float f) {  // RNE, like half::bf16::from_f32
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

// seeded N(mean, std) bf16 fill: hash(index, seed) -> two uniforms -> Box-Muller
__global__ void fill_normal_kernel(Half* out, long n, uint64_t seed, float std, float mean) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u1 = ((float)(z >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)((z >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
    const float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    out[i] = __builtin_bit_cast(uint16_t, static_cast<__bf16>(mean + std * g));
  }
}

struct Layer {
  Half *qkv = nullptr, *o = nullptr, *q_norm = nullptr, *k_norm = nullptr, *gate_up = nullptr, *down = nullptr,
       *ln1 = nullptr, *ln2 = nullptr;
};

// byte offsets of the per-step metadata inside the packed block (all int32 unless noted)
struct MetaLayout {
  size_t token_ids, positions, page_indptr, last_page_len, request_indices, kv_tile_indices, kv_chunk_size,
      split_request_indices, split_kv_tile_indices, split_kv_chunk_size, split_o_indptr, split_valid /*u8*/,
      slot_desc /*8 x int32 per slot*/, page_indices, total;
};

struct Model {
  // config
  int device, H, L, Hq, Hkv, D, I, V, tie, max_pos, num_pages, max_bs, enable_graph, decode_mode, split_policy;
  float eps, theta;
  int q_dim, kv_dim;
  std::string err;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // weights
  Half *embed = nullptr, *lm_head = nullptr, *final_norm = nullptr, *cos = nullptr, *sin = nullptr;
  std::vector<Layer> layers;
  std::vector<void*> owned;
  int64_t weight_bytes = 0;
  bool finalized = false;
  // kv
  KvLayout layout;
  PagePool pool;
  Half* kv_buffer = nullptr;
  int padding_page = 0;
  std::vector<KvState> requests;
  // decode buffers
  Half *normed = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *attn_out = nullptr, *attn_proj = nullptr,
       *gate_up_out = nullptr, *mlp_act = nullptr, *mlp_out = nullptr, *hidden = nullptr, *logits = nullptr,
       *hidden2 = nullptr, *qkv_out = nullptr;
  Half* split_tmp_v = nullptr;
  float* split_tmp_s = nullptr;
  int32_t* tokens_out_d = nullptr;
  uint8_t* top1_state = nullptr;
  // The split-KV partials are merged by the last workgroup of each (request, kv head) inside the attention launch
  // (write-through sc1 partials, relaxed agent ticket, batched sc1 loads) instead of by merge_states_kernel: same
  // bits, 2.47 -> 2.33 ms/step at bs 1 / ctx 1024 once the merge loads were batched.  The counters are re-zeroed at
  // the head of every captured step, so an aborted launch cannot leave one armed.  PEGAINFER_FUSED_MERGE=0 keeps
  // the separate merge launch (A/B).
  bool fused_merge = [] { const char* e = getenv("PEGAINFER_FUSED_MERGE"); return !(e && e[0] == '0'); }();
  // PEGAINFER_MID_BATCH_FUSED=0: decode batches of 17..64 keep the reference op sequence (A/B + bit-equality probe)
  bool mid_batch_fused = [] { const char* e = getenv("PEGAINFER_MID_BATCH_FUSED"); return !(e && e[0] == '0'); }();
  int32_t* merge_ctr = nullptr;  // per (request, kv head) arrival counters of the in-kernel split-KV merge
  // single-request steps: attention + o_proj in ONE launch (pegainfer_fused_decode_attention_oproj): the o_proj rows are
  // requested into registers while the attention's chain of round trips runs.  One arrival counter per layer (a cache
  // line each), cleared once per step; PEGAINFER_ATTN_OPROJ=0 keeps the two launches (A/B, bit-identical).
  bool attn_oproj = [] { const char* e = getenv("PEGAINFER_ATTN_OPROJ"); return !(e && e[0] == '0'); }();
  int32_t* attn_done = nullptr;   // per layer: one arrival counter per kv head group, a cache line each (kDoneInts ints)
  size_t kDoneInts = 0;                       // Hkv x 32 ints per layer (set in init)
  uint32_t *attn_status = nullptr, *attn_status_host = nullptr;
  bool oproj_step = false;   // the step being recorded / run uses the fused launch
  bool oproj_shape_ok = true;   // the model's shape fits the fused launch at bs 1 (asked at init; e.g. hidden 4096 does not)
  bool oproj_shape_ok2 = true;  // ... and at bs 2
  bool oproj_plan = false;   // this step's split plan keeps the padding slots the fused launch needs (single request, <= 2304 tokens)
  int oproj_fallbacks = 0;
  // sampling scratch (ops/sampling.rs)
  float* probs_scratch = nullptr;
  Half* top1_value = nullptr;
  uint8_t* row_states = nullptr;
  uint8_t* valid_scratch = nullptr;
  int32_t* sample_out_d = nullptr;
  // metadata
  MetaLayout ml;
  uint8_t* meta_host = nullptr;  // pinned
  uint8_t* meta_dev = nullptr;
  int32_t* tokens_out_host = nullptr;  // pinned
  hipGraphExec_t graphs[kNumBuckets][3];   // [bucket][0 = non-partition | 1 = split-KV | 2 = split-KV with the fused attention + o_proj launch]
  // prefill workspace (grow-only)
  size_t pf_cap_tokens = 0;
  Half *pf_hidden = nullptr, *pf_hidden_out = nullptr, *pf_normed = nullptr, *pf_q = nullptr, *pf_k = nullptr,
       *pf_v = nullptr, *pf_o = nullptr, *pf_gate_up = nullptr, *pf_act = nullptr, *pf_attn = nullptr;
  uint8_t* pf_meta_dev = nullptr;
  uint8_t* pf_meta_host = nullptr;
  size_t pf_meta_cap = 0;
  Half* pf_last_hidden = nullptr;
  Half* pf_last_normed = nullptr;
  Half* pf_logits = nullptr;
  int pf_logits_rows = 0;
  // tensor parallel (reference TP: weights.rs:121-291,396-405): RCCL communicator, one rank per process/GPU
  pegainfer_comm_t tp_comm = nullptr;   // include/pegainfer_comm.h: RCCL + the one-shot peer-access path for <= 64 KB
  int tp_rank = 0, tp_world = 1;
  bool tp_comm_owned = false;
  uint32_t* tp_status_host = nullptr;   // pinned: the one-shot status block of the step, copied back with the tokens
  // A bounded wait of the one-shot all-reduce expired once: the status block is sticky by design (comm.cpp never clears
  // it), the epochs of the ranks may have diverged and the failing step had already advanced seq_len / appended KV.
  // Nothing here tries to resynchronise a world with a late or dead peer: the model is FAILED for good - every later
  // prefill / decode returns -5 at once, the caller drops the touched requests and rebuilds model + communicator
  // (include/pegainfer_qwen3.h).
  bool tp_failed = false;
  // debug tap (accuracy-parity-playbook.md:15-24: "find the first-diff token, then compare LAYERS"): while enabled every
  // step runs eagerly and the residual stream that leaves layer l - one row per request (prefill: the last prompt
  // position) - is copied to tap[l][row]; pegainfer_qwen3_debug_hidden hands a layer's rows to the host
  Half* tap = nullptr;
  bool tap_on = false;
  int tap_rows = 0;
  // last step
  const Half* last_logits = nullptr;
  int last_rows = 0;
  float last_step_ms = 0.f;
  int last_path = 0;

  Model(int dev, int h, int l, int hq, int hkv, int d, int inter, int vocab, float e, float th, int tie_, int mp,
        int pages, int mbs, int graph, int mode, int pol)
      : device(dev), H(h), L(l), Hq(hq), Hkv(hkv), D(d), I(inter), V(vocab), tie(tie_), max_pos(mp),
        num_pages(pages), max_bs(mbs), enable_graph(graph), decode_mode(mode), split_policy(pol), eps(e),
        theta(th), q_dim(hq * d), kv_dim(hkv * d), layout(l, hkv, d, 16), pool(pages) {
    for (auto& b : graphs) b[0] = b[1] = b[2] = nullptr;
  }

  void set_error(const std::string& s) { err = s; }
  // device counters at the head of a captured step are zeroed by a KERNEL node (pegainfer_zero_words; elementwise.hip says
  // why); PEGAINFER_CTR_RESET=memset restores the hipMemsetAsync node for the A/B
  bool ctr_reset_memset = [] { const char* e = getenv("PEGAINFER_CTR_RESET"); return e && e[0] == 'm'; }();
  int zero_ctrs(void* p, size_t words) {
    if (ctr_reset_memset) { PQ_HIP(hipMemsetAsync(p, 0, words * 4, stream)); return 0; }
    if (pegainfer_zero_words(p, (int32_t)words, S())) { set_error("pegainfer_zero_words failed"); return -1; }
    return 0;
  }
  void* S() const { return reinterpret_cast<void*>(stream); }

  template <typename T>
  int dalloc(T** p, size_t count, bool zero = true) {
    void* raw = nullptr;
    PQ_HIP(hipMalloc(&raw, count * sizeof(T)));
    if (zero) PQ_HIP(hipMemsetAsync(raw, 0, count * sizeof(T), stream));
    owned.push_back(raw);
    *p = static_cast<T*>(raw);
    return 0;
  }

  int init() {
    PQ_HIP(hipSetDevice(device));
    cublas_init();
    PQ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    PQ_HIP(hipEventCreate(&ev0));
    PQ_HIP(hipEventCreate(&ev1));
    layers.resize(L);
    auto W = [&](Half** p, size_t n) {
      weight_bytes += (int64_t)n * 2;
      return dalloc(p, n, false);
    };
    if (W(&embed, (size_t)V * H)) return -1;
    if (tie) lm_head = embed;
    else if (W(&lm_head, (size_t)V * H)) return -1;
    if (W(&final_norm, H)) return -1;
    for (auto& ly : layers) {
      if (W(&ly.qkv, (size_t)(q_dim + 2 * kv_dim) * H) || W(&ly.o, (size_t)H * q_dim) || W(&ly.q_norm, D) ||
          W(&ly.k_norm, D) || W(&ly.gate_up, (size_t)2 * I * H) || W(&ly.down, (size_t)H * I) || W(&ly.ln1, H) ||
          W(&ly.ln2, H))
        return -1;
    }
    if (dalloc(&cos, (size_t)max_pos * D, false) || dalloc(&sin, (size_t)max_pos * D, false)) return -1;
    // KV pool (kv_pool.rs:86-118): zeroed buffer, first page reserved as the graph padding page
    if (dalloc(&kv_buffer, (size_t)num_pages * layout.page_stride)) return -1;
    std::vector<int32_t> pad;
    if (!pool.acquire(1, &pad)) { set_error("pool must have at least 1 page for padding"); return -1; }
    padding_page = pad[0];
    // decode buffers (batch_decode_buffers.rs:103-153)
    const size_t bs = max_bs, slots = bs * kSplitMaxChunksPerRequest;
    if (dalloc(&normed, bs * H) || dalloc(&q, bs * q_dim) || dalloc(&k, bs * kv_dim) || dalloc(&v, bs * kv_dim) ||
        dalloc(&attn_out, bs * q_dim) || dalloc(&attn_proj, bs * H) || dalloc(&gate_up_out, bs * 2 * I) ||
        dalloc(&mlp_act, bs * I) || dalloc(&mlp_out, bs * H) || dalloc(&hidden, bs * H) ||
        dalloc(&hidden2, bs * H) || dalloc(&qkv_out, bs * (size_t)(q_dim + 2 * kv_dim)) ||
        dalloc(&logits, bs * (size_t)V) || dalloc(&split_tmp_v, slots * q_dim) || dalloc(&split_tmp_s, slots * Hq) ||
        dalloc(&top1_state, bs * 16) || dalloc(&merge_ctr, bs * (size_t)Hkv * 32) || dalloc(&probs_scratch, (size_t)V) ||
        dalloc(&top1_value, 1) || dalloc(&row_states, 1024 * 1024) || dalloc(&valid_scratch, 1) ||
        dalloc(&sample_out_d, 1))
      return -1;
    // packed metadata block
    auto al = [](size_t x) { return (x + 63) & ~size_t(63); };
    size_t off = 0;
    ml.token_ids = off; off = al(off + bs * 4);
    ml.positions = off; off = al(off + bs * 4);
    ml.page_indptr = off; off = al(off + (bs + 1) * 4);
    ml.last_page_len = off; off = al(off + bs * 4);
    ml.request_indices = off; off = al(off + bs * 4);
    ml.kv_tile_indices = off; off = al(off + bs * 4);
    ml.kv_chunk_size = off; off = al(off + bs * 4);
    ml.split_request_indices = off; off = al(off + slots * 4);
    ml.split_kv_tile_indices = off; off = al(off + slots * 4);
    ml.split_kv_chunk_size = off; off = al(off + 4);
    ml.split_o_indptr = off; off = al(off + (bs + 1) * 4);
    ml.split_valid = off; off = al(off + slots);
    ml.slot_desc = off; off = al(off + slots * 32);
    ml.page_indices = off; off = al(off + ((size_t)num_pages + bs) * 4);
    ml.total = off;
    PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&meta_host), ml.total, hipHostMallocDefault));
    std::memset(meta_host, 0, ml.total);
    if (dalloc(&meta_dev, ml.total)) return -1;
    // The greedy tokens of a step are written INTO the metadata block's token_ids slot (same 4 * bs bytes, read by the
    // embedding at the head of the step, written by the batched top-1 at its tail): a chained step needs no copy to feed
    // the next one - its metadata upload simply starts behind that slot.
    tokens_out_d = reinterpret_cast<int32_t*>(meta_dev + ml.token_ids);
    PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&tokens_out_host), bs * 4, hipHostMallocDefault));
    {  // the fused attention + o_proj form by configuration: the grid of a step is (256 / Hkv) slots x Hkv workgroups
      const int slots1 = std::max(1, 256 / std::max(1, Hkv)), slots2 = 2 * std::max(1, 256 / std::max(1, 2 * Hkv));
      oproj_shape_ok = pegainfer_fused_decode_attention_oproj_supported(Hq, Hkv, D, H, slots1, slots1 - pq::oproj_fused_max_chunks(), 1) == 1;
      oproj_shape_ok2 = pq::oproj_fused_max_batch() >= 2 &&
                        pegainfer_fused_decode_attention_oproj_supported(Hq, Hkv, D, H, slots2, slots2 - pq::oproj_fused_max_chunks(), 2) == 1;
    }
    if (attn_oproj) {
      kDoneInts = (size_t)2 * Hkv * 32;   // one arrival counter per (request <= 2, kv head group), a cache line each
      if (dalloc(&attn_done, (size_t)L * kDoneInts) || dalloc(&attn_status, 4)) return -1;
      PQ_HIP(hipMemsetAsync(attn_done, 0, (size_t)L * kDoneInts * sizeof(int32_t), stream));
      PQ_HIP(hipMemsetAsync(attn_status, 0, 16, stream));
      PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&attn_status_host), 16, hipHostMallocDefault));
      attn_status_host[0] = 0;
    }
    PQ_HIP(hipStreamSynchronize(stream));
    return 0;
  }

  // Qwen3Model::all_reduce_hidden (weights.rs:396-405): in-place bf16 sum over the TP group, on the model
  // stream (capturable), after O-proj and down-proj.  No-op without a communicator.
  int all_reduce_hidden(Half* buf, size_t count) {
    if (!tp_comm) return 0;
    // decode: 5 KB x bs per call, 72 calls per step - latency-bound, taken by the one-shot xGMI kernel when the peers
    // are mapped; prefill-sized payloads go to RCCL (size dispatch inside the verb)
    if (pegainfer_comm_all_reduce_bf16(tp_comm, buf, (int64_t)count, stream)) {
      set_error(std::string("all_reduce_hidden: ") + pegainfer_comm_last_error(tp_comm));
      return -1;
    }
    return 0;
  }
  // rows of `src` ([rows, H], row r at src + row_index[r] * H, or r * H when row_index is null) -> tap[layer]
  int tap_layer(int layer, const Half* src, int rows, const int32_t* row_index = nullptr) {
    if (!tap_on) return 0;
    rows = std::min(rows, max_bs);
    if (!row_index) {
      PQ_HIP(hipMemcpyAsync(tap + (size_t)layer * max_bs * H, src, (size_t)rows * H * 2, hipMemcpyDeviceToDevice, stream));
    } else {
      for (int r = 0; r < rows; ++r)
        PQ_HIP(hipMemcpyAsync(tap + ((size_t)layer * max_bs + r) * H, src + (size_t)row_index[r] * H, (size_t)H * 2,
                              hipMemcpyDeviceToDevice, stream));
    }
    tap_rows = rows;
    return 0;
  }
  int debug_hidden_enable(int on) {
    if (on && !tap && dalloc(&tap, (size_t)L * max_bs * H)) return -1;
    tap_on = on != 0;
    tap_rows = 0;
    return 0;
  }
  int debug_hidden(int layer, void* host, int max_rows) {
    if (!tap || layer < 0 || layer >= L) { set_error("debug_hidden: tap not enabled or bad layer"); return -1; }
    const int rows = std::min(tap_rows, max_rows);
    PQ_HIP(hipStreamSynchronize(stream));
    if (rows > 0) PQ_HIP(hipMemcpy(host, tap + (size_t)layer * max_bs * H, (size_t)rows * H * 2, hipMemcpyDeviceToHost));
    return rows;
  }
  int attach_tp(int rank, int world, const void* unique_id) {
    if (world < 1 || rank < 0 || rank >= world) { set_error("bad TP rank/world"); return -1; }
    if (tp_comm) { set_error("a TP communicator is already attached"); return -1; }
    tp_rank = rank; tp_world = world;
    if (!unique_id) return world == 1 ? 0 : (set_error("TP world > 1 needs a unique id"), -1);
    tp_comm = pegainfer_comm_create(device, rank, world, unique_id);
    if (!tp_comm) { set_error("pegainfer_comm_create (ncclCommInitRank) failed"); return -1; }
    tp_comm_owned = true;
    // best effort, and collective-safe: a rank whose export / mapping failed still takes part in the exchanges and the
    // path is switched on only when EVERY rank succeeded, so a failure leaves all ranks on RCCL.  PEGAINFER_TP_ONESHOT=0
    // (A/B runs) must be set on every rank or on none: it decides whether the rank enters the collective at all.
    const char* os = getenv("PEGAINFER_TP_ONESHOT");
    if (!(os && os[0] == '0')) (void)pegainfer_comm_oneshot_enable(tp_comm);
    return tp_after_attach();
  }
  // An already-built communicator (include/pegainfer_comm.h) drives the runtime's all-reduces: RCCL ones behave like
  // attach_tp; PEER-ONLY ones (no RCCL, slabs mapped over hipIpc) carry every all-reduce on the one-shot kernel, 64 KB
  // pieces for prefill-sized payloads - which is how the sharded runtime runs with several ranks on ONE GPU.
  // The caller keeps ownership of the communicator and must keep it alive until the model is destroyed.
  int attach_comm(pegainfer_comm_t comm) {
    if (!comm) { set_error("attach_comm: null communicator"); return -1; }
    if (tp_comm) { set_error("a TP communicator is already attached"); return -1; }
    const int world = pegainfer_comm_world(comm), rank = pegainfer_comm_rank(comm);
    if (world < 1 || rank < 0 || rank >= world) { set_error("attach_comm: bad communicator"); return -1; }
    tp_rank = rank; tp_world = world;
    if (world == 1) return 0;
    tp_comm = comm;
    tp_comm_owned = false;
    return tp_after_attach();
  }
  int tp_after_attach() {
    // every graph captured so far has no all-reduce nodes
    for (auto& b : graphs)
      for (auto& g : b)
        if (g) { hipGraphExecDestroy(g); g = nullptr; }
    if (!tp_status_host) PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&tp_status_host), 16, hipHostMallocDefault));
    tp_status_host[0] = tp_status_host[1] = tp_status_host[2] = 0;
    return 0;
  }

  // ------------------------------------------------------------------ weights
  // expected tensors: 3 globals (embed, lm_head unless tied, final norm) + 11 per layer; `loaded` is checked by finalize()
  enum { kSlotEmbed = 0, kSlotLmHead = 1, kSlotNorm = 2, kSlotLayer0 = 3, kSlotsPerLayer = 11 };
  std::vector<uint8_t> loaded;
  // HF tensor name -> device address, logical shape and bookkeeping slot (weights.rs:102-296).  0 ok, 1 = a tensor this
  // model ignores (lm_head of a tied checkpoint), -1 = error
  int resolve_tensor(const std::string& name, Half** dst_out, int64_t* rows_out, int64_t* cols_out, int* slot_out) {
    Half* dst = nullptr;
    int64_t rows = 0, cols = 0;
    int slot = -1;
    auto set = [&](Half* p, int64_t r, int64_t c, int sl) { dst = p; rows = r; cols = c; slot = sl; };
    if (name == "model.embed_tokens.weight") set(embed, V, H, kSlotEmbed);
    else if (name == "lm_head.weight") { if (tie) return 1; set(lm_head, V, H, kSlotLmHead); }
    else if (name == "model.norm.weight") set(final_norm, H, 1, kSlotNorm);
    else if (name.rfind("model.layers.", 0) == 0) {
      const size_t p0 = 13, p1 = name.find('.', p0);
      if (p1 == std::string::npos || p1 == p0 || p1 - p0 > 6) { set_error("bad layer index in tensor name: " + name); return -1; }
      int li = 0;
      for (size_t i = p0; i < p1; ++i) {
        if (name[i] < '0' || name[i] > '9') { set_error("bad layer index in tensor name: " + name); return -1; }
        li = li * 10 + (name[i] - '0');
      }
      if (li < 0 || li >= L) { set_error("layer index out of range: " + name); return -1; }
      const std::string rest = name.substr(p1 + 1);
      Layer& ly = layers[li];
      const int base = kSlotLayer0 + li * kSlotsPerLayer;
      if (rest == "self_attn.q_proj.weight") set(ly.qkv, q_dim, H, base + 0);
      else if (rest == "self_attn.k_proj.weight") set(ly.qkv + (size_t)q_dim * H, kv_dim, H, base + 1);
      else if (rest == "self_attn.v_proj.weight") set(ly.qkv + (size_t)(q_dim + kv_dim) * H, kv_dim, H, base + 2);
      else if (rest == "self_attn.o_proj.weight") set(ly.o, H, q_dim, base + 3);
      else if (rest == "self_attn.q_norm.weight") set(ly.q_norm, D, 1, base + 4);
      else if (rest == "self_attn.k_norm.weight") set(ly.k_norm, D, 1, base + 5);
      else if (rest == "mlp.gate_proj.weight") set(ly.gate_up, I, H, base + 6);
      else if (rest == "mlp.up_proj.weight") set(ly.gate_up + (size_t)I * H, I, H, base + 7);
      else if (rest == "mlp.down_proj.weight") set(ly.down, H, I, base + 8);
      else if (rest == "input_layernorm.weight") set(ly.ln1, H, 1, base + 9);
      else if (rest == "post_attention_layernorm.weight") set(ly.ln2, H, 1, base + 10);
    }
    if (!dst) { set_error("unknown tensor name: " + name); return -1; }
    *dst_out = dst; *rows_out = rows; *cols_out = cols; *slot_out = slot;
    return 0;
  }
  // the bf16 bits of a loaded (or synthetic) tensor back on the host: the checkpoint the engine computes with can be
  // handed to a checker as it is (bench.py's parity leg, tests)
  int export_tensor(const char* name_c, void* host, int64_t numel) {
    Half* src = nullptr;
    int64_t rows = 0, cols = 0;
    int slot = -1;
    const std::string name(name_c);
    if (name == "lm_head.weight" && tie) { src = embed; rows = V; cols = H; }
    else if (resolve_tensor(name, &src, &rows, &cols, &slot)) return -1;
    if (numel != rows * cols) { set_error("shape mismatch for " + name); return -1; }
    PQ_HIP(hipStreamSynchronize(stream));
    PQ_HIP(hipMemcpy(host, src, (size_t)numel * 2, hipMemcpyDeviceToHost));
    return 0;
  }
  int load_tensor(const char* name_c, const void* host, int64_t numel, const int64_t* shape = nullptr, int ndim = 0) {
    const std::string name(name_c);
    Half* dst = nullptr;
    int64_t rows = 0, cols = 0;
    int slot = -1;
    if (const int r = resolve_tensor(name, &dst, &rows, &cols, &slot)) return r == 1 ? 0 : -1;
    if (numel != rows * cols) { set_error("shape mismatch for " + name); return -1; }
    if (shape) {  // a transposed or re-shaped tensor with the right element count is still wrong
      const bool ok = cols == 1 ? (ndim == 1 && shape[0] == rows) : (ndim == 2 && shape[0] == rows && shape[1] == cols);
      if (!ok) { set_error("tensor " + name + " has the wrong shape (expected [" + std::to_string(rows) +
                           (cols == 1 ? "]" : ", " + std::to_string(cols) + "]") + ")"); return -1; }
    }
    PQ_HIP(hipMemcpy(dst, host, (size_t)numel * 2, hipMemcpyHostToDevice));
    if (loaded.empty()) loaded.assign(kSlotLayer0 + (size_t)L * kSlotsPerLayer, 0);
    loaded[slot] = 1;
    return 0;
  }
  // every weight the forward pass reads must have been uploaded (weights are hipMalloc'ed uninitialised)
  int check_all_loaded() {
    if (loaded.empty()) loaded.assign(kSlotLayer0 + (size_t)L * kSlotsPerLayer, 0);
    static const char* per_layer[] = {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                                      "self_attn.q_norm", "self_attn.k_norm", "mlp.gate_proj", "mlp.up_proj",
                                      "mlp.down_proj", "input_layernorm", "post_attention_layernorm"};
    for (size_t sl = 0; sl < loaded.size(); ++sl) {
      if (loaded[sl] || (sl == kSlotLmHead && tie)) continue;
      std::string what = sl == kSlotEmbed ? "model.embed_tokens.weight" : sl == kSlotLmHead ? "lm_head.weight"
                         : sl == kSlotNorm ? "model.norm.weight"
                         : "model.layers." + std::to_string((sl - kSlotLayer0) / kSlotsPerLayer) + "." +
                               per_layer[(sl - kSlotLayer0) % kSlotsPerLayer] + ".weight";
      set_error("checkpoint is missing tensor " + what);
      return -1;
    }
    return 0;
  }

  // Native checkpoint load with the reference's TP slicing (weights.rs:121-291 over weight_loader.rs:109-206):
  // q/k/v/gate/up row-sharded, o/down column-sharded, everything else replicated.  The model was created with
  // the LOCAL head counts / intermediate size; the file holds the full tensors.
  int load_safetensors(const char* path, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) { set_error("bad TP rank/world"); return -1; }
    pst::Checkpoint ck;
    std::string e;
    if (!ck.open(path, &e)) { set_error(e); return -1; }
    auto ends_with = [](const std::string& s, const char* suf) {
      const size_t n = std::strlen(suf);
      return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
    };
    std::vector<uint16_t> staging;
    for (const auto& kv : ck.tensors()) {
      const std::string& name = kv.first;
      const pst::TensorView& t = kv.second;
      if (name == "lm_head.weight" && tie) continue;
      if (t.dtype != "BF16") { set_error("tensor " + name + " is " + t.dtype + ", expected BF16"); return -1; }
      const bool row = ends_with(name, "self_attn.q_proj.weight") || ends_with(name, "self_attn.k_proj.weight") ||
                       ends_with(name, "self_attn.v_proj.weight") || ends_with(name, "mlp.gate_proj.weight") ||
                       ends_with(name, "mlp.up_proj.weight");
      const bool col = ends_with(name, "self_attn.o_proj.weight") || ends_with(name, "mlp.down_proj.weight");
      const uint16_t* src = reinterpret_cast<const uint16_t*>(t.data);
      if ((int64_t)t.nbytes != t.numel() * 2) { set_error("tensor " + name + ": data_offsets do not match its shape"); return -1; }
      if (world == 1 || (!row && !col)) {
        if (load_tensor(name.c_str(), src, t.numel(), t.shape.data(), (int)t.shape.size())) return -1;
        continue;
      }
      if (t.shape.size() != 2) { set_error("tensor " + name + " expected 2-D"); return -1; }
      const int64_t rows = t.shape[0], cols = t.shape[1];
      if (row) {
        if (rows % world) { set_error("rows of " + name + " not divisible by the TP world"); return -1; }
        const int64_t lr = rows / world;
        const int64_t shp[2] = {lr, cols};
        if (load_tensor(name.c_str(), src + (size_t)rank * lr * cols, lr * cols, shp, 2)) return -1;
      } else {
        if (cols % world) { set_error("columns of " + name + " not divisible by the TP world"); return -1; }
        const int64_t lc = cols / world;
        staging.resize((size_t)rows * lc);
        for (int64_t r = 0; r < rows; ++r)
          std::memcpy(staging.data() + (size_t)r * lc, src + (size_t)r * cols + (size_t)rank * lc, (size_t)lc * 2);
        const int64_t shp[2] = {rows, lc};
        if (load_tensor(name.c_str(), staging.data(), rows * lc, shp, 2)) return -1;
      }
    }
    return finalize();
  }

  int fill(Half* p, size_t n, uint64_t seed, float std, float mean) {
    fill_normal_kernel<<<2048, 256, 0, stream>>>(p, (long)n, seed, std, mean);
    return 0;
  }
  int fill_synthetic(uint64_t seed, float std) {
    loaded.assign(kSlotLayer0 + (size_t)L * kSlotsPerLayer, 1);
    uint64_t s = seed * 1000003ull;
    fill(embed, (size_t)V * H, ++s, std, 0.f);
    if (!tie) fill(lm_head, (size_t)V * H, ++s, std, 0.f);
    fill(final_norm, H, ++s, 0.1f, 1.f);
    for (auto& ly : layers) {
      fill(ly.qkv, (size_t)(q_dim + 2 * kv_dim) * H, ++s, std, 0.f);
      fill(ly.o, (size_t)H * q_dim, ++s, std, 0.f);
      fill(ly.q_norm, D, ++s, 0.1f, 1.f);
      fill(ly.k_norm, D, ++s, 0.1f, 1.f);
      fill(ly.gate_up, (size_t)2 * I * H, ++s, std, 0.f);
      fill(ly.down, (size_t)H * I, ++s, std, 0.f);
      fill(ly.ln1, H, ++s, 0.1f, 1.f);
      fill(ly.ln2, H, ++s, 0.1f, 1.f);
    }
    PQ_HIP(hipStreamSynchronize(stream));
    return 0;
  }

  // RoPE tables: fp32 pos * theta^(-2i/D) -> cos/sin -> bf16, duplicated halves (weight_loader.rs:210-244),
  // sized to max_position_embeddings (the reference's 4096-row table is a latent OOB).
  int finalize() {
    if (check_all_loaded()) return -1;
    const int half = D / 2;
    std::vector<float> inv(half);
    for (int i = 0; i < half; ++i) inv[i] = 1.0f / std::pow(theta, (float)i * 2.0f / (float)D);
    std::vector<uint16_t> c((size_t)max_pos * D), s((size_t)max_pos * D);
    for (int pos = 0; pos < max_pos; ++pos)
      for (int i = 0; i < half; ++i) {
        const float f = (float)pos * inv[i];
        const uint16_t cv = host_f2bf(std::cos(f)), sv = host_f2bf(std::sin(f));
        c[(size_t)pos * D + i] = c[(size_t)pos * D + i + half] = cv;
        s[(size_t)pos * D + i] = s[(size_t)pos * D + i + half] = sv;
      }
    PQ_HIP(hipMemcpy(cos, c.data(), c.size() * 2, hipMemcpyHostToDevice));
    PQ_HIP(hipMemcpy(sin, s.data(), s.size() * 2, hipMemcpyHostToDevice));
    finalized = true;
    return 0;
  }

  // ------------------------------------------------------------------ requests
  int new_request() {
    for (size_t i = 0; i < requests.size(); ++i)
      if (!requests[i].live) { requests[i] = KvState(); requests[i].live = true; return (int)i; }
    requests.emplace_back();
    requests.back().live = true;
    return (int)requests.size() - 1;
  }
  KvState* req(int id) {
    if (id < 0 || id >= (int)requests.size() || !requests[id].live) { set_error("bad request id"); return nullptr; }
    return &requests[id];
  }
  int drop_request(int id) {
    KvState* r = req(id);
    if (!r) return -1;
    r->reset(&pool);
    r->live = false;
    return 0;
  }

  // ------------------------------------------------------------------ decode (batch_decode.rs)
  template <typename T>
  T* mh(size_t off) { return reinterpret_cast<T*>(meta_host + off); }
  template <typename T>
  T* md(size_t off) { return reinterpret_cast<T*>(meta_dev + off); }

  // 17..64 requests (decode_mode 1): the same DAG with the launches that have a bit-identical fused form folded:
  // one GEMM over the stacked q|k|v rows, qk-norm + RoPE + KV append inside the attention launch, SwiGLU in the
  // gate_up GEMM epilogue, and (single GPU) the split-K slice sum + residual add + RMSNorm of o_proj / down_proj in
  // one launch.  7 launches per layer instead of 14.  `next_w` = the norm that follows this layer's down_proj.
  int decode_layer_mid_batch(int li, int bs, bool split, int split_slots, const Half* next_w) {
    const Layer& ly = layers[li];
    const float sm = 1.0f / std::sqrt((float)D);
    gemm_graphsafe_cuda(ly.qkv, normed, qkv_out, q_dim + 2 * kv_dim, bs, H, S());
    int rc = pegainfer_fused_decode_attention(
        qkv_out, attn_out, kv_buffer, layout.k_offset(li), layout.v_offset(li), md<int32_t>(ml.page_indices),
        md<int32_t>(ml.page_indptr), md<int32_t>(ml.last_page_len), md<int32_t>(ml.positions), ly.q_norm, ly.k_norm, cos,
        sin, eps, split ? 1 : 0, md<int32_t>(ml.split_request_indices), md<int32_t>(ml.split_kv_tile_indices),
        md<int32_t>(ml.split_kv_chunk_size), md<int32_t>(ml.split_o_indptr), md<uint8_t>(ml.split_valid), split_tmp_v,
        split_tmp_s, Hq, Hkv, D, layout.page_size, bs, split_slots, layout.page_stride, sm, md<int32_t>(ml.slot_desc),
        fused_merge ? merge_ctr : nullptr, S());
    if (rc) { set_error("pegainfer_fused_decode_attention failed"); return -1; }
    if (tp_comm) {
      gemm_graphsafe_cuda(ly.o, attn_out, attn_proj, H, bs, q_dim, S());
      if (all_reduce_hidden(attn_proj, (size_t)bs * H)) return -1;  // batch_decode.rs:266
      fused_add_rms_norm_batched_cuda(hidden, attn_proj, ly.ln2, normed, H, bs, eps, S());
    } else if (pegainfer_gemm_add_rms_norm(ly.o, attn_out, attn_proj, hidden, ly.ln2, normed, H, bs, q_dim, eps, S())) {
      set_error("pegainfer_gemm_add_rms_norm (o_proj) failed"); return -1;
    }
    if (pegainfer_gemm_silu(ly.gate_up, normed, mlp_act, gate_up_out, I, bs, H, S())) { set_error("pegainfer_gemm_silu failed"); return -1; }
    if (tp_comm) {
      gemm_graphsafe_cuda(ly.down, mlp_act, mlp_out, H, bs, I, S());
      if (all_reduce_hidden(mlp_out, (size_t)bs * H)) return -1;    // batch_decode.rs:292
      fused_add_rms_norm_batched_cuda(hidden, mlp_out, next_w, normed, H, bs, eps, S());
    } else if (pegainfer_gemm_add_rms_norm(ly.down, mlp_act, mlp_out, hidden, next_w, normed, H, bs, I, eps, S())) {
      set_error("pegainfer_gemm_add_rms_norm (down_proj) failed"); return -1;
    }
    return tap_layer(li, hidden, bs);
  }

  int decode_layer_reference(int li, int bs, bool split, int split_slots) {
    const Layer& ly = layers[li];
    // Q/K/V as three row-sliced GEMMs on purpose (batch_decode.rs:160-186)
    if (stacked_qkv(bs)) {
      if (pegainfer_gemm_split3(ly.qkv, normed, q, q_dim, k, kv_dim, v, kv_dim, bs, H, S())) {
        set_error("pegainfer_gemm_split3 failed"); return -1;
      }
    } else {  // three GEMMs over row slices of the fused matrix, as the reference does (batch_decode.rs:160-163)
      gemm_graphsafe_cuda(ly.qkv, normed, q, q_dim, bs, H, S());
      gemm_graphsafe_cuda(ly.qkv + (size_t)q_dim * H, normed, k, kv_dim, bs, H, S());
      gemm_graphsafe_cuda(ly.qkv + (size_t)(q_dim + kv_dim) * H, normed, v, kv_dim, bs, H, S());
    }
    qk_norm_rope_batched_decode_cuda(q, k, ly.q_norm, ly.k_norm, cos, sin, md<int32_t>(ml.positions), Hq, Hkv, D, bs,
                                     eps, S());
    int rc = paged_kv_scatter_cuda(kv_buffer, layout.k_offset(li), layout.v_offset(li), md<int32_t>(ml.page_indices),
                                   md<int32_t>(ml.page_indptr), md<int32_t>(ml.last_page_len), k, v,
                                   md<int32_t>(ml.request_indices), md<int32_t>(ml.positions), bs, Hkv, D,
                                   layout.page_size, layout.page_stride, kv_dim, D, S());
    if (rc) { set_error("paged_kv_scatter_cuda failed"); return -1; }
    const float sm = 1.0f / std::sqrt((float)D);
    if (!split) {
      rc = paged_attention_decode_cuda(q, attn_out, kv_buffer, layout.k_offset(li), layout.v_offset(li),
                                       md<int32_t>(ml.page_indices), md<int32_t>(ml.page_indptr),
                                       md<int32_t>(ml.last_page_len), md<int32_t>(ml.request_indices),
                                       md<int32_t>(ml.kv_tile_indices), md<int32_t>(ml.kv_chunk_size), Hq, Hkv, D,
                                       layout.page_size, bs, layout.page_stride, sm, S());
    } else {
      rc = paged_attention_decode_split_kv_cuda(
          q, attn_out, kv_buffer, layout.k_offset(li), layout.v_offset(li), md<int32_t>(ml.page_indices),
          md<int32_t>(ml.page_indptr), md<int32_t>(ml.last_page_len), md<int32_t>(ml.split_request_indices),
          md<int32_t>(ml.split_kv_tile_indices), md<int32_t>(ml.split_kv_chunk_size), md<int32_t>(ml.split_o_indptr),
          md<uint8_t>(ml.split_valid), split_tmp_v, split_tmp_s, Hq, Hkv, D, layout.page_size, bs, split_slots,
          layout.page_stride, sm, S());
    }
    if (rc) { set_error("paged attention decode failed"); return -1; }
    gemm_graphsafe_cuda(ly.o, attn_out, attn_proj, H, bs, q_dim, S());
    if (all_reduce_hidden(attn_proj, (size_t)bs * H)) return -1;  // batch_decode.rs:266
    fused_add_rms_norm_batched_cuda(hidden, attn_proj, ly.ln2, normed, H, bs, eps, S());
    if (stacked_qkv(bs)) {  // mid-batch path: SwiGLU in the tiled GEMM's epilogue (same bits as the pair below)
      if (pegainfer_gemm_silu(ly.gate_up, normed, mlp_act, gate_up_out, I, bs, H, S())) { set_error("pegainfer_gemm_silu failed"); return -1; }
    } else {
      gemm_graphsafe_cuda(ly.gate_up, normed, gate_up_out, 2 * I, bs, H, S());
      silu_mul_fused_cuda(gate_up_out, mlp_act, I, bs, S());
    }
    gemm_graphsafe_cuda(ly.down, mlp_act, mlp_out, H, bs, I, S());
    if (all_reduce_hidden(mlp_out, (size_t)bs * H)) return -1;    // batch_decode.rs:292
    return 0;
  }

  // decode_mode 1: the MI355X launch-lean form of the same DAG.  Per layer 5 launches (+ merge when the KV
  // is partitioned) instead of 14: both fused_add_rms_norm ops are folded into the prologue of the GEMV that
  // consumes them, SwiGLU into the gate_up GEMV epilogue, qk-norm + RoPE + KV append into the attention
  // kernel.  Every fused kernel shares its arithmetic core with the reference-named op it replaces, so the
  // logits are bit-identical to decode_mode 0 (tests/test_gpu_fused.py).
  int decode_kernels_fused(int bs, bool split, int split_slots) {
    if (fused_merge && split && zero_ctrs(merge_ctr, (size_t)max_bs * Hkv * 32)) return -1;
    oproj_step = attn_oproj && attn_done && fused_merge && split && bs <= pq::oproj_fused_max_batch() && !tp_comm && D == 128 && oproj_plan;
    if (oproj_step && zero_ctrs(attn_done, (size_t)L * kDoneInts)) return -1;
    if (embedding_batched_cuda(embed, md<uint32_t>(ml.token_ids), hidden, H, bs, S())) {
      set_error("embedding_batched_cuda failed");
      return -1;
    }
    Half *cur = hidden, *nxt = hidden2;
    const Half* resid = nullptr;
    const float sm = 1.0f / std::sqrt((float)D);
    for (int li = 0; li < L; ++li) {
      const Layer& ly = layers[li];
      int rc = pegainfer_gemv_fused(ly.qkv, cur, qkv_out, q_dim + 2 * kv_dim, bs, H, resid, ly.ln1,
                                    resid ? nxt : nullptr, eps, 0, S());
      if (resid) std::swap(cur, nxt);
      if (resid && tap_layer(li - 1, cur, bs)) return -1;   // cur = hidden + the previous layer's down_proj = its output
      bool fused_o = false;
      if (!rc && oproj_step) {
        const int r2 = pegainfer_fused_decode_attention_oproj(
            qkv_out, attn_out, kv_buffer, layout.k_offset(li), layout.v_offset(li), md<int32_t>(ml.page_indices),
            md<int32_t>(ml.page_indptr), md<int32_t>(ml.last_page_len), md<int32_t>(ml.positions), ly.q_norm,
            ly.k_norm, cos, sin, eps, md<int32_t>(ml.split_request_indices), md<int32_t>(ml.split_kv_tile_indices),
            md<int32_t>(ml.split_kv_chunk_size), md<int32_t>(ml.split_o_indptr), md<uint8_t>(ml.split_valid),
            split_tmp_v, split_tmp_s, Hq, Hkv, D, layout.page_size, bs, split_slots,
            split_slots - pq::oproj_fused_max_chunks(), layout.page_stride, sm, md<int32_t>(ml.slot_desc), merge_ctr, ly.o,
            attn_proj, H, attn_done + (size_t)li * kDoneInts, attn_status, S());
        if (r2 == 0) fused_o = true;
        else if (r2 != (int)hipErrorInvalidValue) rc = r2;
        else { oproj_step = false; oproj_shape_ok = oproj_shape_ok2 = false; }   // cannot happen after init's shape test; kept as the safe path
      }
      if (!rc && !fused_o)
        rc = pegainfer_fused_decode_attention(
            qkv_out, attn_out, kv_buffer, layout.k_offset(li), layout.v_offset(li), md<int32_t>(ml.page_indices),
            md<int32_t>(ml.page_indptr), md<int32_t>(ml.last_page_len), md<int32_t>(ml.positions), ly.q_norm,
            ly.k_norm, cos, sin, eps, split ? 1 : 0, md<int32_t>(ml.split_request_indices),
            md<int32_t>(ml.split_kv_tile_indices), md<int32_t>(ml.split_kv_chunk_size),
            md<int32_t>(ml.split_o_indptr), md<uint8_t>(ml.split_valid), split_tmp_v, split_tmp_s, Hq, Hkv, D,
            layout.page_size, bs, split_slots, layout.page_stride, sm, md<int32_t>(ml.slot_desc),
            fused_merge ? merge_ctr : nullptr, S());
      if (!rc && !fused_o) rc = pegainfer_gemv_fused(ly.o, attn_out, attn_proj, H, bs, q_dim, nullptr, nullptr, nullptr, 0.f, 0, S());
      if (!rc) rc = all_reduce_hidden(attn_proj, (size_t)bs * H);
      if (!rc) {
        rc = pegainfer_gemv_fused(ly.gate_up, cur, mlp_act, 2 * I, bs, H, attn_proj, ly.ln2, nxt, eps, I, S());
        std::swap(cur, nxt);
      }
      if (!rc) rc = pegainfer_gemv_fused(ly.down, mlp_act, mlp_out, H, bs, I, nullptr, nullptr, nullptr, 0.f, 0, S());
      if (!rc) rc = all_reduce_hidden(mlp_out, (size_t)bs * H);
      if (rc) { set_error("fused decode layer failed"); return -1; }
      resid = mlp_out;
    }
    if (pegainfer_gemv_fused(lm_head, cur, logits, V, bs, H, resid, final_norm, nxt, eps, 0, S())) {
      set_error("fused lm_head failed");
      return -1;
    }
    if (tap_layer(L - 1, nxt, bs)) return -1;   // the lm_head prologue wrote the last layer's residual sum to nxt
    if (pegainfer_batched_top1(logits, V, bs, V, top1_state, tokens_out_d, S())) {
      set_error("pegainfer_batched_top1 failed");
      return -1;
    }
    return 0;
  }
  // fused kernels up to 16 columns; 17..64 run the unfused sequence with the stacked q|k|v GEMM (mid-batch path)
  bool fused_ok(int bs) const { return decode_mode >= 1 && bs <= 16 && (H & 31) == 0 && (I & 31) == 0 && D == 128; }
  bool stacked_qkv(int bs) const { return decode_mode >= 1 && bs > 16; }

  int decode_kernels(int bs, bool split, int split_slots) {
    if (fused_ok(bs)) return decode_kernels_fused(bs, split, split_slots);
    if (embedding_batched_cuda(embed, md<uint32_t>(ml.token_ids), hidden, H, bs, S())) {
      set_error("embedding_batched_cuda failed");
      return -1;
    }
    rms_norm_batched_cuda(hidden, layers[0].ln1, normed, H, bs, eps, S());
    const bool mid = stacked_qkv(bs) && D == 128 && mid_batch_fused;
    if (mid && fused_merge && split && zero_ctrs(merge_ctr, (size_t)max_bs * Hkv * 32)) return -1;
    for (int li = 0; li < L; ++li) {
      const Half* next_w = li + 1 < L ? layers[li + 1].ln1 : final_norm;
      if (mid) {
        if (decode_layer_mid_batch(li, bs, split, split_slots, next_w)) return -1;
        continue;
      }
      if (decode_layer_reference(li, bs, split, split_slots)) return -1;
      fused_add_rms_norm_batched_cuda(hidden, mlp_out, next_w, normed, H, bs, eps, S());
      if (tap_layer(li, hidden, bs)) return -1;
    }
    gemm_graphsafe_cuda(lm_head, normed, logits, V, bs, H, S());
    // greedy token for every column inside the captured step
    if (pegainfer_batched_top1(logits, V, bs, V, top1_state, tokens_out_d, S())) {
      set_error("pegainfer_batched_top1 failed");
      return -1;
    }
    return 0;
  }

  // One decode step = step_prepare (validate the batch, advance the KvStates, pack + upload the metadata block) +
  // step_launch (replay - or capture - the (bucket, path) graph).  decode() runs ONE step and synchronises;
  // decode_greedy_chain() enqueues n steps back to back (below).
  struct StepCtx { int n = 0, padded = 0, plan_slots = 0; bool split = false; std::vector<KvState*> st; };
  // token_ids == nullptr: a chained step - the previous step's greedy tokens already sit in the device block's token_ids
  // slot, the upload starts behind it
  int step_prepare(int n, const int32_t* ids, const uint32_t* token_ids, StepCtx* c, uint8_t* host_block) {
    if (!finalized) { set_error("model not finalized"); return -1; }
    if (tp_failed) { set_error("tensor-parallel group failed in an earlier step (bounded wait expired): rebuild the model and its communicator"); return -5; }
    if (n <= 0 || n > max_bs) { set_error("bad batch size"); return -1; }
    std::vector<KvState*>& st = c->st;
    st.assign(n, nullptr);
    std::vector<int> positions(n), seq_lens(n);
    // validate the whole batch first (ids, RoPE range, bucket, total pages): a failed call must leave every
    // request exactly as it was - no seq_len advanced, no page acquired
    const int padded = enable_graph ? bucket_for(n) : n;
    if (padded < 0 || padded > max_bs) { set_error("batch exceeds largest bucket"); return -1; }
    int pages_short = 0;
    for (int i = 0; i < n; ++i) {
      st[i] = req(ids[i]);
      if (!st[i]) return -1;
      for (int j = 0; j < i; ++j)
        if (ids[j] == ids[i]) { set_error("duplicate request id in decode batch"); return -1; }
      if (st[i]->seq_len + 1 > max_pos) { set_error("position exceeds RoPE table"); return -1; }
      pages_short += st[i]->pages_short(st[i]->seq_len + 1, layout.page_size);
    }
    if (pages_short > pool.available()) { set_error("KvState: out of pages"); return -2; }
    for (int i = 0; i < n; ++i) {
      positions[i] = st[i]->seq_len;
      if (!st[i]->ensure_capacity(&pool, positions[i] + 1, layout.page_size)) {
        set_error("KvState: out of pages");
        return -2;
      }
      st[i]->seq_len += 1;
      seq_lens[i] = st[i]->seq_len;
    }
    // ---- pack metadata (batch_decode.rs:45-59, batch_decode_buffers.rs:178-279) into `host_block` (pinned) ----
    auto hb = [&](size_t off) { return host_block + off; };
    auto* tok = reinterpret_cast<uint32_t*>(hb(ml.token_ids));
    auto* pos = reinterpret_cast<int32_t*>(hb(ml.positions));
    auto* indptr = reinterpret_cast<int32_t*>(hb(ml.page_indptr));
    auto* lpl = reinterpret_cast<int32_t*>(hb(ml.last_page_len));
    auto* ri = reinterpret_cast<int32_t*>(hb(ml.request_indices));
    auto* kti = reinterpret_cast<int32_t*>(hb(ml.kv_tile_indices));
    auto* kcs = reinterpret_cast<int32_t*>(hb(ml.kv_chunk_size));
    auto* pgs = reinterpret_cast<int32_t*>(hb(ml.page_indices));
    int np = 0;
    indptr[0] = 0;
    for (int i = 0; i < padded; ++i) {
      if (i < n) {
        tok[i] = token_ids ? token_ids[i] : 0;   // chained steps: the device copies the previous step's tokens in
        pos[i] = positions[i];
        for (int32_t p : st[i]->pages) pgs[np++] = p;
        lpl[i] = st[i]->last_page_len(layout.page_size);
        kcs[i] = st[i]->seq_len;
      } else {  // padding slot -> padding page, seq_len 1
        tok[i] = 0;
        pos[i] = 0;
        pgs[np++] = padding_page;
        lpl[i] = 1;
        kcs[i] = 1;
      }
      indptr[i + 1] = np;
      ri[i] = i;
      kti[i] = 0;
    }
    const bool oproj_usable = attn_oproj && attn_done && fused_merge && !tp_comm && decode_mode >= 1 && D == 128 &&
                              (padded == 1 ? oproj_shape_ok : padded == 2 ? oproj_shape_ok2 : false);
    const SplitPlan plan = make_split_plan(split_policy, seq_lens, padded, Hkv, oproj_usable);
    int max_seq = 0;
    for (int v : seq_lens) max_seq = std::max(max_seq, v);
    // every column of the bucket must be a real request: a padding column's merge would never arrive on its counters
    oproj_plan = oproj_usable && split_policy == 1 && padded <= pq::oproj_fused_max_batch() && n == padded && plan.use_split &&
                 max_seq <= pq::oproj_fused_max_seq() && plan.slots > pq::oproj_fused_max_chunks();
    for (int i = 0; i < n && oproj_plan; ++i)   // ... and must own at least one chunk
      if (plan.o_indptr[i + 1] <= plan.o_indptr[i]) oproj_plan = false;
    std::memcpy(hb(ml.split_request_indices), plan.request_indices.data(), plan.slots * 4);
    std::memcpy(hb(ml.split_kv_tile_indices), plan.kv_tile_indices.data(), plan.slots * 4);
    reinterpret_cast<int32_t*>(hb(ml.split_kv_chunk_size))[0] = plan.chunk;
    std::memcpy(hb(ml.split_o_indptr), plan.o_indptr.data(), (padded + 1) * 4);
    std::memcpy(hb(ml.split_valid), plan.valid.data(), plan.slots);
    // host-resolved slot records for the fused attention kernel (one load instead of four dependent ones)
    {
      auto* sd = reinterpret_cast<int32_t*>(hb(ml.slot_desc));
      const int nslots = plan.use_split ? plan.slots : padded;
      for (int sl = 0; sl < nslots; ++sl) {
        int32_t* r = sd + 8 * sl;
        int b, lo, hi;
        if (plan.use_split) {
          if (!plan.valid[sl]) { r[0] = 0; r[1] = -1; r[2] = -1; r[3] = 0; r[4] = 0; r[5] = 0; r[6] = r[7] = 0; continue; }
          b = plan.request_indices[sl];
          const int len = kcs[b];
          lo = plan.kv_tile_indices[sl] * plan.chunk;
          hi = std::min(lo + plan.chunk, len);
          if (lo > hi) lo = hi;
        } else {
          b = sl; lo = 0; hi = kcs[b];
        }
        r[0] = b; r[1] = lo; r[2] = hi; r[3] = indptr[b]; r[4] = pos[b]; r[5] = kcs[b];
        r[6] = plan.o_indptr[b]; r[7] = plan.o_indptr[b + 1];   // partial slots of this request (in-launch merge)
      }
    }
    const size_t upload = ml.page_indices + (size_t)np * 4, from = token_ids ? 0 : ml.positions;
    PQ_HIP(hipMemcpyAsync(meta_dev + from, host_block + from, upload - from, hipMemcpyHostToDevice, stream));
    c->n = n; c->padded = padded; c->split = plan.use_split; c->plan_slots = plan.slots;
    last_path = c->split ? 1 : 0;
    return 0;
  }
  bool step_uses_oproj_form(const StepCtx& c) const {
    return attn_status && attn_oproj && oproj_plan && c.padded <= pq::oproj_fused_max_batch() && c.split && decode_mode >= 1 && !tp_comm;
  }
  int step_launch(const StepCtx& c) {
    if (enable_graph && !tap_on) {
      const int bi = bucket_index(c.padded);
      const int gv = !c.split ? 0 : (oproj_plan && attn_oproj && c.padded <= pq::oproj_fused_max_batch() && !tp_comm && decode_mode >= 1 ? 2 : 1);
      hipGraphExec_t& exec = graphs[bi][gv];
      if (!exec) {  // capture once (cuda_graph.rs:36-55), thread-local mode
        hipGraph_t graph = nullptr;
        PQ_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        const int rc = decode_kernels(c.padded, c.split, c.plan_slots);
        hipError_t e = hipStreamEndCapture(stream, &graph);
        if (rc || e != hipSuccess) { set_error("graph capture failed: " + err); return -1; }
        PQ_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        PQ_HIP(hipGraphDestroy(graph));
      }
      PQ_HIP(hipGraphLaunch(exec, stream));
      return 0;
    }
    return decode_kernels(c.padded, c.split, c.plan_slots) ? -1 : 0;
  }
  // the fused attention + o_proj launch reported an expired bounded wait: disable the form for good and drop the graphs
  // that captured it (the caller re-runs what it enqueued on the two stand-alone launches)
  int oproj_fall_back() {
    attn_oproj = false;
    oproj_fallbacks += 1;
    PQ_HIP(hipMemsetAsync(attn_status, 0, 4, stream));
    for (int b = 0; b < kNumBuckets; ++b)
      for (int q = 0; q < 3; ++q)
        if (graphs[b][q]) { hipGraphExecDestroy(graphs[b][q]); graphs[b][q] = nullptr; }
    err = "fused attention + o_proj launch: the bounded wait for the attention rows expired (code " +
          std::to_string(attn_status_host[0]) + "); step re-run on two launches, form disabled";
    attn_status_host[0] = 0;
    return 0;
  }

  int decode(int n, const int32_t* ids, const uint32_t* token_ids, int32_t* out_tokens, void* out_logits_host) {
    StepCtx c;
    if (const int rc = step_prepare(n, ids, token_ids, &c, meta_host)) return rc;
    // one attempt = graph replay (or eager launches) + token D2H + sync.  A step whose fused attention + o_proj launch
    // reported an expired bounded wait is re-run in the SAME call on the two stand-alone launches: the metadata block is
    // already on the device, the step recomputes everything from the embedding and the KV append rewrites the same slots
    // with the same bits, so the request state the caller sees (seq_len advanced, pages held) matches the tokens it gets
    // back.  The loop runs until an attempt took no fallback (every fallback disables its cause for good), capped at 3.
    for (int attempt = 0;; ++attempt) {
      if (attempt > 0)   // the failed attempt's top-1 wrote ITS tokens into the block's token_ids slot: put the inputs back
        PQ_HIP(hipMemcpyAsync(meta_dev + ml.token_ids, meta_host + ml.token_ids, (size_t)c.padded * 4, hipMemcpyHostToDevice, stream));
      PQ_HIP(hipEventRecord(ev0, stream));
      const bool oproj_check = step_uses_oproj_form(c);
      if (step_launch(c)) return -1;
      PQ_HIP(hipEventRecord(ev1, stream));
      PQ_HIP(hipMemcpyAsync(tokens_out_host, tokens_out_d, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
      if (oproj_check) PQ_HIP(hipMemcpyAsync(attn_status_host, attn_status, 4, hipMemcpyDeviceToHost, stream));
      // tensor parallel: the 72 one-shot all-reduces of the step share one status block; it travels back with the tokens
      const uint32_t* os_status_d = tp_comm && tp_status_host && pegainfer_comm_oneshot_active(tp_comm)
                                        ? pegainfer_comm_oneshot_status_ptr(tp_comm) : nullptr;
      if (os_status_d) PQ_HIP(hipMemcpyAsync(tp_status_host, os_status_d, 12, hipMemcpyDeviceToHost, stream));
      PQ_HIP(hipStreamSynchronize(stream));
      hipEventElapsedTime(&last_step_ms, ev0, ev1);
      if (os_status_d && tp_status_host[0] != 0) {
        // a peer's flag never arrived within the bound: the sums of this step are undefined on this rank and the peer is
        // late or gone - RCCL would block on it too.  Fail the step loudly; the caller drops the request(s).
        set_error("tensor-parallel one-shot all-reduce: bounded wait expired (missing-rank mask " +
                  std::to_string(tp_status_host[0] & 0xff) + ", epoch " + std::to_string(tp_status_host[1]) +
                  ", segment " + std::to_string(tp_status_host[2]) + "); step failed, model marked failed");
        tp_failed = true;
        return -5;
      }
      if (!(oproj_check && attn_status_host[0] != 0)) break;
      // the o_proj phase of the fused launch never saw the attention rows (workgroups not co-resident): this attempt's
      // outputs are invalid.  Disable the form, drop the graphs that captured it and take another attempt on the two
      // stand-alone launches (same metadata, same KV slots: the request state stays consistent).
      if (oproj_fall_back()) return -1;
      if (attempt >= 2) { set_error("decode step still reported a fallback after every fused form was disabled"); return -4; }
    }
    for (int i = 0; i < n; ++i) out_tokens[i] = tokens_out_host[i];
    last_logits = logits;
    last_rows = n;
    if (out_logits_host) PQ_HIP(hipMemcpy(out_logits_host, logits, (size_t)n * V * 2, hipMemcpyDeviceToHost));
    return 0;
  }

  // n_steps GREEDY decode steps of the same batch enqueued back to back, ONE host synchronisation at the end (round 5).
  // A synchronous step leaves the GPU idle from "tokens on the host" to "next graph launched" (metadata packing, two
  // runtime calls, the caller's own loop: 30-65 us of a 2 ms step).  Greedy needs no host in between: the token of step s
  // is already on the device - the batched top-1 inside the graph writes it into the metadata block's token_ids slot, which
  // step s + 1's upload leaves alone - so the next graph replays as soon as its (token-independent) metadata landed.  Metadata blocks travel
  // through a ring of pinned buffers (the upload of step s must have executed before its buffer is repacked: one event per
  // ring slot).  Same graphs, same kernels, same bits as n_steps calls of decode() (tested); sampling with temperature,
  // stop tokens and logprobs need the host between steps and keep using decode().
  // out_tokens: [n_steps][n].  An expired bounded wait of the fused attention + o_proj form re-runs the whole chain on two
  // launches (the KV appends rewrite the same slots); tensor parallel takes the step-by-step path.
  static constexpr int kChainRing = 8;
  uint8_t* chain_ring[kChainRing] = {};
  hipEvent_t chain_ev[kChainRing] = {};
  int32_t* chain_tokens_host = nullptr;   // pinned, grow-only
  size_t chain_tokens_cap = 0;
  int decode_greedy_chain(int n, const int32_t* ids, const uint32_t* first_tokens, int n_steps, int32_t* out_tokens) {
    if (n_steps <= 0) { set_error("decode_greedy_chain: n_steps must be positive"); return -1; }
    // everything that can be refused is refused BEFORE the first step and before first_tokens is touched
    if (n < 1 || n > max_bs || bucket_for(n) < 0) { set_error("decode batch size out of range"); return -1; }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j)
        if (ids[i] == ids[j]) { set_error("decode_greedy_chain: duplicate request id"); return -1; }
    if (tp_comm || tap_on) {   // step by step: the one-shot status / the taps are per step
      std::vector<uint32_t> tk(first_tokens, first_tokens + n);
      for (int s = 0; s < n_steps; ++s) {
        if (const int rc = decode(n, ids, tk.data(), out_tokens + (size_t)s * n, nullptr)) return rc;
        for (int i = 0; i < n; ++i) tk[i] = (uint32_t)out_tokens[(size_t)s * n + i];
      }
      return 0;
    }
    if (!chain_ring[0]) {
      for (int r = 0; r < kChainRing; ++r) {
        PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&chain_ring[r]), ml.total, hipHostMallocDefault));
        std::memset(chain_ring[r], 0, ml.total);
        PQ_HIP(hipEventCreateWithFlags(&chain_ev[r], hipEventDisableTiming));
      }
    }
    if ((size_t)n_steps * n > chain_tokens_cap) {
      if (chain_tokens_host) PQ_HIP(hipHostFree(chain_tokens_host));
      chain_tokens_cap = std::max<size_t>((size_t)n_steps * n, 4096);   // grow-only, never inside a short timed chain
      PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&chain_tokens_host), chain_tokens_cap * 4, hipHostMallocDefault));
    }
    // a chain must be admissible as a whole before any request is advanced: positions and pages of ALL its steps
    {
      int pages_short = 0;
      for (int i = 0; i < n; ++i) {
        KvState* r = req(ids[i]);
        if (!r) return -1;
        if (r->seq_len + n_steps > max_pos) { set_error("position exceeds RoPE table"); return -1; }
        pages_short += r->pages_short(r->seq_len + n_steps, layout.page_size);
      }
      if (pages_short > pool.available()) { set_error("KvState: out of pages"); return -2; }
    }
    std::vector<int> start_len(n);
    for (int i = 0; i < n; ++i) start_len[i] = req(ids[i])->seq_len;
    for (int attempt = 0;; ++attempt) {
      bool any_oproj = false;
      PQ_HIP(hipEventRecord(ev0, stream));
      for (int s = 0; s < n_steps; ++s) {
        const int slot = s % kChainRing;
        if (s >= kChainRing) PQ_HIP(hipEventSynchronize(chain_ev[slot]));   // its upload has executed: safe to repack
        StepCtx c;
        // a failure at step s > 0 (graph capture, page race) leaves steps 0 .. s - 1 in flight: drain them, put the requests
        // back where the chain found them (their KV appends are simply overwritten later) and only then report - the next
        // call must not repack pinned ring slots whose uploads have not executed
        auto abort_chain = [&](int rc) {
          (void)hipStreamSynchronize(stream);
          for (int i = 0; i < n; ++i) req(ids[i])->seq_len = start_len[i];
          return rc;
        };
        if (const int rc = step_prepare(n, ids, s == 0 ? first_tokens : nullptr, &c, chain_ring[slot])) return abort_chain(rc);
        PQ_HIP(hipEventRecord(chain_ev[slot], stream));
        any_oproj = any_oproj || step_uses_oproj_form(c);
        if (step_launch(c)) return abort_chain(-1);
        PQ_HIP(hipMemcpyAsync(chain_tokens_host + (size_t)s * n, tokens_out_d, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
      }
      PQ_HIP(hipEventRecord(ev1, stream));
      if (any_oproj) PQ_HIP(hipMemcpyAsync(attn_status_host, attn_status, 4, hipMemcpyDeviceToHost, stream));
      PQ_HIP(hipStreamSynchronize(stream));
      hipEventElapsedTime(&last_step_ms, ev0, ev1);
      last_step_ms /= (float)n_steps;
      if (!(any_oproj && attn_status_host[0] != 0)) break;
      if (oproj_fall_back()) return -1;
      if (attempt >= 2) { set_error("decode chain still reported a fallback after every fused form was disabled"); return -4; }
      for (int i = 0; i < n; ++i) req(ids[i])->seq_len = start_len[i];   // pages stay; every step recomputes from the embedding
    }
    std::memcpy(out_tokens, chain_tokens_host, (size_t)n_steps * n * 4);
    last_logits = logits;
    last_rows = n;
    return 0;
  }

  // ------------------------------------------------------------------ prefill (prefill.rs)
  int ensure_prefill_ws(size_t T, int nreq) {
    if (T > pf_cap_tokens) {
      const size_t cap = std::max<size_t>(T, pf_cap_tokens * 2);
      Half** bufs[] = {&pf_hidden, &pf_hidden_out, &pf_normed, &pf_q, &pf_k, &pf_v, &pf_o, &pf_gate_up, &pf_act, &pf_attn};
      const size_t dims[] = {(size_t)H, (size_t)H, (size_t)H, (size_t)q_dim, (size_t)kv_dim, (size_t)kv_dim,
                             (size_t)H, (size_t)2 * I, (size_t)I, (size_t)q_dim};
      PQ_HIP(hipStreamSynchronize(stream));
      for (int i = 0; i < 10; ++i) {
        if (*bufs[i]) PQ_HIP(hipFree(*bufs[i]));
        PQ_HIP(hipMalloc(reinterpret_cast<void**>(bufs[i]), cap * dims[i] * 2));
      }
      pf_cap_tokens = cap;
    }
    if (nreq > pf_logits_rows) {
      PQ_HIP(hipStreamSynchronize(stream));
      if (pf_logits) { PQ_HIP(hipFree(pf_logits)); PQ_HIP(hipFree(pf_last_hidden)); PQ_HIP(hipFree(pf_last_normed)); }
      const int rows = std::max(nreq, 4);
      PQ_HIP(hipMalloc(reinterpret_cast<void**>(&pf_logits), (size_t)rows * V * 2));
      PQ_HIP(hipMalloc(reinterpret_cast<void**>(&pf_last_hidden), (size_t)rows * H * 2));
      PQ_HIP(hipMalloc(reinterpret_cast<void**>(&pf_last_normed), (size_t)rows * H * 2));
      pf_logits_rows = rows;
    }
    return 0;
  }

  // n_decode_tail > 0 = unified_step (unified_forward.rs:78-198): the last n_decode_tail requests contribute ONE
  // token each and attend through the decode kernel; everything else (GEMMs, norms, RoPE, KV append, MLP) is
  // shared by all token columns.
  int prefill(int n, const int32_t* ids, const int32_t* lens, const uint32_t* tokens, int32_t* out_tokens,
              void* out_logits_host, int n_decode_tail = 0, void* out_all_logits_host = nullptr) {
    if (!finalized) { set_error("model not finalized"); return -1; }
    if (tp_failed) { set_error("tensor-parallel group failed in an earlier step (bounded wait expired): rebuild the model and its communicator"); return -5; }
    if (n <= 0) { set_error("empty prefill"); return -1; }
    if (n > max_bs) { set_error("prefill batch larger than max_batch_size"); return -1; }
    const int n_pf = n - n_decode_tail;
    if (n_decode_tail < 0 || n_pf < 1) { set_error("unified step needs >= 1 prefill request"); return -1; }
    std::vector<KvState*> st(n);
    std::vector<int> starts(n);
    size_t T = 0;
    int pages_short = 0;
    for (int i = 0; i < n; ++i) {   // every check before any KvState is advanced
      st[i] = req(ids[i]);
      if (!st[i]) return -1;
      for (int j = 0; j < i; ++j)
        if (ids[j] == ids[i]) { set_error("duplicate request id in prefill batch"); return -1; }
      if (lens[i] <= 0) { set_error("empty prompt"); return -1; }
      if (i >= n_pf && lens[i] != 1) { set_error("decode-tail requests must contribute exactly one token"); return -1; }
      starts[i] = st[i]->seq_len;
      if (starts[i] + lens[i] > max_pos) { set_error("position exceeds RoPE table"); return -1; }
      pages_short += st[i]->pages_short(starts[i] + lens[i], layout.page_size);
      T += lens[i];
    }
    if (pages_short > pool.available()) { set_error("KvState: out of pages"); return -2; }
    // echo stages [tokens, vocab] logits through the gate|up scratch: it must hold at least one vocab row
    const size_t echo_min = out_all_logits_host ? ((size_t)V + (size_t)(2 * I) - 1) / (size_t)(2 * I) : 0;
    if (ensure_prefill_ws(std::max(T, echo_min), n)) return -1;
    for (int i = 0; i < n; ++i) {  // ensure_capacity + advance (prefill.rs:236-240)
      if (!st[i]->ensure_capacity(&pool, starts[i] + lens[i], layout.page_size)) { set_error("KvState: out of pages"); return -2; }
      st[i]->seq_len += lens[i];
    }
    // ---- plan (PrefillPagedPlan::new_batch_with_cta_tile_q, ops/attention.rs:208-302; tile 64 = config.rs:5) ----
    const int group = Hq / Hkv;
    const int P = (int)T - n_decode_tail;  // prefill token columns
    const int cta = batch_prefill_cta_tile_q_with_override(P, Hq, Hkv, D, 64);
    std::vector<int32_t> pages, indptr{0}, lpl, kvc, bidx, pos, qind{0}, rq, qt, kt, dri;
    for (int i = 0; i < n; ++i) {
      pages.insert(pages.end(), st[i]->pages.begin(), st[i]->pages.end());
      indptr.push_back((int32_t)pages.size());
      lpl.push_back(st[i]->last_page_len(layout.page_size));
      kvc.push_back(starts[i] + lens[i]);
      for (int t = 0; t < lens[i]; ++t) { bidx.push_back(i); pos.push_back(starts[i] + t); }
      qind.push_back(qind.back() + lens[i]);
      if (i >= n_pf) { dri.push_back(i - n_pf); continue; }  // decode tail: no prefill tiles
      const int tiles = (lens[i] * group + cta - 1) / cta;
      for (int t = 0; t < tiles; ++t) { rq.push_back(i); qt.push_back(t); kt.push_back(0); }
    }
    if (dri.empty()) dri.push_back(0);
    const int num_tiles = (int)rq.size();
    std::vector<std::pair<const void*, size_t>> parts = {
        {tokens, T * 4}, {pages.data(), pages.size() * 4}, {indptr.data(), indptr.size() * 4},
        {lpl.data(), lpl.size() * 4}, {bidx.data(), bidx.size() * 4}, {pos.data(), pos.size() * 4},
        {qind.data(), qind.size() * 4}, {rq.data(), rq.size() * 4}, {qt.data(), qt.size() * 4},
        {kt.data(), kt.size() * 4}, {kvc.data(), kvc.size() * 4}};
    uint32_t total_rows = (uint32_t)P;
    parts.push_back({&total_rows, 4});
    parts.push_back({dri.data(), dri.size() * 4});                      // [12] decode request_indices 0..n_dec
    std::vector<int32_t> dzero(dri.size(), 0);
    parts.push_back({dzero.data(), dzero.size() * 4});                  // [13] decode kv_tile_indices
    std::vector<size_t> offs;
    size_t off = 0;
    for (auto& p : parts) { offs.push_back(off); off = (off + p.second + 63) & ~size_t(63); }
    if (off > pf_meta_cap) {
      PQ_HIP(hipStreamSynchronize(stream));
      if (pf_meta_dev) { PQ_HIP(hipFree(pf_meta_dev)); PQ_HIP(hipHostFree(pf_meta_host)); }
      pf_meta_cap = off * 2;
      PQ_HIP(hipMalloc(reinterpret_cast<void**>(&pf_meta_dev), pf_meta_cap));
      PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&pf_meta_host), pf_meta_cap, hipHostMallocDefault));
    }
    for (size_t i = 0; i < parts.size(); ++i) std::memcpy(pf_meta_host + offs[i], parts[i].first, parts[i].second);
    PQ_HIP(hipMemcpyAsync(pf_meta_dev, pf_meta_host, off, hipMemcpyHostToDevice, stream));
    auto D32 = [&](int i) { return reinterpret_cast<int32_t*>(pf_meta_dev + offs[i]); };
    const uint32_t* tok_d = reinterpret_cast<uint32_t*>(pf_meta_dev + offs[0]);

    const int Ti = (int)T;
    std::vector<int32_t> tap_last(n);   // debug tap: each request's last token column
    for (int i = 0; i < n; ++i) tap_last[i] = qind[i + 1] - 1;
    if (embedding_batched_cuda(embed, tok_d, pf_hidden, H, Ti, S())) { set_error("embedding failed"); return -1; }
    Half *hid = pf_hidden, *hid_out = pf_hidden_out;
    const float sm = 1.0f / std::sqrt((float)D);
    // prefill launch fusions (bit-identical to the reference op sequence; PEGAINFER_PREFILL_FUSE=0 runs it 1:1):
    //   qk_norm_rope + paged_kv_scatter -> one launch; down_proj's slice sum + residual add + the NEXT layer's input
    //   RMSNorm -> one launch (the norm sees the bf16-rounded sum exactly as add_cuda -> rms_norm does)
    const char* pf_env = getenv("PEGAINFER_PREFILL_FUSE");   // read per call: tests flip it between two engines
    const bool pf_fuse = !(pf_env && pf_env[0] == '0');
    bool normed_ready = false;   // pf_normed already holds rms_norm(hid, this layer's ln1)
    // ---- short prompts (<= 16 token columns, round 4): the launch-lean layer of the fused decode path around the prefill
    //      attention - 6 launches per layer instead of 12.  ONE stacked q|k|v GEMV with the layer-input norm in its prologue
    //      (li > 0: "add, then norm" = prefill.rs:183 + :89, the kGemvRoundSum form), per-head norm + RoPE + KV append +
    //      q de-interleave in one launch, attention, o_proj, gate_up with the fused add + RMSNorm prologue (prefill.rs:157)
    //      and the SwiGLU epilogue, down_proj.  Every piece shares its arithmetic core with the reference-named op it
    //      replaces (row slices of a stacked GEMV == three GEMVs; prologue == stand-alone norm): bit-identical to the 1:1
    //      sequence (PEGAINFER_PREFILL_FUSE=0 / PEGAINFER_PREFILL_SHORT=0), tested.
    const char* short_e = getenv("PEGAINFER_PREFILL_SHORT");   // read per call: tests flip it between two engines
    const bool short_env = !(short_e && short_e[0] == '0');
    const bool short_path = short_env && pf_fuse && Ti <= 16 && D == 128 && !tp_comm && (H & 63) == 0 && (I & 63) == 0 &&
                            (q_dim & 63) == 0 && q_dim + 2 * kv_dim <= 2 * I && !out_all_logits_host;
    if (short_path) {
      Half* qkv_st = pf_gate_up;          // [Ti, q_dim + 2 kv_dim] stacked rows (the gate|up scratch is unused: SwiGLU is fused)
      const Half* resid = nullptr;        // pending residual: the previous layer's down_proj output (in pf_o)
      for (int li = 0; li < L; ++li) {
        const Layer& ly = layers[li];
        int rc = pegainfer_gemv_fused_ex(ly.qkv, hid, qkv_st, q_dim + 2 * kv_dim, Ti, H, resid, ly.ln1,
                                         resid ? hid_out : nullptr, eps, 0, resid ? 2 : 0, S());
        if (rc) { set_error("short prefill: fused qkv GEMV failed"); return -1; }
        if (resid) std::swap(hid, hid_out);
        if (resid && tap_layer(li - 1, hid, n, tap_last.data())) return -1;
        rc = pegainfer_qkv_stacked_norm_rope_scatter(qkv_st, pf_q, ly.q_norm, ly.k_norm, cos, sin, D32(5), D32(4), kv_buffer,
                                                     layout.k_offset(li), layout.v_offset(li), D32(1), D32(2), Hq, Hkv, D,
                                                     layout.page_size, layout.page_stride, Ti, eps, S());
        if (rc) { set_error("pegainfer_qkv_stacked_norm_rope_scatter failed"); return -1; }
        rc = batch_prefill_paged_cuda_with_cta_tile_q(
            pf_q, pf_attn, kv_buffer, layout.k_offset(li), layout.v_offset(li), D32(1), D32(2), D32(3), D32(6), D32(7),
            D32(8), D32(9), D32(10), reinterpret_cast<uint32_t*>(pf_meta_dev + offs[11]), Hq, Hkv, D, layout.page_size,
            P, n_pf, num_tiles, layout.page_stride, sm, cta, S());
        if (rc) { set_error("batch_prefill_paged_cuda failed"); return -1; }
        if (n_decode_tail > 0) {
          rc = paged_attention_decode_cuda(pf_q + (size_t)P * q_dim, pf_attn + (size_t)P * q_dim, kv_buffer,
                                           layout.k_offset(li), layout.v_offset(li), D32(1), D32(2) + n_pf,
                                           D32(3) + n_pf, D32(12), D32(13), D32(10) + n_pf, Hq, Hkv, D,
                                           layout.page_size, n_decode_tail, layout.page_stride, sm, S());
          if (rc) { set_error("paged_attention_decode_cuda (unified) failed"); return -1; }
        }
        rc = pegainfer_gemv_fused(ly.o, pf_attn, pf_o, H, Ti, q_dim, nullptr, nullptr, nullptr, 0.f, 0, S());
        if (!rc) rc = pegainfer_gemv_fused(ly.gate_up, hid, pf_act, 2 * I, Ti, H, pf_o, ly.ln2, hid_out, eps, I, S());
        if (rc) { set_error("short prefill: o_proj / fused gate_up failed"); return -1; }
        std::swap(hid, hid_out);
        rc = pegainfer_gemv_fused(ly.down, pf_act, pf_o, H, Ti, I, nullptr, nullptr, nullptr, 0.f, 0, S());
        if (rc) { set_error("short prefill: down_proj failed"); return -1; }
        resid = pf_o;
      }
      if (add_cuda(hid, pf_o, hid_out, Ti * H, S())) { set_error("add_cuda failed"); return -1; }   // prefill.rs:183, last layer
      std::swap(hid, hid_out);
      if (tap_layer(L - 1, hid, n, tap_last.data())) return -1;
    }
    for (int li = short_path ? L : 0; li < L; ++li) {
      const Layer& ly = layers[li];
      if (!normed_ready) rms_norm_batched_cuda(hid, ly.ln1, pf_normed, H, Ti, eps, S());
      normed_ready = false;
      auto G = [&](const Half* w, const Half* x, Half* y, int M, int K) {
        if (Ti == 1) gemm_graphsafe_cuda(w, x, y, M, 1, K, S());
        else gemm_cuda(w, x, y, M, Ti, K, S());
      };
      // one launch over the stacked q/k/v rows, three outputs (bit-identical to three calls; from 17 tokens on - the
      // tiled kernels' range - since round 4: at 17..64 tokens the three calls were six launches, split-K GEMM + slice sum each)
      static const int split3_min = [] { const char* e = getenv("PEGAINFER_PREFILL_SPLIT3_MIN"); return e && *e ? atoi(e) : 17; }();
      if (Ti >= split3_min) {
        if (pegainfer_gemm_split3(ly.qkv, pf_normed, pf_q, q_dim, pf_k, kv_dim, pf_v, kv_dim, Ti, H, S())) {
          set_error("pegainfer_gemm_split3 failed"); return -1;
        }
      } else {
        G(ly.qkv, pf_normed, pf_q, q_dim, H);
        G(ly.qkv + (size_t)q_dim * H, pf_normed, pf_k, kv_dim, H);
        G(ly.qkv + (size_t)(q_dim + kv_dim) * H, pf_normed, pf_v, kv_dim, H);
      }
      int rc = 0;
      if (pf_fuse && D == 128) {
        rc = pegainfer_qk_norm_rope_scatter(pf_q, pf_k, pf_v, ly.q_norm, ly.k_norm, cos, sin, D32(5), D32(4), kv_buffer,
                                            layout.k_offset(li), layout.v_offset(li), D32(1), D32(2), Hq, Hkv, D,
                                            layout.page_size, layout.page_stride, Ti, eps, S());
        if (rc) { set_error("pegainfer_qk_norm_rope_scatter failed"); return -1; }
      } else {
        if (n == 1 && n_decode_tail == 0)
          prefill_qk_norm_rope_only_cuda(pf_q, pf_k, ly.q_norm, ly.k_norm, cos, sin, Hq, Hkv, D, Ti, starts[0], eps, S());
        else
          qk_norm_rope_batched_decode_cuda(pf_q, pf_k, ly.q_norm, ly.k_norm, cos, sin, D32(5), Hq, Hkv, D, Ti, eps, S());
        rc = paged_kv_scatter_cuda(kv_buffer, layout.k_offset(li), layout.v_offset(li), D32(1), D32(2), D32(3), pf_k,
                                   pf_v, D32(4), D32(5), Ti, Hkv, D, layout.page_size, layout.page_stride, kv_dim, D, S());
        if (rc) { set_error("paged_kv_scatter_cuda failed"); return -1; }
      }
      rc = batch_prefill_paged_cuda_with_cta_tile_q(
          pf_q, pf_attn, kv_buffer, layout.k_offset(li), layout.v_offset(li), D32(1), D32(2), D32(3), D32(6), D32(7),
          D32(8), D32(9), D32(10), reinterpret_cast<uint32_t*>(pf_meta_dev + offs[11]), Hq, Hkv, D, layout.page_size,
          P, n_pf, num_tiles, layout.page_stride, sm, cta, S());
      if (rc) { set_error("batch_prefill_paged_cuda failed"); return -1; }
      if (n_decode_tail > 0) {  // trailing decode columns via pointer offsets (unified_forward.rs:392-396,495-496)
        rc = paged_attention_decode_cuda(pf_q + (size_t)P * q_dim, pf_attn + (size_t)P * q_dim, kv_buffer,
                                         layout.k_offset(li), layout.v_offset(li), D32(1), D32(2) + n_pf,
                                         D32(3) + n_pf, D32(12), D32(13), D32(10) + n_pf, Hq, Hkv, D,
                                         layout.page_size, n_decode_tail, layout.page_stride, sm, S());
        if (rc) { set_error("paged_attention_decode_cuda (unified) failed"); return -1; }
      }
      if (tp_comm) {
        G(ly.o, pf_attn, pf_o, H, q_dim);
        if (all_reduce_hidden(pf_o, (size_t)Ti * H)) return -1;   // prefill.rs:154
        fused_add_rms_norm_batched_cuda(hid, pf_o, ly.ln2, pf_normed, H, Ti, eps, S());
      } else if (pegainfer_gemm_add_rms_norm(ly.o, pf_attn, pf_o, hid, ly.ln2, pf_normed, H, Ti, q_dim, eps, S())) {
        set_error("pegainfer_gemm_add_rms_norm (prefill o_proj) failed"); return -1;   // same two ops, one entry point
      }
      if (Ti > 16) {  // SwiGLU in the GEMM epilogue (bit-identical to gemm + silu_mul_fused)
        if (pegainfer_gemm_silu(ly.gate_up, pf_normed, pf_act, pf_gate_up, I, Ti, H, S())) { set_error("pegainfer_gemm_silu failed"); return -1; }
      } else {
        G(ly.gate_up, pf_normed, pf_gate_up, 2 * I, H);
        silu_mul_fused_cuda(pf_gate_up, pf_act, I, Ti, S());
      }
      if (tp_comm) {
        G(ly.down, pf_act, pf_o, H, I);
        if (all_reduce_hidden(pf_o, (size_t)Ti * H)) return -1;   // prefill.rs:180
        if (add_cuda(hid, pf_o, hid_out, Ti * H, S())) { set_error("add_cuda failed"); return -1; }
      } else if (pf_fuse && li + 1 < L) {
        if (pegainfer_gemm_add_then_rms_norm(ly.down, pf_act, pf_o, hid, hid_out, layers[li + 1].ln1, pf_normed, H, Ti, I,
                                             eps, S())) {
          set_error("pegainfer_gemm_add_then_rms_norm (prefill down_proj) failed"); return -1;
        }
        normed_ready = true;
      } else if (pegainfer_gemm_add(ly.down, pf_act, pf_o, hid, hid_out, H, Ti, I, S())) {
        set_error("pegainfer_gemm_add (prefill down_proj) failed"); return -1;
      }
      std::swap(hid, hid_out);  // prefill.rs:183-185
      if (tap_layer(li, hid, n, tap_last.data())) return -1;
    }
    // echo = true (compute_all_position_logits, prefill.rs:196-212): final RMSNorm + lm_head over EVERY position,
    // [total_tokens, vocab] bf16 to the host.  The GEMM output is staged through the gate|up scratch (pf_cap_tokens x
    // 2I elements) in token chunks, so no T x vocab device buffer is ever allocated (1024 tokens = 311 MB).
    if (out_all_logits_host) {
      if ((size_t)pf_cap_tokens * (size_t)(2 * I) < (size_t)V) { set_error("echo: prefill scratch smaller than one vocab row"); return -1; }
      const size_t chunk = std::max<size_t>(1, std::min<size_t>(T, pf_cap_tokens * (size_t)(2 * I) / (size_t)V));
      for (size_t i0 = 0; i0 < T; i0 += chunk) {
        const int nb = (int)std::min(chunk, T - i0);
        rms_norm_batched_cuda(hid + i0 * H, final_norm, pf_normed + i0 * H, H, nb, eps, S());
        gemm_cuda(lm_head, pf_normed + i0 * H, pf_gate_up, V, nb, H, S());   // void in ffi.rs (same as the reference symbol)
        PQ_HIP(hipMemcpyAsync(static_cast<Half*>(out_all_logits_host) + i0 * V, pf_gate_up, (size_t)nb * V * 2,
                              hipMemcpyDeviceToHost, stream));
        PQ_HIP(hipStreamSynchronize(stream));   // the scratch is reused by the next chunk
      }
    }
    // per request: last token -> final norm -> lm_head GEMV (prefill.rs:267-282); batched in groups that stay
    // on the decode-GEMV path so each column is bit-identical to the reference's per-request call.
    for (int i = 0; i < n; ++i)
      PQ_HIP(hipMemcpyAsync(pf_last_hidden + (size_t)i * H, hid + (size_t)(qind[i + 1] - 1) * H, (size_t)H * 2,
                            hipMemcpyDeviceToDevice, stream));
    rms_norm_batched_cuda(pf_last_hidden, final_norm, pf_last_normed, H, n, eps, S());
    for (int i0 = 0; i0 < n; i0 += 16) {
      const int nb = std::min(16, n - i0);
      gemm_graphsafe_cuda(lm_head, pf_last_normed + (size_t)i0 * H, pf_logits + (size_t)i0 * V, V, nb, H, S());
    }
    if (pegainfer_batched_top1(pf_logits, V, n, V, top1_state, tokens_out_d, S())) { set_error("top1 failed"); return -1; }
    PQ_HIP(hipMemcpyAsync(tokens_out_host, tokens_out_d, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    const uint32_t* os_status_d = tp_comm && tp_status_host && pegainfer_comm_oneshot_active(tp_comm)
                                      ? pegainfer_comm_oneshot_status_ptr(tp_comm) : nullptr;
    if (os_status_d) PQ_HIP(hipMemcpyAsync(tp_status_host, os_status_d, 12, hipMemcpyDeviceToHost, stream));
    PQ_HIP(hipStreamSynchronize(stream));
    if (os_status_d && tp_status_host[0] != 0) {
      set_error("tensor-parallel one-shot all-reduce: bounded wait expired during prefill (missing-rank mask " +
                std::to_string(tp_status_host[0] & 0xff) + "); step failed, model marked failed");
      tp_failed = true;
      return -5;
    }
    for (int i = 0; i < n; ++i) out_tokens[i] = tokens_out_host[i];
    last_logits = pf_logits;
    last_rows = n;
    if (out_logits_host) PQ_HIP(hipMemcpy(out_logits_host, pf_logits, (size_t)n * V * 2, hipMemcpyDeviceToHost));
    return 0;
  }

  // Per-kernel timing for bench.py's roofline: launch one GEMM call site `iters` times, cycling through
  // the layers' real weights (each launch streams a different matrix, as in a decode step), bracketed
  // by hipEvents on the model stream.  which: 0 q+k+v rows (fused qkv matrix), 1 o, 2 gate_up, 3 down, 4 lm_head.
  float bench_gemv(int which, int iters, int bs) {
    if (!finalized || iters <= 0 || bs < 1 || bs > max_bs) return -1.f;
    auto launch = [&](int it) {
      const Layer& ly = layers[which >= 10 ? 0 : it % L];  // which >= 10: same layer every time (cache-warm probe)
      switch (which % 10) {
        case 0: gemm_graphsafe_cuda(ly.qkv, normed, gate_up_out, q_dim + 2 * kv_dim, bs, H, S()); break;
        case 1: gemm_graphsafe_cuda(ly.o, attn_out, attn_proj, H, bs, q_dim, S()); break;
        case 2: gemm_graphsafe_cuda(ly.gate_up, normed, gate_up_out, 2 * I, bs, H, S()); break;
        case 3: gemm_graphsafe_cuda(ly.down, mlp_act, mlp_out, H, bs, I, S()); break;
        case 4: gemm_graphsafe_cuda(lm_head, normed, logits, V, bs, H, S()); break;
        // the kernels the fused decode step really launches (decode_mode 1):
        case 5: pegainfer_gemv_fused(ly.gate_up, hidden, mlp_act, 2 * I, bs, H, attn_proj, ly.ln2, hidden2, eps, I, S()); break;
        case 6: pegainfer_gemv_fused(ly.qkv, hidden, qkv_out, q_dim + 2 * kv_dim, bs, H, mlp_out, ly.ln1, hidden2, eps, 0, S()); break;
        case 7: pegainfer_gemv_fused(lm_head, hidden, logits, V, bs, H, mlp_out, final_norm, hidden2, eps, 0, S()); break;
        // 8 / 9 = 5 / 6 with ONE norm weight for every launch (hot in L2) while the matrices still cycle: isolates
        // what the cold norm weight costs the prologue (tools/gemv_probe.py)
        case 8: pegainfer_gemv_fused(ly.gate_up, hidden, mlp_act, 2 * I, bs, H, attn_proj, layers[0].ln2, hidden2, eps, I, S()); break;
        default: pegainfer_gemv_fused(ly.qkv, hidden, qkv_out, q_dim + 2 * kv_dim, bs, H, mlp_out, layers[0].ln1, hidden2, eps, 0, S()); break;
      }
    };
    for (int i = 0; i < 3; ++i) launch(i);
    hipEventRecord(ev0, stream);
    for (int i = 0; i < iters; ++i) launch(i + 3);
    hipEventRecord(ev1, stream);
    hipStreamSynchronize(stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    return ms / iters;
  }

  // gpu_sample rule (ops/sampling.rs:109-170) on one column of the last step's logits
  int sample(int column, float temperature, int top_k, float top_p, float random_val, int32_t* out) {
    if (!last_logits || column < 0 || column >= last_rows) { set_error("no logits for that column"); return -1; }
    const Half* lg = last_logits + (size_t)column * V;
    if ((temperature <= 0.0f || top_k == 1) && top_p >= 1.0f) {
      flashinfer_top1_cuda(lg, top1_value, row_states, sample_out_d, V, S());
    } else {
      uint32_t bits;
      std::memcpy(&bits, &random_val, 4);
      gpu_sample_flashinfer_cuda(lg, probs_scratch, valid_scratch, sample_out_d, V, 1.0f / temperature, top_k, top_p,
                                 (uint64_t)bits, S());
    }
    PQ_HIP(hipStreamSynchronize(stream));
    PQ_HIP(hipMemcpy(out, sample_out_d, 4, hipMemcpyDeviceToHost));
    return 0;
  }

  // executor.rs:807-816 extract_logprobs: the logits row of request `column` of the last step goes to the host as f32 and
  // compute_logprobs_from_cpu (executor.rs:400-434) runs on it
  std::vector<uint16_t> lp_row_bits;
  std::vector<float> lp_row;
  int logprobs(int column, uint32_t token, int top_k, float* out_lp, uint32_t* top_ids, float* top_lps) {
    if (!last_logits || column < 0 || column >= last_rows) { set_error("no logits for that column"); return -1; }
    if (token >= (uint32_t)V) { set_error("logprobs: token outside the vocabulary"); return -1; }
    lp_row_bits.resize(V);
    lp_row.resize(V);
    PQ_HIP(hipStreamSynchronize(stream));
    PQ_HIP(hipMemcpy(lp_row_bits.data(), last_logits + (size_t)column * V, (size_t)V * 2, hipMemcpyDeviceToHost));
    for (int i = 0; i < V; ++i) {
      const uint32_t u = (uint32_t)lp_row_bits[i] << 16;
      std::memcpy(&lp_row[i], &u, 4);
    }
    return pegainfer_logprobs_from_logits(lp_row.data(), V, token, top_k, out_lp, top_ids, top_lps);
  }

  ~Model() {
    if (stream) hipStreamSynchronize(stream);
    if (tp_comm && tp_comm_owned) pegainfer_comm_destroy(tp_comm);
    if (tp_status_host) hipHostFree(tp_status_host);
    for (auto& b : graphs)
      for (auto& g : b)
        if (g) hipGraphExecDestroy(g);
    for (void* p : owned) hipFree(p);
    Half* pf[] = {pf_hidden, pf_hidden_out, pf_normed, pf_q, pf_k, pf_v, pf_o, pf_gate_up, pf_act, pf_attn,
                  pf_last_hidden, pf_last_normed, pf_logits};
    for (Half* p : pf)
      if (p) hipFree(p);
    if (pf_meta_dev) hipFree(pf_meta_dev);
    if (pf_meta_host) hipHostFree(pf_meta_host);
    if (meta_host) hipHostFree(meta_host);
    if (tokens_out_host) hipHostFree(tokens_out_host);
    if (attn_status_host) hipHostFree(attn_status_host);
    for (int r = 0; r < kChainRing; ++r) {
      if (chain_ring[r]) hipHostFree(chain_ring[r]);
      if (chain_ev[r]) hipEventDestroy(chain_ev[r]);
    }
    if (chain_tokens_host) hipHostFree(chain_tokens_host);
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
    if (stream) hipStreamDestroy(stream);
    cublas_destroy();
  }
};

}  // namespace pq

using pq::Model;
static Model* M(pegainfer_qwen3_t m) { return static_cast<Model*>(m); }
static thread_local std::string g_create_error;

extern "C" {

pegainfer_qwen3_t pegainfer_qwen3_create(int32_t device_ordinal, int32_t hidden_size, int32_t num_layers,
                                         int32_t num_attention_heads, int32_t num_kv_heads, int32_t head_dim,
                                         int32_t intermediate_size, int32_t vocab_size, float rms_norm_eps,
                                         float rope_theta, int32_t tie_word_embeddings,
                                         int32_t max_position_embeddings, int32_t num_kv_pages,
                                         int32_t max_batch_size, int32_t enable_graph, int32_t decode_mode,
                                         int32_t split_policy) {
  if (head_dim != 128 || num_kv_heads <= 0 || num_attention_heads % num_kv_heads != 0 || num_kv_pages < 2 ||
      max_batch_size < 1 || max_batch_size > 64) {
    g_create_error = "unsupported configuration";
    return nullptr;
  }
  if (decode_mode != 0 && decode_mode != 1) {   // 2 was the persistent engine, deleted in round 4: do not run something else silently
    g_create_error = "decode_mode must be 0 (reference op sequence) or 1 (fused decode kernels)";
    return nullptr;
  }
  auto* m = new Model(device_ordinal, hidden_size, num_layers, num_attention_heads, num_kv_heads, head_dim,
                      intermediate_size, vocab_size, rms_norm_eps, rope_theta, tie_word_embeddings,
                      max_position_embeddings, num_kv_pages, max_batch_size, enable_graph, decode_mode, split_policy);
  if (m->init() != 0) {
    g_create_error = m->err;
    fprintf(stderr, "pegainfer_qwen3_create: %s\n", m->err.c_str());
    delete m;
    return nullptr;
  }
  return m;
}
void pegainfer_qwen3_destroy(pegainfer_qwen3_t m) { delete M(m); }
const char* pegainfer_qwen3_last_error(pegainfer_qwen3_t m) { return m ? M(m)->err.c_str() : g_create_error.c_str(); }
int32_t pegainfer_qwen3_load_tensor(pegainfer_qwen3_t m, const char* name, const void* host_bf16, int64_t numel) {
  return M(m)->load_tensor(name, host_bf16, numel);
}
int32_t pegainfer_qwen3_fill_synthetic(pegainfer_qwen3_t m, uint64_t seed, float std) { return M(m)->fill_synthetic(seed, std); }
int32_t pegainfer_qwen3_finalize(pegainfer_qwen3_t m) { return M(m)->finalize(); }
int32_t pegainfer_qwen3_new_request(pegainfer_qwen3_t m) { return M(m)->new_request(); }
int32_t pegainfer_qwen3_drop_request(pegainfer_qwen3_t m, int32_t id) { return M(m)->drop_request(id); }
int32_t pegainfer_qwen3_request_seq_len(pegainfer_qwen3_t m, int32_t id) {
  auto* r = M(m)->req(id);
  return r ? r->seq_len : -1;
}
// host-only probe of the native reader (CPU unit tests): shape / dtype / byte checksum of one tensor
int32_t pegainfer_safetensors_probe(const char* path, const char* name, int64_t* shape4, int32_t* ndim, char* dtype8,
                                    uint64_t* byte_sum, int32_t* num_tensors) {
  pst::Checkpoint ck;
  std::string e;
  if (!ck.open(path, &e)) return -1;
  if (num_tensors) *num_tensors = (int32_t)ck.tensors().size();
  const pst::TensorView* t = ck.find(name);
  if (!t) return -2;
  *ndim = (int32_t)t->shape.size();
  for (int i = 0; i < 4; ++i) shape4[i] = i < *ndim ? t->shape[i] : 0;
  std::memset(dtype8, 0, 8);
  std::strncpy(dtype8, t->dtype.c_str(), 7);
  uint64_t s = 0;
  for (size_t i = 0; i < t->nbytes; ++i) s += t->data[i] * (uint64_t)(1 + (i & 0xFF));
  *byte_sum = s;
  return 0;
}
int32_t pegainfer_qwen3_load_safetensors(pegainfer_qwen3_t m, const char* path, int32_t tp_rank, int32_t tp_world) {
  return M(m)->load_safetensors(path, tp_rank, tp_world);
}
pegainfer_qwen3_t pegainfer_qwen3_from_pretrained(const char* model_dir, int32_t device_ordinal, int32_t tp_rank,
                                                  int32_t tp_world, int32_t num_kv_pages, int32_t max_batch_size,
                                                  int32_t enable_graph, int32_t decode_mode, int32_t split_policy) {
  // Config::from_file (pegainfer-qwen3-4b/src/config.rs): HF config.json
  std::string err;
  auto f = pst::MappedFile::open(std::string(model_dir) + "/config.json", &err);
  pst::Json j;
  if (!f || !pst::JsonParser(reinterpret_cast<const char*>(f->data), f->size).parse(&j) || j.kind != pst::Json::Obj) {
    fprintf(stderr, "pegainfer_qwen3_from_pretrained: cannot read %s/config.json\n", model_dir);
    return nullptr;
  }
  const int H = (int)j.number_or("hidden_size", 0), L = (int)j.number_or("num_hidden_layers", 0);
  const int Hq = (int)j.number_or("num_attention_heads", 0), Hkv = (int)j.number_or("num_key_value_heads", Hq);
  const int D = (int)j.number_or("head_dim", Hq ? H / Hq : 0), I = (int)j.number_or("intermediate_size", 0);
  const int V = (int)j.number_or("vocab_size", 0);
  double theta = j.number_or("rope_theta", 0);
  if (theta == 0) { const pst::Json* rp = j.get("rope_parameters"); theta = rp ? rp->number_or("rope_theta", 1e6) : 1e6; }
  const pst::Json* tie = j.get("tie_word_embeddings");
  const int max_pos = (int)j.number_or("max_position_embeddings", 4096);
  if (tp_world < 1 || !H || !L || !Hq || !I || !V || Hq % tp_world || Hkv % tp_world || I % tp_world) {
    fprintf(stderr, "pegainfer_qwen3_from_pretrained: config not usable at TP world %d\n", tp_world);
    return nullptr;
  }
  pegainfer_qwen3_t m = pegainfer_qwen3_create(device_ordinal, H, L, Hq / tp_world, Hkv / tp_world, D, I / tp_world, V,
                                               (float)j.number_or("rms_norm_eps", 1e-6), (float)theta,
                                               tie && tie->kind == pst::Json::Bool && tie->b ? 1 : 0, max_pos,
                                               num_kv_pages, max_batch_size, enable_graph, decode_mode, split_policy);
  if (!m) return nullptr;
  if (M(m)->load_safetensors(model_dir, tp_rank, tp_world)) {
    fprintf(stderr, "pegainfer_qwen3_from_pretrained: %s\n", M(m)->err.c_str());
    pegainfer_qwen3_destroy(m);
    return nullptr;
  }
  return m;
}
int32_t pegainfer_qwen3_available_pages(pegainfer_qwen3_t m) { return M(m)->pool.available(); }
int32_t pegainfer_qwen3_capacity_pages(pegainfer_qwen3_t m) { return M(m)->pool.capacity(); }
int32_t pegainfer_qwen3_max_batch_size(pegainfer_qwen3_t m) { return M(m)->max_bs; }
int32_t pegainfer_qwen3_vocab_size(pegainfer_qwen3_t m) { return M(m)->V; }
int32_t pegainfer_qwen3_prefill(pegainfer_qwen3_t m, int32_t n, const int32_t* ids, const int32_t* lens,
                                const uint32_t* tokens, int32_t* out_tokens, void* out_logits_host) {
  return M(m)->prefill(n, ids, lens, tokens, out_tokens, out_logits_host);
}
int32_t pegainfer_qwen3_prefill_echo(pegainfer_qwen3_t m, int32_t n, const int32_t* ids, const int32_t* lens,
                                     const uint32_t* tokens, int32_t* out_tokens, void* out_logits_host,
                                     void* out_all_logits_host) {
  return M(m)->prefill(n, ids, lens, tokens, out_tokens, out_logits_host, 0, out_all_logits_host);
}
int32_t pegainfer_qwen3_unified_step(pegainfer_qwen3_t m, int32_t n_prefill, int32_t n_decode, const int32_t* ids,
                                     const int32_t* lens, const uint32_t* tokens, int32_t* out_tokens,
                                     void* out_logits_host) {
  return M(m)->prefill(n_prefill + n_decode, ids, lens, tokens, out_tokens, out_logits_host, n_decode);
}
int32_t pegainfer_qwen3_decode(pegainfer_qwen3_t m, int32_t n, const int32_t* ids, const uint32_t* token_ids,
                               int32_t* out_tokens, void* out_logits_host) {
  return M(m)->decode(n, ids, token_ids, out_tokens, out_logits_host);
}
int32_t pegainfer_qwen3_decode_greedy_chain(pegainfer_qwen3_t m, int32_t n_requests, const int32_t* request_ids,
                                            const uint32_t* first_token_ids, int32_t n_steps, int32_t* out_tokens) {
  return M(m)->decode_greedy_chain(n_requests, request_ids, first_token_ids, n_steps, out_tokens);
}
int32_t pegainfer_qwen3_sample(pegainfer_qwen3_t m, int32_t column, float temperature, int32_t top_k, float top_p,
                               float random_val, int32_t* out_token) {
  return M(m)->sample(column, temperature, top_k, top_p, random_val, out_token);
}
int32_t pegainfer_qwen3_export_tensor(pegainfer_qwen3_t m, const char* name, void* host_bf16, int64_t numel) {
  return M(m)->export_tensor(name, host_bf16, numel);
}
int32_t pegainfer_qwen3_logprobs(pegainfer_qwen3_t m, int32_t column, uint32_t token, int32_t top_k, float* out_logprob,
                                 uint32_t* out_top_ids, float* out_top_logprobs) {
  return M(m)->logprobs(column, token, top_k, out_logprob, out_top_ids, out_top_logprobs);
}
// compute_logprobs_from_cpu (executor.rs:400-434), statement for statement: f32 max fold, f32 sequential sum of
// exp(x - max), log_sum_exp = max + ln(sum); the top list is built by the reference's ordered insertion (strictly-greater
// test against the last entry, insertion at partition_point(v > val): value descending, a later equal value in FRONT of
// an earlier one, none displaced from a full list by an equal value).  Returns the number of top
// entries written (min(top_k, n)), or -1 (empty row / token out of range: the reference returns None)
int32_t pegainfer_logprobs_from_logits(const float* logits_f32, int32_t n, uint32_t token, int32_t top_k,
                                       float* out_logprob, uint32_t* out_top_ids, float* out_top_logprobs) {
  if (!logits_f32 || n <= 0 || token >= (uint32_t)n) return -1;
  float max_val = -INFINITY;
  for (int i = 0; i < n; ++i) max_val = std::fmax(max_val, logits_f32[i]);   // f32::max: NaN-ignoring like fmaxf
  float sum_exp = 0.f;
  for (int i = 0; i < n; ++i) sum_exp += std::exp(logits_f32[i] - max_val);
  const float lse = max_val + std::log(sum_exp);
  if (out_logprob) *out_logprob = logits_f32[token] - lse;
  const int k = std::min(std::max(top_k, 0), n);
  if (k == 0) return 0;
  std::vector<std::pair<uint32_t, float>> best;
  best.reserve((size_t)k + 1);
  for (int i = 0; i < n; ++i) {
    const float val = logits_f32[i];
    if ((int)best.size() < k || val > best.back().second) {
      size_t pos = 0;   // partition_point(|v| v > val)
      size_t lo = 0, hi = best.size();
      while (lo < hi) { const size_t mid = (lo + hi) / 2; if (best[mid].second > val) lo = mid + 1; else hi = mid; }
      pos = lo;
      best.insert(best.begin() + pos, {(uint32_t)i, val});
      if ((int)best.size() > k) best.pop_back();
    }
  }
  for (size_t i = 0; i < best.size(); ++i) {
    if (out_top_ids) out_top_ids[i] = best[i].first;
    if (out_top_logprobs) out_top_logprobs[i] = best[i].second - lse;
  }
  return (int32_t)best.size();
}
int32_t pegainfer_qwen3_rccl_unique_id(void* out_128_bytes) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
  std::memcpy(out_128_bytes, &id, sizeof(id));
  return 0;
}
int32_t pegainfer_qwen3_attach_tp(pegainfer_qwen3_t m, int32_t rank, int32_t world, const void* unique_id_128_bytes) {
  return M(m)->attach_tp(rank, world, unique_id_128_bytes);
}
int32_t pegainfer_qwen3_attach_comm(pegainfer_qwen3_t m, void* comm) { return M(m)->attach_comm(comm); }
int32_t pegainfer_qwen3_tp_oneshot_active(pegainfer_qwen3_t m) {
  return M(m)->tp_comm ? pegainfer_comm_oneshot_active(M(m)->tp_comm) : 0;
}
int32_t pegainfer_qwen3_debug_hidden_enable(pegainfer_qwen3_t m, int32_t enable) { return M(m)->debug_hidden_enable(enable); }
int32_t pegainfer_qwen3_debug_hidden(pegainfer_qwen3_t m, int32_t layer, void* out_host_bf16, int32_t max_rows) {
  return M(m)->debug_hidden(layer, out_host_bf16, max_rows);
}
float pegainfer_qwen3_last_step_ms(pegainfer_qwen3_t m) { return M(m)->last_step_ms; }
float pegainfer_qwen3_bench_gemv(pegainfer_qwen3_t m, int32_t which, int32_t iters, int32_t bs) {
  return M(m)->bench_gemv(which, iters, bs);
}
int32_t pegainfer_qwen3_last_attention_path(pegainfer_qwen3_t m) { return M(m)->last_path; }
int64_t pegainfer_qwen3_weight_bytes(pegainfer_qwen3_t m) { return M(m)->weight_bytes; }
void* pegainfer_qwen3_stream(pegainfer_qwen3_t m) { return M(m)->S(); }

// ---- pure-host hooks ----
void* pegainfer_pagepool_create(int32_t capacity_pages) { return new pq::PagePool(capacity_pages); }
void pegainfer_pagepool_destroy(void* pool) { delete static_cast<pq::PagePool*>(pool); }
int32_t pegainfer_pagepool_available(void* pool) { return static_cast<pq::PagePool*>(pool)->available(); }
int32_t pegainfer_pagepool_acquire(void* pool, int32_t n, int32_t* out_pages) {
  std::vector<int32_t> v;
  if (!static_cast<pq::PagePool*>(pool)->acquire(n, &v)) return -1;
  for (int i = 0; i < n; ++i) out_pages[i] = v[i];
  return n;
}
void pegainfer_pagepool_release(void* pool, const int32_t* pages, int32_t n) {
  static_cast<pq::PagePool*>(pool)->release(pages, n);
}
int32_t pegainfer_split_kv_plan(int32_t policy, int32_t n_requests, const int32_t* seq_lens, int32_t padded_bs,
                                int32_t num_kv_heads, int32_t* request_indices, int32_t* kv_tile_indices,
                                int32_t* o_indptr, uint8_t* block_valid_mask, int32_t* kv_chunk_size,
                                int32_t* use_split) {
  std::vector<int> lens(seq_lens, seq_lens + n_requests);
  const pq::SplitPlan p = pq::make_split_plan(policy, lens, padded_bs, num_kv_heads);
  std::memcpy(request_indices, p.request_indices.data(), p.slots * 4);
  std::memcpy(kv_tile_indices, p.kv_tile_indices.data(), p.slots * 4);
  std::memcpy(o_indptr, p.o_indptr.data(), (padded_bs + 1) * 4);
  std::memcpy(block_valid_mask, p.valid.data(), p.slots);
  *kv_chunk_size = p.chunk;
  *use_split = p.use_split ? 1 : 0;
  return p.slots;
}
int32_t pegainfer_bucket_for(int32_t bs) { return pq::bucket_for(bs); }

}  // extern "C"
