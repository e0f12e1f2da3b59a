// Pure-host paged-KV bookkeeping, mirroring the reference's allocator behaviour exactly:
//   PagePool  - pegainfer-core/src/page_pool.rs:34-127 (LIFO free list seeded so pops yield 0,1,2,...;
//               release pushes the pages back reversed so they are handed out again in order)
//   KvLayout  - pegainfer-core/src/kv_pool.rs:14-54
//   KvState   - kv_pool.rs:147-237 (ensure_capacity / advance / last_page_len / reset)
//   decode buckets + split-KV plan - pegainfer-qwen3-4b/src/batch_decode_buffers.rs:12-46,229-287
// No device code here: unit-tested on CPU through the C hooks in pegainfer_qwen3.h.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace pq {

class PagePool {
 public:
  explicit PagePool(int capacity) : capacity_(capacity) {
    free_.reserve(capacity);
    for (int p = capacity - 1; p >= 0; --p) free_.push_back(p);
  }
  int capacity() const { return capacity_; }
  int available() const { return static_cast<int>(free_.size()); }
  // all-or-nothing (page_pool.rs:46-70)
  bool acquire(int n, std::vector<int32_t>* out) {
    if (n < 0 || static_cast<int>(free_.size()) < n) return false;
    for (int i = 0; i < n; ++i) {
      out->push_back(free_.back());
      free_.pop_back();
    }
    return true;
  }
  void release(const int32_t* pages, int n) {  // page_pool.rs:118-127: extend(pages.rev())
    for (int i = n - 1; i >= 0; --i) free_.push_back(pages[i]);
  }

 private:
  int capacity_;
  std::vector<int32_t> free_;
};

struct KvLayout {
  int page_size, num_layers, num_kv_heads, head_dim;
  int64_t kv_block_len, layer_stride, page_stride;
  KvLayout(int layers, int kv_heads, int hd, int ps)
      : page_size(ps), num_layers(layers), num_kv_heads(kv_heads), head_dim(hd) {
    kv_block_len = static_cast<int64_t>(ps) * kv_heads * hd;
    layer_stride = 2 * kv_block_len;
    page_stride = layers * layer_stride;
  }
  int64_t k_offset(int layer) const { return layer * layer_stride; }
  int64_t v_offset(int layer) const { return layer * layer_stride + kv_block_len; }
};

struct KvState {
  std::vector<int32_t> pages;
  int seq_len = 0;
  bool live = false;
  int last_page_len(int page_size) const {
    if (seq_len == 0) return 0;
    const int rem = seq_len % page_size;
    return rem == 0 ? page_size : rem;
  }
  // pages a later ensure_capacity(tokens) would have to acquire (callers validate a whole batch before mutating)
  int pages_short(int tokens, int page_size) const {
    const int needed = (tokens + page_size - 1) / page_size;
    return std::max(0, needed - static_cast<int>(pages.size()));
  }
  bool ensure_capacity(PagePool* pool, int tokens, int page_size) {
    const int needed = (tokens + page_size - 1) / page_size;
    const int held = static_cast<int>(pages.size());
    if (needed <= held) return true;
    return pool->acquire(needed - held, &pages);
  }
  void reset(PagePool* pool) {
    if (!pages.empty()) pool->release(pages.data(), static_cast<int>(pages.size()));
    pages.clear();
    seq_len = 0;
  }
};

constexpr int kBatchBuckets[] = {1, 2, 4, 8, 16, 32, 64};
constexpr int kNumBuckets = 7;
constexpr int kSplitMaxChunksPerRequest = 64;

inline int bucket_for(int bs) {
  for (int b : kBatchBuckets)
    if (b >= bs) return b;
  return -1;
}
inline int bucket_index(int padded) {
  for (int i = 0; i < kNumBuckets; ++i)
    if (kBatchBuckets[i] == padded) return i;
  return -1;
}

struct SplitPlan {
  std::vector<int32_t> request_indices, kv_tile_indices, o_indptr;
  std::vector<uint8_t> valid;
  int chunk = 0, slots = 0;
  bool use_split = false;
};

// policy 0: the reference's rule. policy 1: MI355X - split whenever it raises the number of
// workgroups towards >= 256 (256 CUs): chunks/request = clamp(ceil(256/(bs*kv_heads)), 1, 64),
// chunk = max(64, round_up16(ceil(L/chunks))); 64 tokens = one 16-token tile per wave, i.e. a workgroup's
// whole K/V fetch is a single round trip.
constexpr int kSplitChunkAlignDefault = 128;  // whole rounds of an 8-wave workgroup (16 = the round-4 plan; measured: see make_split_plan)
constexpr int kOprojFusedMaxSeqDefault = 12288;   // measured: ctx 4096 2.213 -> 2.166 ms per step, ctx 10 000 2.422 -> 2.409 (profiles/r4_oproj_maxseq_ab.txt)
// longest single request whose decode step keeps the fused attention + o_proj launch (20 chunks; PEGAINFER_OPROJ_MAX_SEQ
// is the A/B knob: beyond 2304 tokens a chunk is more than one tile per wave, so the scan gets longer while the o_proj
// launch it hides stays 6 us)
inline int oproj_fused_max_seq() {
  static const int v = [] { const char* e = getenv("PEGAINFER_OPROJ_MAX_SEQ"); const int x = e && *e ? atoi(e) : kOprojFusedMaxSeqDefault; return x < 128 ? 128 : x; }();
  return v;
}
// requests per step the fused attention + o_proj launch serves (PEGAINFER_OPROJ_MAX_BATCH=1 restores the round-4 form)
constexpr int kOprojFusedMaxBatchDefault = 2;
inline int oproj_fused_max_batch() {
  static const int v = [] { const char* e = getenv("PEGAINFER_OPROJ_MAX_BATCH"); const int x = e && *e ? atoi(e) : kOprojFusedMaxBatchDefault; return x < 1 ? 1 : (x > 2 ? 2 : x); }();
  return v;
}
// chunks the requests of such a step are cut into, all together (PEGAINFER_OPROJ_CHUNKS, A/B knob)
inline int oproj_fused_max_chunks() {
  // 20 since round 5 (18 before): 10 000 tokens = 20 x 512 = four balanced tile rounds per workgroup; the 12 padding slots left
  // give 96 o_proj workgroups x 14 rows per wave quad for hidden 2560 (kOprojMaxRows)
  static const int c = [] { const char* e = getenv("PEGAINFER_OPROJ_CHUNKS"); const int v = e && *e ? atoi(e) : 20; return v < 2 ? 2 : (v > 30 ? 30 : v); }();
  return c;
}
// fused_oproj_usable: the caller can run the fused attention + o_proj launch on this step (single GPU, form enabled, shape
// fits) - only then is a short single request capped at oproj_fused_max_chunks() chunks; otherwise the cap would cost
// attention workgroups and buy nothing
inline SplitPlan make_split_plan(int policy, const std::vector<int>& seq_lens, int padded_bs, int num_kv_heads,
                                 bool fused_oproj_usable = true) {
  SplitPlan p;
  int max_seq = 0;
  for (int n : seq_lens) max_seq = std::max(max_seq, n);
  int slots_per_request = kSplitMaxChunksPerRequest;   // launch-grid slots per (padded) request
  if (policy == 0) {
    p.chunk = std::max(256, (max_seq + kSplitMaxChunksPerRequest - 1) / kSplitMaxChunksPerRequest);
    p.use_split = padded_bs <= 2 && max_seq >= 1024;
  } else {
    // one workgroup per CU is the measured optimum (device ms/step at ctx 1024 / 4096 / 10000: target 128 ->
    // 2.46 / 2.84 / 3.42, 256 -> 2.43 / 2.75 / 3.06, 512 -> 2.44 / 2.87 / 3.17: more partials cost more in the merge
    // than they gain in the scan); PEGAINFER_SPLIT_TARGET_WGS is the probe knob
    // Batches: with 32..255 (request, kv head) pairs two workgroups per CU win (device ms/step at ctx 1024, bs 8 / 16:
    // 3.26 / 4.16 at 256 vs 3.13 / 3.82 at 512); from 256 pairs on the un-split scan is best (bs 32: 4.85 vs 5.35).
    static const int target_env = [] { const char* e = getenv("PEGAINFER_SPLIT_TARGET_WGS"); return e ? atoi(e) : 0; }();
    const int pairs = padded_bs * num_kv_heads;
    const int target_wgs = target_env > 0 ? target_env : (pairs >= 32 && pairs < 256 ? 512 : 256);
    int want = pairs >= 256 && target_env <= 0 ? 1 : (target_wgs + pairs - 1) / pairs;
    want = std::min(std::max(want, 1), kSplitMaxChunksPerRequest);
    // single request, up to kOprojFusedMaxSeq tokens: at most kOprojFusedMaxChunks chunks (an 8-wave workgroup scans 128
    // tokens in one tile per wave) - the other slots of the launch grid stay
    // padding, and the fused attention + o_proj launch gives their workgroups the o_proj rows (attn_oproj_kernel)
    const int grid_slots = want;   // the launch grid keeps its one-workgroup-per-CU size
    // (round 5) two requests share the same budget: 20 / 2 = 10 chunks each, the launch grid stays 2 x 16 slots x kv heads
    if (fused_oproj_usable && pairs <= oproj_fused_max_batch() * num_kv_heads && max_seq <= oproj_fused_max_seq() && target_env <= 0)
      want = std::min(want, std::max(1, oproj_fused_max_chunks() / std::max(1, padded_bs)));
    int chunk = (max_seq + want - 1) / want;
    chunk = std::max(64, (chunk + 15) / 16 * 16);
    // Chunks of more than one tile per wave (an 8-wave workgroup scans 128 tokens per round of its waves) are rounded up to
    // whole rounds when the fused form's plan applies: a 560-token chunk is 35 tiles dealt to 8 waves - three waves scan 5,
    // five scan 4 - and the workgroup's merge waits for the slowest (the "publish" phase of the in-kernel stamps grows from
    // 2.4 us at 1 k to 5.7 us at 10 k tokens, profiles/r5_attn_phase_trace.txt).  Same-box A/B (profiles/r5_long_ctx_chunks_ab.txt):
    // ctx 4096 +0.35 %, ctx 10 000 with 20 chunks of 512 tokens +0.55 % (18 chunks of 640: -2 %), ctx 2048 unchanged.
    // PEGAINFER_SPLIT_CHUNK_ALIGN / PEGAINFER_OPROJ_CHUNKS are the A/B knobs.
    static const int align_env = [] { const char* e = getenv("PEGAINFER_SPLIT_CHUNK_ALIGN"); return e && *e ? atoi(e) : 0; }();
    const int align = align_env > 0 ? align_env : kSplitChunkAlignDefault;
    if (fused_oproj_usable && pairs <= oproj_fused_max_batch() * num_kv_heads && chunk > 128 && align > 16 && target_env <= 0)
      chunk = (chunk + align - 1) / align * align;
    p.chunk = chunk;
    p.use_split = want > 1 && max_seq > chunk;
    // a request never gets more than `want` chunks, and `want` depends on the bucket only: the launch grid (fixed
    // at graph capture) needs padded_bs * want slots, not the reference's padded_bs * 64 - at bs 16 that is 512
    // workgroups instead of 8192 of which 7680 exited at once (SQ_WAVES per launch 32768 -> 2048)
    slots_per_request = grid_slots;
  }
  p.slots = padded_bs * slots_per_request;
  p.o_indptr.push_back(0);
  for (size_t r = 0; r < seq_lens.size(); ++r) {
    const int chunks = std::max(1, (seq_lens[r] + p.chunk - 1) / p.chunk);
    for (int c = 0; c < chunks; ++c) {
      p.request_indices.push_back(static_cast<int32_t>(r));
      p.kv_tile_indices.push_back(c);
      p.valid.push_back(1);
    }
    p.o_indptr.push_back(static_cast<int32_t>(p.request_indices.size()));
  }
  for (int r = static_cast<int>(seq_lens.size()); r < padded_bs; ++r)
    p.o_indptr.push_back(static_cast<int32_t>(p.request_indices.size()));
  while (static_cast<int>(p.request_indices.size()) < p.slots) {
    p.request_indices.push_back(0);
    p.kv_tile_indices.push_back(0);
    p.valid.push_back(0);
  }
  return p;
}

}  // namespace pq
