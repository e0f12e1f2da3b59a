// C++ host runtime for the Qwen3.5 hybrid forward pass on MI355X (include/pegainfer_qwen35.h lists the reference
// files each piece mirrors).  Same shape as the Qwen3 runtime: one model = one device + one in-order HIP stream,
// every op goes through the C ABI of libpegainfer_kernels_hip.so in the reference's order
// (prefill.rs:121-449, batch_decode.rs:43-365), decode metadata travels as ONE pinned block, and the decode
// step is captured into a hipGraph.  The linear-attention layers own per-request state (conv window + fp32
// delta-rule matrix), so the graph is keyed by the list of request ids and re-captured when it changes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "kv_pool.h"
#include "safetensors_loader.h"
#include "pegainfer_kernels.h"
#include "pegainfer_kernels_ext.h"
#include "pegainfer_qwen3.h"    // pegainfer_logprobs_from_logits (pure host)
#include "pegainfer_qwen35.h"

namespace pq35 {

using pq::KvLayout;
using pq::KvState;
using pq::PagePool;

#define P35_HIP(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                          \
      return -1;                                                                             \
    }                                                                                        \
  } while (0)

static inline uint16_t host_f2bf(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

template <typename T>
static __global__ void fill_normal35_kernel(T* out, long n, uint64_t seed, float std, float mean) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u1 = ((float)(z >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)((z >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
    const float g = mean + std * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    if constexpr (sizeof(T) == 2) out[i] = __builtin_bit_cast(uint16_t, static_cast<__bf16>(g));
    else out[i] = g;
  }
}

constexpr int LK = 128;  // linear key_dim == value_dim (fixed by the chunk-wise kernels)

struct Layer35 {
  bool full = false;
  Half *ln1 = nullptr, *ln2 = nullptr, *gate = nullptr, *up = nullptr, *down = nullptr;
  Half *q_proj = nullptr, *k_proj = nullptr, *v_proj = nullptr, *o_proj = nullptr, *q_norm = nullptr, *k_norm = nullptr;
  Half *in_qkv = nullptr, *in_z = nullptr, *in_b = nullptr, *in_a = nullptr, *conv_w = nullptr, *dt_bias = nullptr,
       *out_proj = nullptr;
  float *a_log = nullptr, *norm_w = nullptr;
};

struct Request35 {
  KvState kv;
  std::vector<Half*> conv;     // per linear layer [C * (K-1)] bf16
  std::vector<float*> state;   // per linear layer [vh, 128, 128] f32
  bool allocated = false;
};

struct Model35 {
  int device, H, I, L, V, Hq, Hkv, D, kh, vh, convK, rotary, max_pos, num_pages, max_bs, enable_graph, split_policy;
  int decode_mode = [] { const char* e = getenv("PEGAINFER_Q35_DECODE_MODE"); return e ? atoi(e) : 1; }();
  int split_slots = 0;  // slots launched by the partition-KV attention of the current step (= SplitPlan::slots)
  float eps, theta;
  int q_dim, kv_dim, C, Z, n_full = 0, n_lin = 0;
  std::string err;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<Layer35> layers;
  Half *embed = nullptr, *final_norm = nullptr, *cos = nullptr, *sin = nullptr;
  std::vector<void*> owned;
  int64_t weight_bytes = 0;
  KvLayout layout;
  PagePool pool;
  Half* kv_buffer = nullptr;
  std::vector<Request35> requests;
  // decode buffers (decode_buffers.rs:22-140)
  Half *hidden = nullptr, *hidden_mid = nullptr, *normed = nullptr, *attn_res = nullptr, *q_full = nullptr,
       *q_attn = nullptr, *k_attn = nullptr, *v_attn = nullptr, *attn_out = nullptr, *qkv = nullptr,
       *qkv_conv = nullptr, *z = nullptr, *b_proj = nullptr, *a_proj = nullptr, *gdr_out = nullptr,
       *normed_gated = nullptr, *gate_out = nullptr, *up_out = nullptr, *act_out = nullptr, *mlp_out = nullptr,
       *logits = nullptr, *wide = nullptr;  // wide: one token's stacked projection output (fused path)
  int32_t* tokens_out_d = nullptr;
  uint8_t* top1_state = nullptr;
  int32_t* tokens_out_host = nullptr;
  // per-request sampling / logprobs over the LAST step's logits (ops/sampling.rs:109-170, executor.rs:400-434)
  const Half* last_logits = nullptr;
  int last_rows = 0;
  float* probs_scratch = nullptr;
  Half* top1_value = nullptr;
  uint8_t *row_states = nullptr, *valid_scratch = nullptr;
  int32_t* sample_out_d = nullptr;
  std::vector<uint16_t> lp_bits;
  std::vector<float> lp_row;
  uint8_t *meta_host = nullptr, *meta_dev = nullptr;
  size_t m_tok, m_pos, m_indptr, m_lpl, m_ri, m_kti, m_kcs, m_sri, m_skt, m_skc, m_soi, m_sva, m_pages, m_total;
  // partition-KV decode for the full-attention layers (MI355X policy of kv_pool.h: fill the 256 CUs)
  Half* split_tmp_v = nullptr;
  float* split_tmp_s = nullptr;
  int32_t* merge_ctr = nullptr;  // in-launch split-KV merge tickets, zeroed at the head of every step
  int32_t* lin_ticket = nullptr; // pegainfer_linear_attn_decode_fused: one self-resetting word per key head (zero at allocation)
  hipGraphExec_t graph = nullptr;
  std::vector<int> graph_ids;   // request ids of the captured step, then the attention path
  std::vector<std::pair<std::vector<int>, hipGraphExec_t>> graph_cache;   // parked execs of other keys (<= 8)
  // debug tap (include/pegainfer_qwen35.h, accuracy-parity-playbook.md:15-24): the residual stream leaving every layer
  Half* tap = nullptr;
  bool tap_on = false;
  bool linattn_fused = [] { const char* e = getenv("PEGAINFER_LINATTN_FUSED"); return !(e && e[0] == '0'); }();   // read per model: tests build one of each
  int tap_rows = 0;
  // prefill workspace (grow-only)
  size_t pf_cap = 0;
  std::vector<void*> pf_owned;
  Half *pf_hidden = nullptr, *pf_mid = nullptr, *pf_normed = nullptr, *pf_attn_res = nullptr, *pf_big0 = nullptr,
       *pf_big1 = nullptr, *pf_big2 = nullptr, *pf_q_prep = nullptr, *pf_k = nullptr, *pf_v = nullptr, *pf_kc = nullptr,
       *pf_vc = nullptr, *pf_attn = nullptr, *pf_z = nullptr, *pf_b = nullptr, *pf_a = nullptr, *pf_gdr = nullptr;
  // GDR chunk-wise scratch (prefill_buffers.rs)
  float *g_cumsum = nullptr, *beta = nullptr, *a_tril = nullptr, *chunk_state = nullptr;
  Half *q_exp = nullptr, *k_exp = nullptr, *v_raw = nullptr, *a_inv = nullptr, *w = nullptr, *u = nullptr, *v_new = nullptr;
  uint8_t *pf_meta_dev = nullptr, *pf_meta_host = nullptr;
  size_t pf_meta_cap = 0;
  Half *pf_last = nullptr, *pf_last_normed = nullptr, *pf_logits = nullptr;
  int32_t* start_pos_d = nullptr;
  float last_step_ms = 0.f;

  Model35(int dev, int h, int inter, int l, int v, int hq, int hkv, int d, int kh_, int vh_, int ck, float e, float th,
          int rot, const int32_t* is_full, int mp, int pages, int mbs, int graph_, int split_)
      : device(dev), H(h), I(inter), L(l), V(v), Hq(hq), Hkv(hkv), D(d), kh(kh_), vh(vh_), convK(ck), rotary(rot),
        max_pos(mp), num_pages(pages), max_bs(mbs), enable_graph(graph_), split_policy(split_), eps(e), theta(th),
        q_dim(hq * d),
        kv_dim(hkv * d), C(2 * kh_ * LK + vh_ * LK), Z(vh_ * LK), layout(1, hkv, d, 16), pool(pages) {
    layers.resize(l);
    for (int i = 0; i < l; ++i) {
      layers[i].full = is_full[i] != 0;
      (layers[i].full ? n_full : n_lin)++;
    }
    layout = KvLayout(n_full > 0 ? n_full : 1, hkv, d, 16);
  }

  void set_error(const std::string& s) { err = s; }
  void* S() const { return reinterpret_cast<void*>(stream); }

  template <typename T>
  int dalloc(T** p, size_t count, bool zero = true, std::vector<void*>* owner = nullptr) {
    void* raw = nullptr;
    P35_HIP(hipMalloc(&raw, (count ? count : 1) * sizeof(T)));
    if (zero) P35_HIP(hipMemsetAsync(raw, 0, (count ? count : 1) * sizeof(T), stream));
    (owner ? owner : &owned)->push_back(raw);
    *p = static_cast<T*>(raw);
    return 0;
  }

  int tap_layer(int layer, const Half* src, int rows) {
    if (!tap_on) return 0;
    rows = std::min(rows, max_bs);
    P35_HIP(hipMemcpyAsync(tap + (size_t)layer * max_bs * H, src, (size_t)rows * H * 2, hipMemcpyDeviceToDevice, stream));
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
    P35_HIP(hipStreamSynchronize(stream));
    if (rows > 0) P35_HIP(hipMemcpy(host, tap + (size_t)layer * max_bs * H, (size_t)rows * H * 2, hipMemcpyDeviceToHost));
    return rows;
  }

  int init() {
    if (D != 256) { set_error("Qwen3.5 full attention needs head_dim 256"); return -1; }
    P35_HIP(hipSetDevice(device));
    cublas_init();
    P35_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    P35_HIP(hipEventCreate(&ev0));
    P35_HIP(hipEventCreate(&ev1));
    auto W = [&](Half** p, size_t n) { weight_bytes += (int64_t)n * 2; return dalloc(p, n, false); };
    auto WF = [&](float** p, size_t n) { weight_bytes += (int64_t)n * 4; return dalloc(p, n, false); };
    if (W(&embed, (size_t)V * H) || W(&final_norm, H)) return -1;
    for (auto& ly : layers) {
      // projections that read the same activation are stored row-stacked (gate|up, q|k|v, qkv|z|b|a) so the fused
      // decode path can stream them with ONE GEMV; the reference-order path uses the sub-matrices in place
      if (W(&ly.ln1, H) || W(&ly.ln2, H) || W(&ly.gate, (size_t)2 * I * H) || W(&ly.down, (size_t)H * I)) return -1;
      ly.up = ly.gate + (size_t)I * H;
      if (ly.full) {
        if (W(&ly.q_proj, (size_t)(2 * q_dim + 2 * kv_dim) * H) || W(&ly.o_proj, (size_t)H * q_dim) ||
            W(&ly.q_norm, D) || W(&ly.k_norm, D))
          return -1;
        ly.k_proj = ly.q_proj + (size_t)2 * q_dim * H;
        ly.v_proj = ly.k_proj + (size_t)kv_dim * H;
      } else {
        if (W(&ly.in_qkv, (size_t)(C + Z + 2 * vh) * H) || W(&ly.conv_w, (size_t)C * convK) || W(&ly.dt_bias, vh) ||
            W(&ly.out_proj, (size_t)H * Z) || WF(&ly.a_log, vh) || WF(&ly.norm_w, LK))
          return -1;
        ly.in_z = ly.in_qkv + (size_t)C * H;
        ly.in_b = ly.in_z + (size_t)Z * H;
        ly.in_a = ly.in_b + (size_t)vh * H;
      }
    }
    if (dalloc(&cos, (size_t)max_pos * rotary, false) || dalloc(&sin, (size_t)max_pos * rotary, false)) return -1;
    if (dalloc(&kv_buffer, (size_t)num_pages * layout.page_stride)) return -1;
    std::vector<int32_t> pad;
    if (!pool.acquire(1, &pad)) { set_error("pool must have at least 1 page"); return -1; }  // padding page 0
    const size_t bs = max_bs;
    if (dalloc(&hidden, bs * H) || dalloc(&hidden_mid, bs * H) || dalloc(&normed, bs * H) ||
        dalloc(&attn_res, bs * H) || dalloc(&q_full, bs * 2 * q_dim) || dalloc(&q_attn, bs * q_dim) ||
        dalloc(&k_attn, bs * kv_dim) || dalloc(&v_attn, bs * kv_dim) || dalloc(&attn_out, bs * q_dim) ||
        dalloc(&qkv, bs * C) || dalloc(&qkv_conv, bs * C) || dalloc(&z, bs * Z) || dalloc(&b_proj, bs * vh) ||
        dalloc(&a_proj, bs * vh) || dalloc(&gdr_out, bs * Z) || dalloc(&normed_gated, bs * Z) ||
        dalloc(&gate_out, bs * I) || dalloc(&up_out, bs * I) || dalloc(&act_out, bs * I) ||
        dalloc(&mlp_out, bs * H) || dalloc(&logits, bs * (size_t)V) ||
        dalloc(&top1_state, bs * 16) || dalloc(&start_pos_d, 1) || dalloc(&probs_scratch, (size_t)V) || dalloc(&top1_value, 1) ||
        dalloc(&row_states, 1024 * 1024) || dalloc(&valid_scratch, 1) || dalloc(&sample_out_d, 1) ||
        dalloc(&wide, (size_t)std::max(C + Z + 2 * vh, 2 * q_dim + 2 * kv_dim)) ||
        dalloc(&split_tmp_v, bs * pq::kSplitMaxChunksPerRequest * q_dim) ||
        dalloc(&split_tmp_s, bs * pq::kSplitMaxChunksPerRequest * Hq) || dalloc(&merge_ctr, bs * (size_t)Hkv * 32) || dalloc(&lin_ticket, (size_t)kh) || dalloc(&pf_last, H) ||
        dalloc(&pf_last_normed, H) || dalloc(&pf_logits, V))
      return -1;
    auto al = [](size_t x) { return (x + 63) & ~size_t(63); };
    size_t off = 0;
    m_tok = off; off = al(off + bs * 4);
    m_pos = off; off = al(off + bs * 4);
    m_indptr = off; off = al(off + (bs + 1) * 4);
    m_lpl = off; off = al(off + bs * 4);
    m_ri = off; off = al(off + bs * 4);
    m_kti = off; off = al(off + bs * 4);
    m_kcs = off; off = al(off + bs * 4);
    const size_t slots = bs * pq::kSplitMaxChunksPerRequest;
    m_sri = off; off = al(off + slots * 4);
    m_skt = off; off = al(off + slots * 4);
    m_skc = off; off = al(off + 4);
    m_soi = off; off = al(off + (bs + 1) * 4);
    m_sva = off; off = al(off + slots);
    m_pages = off; off = al(off + ((size_t)num_pages + bs) * 4);
    m_total = off;
    P35_HIP(hipHostMalloc(reinterpret_cast<void**>(&meta_host), m_total, hipHostMallocDefault));
    std::memset(meta_host, 0, m_total);
    if (dalloc(&meta_dev, m_total)) return -1;
    tokens_out_d = reinterpret_cast<int32_t*>(meta_dev + m_tok);   // greedy tokens land in the block's token slot (chained steps)
    P35_HIP(hipHostMalloc(reinterpret_cast<void**>(&tokens_out_host), bs * 4, hipHostMallocDefault));
    P35_HIP(hipStreamSynchronize(stream));
    return 0;
  }

  // ------------------------------------------------------------------ weights (weights.rs:118-296)
  // every tensor the forward pass reads, by destination pointer: load_tensor() marks, finalize() checks (the weights
  // are hipMalloc'ed uninitialised, so a checkpoint missing a tensor would otherwise produce silent garbage)
  std::vector<std::pair<const void*, std::string>> expected_tensors() const {
    std::vector<std::pair<const void*, std::string>> v;
    const std::string wp = "model.language_model.";
    v.push_back({embed, wp + "embed_tokens.weight"});
    v.push_back({final_norm, wp + "norm.weight"});
    for (int li = 0; li < L; ++li) {
      const Layer35& ly = layers[li];
      const std::string lp = wp + "layers." + std::to_string(li) + ".";
      v.push_back({ly.ln1, lp + "input_layernorm.weight"});
      v.push_back({ly.ln2, lp + "post_attention_layernorm.weight"});
      v.push_back({ly.gate, lp + "mlp.gate_proj.weight"});
      v.push_back({ly.up, lp + "mlp.up_proj.weight"});
      v.push_back({ly.down, lp + "mlp.down_proj.weight"});
      if (ly.full) {
        v.push_back({ly.q_proj, lp + "self_attn.q_proj.weight"});
        v.push_back({ly.k_proj, lp + "self_attn.k_proj.weight"});
        v.push_back({ly.v_proj, lp + "self_attn.v_proj.weight"});
        v.push_back({ly.o_proj, lp + "self_attn.o_proj.weight"});
        v.push_back({ly.q_norm, lp + "self_attn.q_norm.weight"});
        v.push_back({ly.k_norm, lp + "self_attn.k_norm.weight"});
      } else {
        v.push_back({ly.in_qkv, lp + "linear_attn.in_proj_qkv.weight"});
        v.push_back({ly.in_z, lp + "linear_attn.in_proj_z.weight"});
        v.push_back({ly.in_b, lp + "linear_attn.in_proj_b.weight"});
        v.push_back({ly.in_a, lp + "linear_attn.in_proj_a.weight"});
        v.push_back({ly.conv_w, lp + "linear_attn.conv1d.weight"});
        v.push_back({ly.dt_bias, lp + "linear_attn.dt_bias"});
        v.push_back({ly.a_log, lp + "linear_attn.A_log"});
        v.push_back({ly.norm_w, lp + "linear_attn.norm.weight"});
        v.push_back({ly.out_proj, lp + "linear_attn.out_proj.weight"});
      }
    }
    return v;
  }
  std::set<const void*> loaded;
  bool all_loaded = false;   // fill_synthetic() wrote every tensor
  int check_all_loaded() {
    if (all_loaded) return 0;
    for (const auto& e : expected_tensors())
      if (!loaded.count(e.first)) { set_error("checkpoint is missing tensor " + e.second); return -1; }
    return 0;
  }

  // shape / ndim given (native loader): a transposed or re-shaped tensor with the right element count is still wrong.
  // The depthwise conv weight is [C, 1, K] in HF checkpoints ([C, K] accepted too).
  // reference tensor name (weights.rs:102-296, prefix model.language_model.) -> device address, logical shape, dtype
  int resolve_tensor(const char* name_c, void** dst_out, int64_t* rows_out, int64_t* cols_out, bool* f32_out) {
    std::string name(name_c);
    const std::string wp = "model.language_model.";
    if (name.rfind(wp, 0) != 0) { set_error("tensor outside " + wp + ": " + name); return -1; }
    name = name.substr(wp.size());
    void* dst = nullptr;
    int64_t rows = 0, cols = 0;
    bool f32 = false;
    auto set = [&](void* p, int64_t r, int64_t c = 1, bool isf = false) { dst = p; rows = r; cols = c; f32 = isf; };
    if (name == "embed_tokens.weight") set(embed, V, H);
    else if (name == "norm.weight") set(final_norm, H);
    else if (name.rfind("layers.", 0) == 0) {
      const size_t p0 = 7, p1 = name.find('.', p0);
      if (p1 == std::string::npos || p1 == p0 || p1 - p0 > 6) { set_error("bad layer index in tensor name: " + name); return -1; }
      int li = 0;   // digits only: nothing may throw through the extern "C" boundary
      for (size_t i = p0; i < p1; ++i) {
        if (name[i] < '0' || name[i] > '9') { set_error("bad layer index in tensor name: " + name); return -1; }
        li = li * 10 + (name[i] - '0');
      }
      if (li < 0 || li >= L) { set_error("layer index out of range: " + name); return -1; }
      const std::string r = name.substr(p1 + 1);
      Layer35& ly = layers[li];
      if (r == "input_layernorm.weight") set(ly.ln1, H);
      else if (r == "post_attention_layernorm.weight") set(ly.ln2, H);
      else if (r == "mlp.gate_proj.weight") set(ly.gate, I, H);
      else if (r == "mlp.up_proj.weight") set(ly.up, I, H);
      else if (r == "mlp.down_proj.weight") set(ly.down, H, I);
      else if (ly.full) {
        if (r == "self_attn.q_proj.weight") set(ly.q_proj, (int64_t)2 * q_dim, H);
        else if (r == "self_attn.k_proj.weight") set(ly.k_proj, kv_dim, H);
        else if (r == "self_attn.v_proj.weight") set(ly.v_proj, kv_dim, H);
        else if (r == "self_attn.o_proj.weight") set(ly.o_proj, H, q_dim);
        else if (r == "self_attn.q_norm.weight") set(ly.q_norm, D);
        else if (r == "self_attn.k_norm.weight") set(ly.k_norm, D);
      } else {
        if (r == "linear_attn.in_proj_qkv.weight") set(ly.in_qkv, C, H);
        else if (r == "linear_attn.in_proj_z.weight") set(ly.in_z, Z, H);
        else if (r == "linear_attn.in_proj_b.weight") set(ly.in_b, vh, H);
        else if (r == "linear_attn.in_proj_a.weight") set(ly.in_a, vh, H);
        else if (r == "linear_attn.conv1d.weight") set(ly.conv_w, C, convK);
        else if (r == "linear_attn.dt_bias") set(ly.dt_bias, vh);
        else if (r == "linear_attn.A_log") set(ly.a_log, vh, 1, true);
        else if (r == "linear_attn.norm.weight") set(ly.norm_w, LK, 1, true);
        else if (r == "linear_attn.out_proj.weight") set(ly.out_proj, H, Z);
      }
    }
    if (!dst) { set_error("unknown tensor name: " + std::string(name_c)); return -1; }
    *dst_out = dst; *rows_out = rows; *cols_out = cols; *f32_out = f32;
    return 0;
  }
  // a loaded (or synthetic) tensor back on the host, bf16 bits or f32 as it is stored: hands the checkpoint the engine
  // computes with to a checker (bench.py's Qwen3.5 CPU leg)
  int export_tensor(const char* name_c, void* host, int64_t numel, int is_f32) {
    void* src = nullptr;
    int64_t rows = 0, cols = 0;
    bool f32 = false;
    if (resolve_tensor(name_c, &src, &rows, &cols, &f32)) return -1;
    if (numel != rows * cols || (is_f32 != 0) != f32) { set_error("shape/dtype mismatch for " + std::string(name_c)); return -1; }
    P35_HIP(hipStreamSynchronize(stream));
    P35_HIP(hipMemcpy(host, src, (size_t)numel * (f32 ? 4 : 2), hipMemcpyDeviceToHost));
    return 0;
  }
  int load_tensor(const char* name_c, const void* host, int64_t numel, int is_f32, const int64_t* shape = nullptr,
                  int ndim = 0) {
    void* dst = nullptr;
    int64_t rows = 0, cols = 0;
    bool f32 = false;
    if (resolve_tensor(name_c, &dst, &rows, &cols, &f32)) return -1;
    if (numel != rows * cols || (is_f32 != 0) != f32) { set_error("shape/dtype mismatch for " + std::string(name_c)); return -1; }
    if (shape) {
      bool ok;
      if (cols == 1) ok = ndim == 1 && shape[0] == rows;
      else if (ndim == 3) ok = shape[0] == rows && shape[1] == 1 && shape[2] == cols;   // conv1d [C, 1, K]
      else ok = ndim == 2 && shape[0] == rows && shape[1] == cols;
      if (!ok) { set_error("tensor " + std::string(name_c) + " has the wrong shape (expected [" + std::to_string(rows) +
                           (cols == 1 ? "]" : ", " + std::to_string(cols) + "]") + ")"); return -1; }
    }
    P35_HIP(hipMemcpy(dst, host, (size_t)numel * (f32 ? 4 : 2), hipMemcpyHostToDevice));
    loaded.insert(dst);
    return 0;
  }

  // native checkpoint load (weights.rs:102-296): every tensor under model.language_model., BF16 or F32
  int load_safetensors(const char* path) {
    pst::Checkpoint ck;
    std::string e;
    if (!ck.open(path, &e)) { set_error(e); return -1; }
    for (const auto& kv : ck.tensors()) {
      if (kv.first.rfind("model.language_model.", 0) != 0) continue;   // vision tower / mtp heads: not this path
      const pst::TensorView& t = kv.second;
      const bool f32 = t.dtype == "F32";
      if (!f32 && t.dtype != "BF16") { set_error("tensor " + kv.first + " has dtype " + t.dtype); return -1; }
      if ((int64_t)t.nbytes != t.numel() * (f32 ? 4 : 2)) { set_error("tensor " + kv.first + ": data_offsets do not match its shape"); return -1; }
      if (load_tensor(kv.first.c_str(), t.data, t.numel(), f32 ? 1 : 0, t.shape.data(), (int)t.shape.size())) return -1;
    }
    return finalize();
  }

  void fill(Half* p, size_t n, uint64_t seed, float std, float mean) {
    fill_normal35_kernel<Half><<<2048, 256, 0, stream>>>(p, (long)n, seed, std, mean);
  }
  void fillf(float* p, size_t n, uint64_t seed, float std, float mean) {
    fill_normal35_kernel<float><<<64, 256, 0, stream>>>(p, (long)n, seed, std, mean);
  }
  int fill_synthetic(uint64_t seed, float std) {
    all_loaded = true;
    uint64_t s = seed * 1000003ull;
    fill(embed, (size_t)V * H, ++s, std, 0.f);
    fill(final_norm, H, ++s, 0.1f, 0.f);
    for (auto& ly : layers) {
      fill(ly.ln1, H, ++s, 0.1f, 0.f);
      fill(ly.ln2, H, ++s, 0.1f, 0.f);
      fill(ly.gate, (size_t)I * H, ++s, std, 0.f);
      fill(ly.up, (size_t)I * H, ++s, std, 0.f);
      fill(ly.down, (size_t)H * I, ++s, std, 0.f);
      if (ly.full) {
        fill(ly.q_proj, (size_t)2 * q_dim * H, ++s, std, 0.f);
        fill(ly.k_proj, (size_t)kv_dim * H, ++s, std, 0.f);
        fill(ly.v_proj, (size_t)kv_dim * H, ++s, std, 0.f);
        fill(ly.o_proj, (size_t)H * q_dim, ++s, std, 0.f);
        fill(ly.q_norm, D, ++s, 0.1f, 0.f);
        fill(ly.k_norm, D, ++s, 0.1f, 0.f);
      } else {
        fill(ly.in_qkv, (size_t)C * H, ++s, std, 0.f);
        fill(ly.in_z, (size_t)Z * H, ++s, std, 0.f);
        fill(ly.in_b, (size_t)vh * H, ++s, std, 0.f);
        fill(ly.in_a, (size_t)vh * H, ++s, std, 0.f);
        fill(ly.conv_w, (size_t)C * convK, ++s, 0.3f, 0.f);
        fill(ly.dt_bias, vh, ++s, 0.5f, 0.f);
        fill(ly.out_proj, (size_t)H * Z, ++s, std, 0.f);
        fillf(ly.a_log, vh, ++s, 0.5f, 0.f);
        fillf(ly.norm_w, LK, ++s, 0.1f, 1.f);
      }
    }
    P35_HIP(hipStreamSynchronize(stream));
    return 0;
  }
  // partial-RoPE tables: rows of rotary_dim, cos/sin duplicated in both halves (weights.rs:296-297 ->
  // weight_loader.rs:210-244 with head_dim = rotary_dim)
  int finalize() {
    if (check_all_loaded()) return -1;
    const int half = rotary / 2;
    std::vector<float> inv(half);
    for (int i = 0; i < half; ++i) inv[i] = 1.0f / std::pow(theta, (float)i * 2.0f / (float)rotary);
    std::vector<uint16_t> c((size_t)max_pos * rotary), s((size_t)max_pos * rotary);
    for (int pos = 0; pos < max_pos; ++pos)
      for (int i = 0; i < half; ++i) {
        const float f = (float)pos * inv[i];
        const uint16_t cv = host_f2bf(std::cos(f)), sv = host_f2bf(std::sin(f));
        c[(size_t)pos * rotary + i] = c[(size_t)pos * rotary + i + half] = cv;
        s[(size_t)pos * rotary + i] = s[(size_t)pos * rotary + i + half] = sv;
      }
    P35_HIP(hipMemcpy(cos, c.data(), c.size() * 2, hipMemcpyHostToDevice));
    P35_HIP(hipMemcpy(sin, s.data(), s.size() * 2, hipMemcpyHostToDevice));
    return 0;
  }

  // ------------------------------------------------------------------ requests
  int new_request() {
    int id = -1;
    for (size_t i = 0; i < requests.size(); ++i)
      if (!requests[i].kv.live) { id = (int)i; break; }
    if (id < 0) { requests.emplace_back(); id = (int)requests.size() - 1; }
    Request35& r = requests[id];
    if (!r.allocated) {
      r.conv.resize(n_lin);
      r.state.resize(n_lin);
      for (int i = 0; i < n_lin; ++i)
        if (dalloc(&r.conv[i], (size_t)C * (convK - 1), false) || dalloc(&r.state[i], (size_t)vh * LK * LK, false)) return -1;
      r.allocated = true;
    }
    for (int i = 0; i < n_lin; ++i) {
      P35_HIP(hipMemsetAsync(r.conv[i], 0, (size_t)C * (convK - 1) * 2, stream));
      P35_HIP(hipMemsetAsync(r.state[i], 0, (size_t)vh * LK * LK * 4, stream));
    }
    r.kv = KvState();
    r.kv.live = true;
    return id;
  }
  Request35* req(int id) {
    if (id < 0 || id >= (int)requests.size() || !requests[id].kv.live) { set_error("bad request id"); return nullptr; }
    return &requests[id];
  }
  int drop_request(int id) {
    Request35* r = req(id);
    if (!r) return -1;
    r->kv.reset(&pool);
    r->kv.live = false;
    return 0;
  }

  void G(const Half* wt, const Half* x, Half* y, int M, int T, int K) {
    if (T == 1) gemm_graphsafe_cuda(wt, x, y, M, 1, K, S());
    else gemm_cuda(wt, x, y, M, T, K, S());
  }
  // stacked projections of one activation: one multi-output GEMM when prefill-sized, else the reference's calls
  int GS(const Half* wt, const Half* x, int n_out, Half* const* ys, const int32_t* ms, int T, int K) {
    if (T > 64) {
      if (pegainfer_gemm_split(wt, x, n_out, ys, ms, T, K, S())) { set_error("pegainfer_gemm_split failed"); return -1; }
      return 0;
    }
    size_t row = 0;
    for (int i = 0; i < n_out; ++i) { G(wt + row * K, x, ys[i], ms[i], T, K); row += ms[i]; }
    return 0;
  }
  int mlp(const Layer35& ly, const Half* x, Half* g, Half* up_, Half* act, Half* out, int T) {
    if (T > 64 && (I & 3) == 0) {  // gate|up stacked: SwiGLU (Qwen3.5 rounding) in the GEMM epilogue, same bits
      if (pegainfer_gemm_silu_rounded(ly.gate, x, act, nullptr, I, T, H, S()) == 0) {
        G(ly.down, act, out, H, T, I);
        return 0;
      }
    }
    Half* ys[2] = {g, up_};
    const int32_t ms[2] = {I, I};
    if (GS(ly.gate, x, 2, ys, ms, T, H)) return -1;
    if (silu_mul_triton_aot_cuda(g, up_, act, T * I, S())) { set_error("silu_mul failed"); return -1; }
    G(ly.down, act, out, H, T, I);
    return 0;
  }

  // ------------------------------------------------------------------ decode (batch_decode.rs:198-365)
  int decode_kernels(int bs, const std::vector<Request35*>& rs, bool split) {
    auto md = [&](size_t off) { return reinterpret_cast<int32_t*>(meta_dev + off); };
    // a kernel node, not a memset node (csrc/elementwise.hip: pegainfer_zero_words)
    if (split && pegainfer_zero_words(merge_ctr, (int32_t)((size_t)max_bs * Hkv * 32), S())) { set_error("pegainfer_zero_words failed"); return -1; }
    if (embedding_batched_cuda(embed, reinterpret_cast<uint32_t*>(meta_dev + m_tok), hidden, H, bs, S())) {
      set_error("embedding failed"); return -1;
    }
    const float sm = 1.0f / std::sqrt((float)D);
    int lin = 0, full = 0;
    for (const Layer35& ly : layers) {
      rms_norm_batched_offset_cuda(hidden, ly.ln1, normed, H, bs, eps, S());
      if (ly.full) {
        G(ly.q_proj, normed, q_full, 2 * q_dim, bs, H);
        G(ly.k_proj, normed, k_attn, kv_dim, bs, H);
        G(ly.v_proj, normed, v_attn, kv_dim, bs, H);
        qk_norm_partial_rope_batched_decode_hd256_cuda(q_full, k_attn, ly.q_norm, ly.k_norm, cos, sin, md(m_pos), q_attn,
                                                       Hq, Hkv, bs, rotary, eps, S());
        int rc = paged_kv_scatter_cuda(kv_buffer, layout.k_offset(full), layout.v_offset(full), md(m_pages), md(m_indptr),
                                       md(m_lpl), k_attn, v_attn, md(m_ri), md(m_pos), bs, Hkv, D, layout.page_size,
                                       layout.page_stride, kv_dim, D, S());
        if (!rc && split)
          rc = pegainfer_paged_attention_decode_split_kv_hd256(
              q_attn, attn_out, kv_buffer, layout.k_offset(full), layout.v_offset(full), md(m_pages), md(m_indptr),
              md(m_lpl), md(m_sri), md(m_skt), md(m_skc), md(m_soi), meta_dev + m_sva, split_tmp_v, split_tmp_s, Hq, Hkv,
              D, layout.page_size, bs, split_slots, layout.page_stride, sm, merge_ctr, S());
        else if (!rc)
          rc = paged_attention_decode_cuda_hd256(q_attn, attn_out, kv_buffer, layout.k_offset(full), layout.v_offset(full),
                                                 md(m_pages), md(m_indptr), md(m_lpl), md(m_ri), md(m_kti), md(m_kcs), Hq,
                                                 Hkv, D, layout.page_size, bs, layout.page_stride, sm, S());
        if (rc) { set_error("hd256 decode attention failed"); return -1; }
        attention_gate_batch_hd256_cuda(q_full, attn_out, Hq, bs, S());
        G(ly.o_proj, attn_out, attn_res, H, bs, q_dim);
        ++full;
      } else {
        G(ly.in_qkv, normed, qkv, C, bs, H);
        G(ly.in_z, normed, z, Z, bs, H);
        G(ly.in_b, normed, b_proj, vh, bs, H);
        G(ly.in_a, normed, a_proj, vh, bs, H);
        // per slot (batch_decode.rs:315-345); the reference copies each column out and back, the kernels take
        // the column pointers directly - same kernels, same bits
        for (int i = 0; i < bs; ++i) {
          conv1d_prefill_cuda(qkv + (size_t)i * C, ly.conv_w, rs[i]->conv[lin], qkv_conv + (size_t)i * C, C, 1, convK, S());
          gated_delta_rule_decode_cuda(qkv_conv + (size_t)i * C, b_proj + (size_t)i * vh, a_proj + (size_t)i * vh,
                                       ly.dt_bias, ly.a_log, rs[i]->state[lin], gdr_out + (size_t)i * Z, kh, vh, LK, LK, S());
        }
        rms_norm_gated_cuda(gdr_out, ly.norm_w, z, normed_gated, bs * vh, LK, eps, S());
        G(ly.out_proj, normed_gated, attn_res, H, bs, Z);
        ++lin;
      }
      if (add_cuda(hidden, attn_res, hidden_mid, bs * H, S())) { set_error("add failed"); return -1; }
      rms_norm_batched_offset_cuda(hidden_mid, ly.ln2, normed, H, bs, eps, S());
      if (mlp(ly, normed, gate_out, up_out, act_out, mlp_out, bs)) return -1;
      if (add_cuda(hidden_mid, mlp_out, hidden, bs * H, S())) { set_error("add failed"); return -1; }
      if (tap_layer(lin + full - 1, hidden, bs)) return -1;
    }
    rms_norm_batched_offset_cuda(hidden, final_norm, normed, H, bs, eps, S());
    G(embed, normed, logits, V, bs, H);
    if (pegainfer_batched_top1(logits, V, bs, V, top1_state, tokens_out_d, S())) { set_error("top1 failed"); return -1; }
    return 0;
  }

  // bs == 1 fused step (decode_mode 1): 8-10 launches per layer instead of 16-17, every result bit-identical to
  // decode_kernels().  The two norms and both residual adds ride in GEMV prologues (pegainfer_gemv_fused_ex with the
  // Qwen3.5 rounding points), SwiGLU in the gate|up GEMV epilogue, and the stacked qkv|z|b|a / q|k|v / gate|up
  // weights are streamed by one GEMV each.  Batches > 1 use the reference-order path (the consumers of the
  // stacked outputs take contiguous [bs, dim] tensors in the reference ABI).
  int decode_kernels_fused1(Request35* r, bool split) {
    auto md = [&](size_t off) { return reinterpret_cast<int32_t*>(meta_dev + off); };
    // a kernel node, not a memset node (csrc/elementwise.hip: pegainfer_zero_words)
    if (split && pegainfer_zero_words(merge_ctr, (int32_t)((size_t)max_bs * Hkv * 32), S())) { set_error("pegainfer_zero_words failed"); return -1; }
    if (embedding_batched_cuda(embed, reinterpret_cast<uint32_t*>(meta_dev + m_tok), hidden, H, 1, S())) {
      set_error("embedding failed"); return -1;
    }
    const float sm = 1.0f / std::sqrt((float)D);
    const int OFF = 1, RSUM = 2, SILU2 = 4;
    Half *cur = hidden, *nxt = hidden_mid;
    const Half* resid = nullptr;
    int lin = 0, full = 0, rc = 0;
    for (const Layer35& ly : layers) {
      if (ly.full) {
        rc = pegainfer_gemv_fused_ex(ly.q_proj, cur, wide, 2 * q_dim + 2 * kv_dim, 1, H, resid, ly.ln1,
                                     resid ? nxt : nullptr, eps, 0, OFF | (resid ? RSUM : 0), S());
        if (resid) std::swap(cur, nxt);
        if (resid && tap_layer(lin + full - 1, cur, 1)) return -1;   // cur = the previous layer's output
        Half *qf = wide, *kk = wide + 2 * q_dim, *vv = kk + kv_dim;   // [q|gate per head | k | v] of this token
        if (!rc) {
          qk_norm_partial_rope_batched_decode_hd256_cuda(qf, kk, ly.q_norm, ly.k_norm, cos, sin, md(m_pos), q_attn, Hq,
                                                         Hkv, 1, rotary, eps, S());
          rc = paged_kv_scatter_cuda(kv_buffer, layout.k_offset(full), layout.v_offset(full), md(m_pages), md(m_indptr),
                                     md(m_lpl), kk, vv, md(m_ri), md(m_pos), 1, Hkv, D, layout.page_size,
                                     layout.page_stride, kv_dim, D, S());
        }
        if (!rc && split)
          rc = pegainfer_paged_attention_decode_split_kv_hd256(
              q_attn, attn_out, kv_buffer, layout.k_offset(full), layout.v_offset(full), md(m_pages), md(m_indptr),
              md(m_lpl), md(m_sri), md(m_skt), md(m_skc), md(m_soi), meta_dev + m_sva, split_tmp_v, split_tmp_s, Hq, Hkv,
              D, layout.page_size, 1, split_slots, layout.page_stride, sm, merge_ctr, S());
        else if (!rc)
          rc = paged_attention_decode_cuda_hd256(q_attn, attn_out, kv_buffer, layout.k_offset(full), layout.v_offset(full),
                                                 md(m_pages), md(m_indptr), md(m_lpl), md(m_ri), md(m_kti), md(m_kcs), Hq,
                                                 Hkv, D, layout.page_size, 1, layout.page_stride, sm, S());
        if (rc) { set_error("fused full-attention layer failed"); return -1; }
        attention_gate_batch_hd256_cuda(qf, attn_out, Hq, 1, S());
        gemm_graphsafe_cuda(ly.o_proj, attn_out, attn_res, H, 1, q_dim, S());
        ++full;
      } else {
        rc = pegainfer_gemv_fused_ex(ly.in_qkv, cur, wide, C + Z + 2 * vh, 1, H, resid, ly.ln1, resid ? nxt : nullptr,
                                     eps, 0, OFF | (resid ? RSUM : 0), S());
        if (rc) { set_error("fused linear-attention projection failed"); return -1; }
        if (resid) std::swap(cur, nxt);
        if (resid && tap_layer(lin + full - 1, cur, 1)) return -1;
        // (round 6) conv step + gated delta rule + gated norm in ONE launch (one workgroup per key head, same bits:
        // csrc/qwen35.hip); PEGAINFER_LINATTN_FUSED=0, or a shape that kernel does not take, runs the three calls
        if (!linattn_fused ||
            pegainfer_linear_attn_decode_fused(wide, ly.conv_w, r->conv[lin], wide + C + Z, wide + C + Z + vh, ly.dt_bias,
                                               ly.a_log, r->state[lin], ly.norm_w, wide + C, normed_gated, kh, vh, LK, LK,
                                               convK, eps, lin_ticket, S()) != 0) {
          conv1d_prefill_cuda(wide, ly.conv_w, r->conv[lin], qkv_conv, C, 1, convK, S());
          gated_delta_rule_decode_cuda(qkv_conv, wide + C + Z, wide + C + Z + vh, ly.dt_bias, ly.a_log, r->state[lin],
                                       gdr_out, kh, vh, LK, LK, S());
          rms_norm_gated_cuda(gdr_out, ly.norm_w, wide + C, normed_gated, vh, LK, eps, S());
        }
        gemm_graphsafe_cuda(ly.out_proj, normed_gated, attn_res, H, 1, Z, S());
        ++lin;
      }
      // mid = bf16(cur + attn_res) -> nxt; act = silu_mul(gate, up) of norm_offset(mid)
      rc = pegainfer_gemv_fused_ex(ly.gate, cur, act_out, 2 * I, 1, H, attn_res, ly.ln2, nxt, eps, I, OFF | RSUM | SILU2, S());
      if (rc) { set_error("fused gate|up GEMV failed"); return -1; }
      std::swap(cur, nxt);
      gemm_graphsafe_cuda(ly.down, act_out, mlp_out, H, 1, I, S());
      resid = mlp_out;
    }
    rc = pegainfer_gemv_fused_ex(embed, cur, logits, V, 1, H, resid, final_norm, nxt, eps, 0, OFF | RSUM, S());
    if (!rc) rc = tap_layer(L - 1, nxt, 1);
    if (rc || pegainfer_batched_top1(logits, V, 1, V, top1_state, tokens_out_d, S())) { set_error("fused lm_head failed"); return -1; }
    return 0;
  }
  // bench.py roofline: the dominant kernel of the fused bs = 1 step - the gate|up GEMV with the residual add + (1 + w)
  // RMSNorm prologue and the SwiGLU epilogue - launched `iters` times over the layers' real weights (each launch streams a
  // different 94 MB matrix, as in a decode step), hipEvents on the model stream.  which 0 = that kernel, 1 = the plain
  // down_proj GEMV.  Returns ms per launch (< 0 on error).
  float bench_gemv(int which, int iters) {
    if (iters <= 0 || (H & 7)) return -1.f;
    const int OFF = 1, RSUM = 2, SILU2 = 4;
    auto launch = [&](int it) {
      const Layer35& ly = layers[(size_t)it % layers.size()];
      if (which == 0) (void)pegainfer_gemv_fused_ex(ly.gate, hidden, act_out, 2 * I, 1, H, attn_res, ly.ln2, hidden_mid, eps, I, OFF | RSUM | SILU2, S());
      else gemm_graphsafe_cuda(ly.down, act_out, mlp_out, H, 1, I, S());
    };
    for (int i = 0; i < 3; ++i) launch(i);
    if (hipEventRecord(ev0, stream) != hipSuccess) return -1.f;
    for (int i = 0; i < iters; ++i) launch(i + 3);
    if (hipEventRecord(ev1, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    return ms / iters;
  }
  int run_decode_kernels(int n, const std::vector<Request35*>& rs, bool split) {
    if (decode_mode == 1 && n == 1 && (H & 7) == 0) return decode_kernels_fused1(rs[0], split);
    return decode_kernels(n, rs, split);
  }

  // One decode step = step_prepare (validate, advance the KvStates, pack + upload the metadata) + step_launch (replay or
  // capture the graph keyed by the request ids).  toks == nullptr: a CHAINED step - the previous step's greedy tokens already
  // sit in the device block's token slot (tokens_out_d aliases it), the upload starts behind it.
  struct StepCtx35 { int n = 0; bool split = false; std::vector<Request35*> rs; std::vector<int> key; };
  int step_prepare(int n, const int32_t* ids, const uint32_t* toks, StepCtx35* c, uint8_t* host_block) {
    if (n < 1 || n > max_bs) { set_error("decode batch size out of range"); return -1; }
    std::vector<Request35*>& rs = c->rs;
    rs.assign(n, nullptr);
    c->key.assign(ids, ids + n);
    for (int i = 0; i < n; ++i) {
      rs[i] = req(ids[i]);
      if (!rs[i]) return -1;
      for (int j = 0; j < i; ++j)
        if (ids[j] == ids[i]) { set_error("duplicate request id in decode batch"); return -1; }
    }
    auto mh = [&](size_t off) { return reinterpret_cast<int32_t*>(host_block + off); };
    // validate the whole batch before any KvState is advanced: a failed call leaves every request untouched
    int pages_short = 0;
    for (int i = 0; i < n; ++i) {
      if (rs[i]->kv.seq_len + 1 > max_pos) { set_error("position beyond the RoPE table"); return -1; }
      pages_short += rs[i]->kv.pages_short(rs[i]->kv.seq_len + 1, layout.page_size);
    }
    if (pages_short > pool.available()) { set_error("KV pool exhausted"); return -1; }
    int np = 0;
    mh(m_indptr)[0] = 0;
    for (int i = 0; i < n; ++i) {
      KvState& kv = rs[i]->kv;
      const int pos = kv.seq_len;
      if (!kv.ensure_capacity(&pool, pos + 1, layout.page_size)) { set_error("KV pool exhausted"); return -1; }
      kv.seq_len += 1;
      mh(m_tok)[i] = toks ? (int32_t)toks[i] : 0;
      mh(m_pos)[i] = pos;
      for (int32_t p : kv.pages) mh(m_pages)[np++] = p;
      mh(m_indptr)[i + 1] = np;
      mh(m_lpl)[i] = kv.last_page_len(layout.page_size);
      mh(m_ri)[i] = i;
      mh(m_kti)[i] = 0;
      mh(m_kcs)[i] = kv.seq_len;
    }
    // split-KV plan for the full-attention layers (split_policy 0 = the reference's non-partition call only)
    const bool allow_split = split_policy != 0;
    std::vector<int> seq_lens(n);
    for (int i = 0; i < n; ++i) seq_lens[i] = rs[i]->kv.seq_len;
    // (a lone request keeps the 18-chunk plan it has had since round 3 - 128-token chunks from 1152 tokens on, one tile per
    // wave of the 8-wave workgroups; batches take the un-capped plan)
    const pq::SplitPlan plan = pq::make_split_plan(1, seq_lens, n, Hkv, n == 1);
    const bool split = allow_split && plan.use_split;
    // the launch covers exactly the slots refreshed below: plan.slots = n * (chunks per request), a function of n and
    // Hkv only, so it is constant for a captured graph (the key holds the request ids).  Launching n * 64 slots would
    // run slots whose request / tile / valid entries are left over from a step with a different batch size; such a
    // stale slot naming a live request bumps its merge counter and can fire the in-launch merge early.
    split_slots = plan.slots;
    std::memcpy(mh(m_sri), plan.request_indices.data(), (size_t)plan.slots * 4);
    std::memcpy(mh(m_skt), plan.kv_tile_indices.data(), (size_t)plan.slots * 4);
    mh(m_skc)[0] = plan.chunk;
    std::memcpy(mh(m_soi), plan.o_indptr.data(), (size_t)(n + 1) * 4);
    std::memcpy(host_block + m_sva, plan.valid.data(), (size_t)plan.slots);
    c->key.push_back(split ? 1 : 0);
    c->key.push_back(decode_mode);
    const size_t from = toks ? 0 : m_pos;   // m_tok is the first field of the block
    P35_HIP(hipMemcpyAsync(meta_dev + from, host_block + from, m_total - from, hipMemcpyHostToDevice, stream));
    c->n = n; c->split = split;
    return 0;
  }
  int step_launch(const StepCtx35& c) {
    if (enable_graph && !tap_on) {
      if (!graph || graph_ids != c.key) {
        // a small cache of execs per key: inside a greedy chain the key flips when the attention path toggles, and
        // destroying an exec whose earlier replays are still in flight on the stream is not known to be safe on ROCm 7.2.  The
        // current exec is parked, a parked one with the wanted key is revived; when the cache is full the stream is drained
        // before the oldest exec is destroyed.
        if (graph) { graph_cache.emplace_back(graph_ids, graph); graph = nullptr; }
        for (size_t i = 0; i < graph_cache.size(); ++i)
          if (graph_cache[i].first == c.key) {
            graph = graph_cache[i].second;
            graph_ids = c.key;
            graph_cache.erase(graph_cache.begin() + i);
            break;
          }
        if (graph) { P35_HIP(hipGraphLaunch(graph, stream)); return 0; }
        if (graph_cache.size() >= 8) {
          P35_HIP(hipStreamSynchronize(stream));
          P35_HIP(hipGraphExecDestroy(graph_cache.front().second));
          graph_cache.erase(graph_cache.begin());
        }
        hipGraph_t g = nullptr;
        P35_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_decode_kernels(c.n, c.rs, c.split);
        hipError_t e = hipStreamEndCapture(stream, &g);
        if (rc || e != hipSuccess) { if (!rc) set_error("graph capture failed"); return -1; }
        P35_HIP(hipGraphInstantiate(&graph, g, nullptr, nullptr, 0));
        P35_HIP(hipGraphDestroy(g));
        graph_ids = c.key;
      }
      P35_HIP(hipGraphLaunch(graph, stream));
      return 0;
    }
    return run_decode_kernels(c.n, c.rs, c.split) ? -1 : 0;
  }
  int decode(int n, const int32_t* ids, const uint32_t* toks, int32_t* out_tokens, void* out_logits_host) {
    StepCtx35 c;
    if (const int rc = step_prepare(n, ids, toks, &c, meta_host)) return rc;
    P35_HIP(hipEventRecord(ev0, stream));
    if (step_launch(c)) return -1;
    P35_HIP(hipEventRecord(ev1, stream));
    P35_HIP(hipMemcpyAsync(tokens_out_host, tokens_out_d, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    if (out_logits_host)
      P35_HIP(hipMemcpyAsync(out_logits_host, logits, (size_t)n * V * 2, hipMemcpyDeviceToHost, stream));
    P35_HIP(hipStreamSynchronize(stream));
    P35_HIP(hipEventElapsedTime(&last_step_ms, ev0, ev1));
    if (out_tokens) std::memcpy(out_tokens, tokens_out_host, (size_t)n * 4);
    last_logits = logits;
    last_rows = n;
    return 0;
  }
  // n_steps GREEDY decode steps enqueued back to back, one host synchronisation (the twin of
  // pegainfer_qwen3_decode_greedy_chain; include/pegainfer_qwen35.h).  The recurrent state and the KV append advance on the
  // device exactly as in n_steps calls of decode(); the chain is validated as a whole before any request advances.
  static constexpr int kChainRing = 8;
  uint8_t* chain_ring[kChainRing] = {};
  hipEvent_t chain_ev[kChainRing] = {};
  int32_t* chain_tokens_host = nullptr;
  size_t chain_tokens_cap = 0;
  int decode_greedy_chain(int n, const int32_t* ids, const uint32_t* first_tokens, int n_steps, int32_t* out_tokens) {
    if (n_steps <= 0) { set_error("decode_greedy_chain: n_steps must be positive"); return -1; }
    if (n < 1 || n > max_bs) { set_error("decode batch size out of range"); return -1; }   // before first_tokens is touched
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j)
        if (ids[i] == ids[j]) { set_error("decode_greedy_chain: duplicate request id"); return -1; }
    if (tap_on) {
      std::vector<uint32_t> tk(first_tokens, first_tokens + n);
      for (int s = 0; s < n_steps; ++s) {
        if (const int rc = decode(n, ids, tk.data(), out_tokens + (size_t)s * n, nullptr)) return rc;
        for (int i = 0; i < n; ++i) tk[i] = (uint32_t)out_tokens[(size_t)s * n + i];
      }
      return 0;
    }
    if (!chain_ring[0])
      for (int r = 0; r < kChainRing; ++r) {
        P35_HIP(hipHostMalloc(reinterpret_cast<void**>(&chain_ring[r]), m_total, hipHostMallocDefault));
        std::memset(chain_ring[r], 0, m_total);
        P35_HIP(hipEventCreateWithFlags(&chain_ev[r], hipEventDisableTiming));
      }
    if ((size_t)n_steps * n > chain_tokens_cap) {
      if (chain_tokens_host) P35_HIP(hipHostFree(chain_tokens_host));
      chain_tokens_cap = std::max<size_t>((size_t)n_steps * n, 4096);   // grow-only, never inside a short timed chain
      P35_HIP(hipHostMalloc(reinterpret_cast<void**>(&chain_tokens_host), chain_tokens_cap * 4, hipHostMallocDefault));
    }
    if (n < 1 || n > max_bs) { set_error("decode batch size out of range"); return -1; }
    {
      int pages_short = 0;
      for (int i = 0; i < n; ++i) {
        Request35* r = req(ids[i]);
        if (!r) return -1;
        if (r->kv.seq_len + n_steps > max_pos) { set_error("position beyond the RoPE table"); return -1; }
        pages_short += r->kv.pages_short(r->kv.seq_len + n_steps, layout.page_size);
      }
      if (pages_short > pool.available()) { set_error("KV pool exhausted"); return -1; }
    }
    P35_HIP(hipEventRecord(ev0, stream));
    for (int s = 0; s < n_steps; ++s) {
      const int slot = s % kChainRing;
      if (s >= kChainRing) P35_HIP(hipEventSynchronize(chain_ev[slot]));
      StepCtx35 c;
      // a failure at step s > 0 leaves steps 0 .. s - 1 in flight; the recurrent state they advanced cannot be rolled back, so
      // they are drained, their tokens are handed out, and the error says how far the requests have moved
      auto abort_chain = [&](int rc) {
        (void)hipStreamSynchronize(stream);
        if (s > 0) std::memcpy(out_tokens, chain_tokens_host, (size_t)s * n * 4);
        set_error(err + " (decode_greedy_chain: " + std::to_string(s) + " of " + std::to_string(n_steps) + " steps completed; the requests stand there)");
        return rc;
      };
      if (const int rc = step_prepare(n, ids, s == 0 ? first_tokens : nullptr, &c, chain_ring[slot])) return abort_chain(rc);
      P35_HIP(hipEventRecord(chain_ev[slot], stream));
      if (step_launch(c)) return abort_chain(-1);
      P35_HIP(hipMemcpyAsync(chain_tokens_host + (size_t)s * n, tokens_out_d, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    }
    P35_HIP(hipEventRecord(ev1, stream));
    P35_HIP(hipStreamSynchronize(stream));
    P35_HIP(hipEventElapsedTime(&last_step_ms, ev0, ev1));
    last_step_ms /= (float)n_steps;
    std::memcpy(out_tokens, chain_tokens_host, (size_t)n_steps * n * 4);
    last_logits = logits;
    last_rows = n;
    return 0;
  }
  // gpu_sample rule (ops/sampling.rs:109-170) on one row of the last step's logits - the same symbols the Qwen3 runtime calls
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
    P35_HIP(hipStreamSynchronize(stream));
    P35_HIP(hipMemcpy(out, sample_out_d, 4, hipMemcpyDeviceToHost));
    return 0;
  }
  // extract_logprobs (executor.rs:807-816 of the qwen3 crate; the qwen35 crate shares the engine types): row -> host f32 ->
  // compute_logprobs_from_cpu
  int logprobs(int column, uint32_t token, int top_k, float* out_lp, uint32_t* top_ids, float* top_lps) {
    if (!last_logits || column < 0 || column >= last_rows) { set_error("no logits for that column"); return -1; }
    if (token >= (uint32_t)V) { set_error("logprobs: token outside the vocabulary"); return -1; }
    lp_bits.resize(V);
    lp_row.resize(V);
    P35_HIP(hipStreamSynchronize(stream));
    P35_HIP(hipMemcpy(lp_bits.data(), last_logits + (size_t)column * V, (size_t)V * 2, hipMemcpyDeviceToHost));
    for (int i = 0; i < V; ++i) {
      const uint32_t u = (uint32_t)lp_bits[i] << 16;
      std::memcpy(&lp_row[i], &u, 4);
    }
    return pegainfer_logprobs_from_logits(lp_row.data(), V, token, top_k, out_lp, top_ids, top_lps);
  }

  // ------------------------------------------------------------------ prefill (prefill.rs:21-449)
  int ensure_prefill_ws(size_t T) {
    if (T <= pf_cap) return 0;
    P35_HIP(hipStreamSynchronize(stream));
    for (void* p : pf_owned) P35_HIP(hipFree(p));
    pf_owned.clear();
    const size_t cap = std::max<size_t>(T, 64);
    const size_t big = std::max<size_t>({(size_t)2 * q_dim, (size_t)C, (size_t)I});
    const size_t nch = (cap + 63) / 64;
    auto A = [&](auto** p, size_t n) { return dalloc(p, n, true, &pf_owned); };
    if (A(&pf_hidden, cap * H) || A(&pf_mid, cap * H) || A(&pf_normed, cap * H) || A(&pf_attn_res, cap * H) ||
        A(&pf_big0, cap * big) || A(&pf_big1, cap * big) || A(&pf_big2, cap * big) || A(&pf_q_prep, cap * q_dim) ||
        A(&pf_k, cap * kv_dim) || A(&pf_v, cap * kv_dim) || A(&pf_kc, cap * kv_dim) || A(&pf_vc, cap * kv_dim) ||
        A(&pf_attn, cap * q_dim) || A(&pf_z, cap * Z) || A(&pf_b, cap * vh) || A(&pf_a, cap * vh) ||
        A(&pf_gdr, cap * Z) || A(&g_cumsum, cap * vh) || A(&beta, cap * vh) || A(&a_tril, cap * vh * 64) ||
        A(&chunk_state, nch * vh * LK * LK) || A(&q_exp, cap * Z) || A(&k_exp, cap * Z) || A(&v_raw, cap * Z) ||
        A(&a_inv, cap * vh * 64) || A(&w, cap * Z) || A(&u, cap * Z) || A(&v_new, cap * Z))
      return -1;
    pf_cap = cap;
    return 0;
  }

  int prefill(int id, int T, const uint32_t* tokens, int32_t* out_token, void* out_logits_host) {
    Request35* r = req(id);
    if (!r) return -1;
    if (T < 1) { set_error("prefill needs at least one token"); return -1; }
    KvState& kv = r->kv;
    const int base = kv.seq_len;
    if (base + T > max_pos) { set_error("prompt beyond the RoPE table"); return -1; }
    if (!kv.ensure_capacity(&pool, base + T, layout.page_size)) { set_error("KV pool exhausted"); return -1; }
    kv.seq_len += T;
    if (ensure_prefill_ws(T)) return -1;
    // PrefillPagedPlan::new (ops/attention.rs:17-130), one request
    const int group = Hq / Hkv;
    const int cta = batch_prefill_cta_tile_q(T, Hq, Hkv, D);
    const int tiles = (T * group + cta - 1) / cta;
    std::vector<int32_t> indptr{0, (int32_t)kv.pages.size()}, lpl{kv.last_page_len(layout.page_size)}, kcs{base + T},
        bidx(T, 0), pos(T), qind{0, T}, rq(tiles, 0), qt(tiles), kt(tiles, 0);
    for (int t = 0; t < T; ++t) pos[t] = base + t;
    for (int t = 0; t < tiles; ++t) qt[t] = t;
    uint32_t total_rows = (uint32_t)T;
    int32_t start_pos = base;
    std::vector<std::pair<const void*, size_t>> parts = {
        {tokens, (size_t)T * 4}, {kv.pages.data(), kv.pages.size() * 4}, {indptr.data(), 8}, {lpl.data(), 4},
        {bidx.data(), (size_t)T * 4}, {pos.data(), (size_t)T * 4}, {qind.data(), 8}, {rq.data(), rq.size() * 4},
        {qt.data(), qt.size() * 4}, {kt.data(), kt.size() * 4}, {kcs.data(), 4}, {&total_rows, 4}, {&start_pos, 4}};
    std::vector<size_t> offs;
    size_t off = 0;
    for (auto& p : parts) { offs.push_back(off); off = (off + p.second + 63) & ~size_t(63); }
    if (off > pf_meta_cap) {
      P35_HIP(hipStreamSynchronize(stream));
      if (pf_meta_dev) { P35_HIP(hipFree(pf_meta_dev)); P35_HIP(hipHostFree(pf_meta_host)); }
      pf_meta_cap = off * 2;
      P35_HIP(hipMalloc(reinterpret_cast<void**>(&pf_meta_dev), pf_meta_cap));
      P35_HIP(hipHostMalloc(reinterpret_cast<void**>(&pf_meta_host), pf_meta_cap, hipHostMallocDefault));
    }
    for (size_t i = 0; i < parts.size(); ++i) std::memcpy(pf_meta_host + offs[i], parts[i].first, parts[i].second);
    P35_HIP(hipMemcpyAsync(pf_meta_dev, pf_meta_host, off, hipMemcpyHostToDevice, stream));
    auto D32 = [&](int i) { return reinterpret_cast<int32_t*>(pf_meta_dev + offs[i]); };

    if (embedding_batched_cuda(embed, reinterpret_cast<uint32_t*>(pf_meta_dev + offs[0]), pf_hidden, H, T, S())) {
      set_error("embedding failed"); return -1;
    }
    Half *hid = pf_hidden, *alt = pf_mid;  // hidden_batch / hidden_plus_attn ping-pong
    const float sm = 1.0f / std::sqrt((float)D);
    int lin = 0, full = 0;
    for (const Layer35& ly : layers) {
      rms_norm_batched_offset_cuda(hid, ly.ln1, pf_normed, H, T, eps, S());
      if (ly.full) {
        Half* qf = pf_big0;  // q_full_batch [T, 2*q_dim]
        {
          Half* ys[3] = {qf, pf_k, pf_v};
          const int32_t ms[3] = {2 * q_dim, kv_dim, kv_dim};
          if (GS(ly.q_proj, pf_normed, 3, ys, ms, T, H)) return -1;
        }
        // prep writes the processed K/V into an HND buffer [Hkv][max_seq][256] at rows start_pos + t; here the
        // buffer holds only this call's T rows, so its origin is shifted back by start_pos rows
        Half* kc0 = pf_kc - (size_t)base * D;
        Half* vc0 = pf_vc - (size_t)base * D;
        prefill_attention_hd256_prep_cuda(qf, pf_k, pf_v, ly.q_norm, ly.k_norm, cos, sin, pf_q_prep, kc0, vc0, Hq, Hkv, T,
                                          D32(12), rotary, eps, T, S());
        int rc = paged_kv_scatter_cuda(kv_buffer, layout.k_offset(full), layout.v_offset(full), D32(1), D32(2), D32(3),
                                       pf_kc, pf_vc, D32(4), D32(5), T, Hkv, D, layout.page_size, layout.page_stride, D,
                                       (int64_t)T * D, S());
        if (!rc)
          rc = batch_prefill_paged_cuda_hd256(pf_q_prep, pf_attn, kv_buffer, layout.k_offset(full), layout.v_offset(full),
                                              D32(1), D32(2), D32(3), D32(6), D32(7), D32(8), D32(9), D32(10),
                                              reinterpret_cast<uint32_t*>(pf_meta_dev + offs[11]), Hq, Hkv, D,
                                              layout.page_size, T, 1, tiles, layout.page_stride, sm, S());
        if (rc) { set_error("hd256 prefill attention failed"); return -1; }
        attention_gate_batch_hd256_cuda(qf, pf_attn, Hq, T, S());
        G(ly.o_proj, pf_attn, pf_attn_res, H, T, q_dim);
        ++full;
      } else {
        Half *qkv_b = pf_big0, *qkv_c = pf_big1;
        {
          Half* ys[4] = {qkv_b, pf_z, pf_b, pf_a};
          const int32_t ms[4] = {C, Z, vh, vh};
          if (GS(ly.in_qkv, pf_normed, 4, ys, ms, T, H)) return -1;
        }
        conv1d_prefill_cuda(qkv_b, ly.conv_w, r->conv[lin], qkv_c, C, T, convK, S());
        // gated_delta_rule_prefill_chunkwise_into (recurrent.rs:368-470)
        float* st = r->state[lin];
        int rc = gated_delta_rule_prefill_chunk_prepare_cuda(qkv_c, pf_b, pf_a, ly.dt_bias, ly.a_log, q_exp, k_exp, v_raw,
                                                             g_cumsum, beta, kh, vh, C, T, S());
        if (!rc) rc = gated_delta_rule_prefill_chunk_cumsum_cuda(g_cumsum, g_cumsum, T, vh, S());
        if (!rc) rc = gated_delta_rule_prefill_chunk_a_cuda(k_exp, g_cumsum, beta, a_tril, T, vh, S());
        if (!rc) rc = gated_delta_rule_prefill_chunk_solve_cuda(a_tril, a_inv, T, vh, S());
        if (!rc) rc = gated_delta_rule_prefill_chunk_recompute_cuda(k_exp, v_raw, beta, w, u, a_inv, g_cumsum, T, vh, S());
        if (!rc) rc = gated_delta_rule_prefill_chunk_state_cuda(k_exp, w, u, g_cumsum, st, chunk_state, v_new, st, T, vh, S());
        if (!rc) rc = gated_delta_rule_prefill_chunk_o_cuda(q_exp, k_exp, v_new, chunk_state, g_cumsum, pf_gdr, T, vh,
                                                            1.0f / std::sqrt((float)LK), S());
        if (rc) { set_error("chunk-wise gated delta rule failed"); return -1; }
        rms_norm_gated_cuda(pf_gdr, ly.norm_w, pf_z, qkv_b /* normed_out */, T * vh, LK, eps, S());
        G(ly.out_proj, qkv_b, pf_attn_res, H, T, Z);
        ++lin;
      }
      if (add_cuda(hid, pf_attn_res, alt, T * H, S())) { set_error("add failed"); return -1; }
      rms_norm_batched_offset_cuda(alt, ly.ln2, pf_normed, H, T, eps, S());
      if (mlp(ly, pf_normed, pf_big0, pf_big1, pf_big2, pf_attn_res, T)) return -1;
      if (add_cuda(alt, pf_attn_res, hid, T * H, S())) { set_error("add failed"); return -1; }
      if (tap_layer(lin + full - 1, hid + (size_t)(T - 1) * H, 1)) return -1;
    }
    P35_HIP(hipMemcpyAsync(pf_last, hid + (size_t)(T - 1) * H, (size_t)H * 2, hipMemcpyDeviceToDevice, stream));
    rms_norm_offset_cuda(pf_last, final_norm, pf_last_normed, H, eps, S());
    gemm_graphsafe_cuda(embed, pf_last_normed, pf_logits, V, 1, H, S());
    if (pegainfer_batched_top1(pf_logits, V, 1, V, top1_state, tokens_out_d, S())) { set_error("top1 failed"); return -1; }
    P35_HIP(hipMemcpyAsync(tokens_out_host, tokens_out_d, 4, hipMemcpyDeviceToHost, stream));
    if (out_logits_host) P35_HIP(hipMemcpyAsync(out_logits_host, pf_logits, (size_t)V * 2, hipMemcpyDeviceToHost, stream));
    P35_HIP(hipStreamSynchronize(stream));
    if (out_token) *out_token = tokens_out_host[0];
    last_logits = pf_logits;
    last_rows = 1;
    return 0;
  }

  ~Model35() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (graph) (void)hipGraphExecDestroy(graph);
    for (auto& ge : graph_cache) (void)hipGraphExecDestroy(ge.second);
    for (void* p : owned) (void)hipFree(p);
    for (void* p : pf_owned) (void)hipFree(p);
    if (pf_meta_dev) (void)hipFree(pf_meta_dev);
    if (pf_meta_host) (void)hipHostFree(pf_meta_host);
    if (meta_host) (void)hipHostFree(meta_host);
    if (tokens_out_host) (void)hipHostFree(tokens_out_host);
    for (int r = 0; r < kChainRing; ++r) {
      if (chain_ring[r]) (void)hipHostFree(chain_ring[r]);
      if (chain_ev[r]) (void)hipEventDestroy(chain_ev[r]);
    }
    if (chain_tokens_host) (void)hipHostFree(chain_tokens_host);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

}  // namespace pq35

using pq35::Model35;
static Model35* M35(pegainfer_qwen35_t m) { return static_cast<Model35*>(m); }

extern "C" {

pegainfer_qwen35_t pegainfer_qwen35_create(int32_t device_ordinal, int32_t hidden_size, int32_t intermediate_size,
                                           int32_t num_layers, int32_t vocab_size, int32_t num_attention_heads,
                                           int32_t num_kv_heads, int32_t head_dim, int32_t linear_num_key_heads,
                                           int32_t linear_num_value_heads, int32_t linear_conv_kernel_dim,
                                           float rms_norm_eps, float rope_theta, int32_t rotary_dim,
                                           const int32_t* layer_is_full, int32_t max_position_embeddings,
                                           int32_t num_kv_pages, int32_t max_batch_size, int32_t enable_graph,
                                           int32_t split_policy) {
  if (!layer_is_full || num_layers < 1 || num_kv_pages < 2 || max_batch_size < 1) return nullptr;
  Model35* m = new Model35(device_ordinal, hidden_size, intermediate_size, num_layers, vocab_size, num_attention_heads,
                           num_kv_heads, head_dim, linear_num_key_heads, linear_num_value_heads, linear_conv_kernel_dim,
                           rms_norm_eps, rope_theta, rotary_dim, layer_is_full, max_position_embeddings, num_kv_pages,
                           max_batch_size, enable_graph, split_policy);
  if (m->init()) {
    fprintf(stderr, "pegainfer_qwen35_create: %s\n", m->err.c_str());
    delete m;
    return nullptr;
  }
  return m;
}
void pegainfer_qwen35_destroy(pegainfer_qwen35_t m) { delete M35(m); }
const char* pegainfer_qwen35_last_error(pegainfer_qwen35_t m) { return M35(m)->err.c_str(); }
int32_t pegainfer_qwen35_load_tensor(pegainfer_qwen35_t m, const char* name, const void* host, int64_t numel,
                                     int32_t is_f32) {
  return M35(m)->load_tensor(name, host, numel, is_f32);
}
int32_t pegainfer_qwen35_vocab_size(pegainfer_qwen35_t m) { return M35(m)->V; }
int32_t pegainfer_qwen35_sample(pegainfer_qwen35_t m, int32_t column, float temperature, int32_t top_k, float top_p,
                                float random_val, int32_t* out_token) {
  return M35(m)->sample(column, temperature, top_k, top_p, random_val, out_token);
}
int32_t pegainfer_qwen35_logprobs(pegainfer_qwen35_t m, int32_t column, uint32_t token, int32_t top_k, float* out_logprob,
                                  uint32_t* out_top_ids, float* out_top_logprobs) {
  return M35(m)->logprobs(column, token, top_k, out_logprob, out_top_ids, out_top_logprobs);
}
int32_t pegainfer_qwen35_export_tensor(pegainfer_qwen35_t m, const char* name, void* host, int64_t numel, int32_t is_f32) {
  return M35(m)->export_tensor(name, host, numel, is_f32);
}
int32_t pegainfer_qwen35_fill_synthetic(pegainfer_qwen35_t m, uint64_t seed, float std) { return M35(m)->fill_synthetic(seed, std); }
int32_t pegainfer_qwen35_finalize(pegainfer_qwen35_t m) { return M35(m)->finalize(); }
int32_t pegainfer_qwen35_load_safetensors(pegainfer_qwen35_t m, const char* path) { return M35(m)->load_safetensors(path); }
int32_t pegainfer_qwen35_new_request(pegainfer_qwen35_t m) { return M35(m)->new_request(); }
int32_t pegainfer_qwen35_drop_request(pegainfer_qwen35_t m, int32_t id) { return M35(m)->drop_request(id); }
int32_t pegainfer_qwen35_request_seq_len(pegainfer_qwen35_t m, int32_t id) {
  auto* r = M35(m)->req(id);
  return r ? r->kv.seq_len : -1;
}
int32_t pegainfer_qwen35_prefill(pegainfer_qwen35_t m, int32_t request_id, int32_t n_tokens, const uint32_t* tokens,
                                 int32_t* out_token, void* out_logits_host) {
  return M35(m)->prefill(request_id, n_tokens, tokens, out_token, out_logits_host);
}
int32_t pegainfer_qwen35_decode(pegainfer_qwen35_t m, int32_t n_requests, const int32_t* request_ids,
                                const uint32_t* token_ids, int32_t* out_tokens, void* out_logits_host) {
  return M35(m)->decode(n_requests, request_ids, token_ids, out_tokens, out_logits_host);
}
int32_t pegainfer_qwen35_decode_greedy_chain(pegainfer_qwen35_t m, int32_t n_requests, const int32_t* request_ids,
                                             const uint32_t* first_token_ids, int32_t n_steps, int32_t* out_tokens) {
  return M35(m)->decode_greedy_chain(n_requests, request_ids, first_token_ids, n_steps, out_tokens);
}
int32_t pegainfer_qwen35_available_pages(pegainfer_qwen35_t m) { return M35(m)->pool.available(); }
int32_t pegainfer_qwen35_capacity_pages(pegainfer_qwen35_t m) { return M35(m)->pool.capacity(); }
int32_t pegainfer_qwen35_max_batch_size(pegainfer_qwen35_t m) { return M35(m)->max_bs; }
float pegainfer_qwen35_last_step_ms(pegainfer_qwen35_t m) { return M35(m)->last_step_ms; }
float pegainfer_qwen35_bench_gemv(pegainfer_qwen35_t m, int32_t which, int32_t iters) { return M35(m)->bench_gemv(which, iters); }
int32_t pegainfer_qwen35_debug_hidden_enable(pegainfer_qwen35_t m, int32_t enable) { return M35(m)->debug_hidden_enable(enable); }
int32_t pegainfer_qwen35_debug_hidden(pegainfer_qwen35_t m, int32_t layer, void* out_host_bf16, int32_t max_rows) {
  return M35(m)->debug_hidden(layer, out_host_bf16, max_rows);
}
int64_t pegainfer_qwen35_weight_bytes(pegainfer_qwen35_t m) { return M35(m)->weight_bytes; }

}  // extern "C"
