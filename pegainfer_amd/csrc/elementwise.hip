// Elementwise / gather ops of the pegainfer-kernels ABI, written for gfx950:
// HBM-bound byte movers - 16-byte (8 x bf16) accesses per lane, grid-stride, fp32 math,
// one rounding at the store exactly where the reference rounds.
//   add, silu_mul (two rounding variants), embedding (3 variants), bf16<->f32 casts.
#include "common.h"

namespace pk {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 blocks (guide: cap + grid-stride)


// ---- add: out = bf16(f32(a)+f32(b))  (reference csrc/elementwise.cu:8-20) ----
__global__ void add_vec_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                               u32x4* __restrict__ out, int nvec) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < nvec; i += gridDim.x * kBlock) {
    u32x4 va = a[i], vb = b[i], vo;
    vo.x = pack_bf2(bf_lo(va.x) + bf_lo(vb.x), bf_hi(va.x) + bf_hi(vb.x));
    vo.y = pack_bf2(bf_lo(va.y) + bf_lo(vb.y), bf_hi(va.y) + bf_hi(vb.y));
    vo.z = pack_bf2(bf_lo(va.z) + bf_lo(vb.z), bf_hi(va.z) + bf_hi(vb.z));
    vo.w = pack_bf2(bf_lo(va.w) + bf_lo(vb.w), bf_hi(va.w) + bf_hi(vb.w));
    out[i] = vo;
  }
}
__global__ void add_scalar_kernel(const Half* __restrict__ a, const Half* __restrict__ b,
                                  Half* __restrict__ out, int start, int n) {
  for (int i = start + blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    out[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}

// ---- silu_mul, separate gate/up; silu rounded to bf16 BEFORE the multiply
//      (reference csrc/elementwise.cu:28-42, Qwen3.5 MLP) ----
__global__ void silu_mul_kernel(const Half* __restrict__ gate, const Half* __restrict__ up,
                                Half* __restrict__ out, int n) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float g = bf2f(gate[i]), u = bf2f(up[i]);
    out[i] = f2bf(bf16_round_f(silu_f(g)) * u);
  }
}
__global__ void silu_mul_vec_kernel(const u32x4* __restrict__ gate, const u32x4* __restrict__ up,
                                    u32x4* __restrict__ out, int nvec) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < nvec; i += gridDim.x * kBlock) {
    u32x4 g = gate[i], u = up[i], o;
    auto f = [](uint32_t gw, uint32_t uw) {
      return pack_bf2(bf16_round_f(silu_f(bf_lo(gw))) * bf_lo(uw),
                      bf16_round_f(silu_f(bf_hi(gw))) * bf_hi(uw));
    };
    o.x = f(g.x, u.x); o.y = f(g.y, u.y); o.z = f(g.z, u.z); o.w = f(g.w, u.w);
    out[i] = o;
  }
}

// ---- silu_mul_fused on [bs, 2I]: out[j,i] = bf16(silu(g)*u), ONE rounding
//      (reference csrc/fused_proj.cu:44-63) ----
__global__ void silu_mul_fused_vec_kernel(const u32x4* __restrict__ gate_up, u32x4* __restrict__ out,
                                          int ivec /* I/8 */, int bs) {
  const int total = ivec * bs;
  for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < total; idx += gridDim.x * kBlock) {
    const int col = idx / ivec, row = idx - col * ivec;
    u32x4 g = gate_up[(size_t)col * 2 * ivec + row];
    u32x4 u = gate_up[(size_t)col * 2 * ivec + ivec + row];
    u32x4 o;
    auto f = [](uint32_t gw, uint32_t uw) {
      return pack_bf2(silu_f(bf_lo(gw)) * bf_lo(uw), silu_f(bf_hi(gw)) * bf_hi(uw));
    };
    o.x = f(g.x, u.x); o.y = f(g.y, u.y); o.z = f(g.z, u.z); o.w = f(g.w, u.w);
    out[idx] = o;
  }
}
__global__ void silu_mul_fused_scalar_kernel(const Half* __restrict__ gate_up, Half* __restrict__ out,
                                             int I, int bs) {
  const int total = I * bs;
  for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < total; idx += gridDim.x * kBlock) {
    const int col = idx / I, row = idx - col * I;
    const size_t src = (size_t)col * 2 * I;
    out[idx] = f2bf(silu_f(bf2f(gate_up[src + row])) * bf2f(gate_up[src + I + row]));
  }
}

// ---- embedding gathers (reference csrc/elementwise.cu:49-112) ----
// grid.x = token, grid.y tiles the row; 16 B per lane when hidden % 8 == 0.
template <bool SHARD>
__global__ void embedding_kernel(const Half* __restrict__ embed, const uint32_t* __restrict__ token_ids,
                                 Half* __restrict__ out, int hidden, int seq_len, uint32_t vocab_start,
                                 uint32_t part_vocab, int vec_ok) {
  const int tok_i = blockIdx.x;
  uint32_t tok = token_ids[tok_i];
  bool inside = true;
  if (SHARD) {
    inside = tok >= vocab_start && tok < vocab_start + part_vocab;
    tok -= vocab_start;
  }
  const Half* src = embed + (size_t)tok * hidden;
  Half* dst = out + (size_t)tok_i * hidden;
  if (vec_ok) {
    const int nvec = hidden >> 3;
    for (int i = blockIdx.y * kBlock + threadIdx.x; i < nvec; i += gridDim.y * kBlock) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (inside) v = reinterpret_cast<const u32x4*>(src)[i];
      reinterpret_cast<u32x4*>(dst)[i] = v;
    }
  } else {
    for (int i = blockIdx.y * kBlock + threadIdx.x; i < hidden; i += gridDim.y * kBlock)
      dst[i] = inside ? src[i] : (Half)0;
  }
}

// ---- bf16 <-> f32 casts around the MP8 collectives
//      (reference csrc/deepseek_v4: deepseek_bf16_to_f32_cuda / deepseek_f32_to_bf16_cuda) ----
__global__ void bf16_to_f32_kernel(const Half* __restrict__ in, float* __restrict__ out, int n) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = bf2f(in[i]);
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, Half* __restrict__ out, int n) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = f2bf(in[i]);
}

static inline int grid_for(long work) {
  int g = ceil_div(work, kBlock);
  return g < 1 ? 1 : (g > kMaxGrid ? kMaxGrid : g);
}
static inline pegainfer_status_t last_error() { return static_cast<pegainfer_status_t>(hipGetLastError()); }

}  // namespace pk

using namespace pk;

// zero `n` 32-bit words (device counters at the head of a captured step).  A KERNEL on purpose: round 5 found that a
// hipMemsetAsync node inside a replayed hipGraph is not reliably ordered against the graph's kernels once any eager
// kernel launch has run on the stream between two replays (tools/diag_8b.py: the in-launch split-KV merge tickets were
// zeroed mid-flight, for good) - kernel nodes of a captured stream are.
__global__ void zero_words_kernel(uint32_t* __restrict__ p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}

extern "C" {

pegainfer_status_t pegainfer_zero_words(void* ptr, int32_t n_words, pegainfer_stream_t stream) {
  if (n_words > 0)
    zero_words_kernel<<<n_words >= 4096 ? 16 : 1, 256, 0, as_stream(stream)>>>(static_cast<uint32_t*>(ptr), n_words);
  return last_error();
}

pegainfer_status_t add_cuda(const Half* a, const Half* b, Half* out, int32_t n, pegainfer_stream_t stream) {
  if (n <= 0) return 0;
  hipStream_t s = as_stream(stream);
  int done = 0;
  if (host_aligned16(a) && host_aligned16(b) && host_aligned16(out) && n >= 8) {
    const int nvec = n >> 3;
    add_vec_kernel<<<grid_for(nvec), kBlock, 0, s>>>(reinterpret_cast<const u32x4*>(a),
                                                     reinterpret_cast<const u32x4*>(b),
                                                     reinterpret_cast<u32x4*>(out), nvec);
    done = nvec << 3;
  }
  if (done < n) add_scalar_kernel<<<grid_for(n - done), kBlock, 0, s>>>(a, b, out, done, n);
  return last_error();
}

pegainfer_status_t silu_mul_triton_aot_cuda(const Half* gate, const Half* up, Half* out, int32_t n,
                                            pegainfer_stream_t stream) {
  if (n <= 0) return 0;
  hipStream_t s = as_stream(stream);
  if ((n & 7) == 0 && host_aligned16(gate) && host_aligned16(up) && host_aligned16(out)) {
    silu_mul_vec_kernel<<<grid_for(n >> 3), kBlock, 0, s>>>(reinterpret_cast<const u32x4*>(gate),
                                                            reinterpret_cast<const u32x4*>(up),
                                                            reinterpret_cast<u32x4*>(out), n >> 3);
  } else {
    silu_mul_kernel<<<grid_for(n), kBlock, 0, s>>>(gate, up, out, n);
  }
  return last_error();
}

void silu_mul_fused_cuda(const Half* gate_up, Half* out, int32_t intermediate_size, int32_t bs,
                         pegainfer_stream_t stream) {
  if (intermediate_size <= 0 || bs <= 0) return;
  hipStream_t s = as_stream(stream);
  if ((intermediate_size & 7) == 0 && host_aligned16(gate_up) && host_aligned16(out)) {
    const int ivec = intermediate_size >> 3;
    silu_mul_fused_vec_kernel<<<grid_for((long)ivec * bs), kBlock, 0, s>>>(
        reinterpret_cast<const u32x4*>(gate_up), reinterpret_cast<u32x4*>(out), ivec, bs);
  } else {
    silu_mul_fused_scalar_kernel<<<grid_for((long)intermediate_size * bs), kBlock, 0, s>>>(
        gate_up, out, intermediate_size, bs);
  }
}

static void launch_embedding(bool shard, const Half* embed, const uint32_t* token_ids, Half* out,
                             int hidden, int seq_len, uint32_t vocab_start, uint32_t part_vocab,
                             hipStream_t s) {
  if (hidden <= 0 || seq_len <= 0) return;
  const int vec_ok = ((hidden & 7) == 0 && host_aligned16(embed) && host_aligned16(out)) ? 1 : 0;
  const int per_row = vec_ok ? (hidden >> 3) : hidden;
  dim3 grid(seq_len, ceil_div(per_row, kBlock) > 64 ? 64 : ceil_div(per_row, kBlock));
  if (shard)
    embedding_kernel<true><<<grid, kBlock, 0, s>>>(embed, token_ids, out, hidden, seq_len, vocab_start,
                                                   part_vocab, vec_ok);
  else
    embedding_kernel<false><<<grid, kBlock, 0, s>>>(embed, token_ids, out, hidden, seq_len, 0, 0, vec_ok);
}

pegainfer_status_t embedding_batched_cuda(const Half* embed, const uint32_t* token_ids, Half* out,
                                          int32_t hidden_size, int32_t seq_len, pegainfer_stream_t stream) {
  launch_embedding(false, embed, token_ids, out, hidden_size, seq_len, 0, 0, as_stream(stream));
  return last_error();
}

pegainfer_status_t embedding_batched_vocab_shard_cuda(const Half* embed, const uint32_t* token_ids, Half* out,
                                                      int32_t hidden_size, int32_t seq_len,
                                                      uint32_t vocab_start, uint32_t part_vocab_size,
                                                      pegainfer_stream_t stream) {
  launch_embedding(true, embed, token_ids, out, hidden_size, seq_len, vocab_start, part_vocab_size,
                   as_stream(stream));
  return last_error();
}

pegainfer_status_t embedding_decode_cuda(const Half* embed, const uint32_t* token_id, Half* out,
                                         int32_t hidden_size, pegainfer_stream_t stream) {
  launch_embedding(false, embed, token_id, out, hidden_size, 1, 0, 0, as_stream(stream));
  return last_error();
}

pegainfer_status_t deepseek_bf16_to_f32_cuda(const Half* input, float* output, int32_t n,
                                             pegainfer_stream_t stream) {
  if (n > 0) bf16_to_f32_kernel<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(input, output, n);
  return last_error();
}

pegainfer_status_t deepseek_f32_to_bf16_cuda(const float* input, Half* output, int32_t n,
                                             pegainfer_stream_t stream) {
  if (n > 0) f32_to_bf16_kernel<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(input, output, n);
  return last_error();
}

}  // extern "C"
