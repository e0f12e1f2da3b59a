// Per-head QK RMSNorm + NeoX (half-split) RoPE, in place, for gfx950.
// One wave per (head, token): lane l owns the RoPE pair (d, d+half) for d = l, l+64, ...,
// the sum of squares is a single wave64 butterfly (no LDS, no barrier - the reference needs
// 3 __syncthreads and a 128-thread block because its warp is 32 wide).
// Rounding sequence is the reference's (csrc/prefill_attention.cu:55-84):
//   n = bf16(x*inv_rms);  m = bf16(f32(n)*f32(w));
//   out[d]      = bf16(m[d]*c - m[d+half]*s)
//   out[d+half] = bf16(m[d]*s + m[d+half]*c)          c,s = bf16 table[pos*head_dim + d]
#include "common.h"
#include "rope_core.h"
#include "pegainfer_kernels_ext.h"

namespace pk {

constexpr int kMaxPairsPerLane = 4;  // head_dim <= 512

__global__ __launch_bounds__(256) void qk_norm_rope_kernel(
    Half* __restrict__ q, Half* __restrict__ k, const Half* __restrict__ q_w, const Half* __restrict__ k_w,
    const Half* __restrict__ cos_cache, const Half* __restrict__ sin_cache, int num_q_heads, int num_kv_heads,
    int head_dim, int tokens, int start_pos, const int* __restrict__ positions, float eps) {
  const int heads = num_q_heads + num_kv_heads;
  const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (token, head) pair index
  if (unit >= (long)tokens * heads) return;
  const int token = (int)(unit / heads);
  const int hg = (int)(unit - (long)token * heads);
  const int lane = threadIdx.x & 63;
  const bool is_q = hg < num_q_heads;
  const int head = is_q ? hg : hg - num_q_heads;
  Half* data = (is_q ? q : k) + (size_t)token * (is_q ? num_q_heads : num_kv_heads) * head_dim +
               (size_t)head * head_dim;
  const Half* w = is_q ? q_w : k_w;
  const int half = head_dim >> 1;
  const int pos = positions ? positions[token] : start_pos + token;
  const Half* crow = cos_cache + (size_t)pos * head_dim;
  const Half* srow = sin_cache + (size_t)pos * head_dim;

  float lo[kMaxPairsPerLane], hi[kMaxPairsPerLane];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxPairsPerLane; ++j) {
    const int d = lane + j * 64;
    lo[j] = hi[j] = 0.f;
    if (d < half) {
      lo[j] = bf2f(data[d]);
      hi[j] = bf2f(data[d + half]);
      ss += lo[j] * lo[j];
      ss += hi[j] * hi[j];
    }
  }
  ss = wave_sum(ss);
  const float inv = rsqrtf(ss / (float)head_dim + eps);
#pragma unroll
  for (int j = 0; j < kMaxPairsPerLane; ++j) {
    const int d = lane + j * 64;
    if (d < half) {
      const float ml = bf16_round_f(bf16_round_f(lo[j] * inv) * bf2f(w[d]));
      const float mh = bf16_round_f(bf16_round_f(hi[j] * inv) * bf2f(w[d + half]));
      const float c = bf2f(crow[d]), s = bf2f(srow[d]);
      data[d] = f2bf(ml * c - mh * s);
      data[d + half] = f2bf(ml * s + mh * c);
    }
  }
}

// head_dim == 128 fast path: 16 lanes per head (one 16-byte load each), 4 heads per wave, 16 per block.
__global__ __launch_bounds__(256) void qk_norm_rope128_kernel(
    Half* __restrict__ q, Half* __restrict__ k, const Half* __restrict__ q_w, const Half* __restrict__ k_w,
    const Half* __restrict__ cos_cache, const Half* __restrict__ sin_cache, int num_q_heads, int num_kv_heads,
    int tokens, int start_pos, const int* __restrict__ positions, float eps) {
  const int heads = num_q_heads + num_kv_heads;
  const long unit = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (unit >= (long)tokens * heads) return;  // whole 16-lane rows drop out together (DPP stays inside a row)
  const int token = (int)(unit / heads);
  const int hg = (int)(unit - (long)token * heads);
  const int sub = threadIdx.x & 15;
  const bool is_q = hg < num_q_heads;
  Half* data = is_q ? q + ((size_t)token * num_q_heads + hg) * 128
                    : k + ((size_t)token * num_kv_heads + (hg - num_q_heads)) * 128;
  const int pos = positions ? positions[token] : start_pos + token;
  const u32x4 x = *reinterpret_cast<const u32x4*>(data + sub * 8);
  const u32x4 r = head_norm_rope16(x, is_q ? q_w : k_w, cos_cache + (size_t)pos * 128, sin_cache + (size_t)pos * 128,
                                   sub, eps);
  *reinterpret_cast<u32x4*>(data + sub * 8) = r;
}

// qk_norm_rope128_kernel + paged_kv_scatter_kernel in one launch: 16-lane rows over (token, q heads | k heads | v heads).
// A k row is normalised + rotated, written back in place AND into its cache slot; a v row is copied into its slot.
__global__ __launch_bounds__(256) void qk_norm_rope128_scatter_kernel(
    Half* __restrict__ q, Half* __restrict__ k, const Half* __restrict__ v, const Half* __restrict__ q_w,
    const Half* __restrict__ k_w, const Half* __restrict__ cos_cache, const Half* __restrict__ sin_cache,
    const int* __restrict__ positions, const int* __restrict__ batch_indices, Half* __restrict__ kv, long k_off, long v_off,
    const int* __restrict__ page_indices, const int* __restrict__ page_indptr, int num_q_heads, int num_kv_heads,
    int page_size, long stride_page, int tokens, float eps) {
  const int heads = num_q_heads + 2 * num_kv_heads;
  const long unit = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (unit >= (long)tokens * heads) return;  // whole 16-lane rows drop out together (DPP stays inside a row)
  const int token = (int)(unit / heads);
  const int hg = (int)(unit - (long)token * heads);
  const int sub = threadIdx.x & 15;
  const int pos = positions[token];
  if (hg < num_q_heads) {
    Half* data = q + ((size_t)token * num_q_heads + hg) * 128;
    const u32x4 x = *reinterpret_cast<const u32x4*>(data + sub * 8);
    *reinterpret_cast<u32x4*>(data + sub * 8) =
        head_norm_rope16(x, q_w, cos_cache + (size_t)pos * 128, sin_cache + (size_t)pos * 128, sub, eps);
    return;
  }
  const bool is_v = hg >= num_q_heads + num_kv_heads;
  const int h = hg - num_q_heads - (is_v ? num_kv_heads : 0);
  const int page = page_indices[page_indptr[batch_indices[token]] + pos / page_size];
  const long dst = (long)page * stride_page + (is_v ? v_off : k_off) + ((long)(pos % page_size) * num_kv_heads + h) * 128;
  u32x4 r;
  if (is_v) {
    r = *reinterpret_cast<const u32x4*>(v + ((size_t)token * num_kv_heads + h) * 128 + sub * 8);
  } else {
    Half* data = k + ((size_t)token * num_kv_heads + h) * 128;
    const u32x4 x = *reinterpret_cast<const u32x4*>(data + sub * 8);
    r = head_norm_rope16(x, k_w, cos_cache + (size_t)pos * 128, sin_cache + (size_t)pos * 128, sub, eps);
    *reinterpret_cast<u32x4*>(data + sub * 8) = r;
  }
  *reinterpret_cast<u32x4*>(kv + dst + sub * 8) = r;
}

// The same work on the STACKED q|k|v rows of one GEMM (qkv [tokens][(Hq + 2 Hkv) * 128], the layout of the fused decode
// path): q heads are normalised + rotated into the dense q_out [tokens][Hq * 128] the prefill attention reads, k heads are
// normalised + rotated straight into their cache slot, v heads are copied into theirs.  The de-interleave costs nothing
// extra: every byte is read once and written once, as in the three-buffer form.  Same per-head arithmetic (rope_core.h).
__global__ __launch_bounds__(256) void qkv_stacked_norm_rope128_scatter_kernel(
    const Half* __restrict__ qkv, Half* __restrict__ q_out, const Half* __restrict__ q_w, const Half* __restrict__ k_w,
    const Half* __restrict__ cos_cache, const Half* __restrict__ sin_cache, const int* __restrict__ positions,
    const int* __restrict__ batch_indices, Half* __restrict__ kv, long k_off, long v_off,
    const int* __restrict__ page_indices, const int* __restrict__ page_indptr, int num_q_heads, int num_kv_heads,
    int page_size, long stride_page, int tokens, float eps) {
  const int heads = num_q_heads + 2 * num_kv_heads;
  const long unit = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (unit >= (long)tokens * heads) return;  // whole 16-lane rows drop out together (DPP stays inside a row)
  const int token = (int)(unit / heads);
  const int hg = (int)(unit - (long)token * heads);
  const int sub = threadIdx.x & 15;
  const int pos = positions[token];
  const u32x4 x = *reinterpret_cast<const u32x4*>(qkv + ((size_t)token * heads + hg) * 128 + sub * 8);
  if (hg < num_q_heads) {
    *reinterpret_cast<u32x4*>(q_out + ((size_t)token * num_q_heads + hg) * 128 + sub * 8) =
        head_norm_rope16(x, q_w, cos_cache + (size_t)pos * 128, sin_cache + (size_t)pos * 128, sub, eps);
    return;
  }
  const bool is_v = hg >= num_q_heads + num_kv_heads;
  const int h = hg - num_q_heads - (is_v ? num_kv_heads : 0);
  const int page = page_indices[page_indptr[batch_indices[token]] + pos / page_size];
  const long dst = (long)page * stride_page + (is_v ? v_off : k_off) + ((long)(pos % page_size) * num_kv_heads + h) * 128;
  const u32x4 r = is_v ? x : head_norm_rope16(x, k_w, cos_cache + (size_t)pos * 128, sin_cache + (size_t)pos * 128, sub, eps);
  *reinterpret_cast<u32x4*>(kv + dst + sub * 8) = r;
}

static void launch(Half* q, Half* k, const Half* qw, const Half* kw, const Half* c, const Half* s, int hq,
                   int hkv, int hd, int tokens, int start_pos, const int* positions, float eps,
                   hipStream_t stream) {
  if (tokens <= 0 || hq + hkv <= 0) return;
  const long units = (long)tokens * (hq + hkv);
  if (hd == 128 && host_aligned16(q) && host_aligned16(k) && host_aligned16(qw) && host_aligned16(kw) &&
      host_aligned16(c) && host_aligned16(s)) {
    qk_norm_rope128_kernel<<<ceil_div(units, 16), 256, 0, stream>>>(q, k, qw, kw, c, s, hq, hkv, tokens, start_pos,
                                                                    positions, eps);
    return;
  }
  qk_norm_rope_kernel<<<ceil_div(units, 4), 256, 0, stream>>>(q, k, qw, kw, c, s, hq, hkv, hd, tokens,
                                                              start_pos, positions, eps);
}

}  // namespace pk

extern "C" {

// ffi.rs:164 - prefill: positions are start_pos + token
void prefill_qk_norm_rope_only_cuda(Half* q_batch, Half* k_batch, const Half* q_norm_weight,
                                    const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache,
                                    int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                    int32_t seq_len, int32_t start_pos, float rms_eps,
                                    pegainfer_stream_t stream) {
  pk::launch(q_batch, k_batch, q_norm_weight, k_norm_weight, cos_cache, sin_cache, num_q_heads, num_kv_heads,
             head_dim, seq_len, start_pos, nullptr, rms_eps, pk::as_stream(stream));
}

// ffi.rs:1143 - decode / multi-request prefill: per-token positions from a device array
void qk_norm_rope_batched_decode_cuda(Half* q, Half* k, const Half* q_norm_weight, const Half* k_norm_weight,
                                      const Half* cos_cache, const Half* sin_cache, const int32_t* positions,
                                      int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                      int32_t batch_size, float rms_eps, pegainfer_stream_t stream) {
  pk::launch(q, k, q_norm_weight, k_norm_weight, cos_cache, sin_cache, num_q_heads, num_kv_heads, head_dim,
             batch_size, 0, positions, rms_eps, pk::as_stream(stream));
}

// extension (include/pegainfer_kernels_ext.h): the two reference calls above + paged_kv_scatter_cuda in one launch
int32_t pegainfer_qk_norm_rope_scatter(Half* q, Half* k, const Half* v, const Half* q_norm_weight,
                                       const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache,
                                       const int32_t* positions, const int32_t* batch_indices, Half* kv_data,
                                       int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices,
                                       const int32_t* page_indptr, int32_t num_q_heads, int32_t num_kv_heads,
                                       int32_t head_dim, int32_t page_size, int64_t stride_page, int32_t tokens,
                                       float rms_eps, pegainfer_stream_t stream) {
  using namespace pk;
  if (tokens <= 0) return 0;
  if (head_dim != 128 || num_q_heads < 0 || num_kv_heads <= 0 || page_size <= 0 || !positions || !batch_indices ||
      !host_aligned16(q) || !host_aligned16(k) || !host_aligned16(v) || !host_aligned16(kv_data) ||
      !host_aligned16(q_norm_weight) || !host_aligned16(k_norm_weight) || !host_aligned16(cos_cache) ||
      !host_aligned16(sin_cache) || (stride_page & 7) || (k_offset_elems & 7) || (v_offset_elems & 7))
    return static_cast<int32_t>(hipErrorInvalidValue);
  const long units = (long)tokens * (num_q_heads + 2 * num_kv_heads);
  qk_norm_rope128_scatter_kernel<<<ceil_div(units, 16), 256, 0, as_stream(stream)>>>(
      q, k, v, q_norm_weight, k_norm_weight, cos_cache, sin_cache, positions, batch_indices, kv_data, k_offset_elems,
      v_offset_elems, page_indices, page_indptr, num_q_heads, num_kv_heads, page_size, stride_page, tokens, rms_eps);
  return static_cast<int32_t>(hipGetLastError());
}

// pegainfer_qk_norm_rope_scatter over the row-stacked output of ONE q|k|v GEMM (extension; short-prompt prefill): qkv
// [tokens][(Hq + 2 Hkv) * 128] in, dense q_out [tokens][Hq * 128] + the K / V cache slots out.  The same bytes the
// three-buffer form leaves in q and in the cache (k is not written back: prefill attention reads it from the cache).
int32_t pegainfer_qkv_stacked_norm_rope_scatter(const Half* qkv, Half* q_out, const Half* q_norm_weight,
                                                const Half* k_norm_weight, const Half* cos_cache, const Half* sin_cache,
                                                const int32_t* positions, const int32_t* batch_indices, Half* kv_data,
                                                int64_t k_offset_elems, int64_t v_offset_elems, const int32_t* page_indices,
                                                const int32_t* page_indptr, int32_t num_q_heads, int32_t num_kv_heads,
                                                int32_t head_dim, int32_t page_size, int64_t stride_page, int32_t tokens,
                                                float rms_eps, pegainfer_stream_t stream) {
  using namespace pk;
  if (tokens <= 0) return 0;
  if (head_dim != 128 || page_size <= 0 || !host_aligned16(qkv) || !host_aligned16(q_out) || !host_aligned16(kv_data) ||
      !host_aligned16(q_norm_weight) || !host_aligned16(k_norm_weight) || !host_aligned16(cos_cache) ||
      !host_aligned16(sin_cache) || (stride_page & 7) || (k_offset_elems & 7) || (v_offset_elems & 7))
    return static_cast<int32_t>(hipErrorInvalidValue);
  const long units = (long)tokens * (num_q_heads + 2 * num_kv_heads);
  qkv_stacked_norm_rope128_scatter_kernel<<<ceil_div(units, 16), 256, 0, as_stream(stream)>>>(
      qkv, q_out, q_norm_weight, k_norm_weight, cos_cache, sin_cache, positions, batch_indices, kv_data, k_offset_elems,
      v_offset_elems, page_indices, page_indptr, num_q_heads, num_kv_heads, page_size, stride_page, tokens, rms_eps);
  return static_cast<int32_t>(hipGetLastError());
}

}  // extern "C"
