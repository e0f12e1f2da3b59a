// RMSNorm family for gfx950.  One wave64 per row (4 rows per 256-thread workgroup), 16-byte loads,
// fp32 sum of squares reduced with a DPP/bpermute butterfly (norm_core.h), the row re-read from L1/L2
// for the scale pass (rows are <= 16 KB); no LDS, no barrier.  Rounding points follow the reference exactly:
//   rms_norm            out = bf16(f32(x) * inv_rms * f32(w))                (single rounding)
//   rms_norm_offset     same with (1 + w)                                     (Gemma / Qwen3.5)
//   fused_add_rms_norm  hidden = bf16(h + r);  out = bf16((h + r)_fp32 * inv_rms * w)
//                       - ONE kernel, no memcpy node, `out` is never used as scratch
//   rms_norm_gated      per head: bf16(x * inv_rms * w_f32 * silu(gate))     (csrc/norm.cu:17-61)
// Reference: csrc/flashinfer_norm.cu:49-133 (FlashInfer RMSNorm / FusedAddRMSNorm / GemmaRMSNorm).
#include "common.h"
#include "norm_core.h"

namespace pk {

// VEC kernels: one wave per row, 4 rows per 256-thread block.  Scalar fallback (d % 8 != 0): one block per row.
template <bool OFFSET>
__global__ __launch_bounds__(kNormBlock) void rms_norm_vec_kernel(const Half* __restrict__ x,
                                                                  const Half* __restrict__ w,
                                                                  Half* __restrict__ out, int d, int rows, float eps) {
  const int row = blockIdx.x * kNormWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const Half* xr = x + (size_t)row * d;
  Half* orow = out + (size_t)row * d;
  const float bias = OFFSET ? 1.0f : 0.0f;
  const float inv = wave_row_inv_rms(xr, nullptr, d, eps);
  const int nvec = d >> 3;
  for (int i = lane; i < nvec; i += 64)
    reinterpret_cast<u32x4*>(orow)[i] = norm_scale8(reinterpret_cast<const u32x4*>(xr)[i], nullptr,
                                                    reinterpret_cast<const u32x4*>(w)[i], inv, bias, nullptr);
}

template <bool OFFSET>
__global__ __launch_bounds__(kNormBlock) void rms_norm_scalar_kernel(const Half* __restrict__ x,
                                                                     const Half* __restrict__ w,
                                                                     Half* __restrict__ out, int d, float eps) {
  __shared__ float red[kNormWaves];
  const Half* xr = x + (size_t)blockIdx.x * d;
  Half* orow = out + (size_t)blockIdx.x * d;
  const float bias = OFFSET ? 1.0f : 0.0f;
  float ss = 0.f;
  for (int i = threadIdx.x; i < d; i += kNormBlock) {
    float a = bf2f(xr[i]);
    ss += a * a;
  }
  ss = block_sum<kNormWaves>(ss, red);
  const float inv = rsqrtf(ss / (float)d + eps);
  for (int i = threadIdx.x; i < d; i += kNormBlock) orow[i] = f2bf(bf2f(xr[i]) * inv * (bias + bf2f(w[i])));
}

// hidden = bf16(h + r); out = bf16((h + r)_fp32 * inv * w): the sum of squares uses the UNROUNDED fp32 sum and is
// taken before hidden is modified; each lane then re-reads exactly the elements it overwrites.
template <bool OFFSET>
__global__ __launch_bounds__(kNormBlock) void fused_add_rms_norm_vec_kernel(Half* __restrict__ hidden,
                                                                            const Half* __restrict__ residual,
                                                                            const Half* __restrict__ w,
                                                                            Half* __restrict__ out, int d, int rows,
                                                                            float eps) {
  const int row = blockIdx.x * kNormWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  Half* hr = hidden + (size_t)row * d;
  const Half* rr = residual + (size_t)row * d;
  Half* orow = out + (size_t)row * d;
  const float bias = OFFSET ? 1.0f : 0.0f;
  const float inv = wave_row_inv_rms(hr, rr, d, eps);
  const int nvec = d >> 3;
  for (int i = lane; i < nvec; i += 64) {
    const u32x4 r = reinterpret_cast<const u32x4*>(rr)[i];
    u32x4 nh;
    const u32x4 o = norm_scale8(reinterpret_cast<const u32x4*>(hr)[i], &r, reinterpret_cast<const u32x4*>(w)[i], inv,
                                bias, &nh);
    reinterpret_cast<u32x4*>(hr)[i] = nh;
    reinterpret_cast<u32x4*>(orow)[i] = o;
  }
}

// the same, one memory pass (rows of a long prompt: launch_fused below picks it from 256 rows on)
template <int VPL>
__global__ __launch_bounds__(kNormBlock) void fused_add_rms_norm_rows_kernel(Half* __restrict__ hidden, const Half* __restrict__ residual,
                                                                             const Half* __restrict__ w, Half* __restrict__ out, int d,
                                                                             int rows, float eps) {
  const int row = blockIdx.x * kNormWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  Half* hr = hidden + (size_t)row * d;
  wave_add_norm_row_cached<VPL, false>(hr, residual + (size_t)row * d, w, hr, out + (size_t)row * d, d, eps, 0.0f);
}

template <bool OFFSET>
__global__ __launch_bounds__(kNormBlock) void fused_add_rms_norm_scalar_kernel(Half* __restrict__ hidden,
                                                                               const Half* __restrict__ residual,
                                                                               const Half* __restrict__ w,
                                                                               Half* __restrict__ out, int d,
                                                                               float eps) {
  __shared__ float red[kNormWaves];
  Half* hr = hidden + (size_t)blockIdx.x * d;
  const Half* rr = residual + (size_t)blockIdx.x * d;
  Half* orow = out + (size_t)blockIdx.x * d;
  const float bias = OFFSET ? 1.0f : 0.0f;
  float ss = 0.f;
  for (int i = threadIdx.x; i < d; i += kNormBlock) {
    float a = bf2f(hr[i]) + bf2f(rr[i]);
    ss += a * a;
  }
  ss = block_sum<kNormWaves>(ss, red);
  const float inv = rsqrtf(ss / (float)d + eps);
  for (int i = threadIdx.x; i < d; i += kNormBlock) {
    float s = bf2f(hr[i]) + bf2f(rr[i]);
    hr[i] = f2bf(s);
    orow[i] = f2bf(s * inv * (bias + bf2f(w[i])));
  }
}

// one wave per head; block = 4 heads
__global__ __launch_bounds__(256) void rms_norm_gated_kernel(const Half* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const Half* __restrict__ gate,
                                                             Half* __restrict__ out, int num_heads,
                                                             int head_dim, float eps) {
  const int head = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (head >= num_heads) return;
  const int lane = threadIdx.x & 63;
  const size_t base = (size_t)head * head_dim;
  float ss = 0.f;
  for (int i = lane; i < head_dim; i += 64) {
    float a = bf2f(x[base + i]);
    ss += a * a;
  }
  ss = wave_sum(ss);
  const float inv = rsqrtf(ss / (float)head_dim + eps);
  for (int i = lane; i < head_dim; i += 64) {
    float normed = bf2f(x[base + i]) * inv * w[i];
    float g = bf2f(gate[base + i]);
    out[base + i] = f2bf(normed * (g / (1.0f + expf(-g))));
  }
}

template <bool OFFSET>
static void launch_rms(const Half* x, const Half* w, Half* out, int d, int rows, float eps, hipStream_t s) {
  if (d <= 0 || rows <= 0) return;
  const bool vec = (d & 7) == 0 && host_aligned16(x) && host_aligned16(w) && host_aligned16(out);
  if (vec) rms_norm_vec_kernel<OFFSET><<<ceil_div(rows, kNormWaves), kNormBlock, 0, s>>>(x, w, out, d, rows, eps);
  else rms_norm_scalar_kernel<OFFSET><<<rows, kNormBlock, 0, s>>>(x, w, out, d, eps);
}

static void launch_fused(Half* hidden, const Half* residual, const Half* w, Half* out, int d, int rows,
                         float eps, hipStream_t s) {
  if (d <= 0 || rows <= 0) return;
  const bool vec = (d & 7) == 0 && host_aligned16(hidden) && host_aligned16(residual) &&
                   host_aligned16(w) && host_aligned16(out);
  // PEGAINFER_NORM_ROWS_CACHED=0: the two-pass kernel at every row count (A/B)
  static const bool cached = [] { const char* e = getenv("PEGAINFER_NORM_ROWS_CACHED"); return !(e && e[0] == '0'); }();
  if (vec && cached && rows >= 256 && d <= 8 * 64 * 5)
    fused_add_rms_norm_rows_kernel<5><<<ceil_div(rows, kNormWaves), kNormBlock, 0, s>>>(hidden, residual, w, out, d, rows, eps);
  else if (vec && cached && rows >= 256 && d <= 8 * 64 * 8)
    fused_add_rms_norm_rows_kernel<8><<<ceil_div(rows, kNormWaves), kNormBlock, 0, s>>>(hidden, residual, w, out, d, rows, eps);
  else if (vec)
    fused_add_rms_norm_vec_kernel<false><<<ceil_div(rows, kNormWaves), kNormBlock, 0, s>>>(hidden, residual, w, out, d,
                                                                                          rows, eps);
  else fused_add_rms_norm_scalar_kernel<false><<<rows, kNormBlock, 0, s>>>(hidden, residual, w, out, d, eps);
}

}  // namespace pk

using namespace pk;

extern "C" {

void rms_norm_cuda(const Half* x, const Half* weight, Half* out, int32_t n, float eps, pegainfer_stream_t stream) {
  launch_rms<false>(x, weight, out, n, 1, eps, as_stream(stream));
}
void rms_norm_batched_cuda(const Half* x, const Half* weight, Half* out, int32_t hidden_dim, int32_t seq_len,
                           float eps, pegainfer_stream_t stream) {
  launch_rms<false>(x, weight, out, hidden_dim, seq_len, eps, as_stream(stream));
}
void rms_norm_offset_cuda(const Half* x, const Half* weight, Half* out, int32_t n, float eps,
                          pegainfer_stream_t stream) {
  launch_rms<true>(x, weight, out, n, 1, eps, as_stream(stream));
}
void rms_norm_batched_offset_cuda(const Half* x, const Half* weight, Half* out, int32_t hidden_dim,
                                  int32_t seq_len, float eps, pegainfer_stream_t stream) {
  launch_rms<true>(x, weight, out, hidden_dim, seq_len, eps, as_stream(stream));
}
void fused_add_rms_norm_cuda(Half* hidden, const Half* residual, const Half* weight, Half* out, int32_t n,
                             float eps, pegainfer_stream_t stream) {
  launch_fused(hidden, residual, weight, out, n, 1, eps, as_stream(stream));
}
void fused_add_rms_norm_batched_cuda(Half* hidden, const Half* residual, const Half* weight, Half* out,
                                     int32_t hidden_dim, int32_t batch_size, float eps,
                                     pegainfer_stream_t stream) {
  launch_fused(hidden, residual, weight, out, hidden_dim, batch_size, eps, as_stream(stream));
}
void rms_norm_gated_cuda(const Half* x, const float* weight, const Half* gate, Half* out, int32_t num_heads,
                         int32_t head_dim, float eps, pegainfer_stream_t stream) {
  if (num_heads <= 0 || head_dim <= 0) return;
  rms_norm_gated_kernel<<<ceil_div(num_heads, 4), 256, 0, as_stream(stream)>>>(x, weight, gate, out,
                                                                              num_heads, head_dim, eps);
}

}  // extern "C"
