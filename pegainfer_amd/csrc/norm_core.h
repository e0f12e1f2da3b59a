// Canonical fp32 sum-of-squares of one token row for a 256-thread workgroup.  Shared by the
// stand-alone norm kernels (norm.hip) and the GEMV prologues (gemv_core.h) so that
// "fused_add_rms_norm then GEMV" and "GEMV with the norm folded into its prologue" round
// identically: same per-thread element order, same wave butterfly, same 4-wave combine.
#pragma once

#include "common.h"

namespace pk {

constexpr int kNormBlock = 256;
constexpr int kNormWaves = kNormBlock / 64;

__device__ __forceinline__ void sq8(const u32x4& v, float& ss) {
  float a;
  a = bf_lo(v.x); ss += a * a; a = bf_hi(v.x); ss += a * a;
  a = bf_lo(v.y); ss += a * a; a = bf_hi(v.y); ss += a * a;
  a = bf_lo(v.z); ss += a * a; a = bf_hi(v.z); ss += a * a;
  a = bf_lo(v.w); ss += a * a; a = bf_hi(v.w); ss += a * a;
}
__device__ __forceinline__ void add_sq8(const u32x4& h, const u32x4& r, float& ss) {
  float a;
  a = bf_lo(h.x) + bf_lo(r.x); ss += a * a; a = bf_hi(h.x) + bf_hi(r.x); ss += a * a;
  a = bf_lo(h.y) + bf_lo(r.y); ss += a * a; a = bf_hi(h.y) + bf_hi(r.y); ss += a * a;
  a = bf_lo(h.z) + bf_lo(r.z); ss += a * a; a = bf_hi(h.z) + bf_hi(r.z); ss += a * a;
  a = bf_lo(h.w) + bf_lo(r.w); ss += a * a; a = bf_hi(h.w) + bf_hi(r.w); ss += a * a;
}

// inv_rms of row `hr` (+ `rr` when non-null), d % 8 == 0, all 256 threads participate.
__device__ __forceinline__ float row_inv_rms_vec(const Half* __restrict__ hr, const Half* __restrict__ rr, int d,
                                                 float eps, float* red /* >= 4 floats of LDS */) {
  float ss = 0.f;
  const int nvec = d >> 3;
  if (rr) {
    for (int i = threadIdx.x; i < nvec; i += kNormBlock)
      add_sq8(reinterpret_cast<const u32x4*>(hr)[i], reinterpret_cast<const u32x4*>(rr)[i], ss);
  } else {
    for (int i = threadIdx.x; i < nvec; i += kNormBlock) sq8(reinterpret_cast<const u32x4*>(hr)[i], ss);
  }
  ss = block_sum<kNormWaves>(ss, red);
  return rsqrtf(ss / (float)d + eps);
}

// out = bf16(s * inv * (bias + w)) on 8 packed elements; s = h (+ r).  Also returns bf16(s) in `nh`.
__device__ __forceinline__ u32x4 norm_scale8(const u32x4& h, const u32x4* r, const u32x4& g, float inv, float bias,
                                             u32x4* nh) {
  u32x4 o, n;
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, gw[4] = {g.x, g.y, g.z, g.w};
  uint32_t rw[4] = {0u, 0u, 0u, 0u};
  if (r) { rw[0] = r->x; rw[1] = r->y; rw[2] = r->z; rw[3] = r->w; }
  uint32_t ow[4], nw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s0 = bf_lo(hw[j]), s1 = bf_hi(hw[j]);
    if (r) { s0 += bf_lo(rw[j]); s1 += bf_hi(rw[j]); }
    nw[j] = pack_bf2(s0, s1);
    ow[j] = pack_bf2(s0 * inv * (bias + bf_lo(gw[j])), s1 * inv * (bias + bf_hi(gw[j])));
  }
  o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
  n.x = nw[0]; n.y = nw[1]; n.z = nw[2]; n.w = nw[3];
  if (nh) *nh = n;
  return o;
}

}  // namespace pk
