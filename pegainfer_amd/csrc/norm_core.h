// Canonical fp32 sum-of-squares of one token row, computed by ONE wave64: lane l folds the 16-byte vectors
// l, l+64, l+128, ... in order, then a wave butterfly (4 DPP steps + 2 bpermutes).  Shared by the stand-alone
// norm kernels (norm.hip) and the GEMV / skinny-GEMM prologues so that "fused_add_rms_norm then GEMM" and
// "GEMM with the norm folded into its prologue" round identically.  One wave per row means a workgroup
// normalises several token rows concurrently with no barrier (a 256-thread-per-row reduction serialised the
// prologue over the batch: T x (loads + 2 barriers)).
#pragma once

#include "common.h"

namespace pk {

constexpr int kNormBlock = 256;
constexpr int kNormWaves = kNormBlock / 64;

__device__ __forceinline__ void sq8(const u32x4& v, float& ss) {
  float a;
  a = bf_lo(v.x); ss = fmaf(a, a, ss); a = bf_hi(v.x); ss = fmaf(a, a, ss);
  a = bf_lo(v.y); ss = fmaf(a, a, ss); a = bf_hi(v.y); ss = fmaf(a, a, ss);
  a = bf_lo(v.z); ss = fmaf(a, a, ss); a = bf_hi(v.z); ss = fmaf(a, a, ss);
  a = bf_lo(v.w); ss = fmaf(a, a, ss); a = bf_hi(v.w); ss = fmaf(a, a, ss);
}
__device__ __forceinline__ void add_sq8(const u32x4& h, const u32x4& r, float& ss) {
  float a;
  a = bf_lo(h.x) + bf_lo(r.x); ss = fmaf(a, a, ss); a = bf_hi(h.x) + bf_hi(r.x); ss = fmaf(a, a, ss);
  a = bf_lo(h.y) + bf_lo(r.y); ss = fmaf(a, a, ss); a = bf_hi(h.y) + bf_hi(r.y); ss = fmaf(a, a, ss);
  a = bf_lo(h.z) + bf_lo(r.z); ss = fmaf(a, a, ss); a = bf_hi(h.z) + bf_hi(r.z); ss = fmaf(a, a, ss);
  a = bf_lo(h.w) + bf_lo(r.w); ss = fmaf(a, a, ss); a = bf_hi(h.w) + bf_hi(r.w); ss = fmaf(a, a, ss);
}
// same with the sum rounded to bf16 first: "add_cuda, then rms_norm on its bf16 output" (the Qwen3.5 residual
// chain, batch_decode.rs:246-262) as opposed to FlashInfer's fused add+norm over the un-rounded sum
__device__ __forceinline__ void add_round_sq8(const u32x4& h, const u32x4& r, float& ss) {
  float a;
  a = bf16_round_f(bf_lo(h.x) + bf_lo(r.x)); ss = fmaf(a, a, ss); a = bf16_round_f(bf_hi(h.x) + bf_hi(r.x)); ss = fmaf(a, a, ss);
  a = bf16_round_f(bf_lo(h.y) + bf_lo(r.y)); ss = fmaf(a, a, ss); a = bf16_round_f(bf_hi(h.y) + bf_hi(r.y)); ss = fmaf(a, a, ss);
  a = bf16_round_f(bf_lo(h.z) + bf_lo(r.z)); ss = fmaf(a, a, ss); a = bf16_round_f(bf_hi(h.z) + bf_hi(r.z)); ss = fmaf(a, a, ss);
  a = bf16_round_f(bf_lo(h.w) + bf_lo(r.w)); ss = fmaf(a, a, ss); a = bf16_round_f(bf_hi(h.w) + bf_hi(r.w)); ss = fmaf(a, a, ss);
}

// inv_rms of row `hr` (+ `rr` when non-null), d % 8 == 0; call from ALL 64 lanes of one wave.
__device__ __forceinline__ float wave_row_inv_rms(const Half* __restrict__ hr, const Half* __restrict__ rr, int d,
                                                  float eps, bool round_sum = false) {
  float ss = 0.f;
  const int nvec = d >> 3;
  const int lane = threadIdx.x & 63;
  if (rr && round_sum) {
    for (int i = lane; i < nvec; i += 64)
      add_round_sq8(reinterpret_cast<const u32x4*>(hr)[i], reinterpret_cast<const u32x4*>(rr)[i], ss);
  } else if (rr) {
    for (int i = lane; i < nvec; i += 64)
      add_sq8(reinterpret_cast<const u32x4*>(hr)[i], reinterpret_cast<const u32x4*>(rr)[i], ss);
  } else {
    for (int i = lane; i < nvec; i += 64) sq8(reinterpret_cast<const u32x4*>(hr)[i], ss);
  }
  ss = wave_sum(ss);
  return rsqrtf(__fadd_rn(ss / (float)d, eps));
}

// out = bf16(s * inv * (bias + w)) on 8 packed elements; s = h (+ r).  Also returns bf16(s) in `nh`.
__device__ __forceinline__ u32x4 norm_scale8(const u32x4& h, const u32x4* r, const u32x4& g, float inv, float bias,
                                             u32x4* nh, bool round_sum = false) {
  u32x4 o, n;
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, gw[4] = {g.x, g.y, g.z, g.w};
  uint32_t rw[4] = {0u, 0u, 0u, 0u};
  if (r) { rw[0] = r->x; rw[1] = r->y; rw[2] = r->z; rw[3] = r->w; }
  uint32_t ow[4], nw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s0 = bf_lo(hw[j]), s1 = bf_hi(hw[j]);
    if (r) { s0 += bf_lo(rw[j]); s1 += bf_hi(rw[j]); }
    if (round_sum) { s0 = bf16_round_f(s0); s1 = bf16_round_f(s1); }
    nw[j] = pack_bf2(s0, s1);
    ow[j] = pack_bf2(__fmul_rn(__fmul_rn(s0, inv), __fadd_rn(bias, bf_lo(gw[j]))),
                     __fmul_rn(__fmul_rn(s1, inv), __fadd_rn(bias, bf_hi(gw[j]))));
  }
  o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
  n.x = nw[0]; n.y = nw[1]; n.z = nw[2]; n.w = nw[3];
  if (nh) *nh = n;
  return o;
}

// (round 6) One row's residual add + RMSNorm with the row held in REGISTERS between the sum of squares and the scaling pass (d / 8 <=
// 64 * VPL vectors, lane l holds vectors l, l + 64, ... - the canonical order above, so the same bits as the two-pass kernels).  For
// the long-prompt rows kernels: with 10 000 rows in flight the second pass of the two-pass form misses the XCD's L2 and goes back to
// memory (FETCH_SIZE 198 MB per launch for 102 MB of inputs).  ROUND = the norm over the bf16-rounded sum ("add, then norm").
template <int VPL, bool ROUND>
__device__ __forceinline__ void wave_add_norm_row_cached(const Half* hr, const Half* rr, const Half* __restrict__ w, Half* hout,
                                                         Half* nout, int d, float eps, float bias) {
  const int lane = threadIdx.x & 63, nvec = d >> 3;
  u32x4 hv[VPL], rv[VPL];
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const int i = lane + 64 * u;
    hv[u] = rv[u] = u32x4{0u, 0u, 0u, 0u};
    if (i < nvec) {
      hv[u] = reinterpret_cast<const u32x4*>(hr)[i];
      rv[u] = reinterpret_cast<const u32x4*>(rr)[i];
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    if (lane + 64 * u >= nvec) continue;
    if (ROUND) add_round_sq8(hv[u], rv[u], ss);
    else add_sq8(hv[u], rv[u], ss);
  }
  ss = wave_sum(ss);
  const float inv = rsqrtf(__fadd_rn(ss / (float)d, eps));
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const int i = lane + 64 * u;
    if (i >= nvec) continue;
    u32x4 nh;
    const u32x4 o = norm_scale8(hv[u], &rv[u], reinterpret_cast<const u32x4*>(w)[i], inv, bias, &nh, ROUND);
    reinterpret_cast<u32x4*>(hout)[i] = nh;
    reinterpret_cast<u32x4*>(nout)[i] = o;
  }
}

}  // namespace pk
