// Canonical per-head RMSNorm + NeoX RoPE for head_dim = 128 on a 16-lane DPP row: lane `sub` owns the
// 8 contiguous dims sub*8..sub*8+7 (one 16-byte load), lanes 0-7 hold the first half of the head and
// lanes 8-15 the second, so the RoPE partner (d, d+64) sits in lane sub^8 at the same element index
// (DPP row_ror:8, no LDS).  Shared by qk_norm_rope.hip and the fused decode attention so both round
// identically.  Rounding sequence = reference csrc/prefill_attention.cu:55-84:
//   n = bf16(x*inv_rms); m = bf16(n*w); out_lo = bf16(m_lo*c - m_hi*s); out_hi = bf16(m_lo*s + m_hi*c)
#pragma once

#include "common.h"

namespace pk {

constexpr int kDppRowRor8 = 0x128;

__device__ __forceinline__ u32x4 head_norm_rope16(const u32x4& x, const Half* __restrict__ w,
                                                  const Half* __restrict__ cos_row,
                                                  const Half* __restrict__ sin_row, int sub, float eps) {
  const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
  float v[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf_lo(xw[j]); v[2 * j + 1] = bf_hi(xw[j]); }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ss = fmaf(v[j], v[j], ss);
  ss = row16_sum(ss);
  const float inv = rsqrtf(__fadd_rn(ss / 128.0f, eps));
  const u32x4 wv = *reinterpret_cast<const u32x4*>(w + sub * 8);
  const u32x4 cv = *reinterpret_cast<const u32x4*>(cos_row + (sub & 7) * 8);
  const u32x4 sv = *reinterpret_cast<const u32x4*>(sin_row + (sub & 7) * 8);
  const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w}, cw[4] = {cv.x, cv.y, cv.z, cv.w}, sw[4] = {sv.x, sv.y, sv.z, sv.w};
  const bool first_half = sub < 8;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float wj = (j & 1) ? bf_hi(ww[j >> 1]) : bf_lo(ww[j >> 1]);
    const float c = (j & 1) ? bf_hi(cw[j >> 1]) : bf_lo(cw[j >> 1]);
    const float s = (j & 1) ? bf_hi(sw[j >> 1]) : bf_lo(sw[j >> 1]);
    const float m = bf16_round_f(__fmul_rn(bf16_round_f(__fmul_rn(v[j], inv)), wj));
    const float pm = dpp_mov<kDppRowRor8>(m);  // partner lane sub^8
    // named fused forms: a difference / sum of two products can be contracted either way, and every caller of this
    // core must round identically
    o[j] = first_half ? fmaf(m, c, -__fmul_rn(pm, s)) : fmaf(pm, s, __fmul_rn(m, c));
  }
  u32x4 r;
  r.x = pack_bf2(o[0], o[1]); r.y = pack_bf2(o[2], o[3]); r.z = pack_bf2(o[4], o[5]); r.w = pack_bf2(o[6], o[7]);
  return r;
}

}  // namespace pk
