// Batched-decode ("skinny") GEMM for gfx950: 2 <= T <= 64 token columns, weights streamed ONCE.
//
//   Y[T, M] = X[T, K] . W[M, K]^T        (same contract / prologue / epilogue options as gemv_core.h)
//
// At T >= 4 the dot2 GEMV turns VALU-bound (v_dot2c is a quarter-rate op), so the contraction moves to the
// matrix cores while the kernel stays HBM-bound: one workgroup = 8 waves = one 16-row block of W; wave w owns
// the 64-wide K step PAIRS p == w (mod 8) (both 64-byte halves of a row's 128-byte line are fetched by the same
// wave back to back) - a mapping that does not depend on T or on the tile width, so every column's result is
// bit-identical for any batch size routed here.  Per K step a wave issues ONE fragment-shaped
// 16-byte load per lane (16 rows x 64 B, straight HBM -> VGPR, non-temporal; all loads of a tile are in
// flight before the x tile is even staged), reads the NB x-fragments from LDS (x tile [T][KT] with the
// 16-byte chunk index XOR-swizzled by token so the 16 token rows hit 16 different slots) and issues NB
// v_mfma_f32_16x16x32_bf16.  The 8 partial accumulators are combined through LDS in fixed wave order.
// MFMA utilisation is tiny on purpose - the op is priced against the HBM roofline, not the MFMA one.
#pragma once

#include "common.h"
#include "gemv_core.h"
#include "norm_core.h"

namespace pk {

constexpr int kSkinnyWaves = 8;
// LDS bytes of the x tile region: T staged rows, but never less than the cross-wave reduction buffer
// ([8 waves][<=2 weight sets][NB][64 lanes] f32x4) that reuses the same memory after the K loop.
__host__ __device__ inline int skinny_xs_bytes(int NB, int T, int KT) {
  const int x = T * KT * 2, r = 8 * 2 * NB * 64 * 16;
  return ((x > r ? x : r) + 15) & ~15;
}
constexpr int kSkinnyThreads = kSkinnyWaves * 64;

// x tile [T][KT]: only the T real token rows are staged; KT = largest multiple of 512 (8 pairs) with
// T*KT*2 <= 64 KB, capped at 2048 (<= 8 fragment loads per weight set in flight per wave per tile).
inline int skinny_pick_kt(int T, int K) {
  int kt = (64 * 1024 / (T * 2)) / 512 * 512;
  kt = kt > 2048 ? 2048 : (kt < 512 ? 512 : kt);
  const int kr = (K + 511) / 512 * 512;
  return kt > kr ? kr : kt;
}

// RB = 16-row blocks of W per workgroup.  The x tile width shrinks as T grows (T*KT*2 <= 64 KB), so at T = 32 / 64 a
// wave has only 4 / 2 K steps of one row block per tile: too few loads in flight, and x re-staged by every 16 rows.
// RB = 2 / 4 row blocks per workgroup keep 8 fragment loads per weight set in flight per wave and cut the x
// staging traffic by RB.  The pair -> wave mapping and the per-wave accumulation order are unchanged, so results
// are bit-identical to RB = 1.
template <int NB, int EPI, int RB>
__global__ __launch_bounds__(kSkinnyThreads) void skinny_mfma_kernel(const GemvFusedArgs a) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  constexpr int JM = 8 / RB;  // K steps per wave per tile per row block; host guarantees KT <= JM * 256
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
  const int KT = a.KT, K = a.K, T = a.T;
  const int pitch = KT >> 3;  // 16-byte chunks per token row
  float* sm_inv = reinterpret_cast<float*>(smem_raw + (size_t)skinny_xs_bytes(NB, T, KT));  // [64] inverse RMS per token
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int rows_total = EPI == kEpiSilu ? a.I : a.M;
  const int row0 = blockIdx.x * 16 * RB;
  const Half* wptr[NW][RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int row = row0 + rb * 16 + l15;
    row = row < rows_total ? row : rows_total - 1;
#pragma unroll
    for (int s = 0; s < NW; ++s) wptr[s][rb] = a.W + ((size_t)row + (size_t)s * a.I) * K + g * 8;
  }

  f32x4 acc[NW][RB][NB];
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[s][rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: per-token inverse RMS, canonical one-wave-per-row order, 8 tokens at a time ----
  if (a.norm_w) {
    for (int t = wave; t < T; t += kSkinnyWaves) {
      const float v = wave_row_inv_rms(a.X + (size_t)t * K, a.residual ? a.residual + (size_t)t * K : nullptr, K, a.eps);
      if (lane == 0) sm_inv[t] = v;
    }
    __syncthreads();
  }

  for (int k0 = 0; k0 < K; k0 += KT) {
    const int kt = (K - k0) < KT ? (K - k0) : KT;
    const int nsteps = kt >> 5;  // K % 32 == 0 (dispatch guarantees)
    // ---- this tile's weight fragments: all loads leave before the x tile is staged ----
    u32x4 av[NW][RB][JM];
    const int jmax = KT >> 8;  // steps per wave in a full tile (uniform): no wasted loads on narrow tiles
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        if (j < jmax) {
          int s = 2 * (wave + kSkinnyWaves * (j >> 1)) + (j & 1);  // steps of pair wave + 8*(j/2)
          s = s < nsteps ? s : nsteps - 1;  // clamp on the last, shorter tile (result unused)
#pragma unroll
          for (int w = 0; w < NW; ++w)
            av[w][rb][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wptr[w][rb] + k0 + s * 32));
        }
      }
    // ---- stage x[:, k0:k0+kt] (optionally normalised on the fly) ----
    __syncthreads();
    const int nvec = kt >> 3;
    for (int idx = tid; idx < T * nvec; idx += kSkinnyThreads) {
      const int t = idx / nvec, c = idx - t * nvec;
      const size_t off = (size_t)t * K + k0 + c * 8;
      const u32x4 h = *reinterpret_cast<const u32x4*>(a.X + off);
      u32x4 v;
      if (a.norm_w) {
        const u32x4 gw = *reinterpret_cast<const u32x4*>(a.norm_w + k0 + c * 8);
        const float inv = sm_inv[t];
        if (a.residual) {
          const u32x4 r = *reinterpret_cast<const u32x4*>(a.residual + off);
          u32x4 nh;
          v = norm_scale8(h, &r, gw, inv, 0.f, &nh);
          if (blockIdx.x == 0) *reinterpret_cast<u32x4*>(a.hidden_out + off) = nh;
        } else {
          v = norm_scale8(h, nullptr, gw, inv, 0.f, nullptr);
        }
      } else {
        v = h;
      }
      xs[t * pitch + (c ^ (t & 15))] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int s = 2 * (wave + kSkinnyWaves * (j >> 1)) + (j & 1);
      if (j < jmax && s < nsteps) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          int t = nb * 16 + l15;
          t = t < T ? t : T - 1;  // absent token columns re-read a staged row; their results are never stored
          const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xs[t * pitch + ((s * 4 + g) ^ l15)]);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int w = 0; w < NW; ++w)
              acc[w][rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[w][rb][j]), b,
                                                                       acc[w][rb][nb], 0, 0, 0);
        }
      }
    }
  }

  // ---- combine the 8 waves (fixed order) and store, one row block at a time through the same LDS buffer ----
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);  // [wave][NW][NB][64 lanes]
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) red[((wave * NW + w) * NB + nb) * 64 + lane] = acc[w][rb][nb];
    __syncthreads();
    for (int e = tid; e < NB * 64; e += kSkinnyThreads) {
      const int nb = e >> 6, ln = e & 63;
      f32x4 tot[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        f32x4 v = red[((0 * NW + w) * NB + nb) * 64 + ln];
#pragma unroll
        for (int wv = 1; wv < kSkinnyWaves; ++wv) v += red[((wv * NW + w) * NB + nb) * 64 + ln];
        tot[w] = v;
      }
      const int t = nb * 16 + (ln & 15);
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = row0 + rb * 16 + (ln >> 4) * 4 + i;
        if (r >= rows_total) continue;
        if (EPI == kEpiSilu) {
          const float gt = bf16_round_f(tot[0][i]), up = bf16_round_f(tot[NW - 1][i]);
          a.Y[(size_t)t * a.I + r] = f2bf(silu_f(gt) * up);
        } else {
          a.Y[(size_t)t * a.M + r] = f2bf(tot[0][i]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Resident-x variant (T*K*2 <= kSkinnyResidentBytes): the whole (normalised) x block is staged ONCE per
// workgroup; the workgroup then persists over 16-row blocks rb = blockIdx.x, + gridDim.x, ... and every wave
// walks its K-step pairs of block after block as one flat stream of work items, always keeping the next chunk
// of fragment loads (CH pairs = 2*CH KB per weight set) in flight - no per-tile barrier, no re-staging.
// Same pair -> wave mapping and the same per-wave accumulation order as the tiled kernel: bit-identical.
// ---------------------------------------------------------------------------------------------------
constexpr int kSkinnyResidentBytes = 100 * 1024;

template <int NB, int EPI>
__global__ __launch_bounds__(kSkinnyThreads) void skinny_resident_kernel(const GemvFusedArgs a) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  constexpr int CH = NW == 1 ? 4 : 2;  // pairs per chunk: 8 (store) / 2x4 (silu) fragment loads in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int K = a.K, T = a.T;
  const int pitch = K >> 3;
  u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
  const int xs_bytes = (T * K * 2 + 15) & ~15;
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + xs_bytes);                          // [8][NW][NB][64]
  float* sm_inv = reinterpret_cast<float*>(smem_raw + xs_bytes + 8 * NW * NB * 64 * 16);  // [64]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int rows_total = EPI == kEpiSilu ? a.I : a.M;
  const int nrb = (rows_total + 15) >> 4;
  const int np = K >> 6;                       // 64-wide K step pairs (K % 64 == 0)
  const int my_np = (np + kSkinnyWaves - 1) / kSkinnyWaves;   // items per row block, same for every wave
  const int my_rb = (nrb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_rb * my_np;

  auto item_ptr = [&](int item, int w) -> const Half* {
    const int rbi = item / my_np, pi = item - rbi * my_np;
    int row = ((int)blockIdx.x + rbi * (int)gridDim.x) * 16 + l15;
    row = row < rows_total ? row : rows_total - 1;
    int pair = wave + kSkinnyWaves * pi;
    pair = pair < np ? pair : np - 1;          // clamped load of a valid address; skipped in the MFMA loop
    return a.W + ((size_t)row + (size_t)w * a.I) * K + pair * 64 + g * 8;
  };
  u32x4 avA[NW][CH][2], avB[NW][CH][2];
  auto issue = [&](int item0, u32x4 (&av)[NW][CH][2]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int item = item0 + c < total ? item0 + c : total - 1;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const Half* p = item_ptr(item, w);
        av[w][c][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        av[w][c][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 32));
      }
    }
  };
  if (total > 0) issue(0, avA);   // first chunk leaves before the prologue

  // ---- prologue + one-time staging of x ----
  if (a.norm_w) {
    for (int t = wave; t < T; t += kSkinnyWaves) {
      const float v = wave_row_inv_rms(a.X + (size_t)t * K, a.residual ? a.residual + (size_t)t * K : nullptr, K, a.eps);
      if (lane == 0) sm_inv[t] = v;
    }
    __syncthreads();
  }
  {
    const int nvec = K >> 3;
    for (int idx = tid; idx < T * nvec; idx += kSkinnyThreads) {
      const int t = idx / nvec, c = idx - t * nvec;
      const size_t off = (size_t)t * K + c * 8;
      const u32x4 h = *reinterpret_cast<const u32x4*>(a.X + off);
      u32x4 v;
      if (a.norm_w) {
        const u32x4 gw = *reinterpret_cast<const u32x4*>(a.norm_w + c * 8);
        const float inv = sm_inv[t];
        if (a.residual) {
          const u32x4 r = *reinterpret_cast<const u32x4*>(a.residual + off);
          u32x4 nh;
          v = norm_scale8(h, &r, gw, inv, 0.f, &nh);
          if (blockIdx.x == 0) *reinterpret_cast<u32x4*>(a.hidden_out + off) = nh;
        } else {
          v = norm_scale8(h, nullptr, gw, inv, 0.f, nullptr);
        }
      } else {
        v = h;
      }
      xs[t * pitch + (c ^ (t & 15))] = v;
    }
  }
  __syncthreads();

  f32x4 acc[NW][NB];
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[w][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto flush = [&](int rbi) {   // combine the 8 waves (fixed order), store, reset
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        red[((wave * NW + w) * NB + nb) * 64 + lane] = acc[w][nb];
        acc[w][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    __syncthreads();
    const int row0 = ((int)blockIdx.x + rbi * (int)gridDim.x) * 16;
    for (int e = tid; e < NB * 64; e += kSkinnyThreads) {
      const int nb = e >> 6, ln = e & 63;
      f32x4 tot[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        f32x4 v = red[((0 * NW + w) * NB + nb) * 64 + ln];
#pragma unroll
        for (int wv = 1; wv < kSkinnyWaves; ++wv) v += red[((wv * NW + w) * NB + nb) * 64 + ln];
        tot[w] = v;
      }
      const int t = nb * 16 + (ln & 15);
      if (t < T) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = row0 + (ln >> 4) * 4 + i;
          if (r < rows_total) {
            if (EPI == kEpiSilu) {
              const float gt = bf16_round_f(tot[0][i]), up = bf16_round_f(tot[NW - 1][i]);
              a.Y[(size_t)t * a.I + r] = f2bf(silu_f(gt) * up);
            } else {
              a.Y[(size_t)t * a.M + r] = f2bf(tot[0][i]);
            }
          }
        }
      }
    }
    __syncthreads();
  };
  auto compute = [&](int item0, const u32x4 (&av)[NW][CH][2]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int item = item0 + c;
      if (item < total) {
        const int rbi = item / my_np, pi = item - rbi * my_np;
        const int pair = wave + kSkinnyWaves * pi;
        if (pair < np) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int s = 2 * pair + h;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              int t = nb * 16 + l15;
              t = t < T ? t : T - 1;
              const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xs[t * pitch + ((s * 4 + g) ^ l15)]);
#pragma unroll
              for (int w = 0; w < NW; ++w)
                acc[w][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[w][c][h]), b,
                                                                     acc[w][nb], 0, 0, 0);
            }
          }
        }
        if (pi == my_np - 1) flush(rbi);   // uniform: every wave reaches the same items in the same order
      }
    }
  };
  for (int item0 = 0; item0 < total; item0 += 2 * CH) {
    if (item0 + CH < total) issue(item0 + CH, avB);
    compute(item0, avA);
    if (item0 + CH >= total) break;
    if (item0 + 2 * CH < total) issue(item0 + 2 * CH, avA);
    compute(item0 + CH, avB);
  }
}

template <int NB, int EPI>
inline void skinny_launch_resident(GemvFusedArgs a, hipStream_t s) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  a.KT = a.K;
  const int rows = EPI == kEpiSilu ? a.I : a.M;
  const int nrb = ceil_div(rows, 16);
  const int lds = ((a.T * a.K * 2 + 15) & ~15) + 8 * NW * NB * 64 * 16 + 64 * 4;
  auto kern = &skinny_resident_kernel<NB, EPI>;
  static const bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  static thread_local int cached_lds = -1, cached_cap = 0;
  if (cached_lds != lds) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kSkinnyThreads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cached_cap = per_cu * (cus > 0 ? cus : 256);
    cached_lds = lds;
  }
  int grid = nrb;
  if (nrb > cached_cap) grid = ceil_div(nrb, ceil_div(nrb, cached_cap));   // equal number of row blocks each
  kern<<<grid, kSkinnyThreads, lds, s>>>(a);
}

template <int NB, int EPI, int RB>
inline void skinny_launch_rb(const GemvFusedArgs& a, hipStream_t s) {
  const int rows = EPI == kEpiSilu ? a.I : a.M;
  const int lds = skinny_xs_bytes(NB, a.T, a.KT) + (64 + 4) * 4;
  auto kern = &skinny_mfma_kernel<NB, EPI, RB>;
  static const bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  kern<<<ceil_div(rows, 16 * RB), kSkinnyThreads, lds, s>>>(a);
}
template <int NB, int EPI>
inline void skinny_launch(GemvFusedArgs a, hipStream_t s) {
  a.KT = skinny_pick_kt(a.T, a.K);
  // row blocks per workgroup: keep ~8 K steps per wave per tile in flight (PEGAINFER_SKINNY_RB=1 forces 1: A/B)
  static const bool rb1 = [] { const char* e = getenv("PEGAINFER_SKINNY_RB"); return e && e[0] == '1'; }();
  const int rows = EPI == kEpiSilu ? a.I : a.M;
  // largest RB the tile allows (KT <= (8/RB)*256; the SwiGLU form streams two weight sets, so RB <= 2 there keeps
  // it under 256 VGPRs) that still leaves at least one workgroup per CU
  int rb = a.KT >= 2048 || rb1 ? 1 : a.KT >= 1024 ? 2 : 4;
  if (EPI == kEpiSilu && rb > 2) rb = 2;
  while (rb > 1 && ceil_div(rows, 16 * rb) < 256) rb >>= 1;
  if (rb == 4) { skinny_launch_rb<NB, EPI, 4>(a, s); }
  else if (rb == 2) { if (a.KT > 1024) a.KT = 1024; skinny_launch_rb<NB, EPI, 2>(a, s); }
  else skinny_launch_rb<NB, EPI, 1>(a, s);
}

// 2 <= T <= 64, K % 32 == 0
template <int EPI>
inline bool skinny_dispatch(const GemvFusedArgs& a, hipStream_t s) {
  if (a.T < 2 || a.T > 64 || (a.K & 31) != 0) return false;
  const bool resident = (a.K & 63) == 0 && (long)a.T * a.K * 2 <= kSkinnyResidentBytes;
  if (a.T <= 16) { if (resident) skinny_launch_resident<1, EPI>(a, s); else skinny_launch<1, EPI>(a, s); }
  else if (a.T <= 32) { if (resident) skinny_launch_resident<2, EPI>(a, s); else skinny_launch<2, EPI>(a, s); }
  else { if (resident) skinny_launch_resident<4, EPI>(a, s); else skinny_launch<4, EPI>(a, s); }
  return true;
}

}  // namespace pk
