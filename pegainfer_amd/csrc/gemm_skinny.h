// Batched-decode ("skinny") GEMM for gfx950: 2 <= T <= 64 token columns, weights streamed ONCE.
//
//   Y[T, M] = X[T, K] . W[M, K]^T        (same contract / prologue / epilogue options as gemv_core.h)
//
// At T >= 4 the dot2 GEMV turns VALU-bound (v_dot2c is a quarter-rate op), so the contraction moves to the
// matrix cores while the kernel stays HBM-bound: one workgroup = 8 waves = one 16-row block of W; wave w owns
// the 64-wide K "pairs" p == w (mod 8) (128 bytes of every row) - a mapping that does not depend on T or on the
// tile width, so every column's result is bit-identical for any batch size routed here.
//
// Load shape (measured, tools/bench_skinny.py): what the HBM stream cares about is that ONE load instruction
// covers whole 128-byte lines.  The natural MFMA A-fragment load (lane = row, 16 B per lane: 16 rows x 64 B per
// instruction, half a line per row) streams at 4.8 TB/s on the 778 MB lm_head and 2-3 TB/s on the layer
// matrices; 8 rows x 128 B per instruction streams at 6.9 TB/s / 3.5-5.7 TB/s.  So a weight register is loaded as
//     row-in-8 r = (lane >> 1) & 7,   16-byte chunk of the row's 128-byte segment c = 2 * (lane >> 4) + (lane & 1)
// and fed to v_mfma_f32_16x16x32_bf16 unpermuted: as an A operand (row index i = lane & 15, k group g = lane >> 4)
// it is 16 "rows" i = 2 r + h, where h = chunk parity and row i carries the chunks {2 g + h}.  The MFMA against
// the x fragment of parity h (x chunks 2 g + h) is therefore exact in the rows of that parity and garbage in the
// others: two MFMAs and two accumulators (even / odd) per register, and the result of real row r is
// acc_even[i = 2 r] + acc_odd[i = 2 r + 1] - both in the same lane of the D layout, so no cross-lane traffic.
// Half of the MFMA work is discarded on purpose: the op is priced against the HBM roofline, not the MFMA one.
//
// Row blocks are at most 16 rows high, but not necessarily 16 (round 4): a launch whose 16-row blocks would not deal
// evenly onto the 256 CUs takes the block height that does (2560 rows: 160 blocks of 16 -> 256 blocks of 10; a streaming
// CU ingests ~24 GB/s, so 160 busy CUs cap the launch at 3.8 TB/s).  Rows past a block's height are clamped duplicates
// of its last row in the loads (same cache lines, no extra HBM traffic) and never stored; a row's K order does not
// depend on the block it sits in, so the bits do not change.
//
// x sits in LDS ([T][KT] tile, or the whole [T][K] block in the resident variant) with the 16-byte chunk index
// XOR-swizzled by token so the 16 token rows of a fragment read hit 16 different bank groups.  The 8 waves'
// partial sums meet in LDS and are added in fixed wave order.
#pragma once

#include <cstdio>

#include <cstdlib>

#include "common.h"
#include "gemv_core.h"
#include "norm_core.h"

namespace pk {

constexpr int kSkinnyWaves = 8;
constexpr int kSkinnyThreads = kSkinnyWaves * 64;

// LDS bytes of the x tile region: T staged rows, but never less than the cross-wave reduction buffer
// ([8 waves][<=2 weight sets][NB][64 lanes] f32x4) that reuses the same memory after the K loop.
__host__ __device__ inline int skinny_xs_bytes(int NB, int T, int KT) {
  const int x = T * KT * 2, r = 8 * 2 * NB * 64 * 16;
  return ((x > r ? x : r) + 15) & ~15;
}

// x tile [T][KT]: only the T real token rows are staged; KT = largest multiple of 512 (8 pairs) with
// T*KT*2 <= 64 KB, capped at 2048 (<= 8 weight loads per weight set in flight per wave per tile).
inline int skinny_pick_kt(int T, int K) {
  int kt = (64 * 1024 / (T * 2)) / 512 * 512;
  kt = kt > 2048 ? 2048 : (kt < 512 ? 512 : kt);
  const int kr = (K + 511) / 512 * 512;
  return kt > kr ? kr : kt;
}

// ---- shared pieces ----
struct SkinnyLane {
  int wave, lane, l15, g;   // MFMA view: token / row index, k group
  int prow, pchunk;         // load view: row within the 8-row register, 16-byte chunk of the 128-byte segment
  __device__ SkinnyLane() {
    wave = threadIdx.x >> 6; lane = threadIdx.x & 63; l15 = lane & 15; g = lane >> 4;
    prow = (lane >> 1) & 7; pchunk = 2 * g + (lane & 1);
  }
};
// acc[2 row halves][2 parities] of one 16-row block -> the wave's partial sums in "row slot" order:
// slot i of lane ln = row (i >> 1) * 8 + 2 * (ln >> 4) + (i & 1) of the block, token ln & 15
__device__ inline f32x4 skinny_fold(const f32x4 (&acc)[2][2]) {
  return f32x4{acc[0][0][0] + acc[0][1][1], acc[0][0][2] + acc[0][1][3],
               acc[1][0][0] + acc[1][1][1], acc[1][0][2] + acc[1][1][3]};
}
__device__ inline int skinny_slot_row(int ln, int i) { return (i >> 1) * 8 + 2 * (ln >> 4) + (i & 1); }

// Combine the 8 waves' folded partials of one row block (fixed order) and store.  red: [8][NW][NB][64] f32x4.
template <int NB, int EPI>
__device__ inline void skinny_reduce_store(const GemvFusedArgs& a, const f32x4* red, int row0, int rows_total, int tid,
                                           int nthreads = kSkinnyThreads) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  for (int e = tid; e < NB * 64; e += nthreads) {
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
    if (t >= a.T) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + skinny_slot_row(ln, i);
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

// Stage x[0:T, k0:k0+kt] (optionally normalised on the fly) into the swizzled tile: wave w takes the token rows
// t == w (mod 8), a lane the 16-byte vectors lane, lane + 64, ... of the row, four at a time so that the global
// loads of a batch are all in flight before the first LDS write (the one-vector-at-a-time loop cost 2.5-7 us).
template <int B, bool NORM>
__device__ inline void skinny_stage_rows(const GemvFusedArgs& a, u32x4* xs, const float* sm_inv, int pitch, int k0,
                                         int kt, int wave, int lane) {
  const int nvec = kt >> 3, T = a.T, K = a.K;
  for (int t = wave; t < T; t += kSkinnyWaves) {
    const size_t row = (size_t)t * K + k0;
    const float inv = NORM ? sm_inv[t] : 0.f;
    for (int c0 = lane; c0 < nvec; c0 += 64 * B) {
      u32x4 h[B], r[B], gw[B];
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const int c = c0 + 64 * j;
        if (c < nvec) {
          h[j] = *reinterpret_cast<const u32x4*>(a.X + row + c * 8);
          if (NORM) gw[j] = *reinterpret_cast<const u32x4*>(a.norm_w + k0 + c * 8);
          if (NORM && a.residual) r[j] = *reinterpret_cast<const u32x4*>(a.residual + row + c * 8);
        }
      }
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const int c = c0 + 64 * j;
        if (c < nvec) {
          u32x4 v = h[j];
          if (NORM) {
            if (a.residual) {
              u32x4 nh;
              v = norm_scale8(h[j], &r[j], gw[j], inv, 0.f, &nh, (a.flags & kGemvRoundSum) != 0);
              if (blockIdx.x == 0) *reinterpret_cast<u32x4*>(a.hidden_out + row + c * 8) = nh;
            } else {
              v = norm_scale8(h[j], nullptr, gw[j], inv, 0.f, nullptr);
            }
          }
          xs[t * pitch + (c ^ (t & 15))] = v;
        }
      }
    }
  }
}
template <int NB>
__device__ inline void skinny_stage_x(const GemvFusedArgs& a, u32x4* xs, const float* sm_inv, int pitch, int k0, int kt,
                                      int wave, int lane) {
  if (a.norm_w) skinny_stage_rows<2, true>(a, xs, sm_inv, pitch, k0, kt, wave, lane);
  else skinny_stage_rows<(NB >= 4 ? 2 : 4), false>(a, xs, sm_inv, pitch, k0, kt, wave, lane);
}

// ---------------------------------------------------------------------------------------------------
// Tiled kernel: x does not fit in LDS as a whole; K is walked in KT-wide tiles (stage, barrier, compute).
// RB = 16-row blocks of W per workgroup.  The x tile width shrinks as T grows (T*KT*2 <= 64 KB), so at T = 32 / 64 a
// wave has only 2 / 1 pairs of one row block per tile: RB = 2 / 4 row blocks per workgroup keep 8 weight loads
// per weight set in flight per wave and cut the x staging traffic by RB.  The pair -> wave mapping and the per-wave
// accumulation order do not depend on RB: bit-identical results.
// ---------------------------------------------------------------------------------------------------
template <int NB, int EPI, int RB>
__global__ __launch_bounds__(kSkinnyThreads) void skinny_mfma_kernel(const GemvFusedArgs a) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  constexpr int JP = 4 / RB;  // pairs per wave per tile per row block; host guarantees KT <= JP * 512
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
  const int KT = a.KT, K = a.K, T = a.T;
  const int pitch = KT >> 3;  // 16-byte chunks per token row
  float* sm_inv = reinterpret_cast<float*>(smem_raw + (size_t)skinny_xs_bytes(NB, T, KT));  // [64] inverse RMS per token
  const SkinnyLane L;
  const int tid = threadIdx.x;
  const int rpb = RB == 1 && a.rpb ? a.rpb : 16;   // rows per row block (launcher: skinny_pick_rpb)
  const int row0 = blockIdx.x * rpb * RB;
  const int rows_all = EPI == kEpiSilu ? a.I : a.M;
  const int rows_total = RB == 1 ? (row0 + rpb < rows_all ? row0 + rpb : rows_all) : rows_all;   // this block's row limit
  const Half* wptr[NW][RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      int row = row0 + rb * 16 + rh * 8 + L.prow;
      row = row < rows_total ? row : rows_total - 1;
#pragma unroll
      for (int s = 0; s < NW; ++s) wptr[s][rb][rh] = a.W + ((size_t)row + (size_t)s * a.I) * K + L.pchunk * 8;
    }

  f32x4 acc[NW][RB][NB][2][2];
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[s][rb][nb][q >> 1][q & 1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: per-token inverse RMS, canonical one-wave-per-row order, 8 tokens at a time ----
  if (a.norm_w) {
    for (int t = L.wave; t < T; t += kSkinnyWaves) {
      const float v = wave_row_inv_rms(a.X + (size_t)t * K, a.residual ? a.residual + (size_t)t * K : nullptr, K, a.eps,
                                       (a.flags & kGemvRoundSum) != 0);
      if (L.lane == 0) sm_inv[t] = v;
    }
    __syncthreads();
  }

  // The next tile's weight loads leave BEFORE the current tile is multiplied (second register set), whenever the
  // accumulators leave room for it: the stream then never drains between tiles.
  constexpr bool kPipe = NB < 4 && NW * RB * NB * 16 + 2 * NW * 32 <= 128;
  const int jmax = KT >> 9;  // pairs per wave in a full tile (uniform): no wasted loads on narrow tiles
  u32x4 avA[NW][RB][JP][2], avB[NW][RB][JP][2];
  auto issue = [&](int k0, u32x4 (&av)[NW][RB][JP][2]) {
    const int kt = (K - k0) < KT ? (K - k0) : KT;
    const int npairs = kt >> 6;  // K % 64 == 0 (dispatch guarantees)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < JP; ++j) {
        if (j < jmax) {
          int p = L.wave + kSkinnyWaves * j;
          p = p < npairs ? p : npairs - 1;  // clamp on the last, shorter tile (result unused)
#pragma unroll
          for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
              av[w][rb][j][rh] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wptr[w][rb][rh] + k0 + p * 64));
        }
      }
  };
  auto compute = [&](int k0, const u32x4 (&av)[NW][RB][JP][2]) {
    const int kt = (K - k0) < KT ? (K - k0) : KT;
    const int npairs = kt >> 6;
#pragma unroll
    for (int j = 0; j < JP; ++j) {
      const int p = L.wave + kSkinnyWaves * j;
      if (j < jmax && p < npairs) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          int t = nb * 16 + L.l15;
          t = t < T ? t : T - 1;  // absent token columns re-read a staged row; their results are never stored
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xs[t * pitch + ((p * 8 + 2 * L.g + h) ^ L.l15)]);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
              for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
                  acc[w][rb][nb][rh][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      __builtin_bit_cast(bf16x8_t, av[w][rb][j][rh]), b, acc[w][rb][nb][rh][h], 0, 0, 0);
          }
        }
      }
    }
  };
  auto stage = [&](int k0) {
    const int kt = (K - k0) < KT ? (K - k0) : KT;
    __syncthreads();   // the previous tile's fragment reads are done
    skinny_stage_x<NB>(a, xs, sm_inv, pitch, k0, kt, L.wave, L.lane);
    __syncthreads();
  };
  // Plain x, <= 16 token columns, 2048-wide tiles (down_proj at 7..16 requests): the x tiles travel by LDS-DMA into
  // TWO tile buffers, each tile's DMA issued BEFORE the weight loads of the same tile (loads retire in order: an x tile
  // staged behind two tiles of in-flight weights waited for all of them, every tile).  Every wave issues exactly
  // kDmaPerTile DMAs per tile (absent token rows / pieces past a short last tile re-send a valid piece), so the waits
  // are counted: x(k) has landed when at most W(k) [+ x(k+1) + W(k+1)] are outstanding.
  constexpr bool kDmaForm = kPipe && NB == 1 && RB == 1 && NW == 1;
  constexpr int kDmaPerTile = 2 * 4;   // 2 token rows per wave x 4 pieces of 1 KiB
  if constexpr (kDmaForm) {
    if (!a.norm_w && KT == 2048 && (K & 511) == 0 && T <= 2 * kSkinnyWaves) {   // launcher doubled the tile region
      typedef __attribute__((address_space(3))) void* lds_ptr_t;
      const uint32_t xs_lds = (uint32_t)(uintptr_t)(lds_ptr_t)xs;
      const uint32_t buf_bytes = (uint32_t)skinny_xs_bytes(NB, T, KT);
      u32x4* xbuf[2] = {xs, reinterpret_cast<u32x4*>(smem_raw + buf_bytes)};
      auto dma = [&](int k0, int b) {
        const int kt = (K - k0) < KT ? (K - k0) : KT;
        const int pieces = kt >> 9;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int t = L.wave + u * kSkinnyWaves;
          t = t < T ? t : T - 1;
#pragma unroll
          for (int q0 = 0; q0 < 4; ++q0) {
            const int q = q0 < pieces ? q0 : pieces - 1;
            const Half* src = a.X + (size_t)t * K + k0 + (size_t)(((q << 6) + L.lane) ^ (t & 15)) * 8;
            const uint32_t dst = xs_lds + (uint32_t)b * buf_bytes + (uint32_t)(t * pitch + (q << 6)) * 16u;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory", "m0");
          }
        }
      };
      auto compute_b = [&](int k0, const u32x4 (&av)[NW][RB][JP][2], const u32x4* xb) {
        const int kt = (K - k0) < KT ? (K - k0) : KT;
        const int npairs = kt >> 6;
#pragma unroll
        for (int j = 0; j < JP; ++j) {
          const int p = L.wave + kSkinnyWaves * j;
          if (p < npairs) {
            int t = L.l15;
            t = t < T ? t : T - 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xb[t * pitch + ((p * 8 + 2 * L.g + h) ^ L.l15)]);
#pragma unroll
              for (int rh = 0; rh < 2; ++rh)
                acc[0][0][0][rh][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, av[0][0][j][rh]), b, acc[0][0][0][rh][h], 0, 0, 0);
            }
          }
        }
      };
      constexpr int kW = NW * RB * JP * 2;   // weight loads per tile and wave (8)
      dma(0, 0);
      issue(0, avA);
      if (KT < K) { dma(KT, 1); issue(KT, avB); }
      for (int k0 = 0; k0 < K; k0 += 2 * KT) {
        if (k0 + KT < K) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kW + kDmaPerTile) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kW) : "memory");
        __syncthreads();                       // x(k0) of every wave is in buffer 0
        compute_b(k0, avA, xbuf[0]);
        __syncthreads();                       // buffer 0 is free again
        if (k0 + 2 * KT < K) { dma(k0 + 2 * KT, 0); issue(k0 + 2 * KT, avA); }
        if (k0 + KT >= K) break;
        if (k0 + 2 * KT < K) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kW + kDmaPerTile) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kW) : "memory");
        __syncthreads();
        compute_b(k0 + KT, avB, xbuf[1]);
        __syncthreads();
        if (k0 + 3 * KT < K) { dma(k0 + 3 * KT, 1); issue(k0 + 3 * KT, avB); }
      }
      goto skinny_tiles_done;
    }
  }
  if constexpr (kPipe) {
    issue(0, avA);   // the loads of the first TWO tiles leave before the first x tile is staged; after that a
    if (KT < K) issue(KT, avB);   // register set is refilled (two tiles ahead) as soon as its tile is multiplied
    for (int k0 = 0; k0 < K; k0 += 2 * KT) {
      stage(k0);
      compute(k0, avA);
      if (k0 + 2 * KT < K) issue(k0 + 2 * KT, avA);
      if (k0 + KT >= K) break;
      stage(k0 + KT);
      compute(k0 + KT, avB);
      if (k0 + 3 * KT < K) issue(k0 + 3 * KT, avB);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += KT) {
      issue(k0, avA);
      stage(k0);
      compute(k0, avA);
    }
  }

skinny_tiles_done:
  // ---- combine the 8 waves (fixed order) and store, one row block at a time through the same LDS buffer ----
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);  // [wave][NW][NB][64 lanes]
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) red[((L.wave * NW + w) * NB + nb) * 64 + L.lane] = skinny_fold(acc[w][rb][nb]);
    __syncthreads();
    skinny_reduce_store<NB, EPI>(a, red, row0 + rb * 16, rows_total, tid);
  }
}

// ---------------------------------------------------------------------------------------------------
// Resident-x variant (T*K*2 <= kSkinnyResidentBytes): the whole (normalised) x block is staged ONCE per
// workgroup; the workgroup then persists over 16-row blocks rb = blockIdx.x, + gridDim.x, ... and every wave
// walks its pairs of block after block as one flat stream of work items, always keeping the next chunk of
// weight loads (CH pairs = 2*CH KB per weight set) in flight - no per-tile barrier, no re-staging.
// Same pair -> wave mapping and the same per-wave accumulation order as the tiled kernel: bit-identical.
// ---------------------------------------------------------------------------------------------------
constexpr int kSkinnyResidentBytes = 128 * 1024;   // o_proj at 16 columns (16 x 4096 x 2 B) is the largest resident x

template <int NB, int EPI>
__global__ __launch_bounds__(kSkinnyThreads) void skinny_resident_kernel(const GemvFusedArgs a) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  constexpr int CH = NW == 1 ? 4 : 2;  // pairs per chunk: 8 (store) / 2x4 (silu) weight loads in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int K = a.K, T = a.T;
  const int pitch = K >> 3;
  u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
  const int xs_bytes = (T * K * 2 + 15) & ~15;
  // [8][NW][NB][64] per buffer.  One buffer in the two-barrier flush form; two in the one-barrier form (a.variant bit 0: row
  // block n's partials go to buffer n & 1); a ring of 2..4 in the ticket form (bit 2, ring size in bits 4..7: block n -> buffer n % ring)
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + xs_bytes);
  constexpr int kRedQuads = 8 * NW * NB * 64;
  const bool flush1 = (a.variant & 1) != 0;
  const bool ticket = NB == 1 && (a.variant & 4) != 0;   // <= 16 columns only (the launcher never asks for it above: registers)
  const int ring = ticket ? (a.variant >> 4) & 15 : (flush1 ? 2 : 1);
  float* sm_inv = reinterpret_cast<float*>(smem_raw + xs_bytes + ring * kRedQuads * 16);  // [64]
  int* sm_cnt = reinterpret_cast<int*>(sm_inv + 64);   // ticket form: [ring] arrivals, [ring] finished reductions (monotonic)
  const SkinnyLane L;
  if (ticket && threadIdx.x < 2 * ring) sm_cnt[threadIdx.x] = 0;   // every prologue form below has a barrier before the K loop
  const int tid = threadIdx.x;
  const int rows_total = EPI == kEpiSilu ? a.I : a.M;
  const int rpb = a.rpb ? a.rpb : 16;          // rows per row block (launcher: skinny_pick_rpb)
  const int nrb = (rows_total + rpb - 1) / rpb;
  const int np = K >> 6;                       // 64-wide K pairs (K % 64 == 0)
  const int my_np = (np + kSkinnyWaves - 1) / kSkinnyWaves;   // items per row block, same for every wave
  const int my_rb = (nrb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_rb * my_np;

  u32x4 avA[NW][CH][2], avB[NW][CH][2];
  auto issue = [&](int item0, u32x4 (&av)[NW][CH][2]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int item = item0 + c < total ? item0 + c : total - 1;
      const int rbi = item / my_np, pi = item - rbi * my_np;
      const int blk0 = ((int)blockIdx.x + rbi * (int)gridDim.x) * rpb;
      const int lim = blk0 + rpb < rows_total ? blk0 + rpb : rows_total;   // rows of this block: [blk0, lim)
      const int row0 = blk0 + L.prow;
      int pair = L.wave + kSkinnyWaves * pi;
      pair = pair < np ? pair : np - 1;          // clamped load of a valid address; skipped in the MFMA loop
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        int row = row0 + rh * 8;
        row = row < lim ? row : lim - 1;
#pragma unroll
        for (int w = 0; w < NW; ++w)
          av[w][c][rh] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(
              a.W + ((size_t)row + (size_t)w * a.I) * K + pair * 64 + L.pchunk * 8));
      }
    }
  };
  // ---- prologue + one-time staging of x.  A wave's loads retire in order: x-side loads issued behind the first
  //      weight chunks cannot be consumed before those have landed from cold HBM (the finding of the GEMV phase
  //      trace), so wherever it fits the x side goes FIRST:
  //        plain x (o_proj, down_proj): rows -> swizzled LDS by LDS-DMA (lane l of piece q fetches chunk
  //          (64 q + l) ^ (t & 15), which is what the swizzled slot 64 q + l holds), weights behind, one counted wait;
  //        add + RMSNorm (qkv, gate_up; K <= 2560, T <= 16): wave w's token rows w, w + 8 (+ residual, + norm weight)
  //          into registers first, weights behind; the wave holds its rows in the canonical distribution of
  //          norm_core.h (lane l: vectors l, l + 64, ...), so the sum of squares needs no second pass - same bits.
  // kGemvRoundSum (the only flag these kernels take): "add, then norm" - hidden_out = bf16(X + residual) and the norm
  // runs over that ROUNDED sum (add_cuda then rms_norm_batched_cuda: the prefill residual chain, prefill.rs:183 + :89)
  const bool round_sum = (a.flags & kGemvRoundSum) != 0;
  constexpr int XV = 5;   // 16-byte vectors per lane and row of the register-staged form (K <= 2560)
  const int nvec_row = K >> 3;
  const bool early_plain = !a.norm_w && (K & 511) == 0;
  const bool early_norm = a.norm_w && nvec_row <= 64 * XV && T <= 2 * kSkinnyWaves;
  if (early_plain) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const uint32_t xs_lds = (uint32_t)(uintptr_t)(lds_ptr_t)xs;
    const int pieces = K >> 9;
    for (int t = L.wave; t < T; t += kSkinnyWaves)
      for (int q = 0; q < pieces; ++q) {
        const Half* src = a.X + (size_t)t * K + (size_t)(((q << 6) + L.lane) ^ (t & 15)) * 8;
        const uint32_t dst = xs_lds + (uint32_t)(t * pitch + (q << 6)) * 16u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory", "m0");
      }
    if (total > 0) issue(0, avA);
    if (total > CH) issue(CH, avB);
    // everything older than the weight loads just issued (the DMAs) has landed
    if (total > CH) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * CH * 2 * NW) : "memory");
    else if (total > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CH * 2 * NW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else if (early_norm) {
    u32x4 hx[2][XV], rx[2][XV], gx[XV];
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int c = L.lane + 64 * j;
      if (c < nvec_row) {
        gx[j] = reinterpret_cast<const u32x4*>(a.norm_w)[c];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = L.wave + u * kSkinnyWaves;
          if (t < T) {
            hx[u][j] = reinterpret_cast<const u32x4*>(a.X + (size_t)t * K)[c];
            if (a.residual) rx[u][j] = reinterpret_cast<const u32x4*>(a.residual + (size_t)t * K)[c];
          }
        }
      }
    }
    asm volatile("" ::: "memory");   // the x-side loads stay in front of the weight loads
    if (total > 0) issue(0, avA);
    if (total > CH) issue(CH, avB);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = L.wave + u * kSkinnyWaves;
      if (t >= T) continue;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        if (L.lane + 64 * j < nvec_row) {
          if (a.residual && round_sum) add_round_sq8(hx[u][j], rx[u][j], ss);
          else if (a.residual) add_sq8(hx[u][j], rx[u][j], ss);
          else sq8(hx[u][j], ss);
        }
      }
      ss = wave_sum(ss);
      const float inv = rsqrtf(__fadd_rn(ss / (float)K, a.eps));
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        const int c = L.lane + 64 * j;
        if (c < nvec_row) {
          u32x4 v;
          if (a.residual) {
            u32x4 nh;
            v = norm_scale8(hx[u][j], &rx[u][j], gx[j], inv, 0.f, &nh, round_sum);
            if (blockIdx.x == 0) reinterpret_cast<u32x4*>(a.hidden_out + (size_t)t * K)[c] = nh;
          } else {
            v = norm_scale8(hx[u][j], nullptr, gx[j], inv, 0.f, nullptr);
          }
          xs[t * pitch + (c ^ (t & 15))] = v;
        }
      }
    }
    __syncthreads();
  } else {
    if (total > 0) issue(0, avA);   // the first two chunks leave before the prologue
    if (total > CH) issue(CH, avB);
    if (a.norm_w) {
      for (int t = L.wave; t < T; t += kSkinnyWaves) {
        const float v = wave_row_inv_rms(a.X + (size_t)t * K, a.residual ? a.residual + (size_t)t * K : nullptr, K, a.eps,
                                         round_sum);
        if (L.lane == 0) sm_inv[t] = v;
      }
      __syncthreads();
    }
    skinny_stage_x<NB>(a, xs, sm_inv, pitch, 0, K, L.wave, L.lane);
    __syncthreads();
  }

  f32x4 acc[NW][NB][2][2];
  auto zero = [&]() {
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[w][nb][q >> 1][q & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero();

  // Combine the 8 waves (fixed order), store, reset.  Four forms, same arithmetic per element:
  //  * two barriers around a reduction by the first NB waves (rounds 1-4);
  //  * one barrier (round 5): the partials of row block n go to buffer n & 1 and wave n & 7 ALONE adds and stores them
  //    while the other seven walk on into block n + 1.  Buffer n & 1 is written again for block n + 2, i.e. after
  //    barrier n + 1, which the reducing wave reaches only after its reduction of block n.
  //  * no barrier (ticket form): a ring of buffers, block n -> buffer n % ring, its u-th use (u = n / ring).  A wave waits
  //    until the buffer's previous use has been reduced (done == u), writes its partial and adds 1 to the buffer's arrival
  //    count; wave n & 7 alone then waits for 8 (u + 1) arrivals, adds and stores, and publishes done = u + 1.  LDS executes
  //    a wave's instructions in order, so the count is added after the partial is written and a reader that has seen the
  //    count reads the partials behind it; no fence (a workgroup-scope fence would also drain the weight loads in flight).
  //    The earliest unreduced block's writers wait only for blocks before it: no cycle.
  //  * lazy tickets (bits 0 + 2): as above, but wave n & 7 does not wait at all: it remembers that it owes block n and adds
  //    it up at one of its later flushes once the count is complete - at the latest when buffer n % ring is needed again
  //    (block n + ring, by which time every wave has written block n... or it waits for that) or after its last item.  A wave
  //    owes at most one block at a time (its blocks are 8 apart, the ring is <= 4 deep), and a wave at block r has added up
  //    everything it owed up to r - ring, so the earliest unreduced block never waits on a later one.
  const int wave_u = __builtin_amdgcn_readfirstlane(L.wave);
  const bool lazy = ticket && flush1;
  int pend = -1;                                  // lazy form: the row block this wave still has to add up
  auto complete = [&](int p) {
    return __hip_atomic_load(&sm_cnt[p % ring], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= kSkinnyWaves * (p / ring + 1);
  };
  auto reduce_blk = [&](int p) {                  // ticket forms: this wave alone adds block p's partials and stores
    const int b0 = ((int)blockIdx.x + p * (int)gridDim.x) * rpb;
    asm volatile("" ::: "memory");
    skinny_reduce_store<NB, EPI>(a, red + (p % ring) * kRedQuads, b0, b0 + rpb < rows_total ? b0 + rpb : rows_total, L.lane, 64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reduction's LDS reads have returned
    if (L.lane == 0) __hip_atomic_store(&sm_cnt[ring + p % ring], p / ring + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto flush = [&](int rbi) {
    const int buf = ticket ? rbi % ring : (flush1 ? rbi & 1 : 0);
    const int use = ticket ? rbi / ring : 0;
    f32x4* rb = red + buf * kRedQuads;
    if (lazy && pend >= 0 && pend != rbi - ring && complete(pend)) { reduce_blk(pend); pend = -1; }
    if (ticket && use > 0) {
      if (lazy && pend == rbi - ring) {           // this buffer's previous block is the one this wave owes
        while (!complete(pend)) __builtin_amdgcn_s_sleep(1);
        reduce_blk(pend);
        pend = -1;
      } else {
        while (__hip_atomic_load(&sm_cnt[ring + buf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < use) __builtin_amdgcn_s_sleep(1);
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) rb[((L.wave * NW + w) * NB + nb) * 64 + L.lane] = skinny_fold(acc[w][nb]);
    zero();
    if (a.variant & 2) return;   // timing probe (PEGAINFER_SKINNY_FLUSH=2): no barrier, nothing stored
    if (ticket) {
      asm volatile("" ::: "memory");
      if (L.lane == 0) (void)__hip_atomic_fetch_add(&sm_cnt[buf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (wave_u == (rbi & 7)) {
        if (lazy) {
          pend = rbi;
        } else {
          while (!complete(rbi)) __builtin_amdgcn_s_sleep(1);
          reduce_blk(rbi);
        }
      }
      return;
    }
    const int blk0 = ((int)blockIdx.x + rbi * (int)gridDim.x) * rpb;
    const int lim = blk0 + rpb < rows_total ? blk0 + rpb : rows_total;
    __syncthreads();
    if (flush1) {
      if (L.wave == (rbi & 7)) skinny_reduce_store<NB, EPI>(a, rb, blk0, lim, L.lane, 64);
    } else {
      skinny_reduce_store<NB, EPI>(a, rb, blk0, lim, tid);
      __syncthreads();
    }
  };
  auto compute = [&](int item0, const u32x4 (&av)[NW][CH][2]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int item = item0 + c;
      if (item < total) {
        const int rbi = item / my_np, pi = item - rbi * my_np;
        const int pair = L.wave + kSkinnyWaves * pi;
        if (pair < np) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            int t = nb * 16 + L.l15;
            t = t < T ? t : T - 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xs[t * pitch + ((pair * 8 + 2 * L.g + h) ^ L.l15)]);
#pragma unroll
              for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
                  acc[w][nb][rh][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      __builtin_bit_cast(bf16x8_t, av[w][c][rh]), b, acc[w][nb][rh][h], 0, 0, 0);
            }
          }
        }
        if (pi == my_np - 1) flush(rbi);   // uniform: every wave reaches the same items in the same order
      }
    }
  };
  for (int item0 = 0; item0 < total; item0 += 2 * CH) {   // a register set is refilled as soon as it is consumed
    compute(item0, avA);
    if (item0 + 2 * CH < total) issue(item0 + 2 * CH, avA);
    if (item0 + CH >= total) break;
    compute(item0 + CH, avB);
    if (item0 + 3 * CH < total) issue(item0 + 3 * CH, avB);
  }
  if (pend >= 0) {                                // lazy tickets: the block this wave still owes
    while (!complete(pend)) __builtin_amdgcn_s_sleep(1);
    reduce_blk(pend);
  }
}

// Row-block height (<= 16) that deals the row blocks evenly onto the CUs: k = the rounds 16-row blocks would need,
// height = ceil(rows / (k * CUs)).  2560 rows -> 10 (256 blocks), 6144 -> 12 (512), 9728 -> 13 (749 of 768), 4096 -> 16.
// PEGAINFER_SKINNY_RPB=0 keeps 16-row blocks (A/B); a positive value forces that height.
inline int skinny_pick_rpb(int rows) {
  static const int env = [] { const char* e = getenv("PEGAINFER_SKINNY_RPB"); return e && *e ? atoi(e) : -1; }();
  if (env == 0) return 16;
  if (env > 0) return env > 16 ? 16 : env;
  const int cus = device_cus();
  const int rounds = ceil_div(ceil_div(rows, 16), cus);
  int h = ceil_div(rows, rounds * cus);
  return h < 1 ? 1 : (h > 16 ? 16 : h);
}

// How the resident kernel combines its 8 waves per row block (bit-identical forms, profiles/r5_skinny_flush_ab.txt, T = 4 / 8 /
// 16, cold weights): two barriers (rounds 1-4) -> gate_up 21.5-21.9 us, lm_head 131-143; one barrier -> 21.0-21.1, 129-142;
// tickets -> 20.2-20.7, 121-133, but qkv / o_proj (two / one row block per workgroup: nothing to overlap) +0.1 us.  A probe
// without barriers and stores (FLUSH=2) measured 18.2 us: what is left is the reducing wave's own wait and the stores.
// A ticket form in which the LAST ARRIVER adds (nobody waits; the slowest wave gets the extra work) measured worse than the
// designated wave: gate_up 21.8-22.2 us (profiles/r5_skinny_flush_ab2.txt).  In the pipeline (bench.py --batch, same box,
// alternating, two-barrier form vs this default): bs 4 2.68 / 2.67 -> 2.59 / 2.60 ms per step, bs 16 3.564 / 3.567 -> 3.464 / 3.490.
// Lazy tickets (5: the designated wave does not wait either - it adds a block up at one of its later flushes, once the count
// is complete): gate_up 19.9-20.2 us against 20.7-21.1 with waiting tickets, the fused form 22.3 / 24.1-24.7 against 22.7 /
// 25.4-25.6, lm_head 122-131 against 127-137 (profiles/r5_skinny_flush_ab3.txt).
// Default: lazy tickets where a workgroup walks more than two row blocks, else the one-barrier form.
// PEGAINFER_SKINNY_FLUSH: 0 = two barriers, 1 = one barrier, 4 = tickets, 5 = lazy tickets; 2 = the timing probe (nothing stored).
extern int g_skinny_flush_override;   // linear.hip: pegainfer_debug_skinny_flush (tests compare the forms in one process); -1 = none
// The timing probe (bit 1: partials written, nothing reduced or stored) exists only WITHOUT the ticket bit: with tickets its
// early return would skip the arrival / done counters and every later use of a ring buffer would spin for ever on them - a GPU
// hang, not garbage.  Modes 6 / 7 are therefore folded to 2; the probe is announced on stderr once.
inline int skinny_flush_sanitise(int mode) {
  mode &= 7;
  if ((mode & 2) && (mode & 4)) mode = 2;
  if (mode & 2) {
    static const bool said = (fprintf(stderr, "pegainfer: PEGAINFER_SKINNY_FLUSH probe mode - the resident skinny GEMMs store NOTHING\n"), true);
    (void)said;
  }
  return mode;
}
inline int skinny_flush_mode(int nrb) {
  static const int env = [] { const char* e = getenv("PEGAINFER_SKINNY_FLUSH"); return e && *e ? atoi(e) & 7 : -1; }();
  if (g_skinny_flush_override >= 0) return skinny_flush_sanitise(g_skinny_flush_override);
  if (env >= 0) return skinny_flush_sanitise(env);
  return nrb > 2 * device_cus() ? 5 : 1;
}

// The flush form a resident launch takes and the partial buffers it needs (one function for the launcher and for
// pegainfer_debug_gemm_route, which tests/test_gemm_routing.py pins on the CPU): nw weight sets, nb 16-token blocks.
struct SkinnyFlushPlan { int variant, bufs, rpb, nrb; };
inline SkinnyFlushPlan skinny_flush_plan(int nw, int nb, int T, int K, int rows) {
  SkinnyFlushPlan p;
  p.rpb = skinny_pick_rpb(rows);
  p.nrb = ceil_div(rows, p.rpb);
  const int xs_bytes = (T * K * 2 + 15) & ~15, red_bytes = 8 * nw * nb * 64 * 16;
  p.variant = skinny_flush_mode(p.nrb);
  if (nb > 1 && (p.variant & 4)) p.variant = (p.variant & 2) | 1;          // wider tiles: the one-barrier form
  const int room = (160 * 1024 - xs_bytes - 64 * 4 - 64) / red_bytes;   // buffers that fit beside x
  p.bufs = 1;
  if (p.variant & 4) {
    if (room >= 2) { p.bufs = room > 4 ? 4 : room; p.variant = (p.variant & 7) | (p.bufs << 4); }
    else p.variant = (p.variant & 2) | 1;                               // no room for a ring: the one-barrier form ...
  }
  if ((p.variant & 5) == 1) { if (room >= 2) p.bufs = 2; else p.variant &= ~1; }   // ... or the two-barrier form
  return p;
}

template <int NB, int EPI>
inline void skinny_launch_resident(GemvFusedArgs a, hipStream_t s) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  a.KT = a.K;
  const int rows = EPI == kEpiSilu ? a.I : a.M;
  const SkinnyFlushPlan fp = skinny_flush_plan(NW, NB, a.T, a.K, rows);
  a.rpb = fp.rpb;
  a.variant = fp.variant;
  const int nrb = fp.nrb, bufs = fp.bufs;
  const int xs_bytes = (a.T * a.K * 2 + 15) & ~15, red_bytes = 8 * NW * NB * 64 * 16;
  const int lds = xs_bytes + bufs * red_bytes + 64 * 4 + 64;
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
inline void skinny_launch_rb(GemvFusedArgs a, hipStream_t s) {
  const int rows = EPI == kEpiSilu ? a.I : a.M;
  a.rpb = RB == 1 ? skinny_pick_rpb(rows) : 16;
  // the LDS-DMA form of the kernel (plain x, <= 16 columns, 2048-wide tiles) keeps two x tile buffers
  const bool dma_form = NB == 1 && RB == 1 && EPI != kEpiSilu && !a.norm_w && a.KT == 2048 && (a.K & 511) == 0 && a.T <= 16;
  const int lds = skinny_xs_bytes(NB, a.T, a.KT) * (dma_form ? 2 : 1) + (64 + 4) * 4;
  auto kern = &skinny_mfma_kernel<NB, EPI, RB>;
  static const bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  kern<<<ceil_div(rows, a.rpb * RB), kSkinnyThreads, lds, s>>>(a);
}
template <int NB, int EPI>
inline void skinny_launch(GemvFusedArgs a, hipStream_t s) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  a.KT = skinny_pick_kt(a.T, a.K);
  // row blocks per workgroup: keep ~4 pairs per wave per tile in flight (PEGAINFER_SKINNY_RB=1 forces 1: A/B)
  static const bool rb1 = [] { const char* e = getenv("PEGAINFER_SKINNY_RB"); return e && e[0] == '1'; }();
  const int rows = EPI == kEpiSilu ? a.I : a.M;
  // largest RB the tile allows (KT <= (4/RB)*512) whose accumulators (NW*RB*NB*4 f32x4) + weight registers fit
  // the 256-VGPR budget without spilling, and that still leaves at least one workgroup per CU
  constexpr int kMaxRb = NW == 1 ? (NB <= 2 ? 4 : 2) : (NB == 1 ? 2 : 1);
  int rb = a.KT >= 2048 || rb1 ? 1 : a.KT >= 1024 ? 2 : 4;
  if (rb > kMaxRb) rb = kMaxRb;
  while (rb > 1 && ceil_div(rows, 16 * rb) < 256) rb >>= 1;
  if (rb == 4) {
    if constexpr (kMaxRb >= 4) { if (a.KT > 512) a.KT = 512; skinny_launch_rb<NB, EPI, 4>(a, s); }
  } else if (rb == 2) {
    if constexpr (kMaxRb >= 2) { if (a.KT > 1024) a.KT = 1024; skinny_launch_rb<NB, EPI, 2>(a, s); }
  } else {
    skinny_launch_rb<NB, EPI, 1>(a, s);
  }
}

inline bool skinny_is_resident(int nw, int T, int K) {
  const int nb = T <= 16 ? 1 : T <= 32 ? 2 : 4;
  return (long)T * K * 2 <= kSkinnyResidentBytes && (long)T * K * 2 + 8 * nw * nb * 64 * 16 + 256 <= 160 * 1024;
}
// 2 <= T <= 64, K % 64 == 0
template <int EPI>
inline bool skinny_dispatch(const GemvFusedArgs& a, hipStream_t s) {
  if (a.T < 2 || a.T > 64 || (a.K & 63) != 0) return false;
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  const bool resident = skinny_is_resident(NW, a.T, a.K);
  if (a.T <= 16) { if (resident) skinny_launch_resident<1, EPI>(a, s); else skinny_launch<1, EPI>(a, s); }
  else if (a.T <= 32) { if (resident) skinny_launch_resident<2, EPI>(a, s); else skinny_launch<2, EPI>(a, s); }
  else { if (resident) skinny_launch_resident<4, EPI>(a, s); else skinny_launch<4, EPI>(a, s); }
  return true;
}

}  // namespace pk
