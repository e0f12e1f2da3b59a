// Weight-streaming GEMM for 17..128 token columns (round 4): Y[T, M] = X[T, K] . W[M, K]^T with the row tiles sized to the
// CU COUNT instead of to a power of two.
//
// At these widths the op is bound by how fast the chip INGESTS the weight matrix, and a streaming CU ingests ~24 GB/s: the
// 128-row tiles of mfma_gemm_glds_kernel give gate_up (19 456 rows) 152 workgroups - 152 of 256 CUs carry 655 KB each while
// 104 idle (28 us at 32 columns for 99.6 MB = 3.6 TB/s) - and the small matrices (qkv 48 tiles, o_proj / down_proj 20) had
// to split K over workgroups and pay fp32 partials + a slice-sum launch.  Here a tile is RT x 16 rows with
// RT = ceil(ceil(M / CUs) / 16): gate_up 80 rows -> 244 workgroups, qkv 32 -> 192, o_proj / down_proj 16 -> 160, every one
// walking the FULL K (no partials, no second launch), all token columns in the workgroup.
//
// Structure = mfma_gemm_glds_kernel's (LDS-DMA rings of K tiles of 64, source-side swizzle, counted vmcnt, one barrier per
// K tile), with the four waves side by side along the TOKEN axis (TT / 4 tokens each, all RT row blocks) instead of 2 x 2:
// any RT works.  Two rings of different depth, fed by different waves (see the kernel): what must be deep is the W stream.  Per-element K order is that of every un-split tiled kernel (K
// tiles ascending, two 32-wide MFMA steps each): bit-identical to mfma_gemm_glds_kernel<.., false>.
//
// Forms (SplitOut as in linear.hip): plain / row-segmented output (rows_per_tile distinct W rows, a multiple of 4);
// SwiGLU (silu_I > 0): RT is even, the first RT / 2 blocks hold `cols_per_tile` gate rows, the last RT / 2 their up rows
// (rows past cols_per_tile are clamped duplicates: same cache lines, never stored), so 40 + 40 rows run as 3 + 3 blocks.
#pragma once

namespace pk {

template <int RT, int TT, int STW, int STX>
__global__ __launch_bounds__(256) void stream_gemm_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                          Half* __restrict__ Y, int M, int T, int K, int m_tiles,
                                                          int t_tiles, int rows_per_tile, SplitOut so) {
  constexpr int TJ = TT / 64;                 // 16-token blocks per wave
  constexpr int NWG = RT * 2;                 // 8-row groups of the W tile
  constexpr int DWW = (NWG + 1) / 2;          // W DMAs per W wave per K tile (groups past NWG re-send the last one)
  constexpr int DXX = TT / 16;                // X DMAs per x wave per K tile (TT / 8 row groups over two waves)
  extern __shared__ __attribute__((aligned(16))) u32x4 stream_smem[];   // STW x W tile | STX x X tile
  u32x4(*ws)[RT * 16 * 8] = reinterpret_cast<u32x4(*)[RT * 16 * 8]>(stream_smem);
  u32x4(*xs)[TT * 8] = reinterpret_cast<u32x4(*)[TT * 8]>(stream_smem + STW * RT * 16 * 8);
  const int ntiles = m_tiles * t_tiles;
  int tile = blockIdx.x;
  {   // XCD-aware order (hardware places block b on XCD b % 8): each XCD gets a contiguous run of tiles
    const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = tile / t_tiles, tt = tile - mt * t_tiles;
  const int t0 = tt * TT;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int l15 = lane & 15, g = lane >> 4;
  const int lr = lane >> 3, ls = lane & 7;
  const bool silu = so.silu_I > 0;
  constexpr int HB = RT / 2 * 16;             // MFMA rows of one half in the SwiGLU form
  const int m0 = mt * rows_per_tile;          // plain: first W row; SwiGLU: first activation column (relative to silu_c0)
  // Waves 0, 1 feed the W ring, waves 2, 3 the X ring (all four compute).  A wave's loads retire IN ORDER, so one wave
  // cannot keep a deep W prefetch and a shallow X prefetch at once: the X tile of the next K step would wait behind W tiles
  // requested for ten steps ahead.  Split by role, each ring has its own depth: W (cold HBM, ~2 us away) 40-48 KB ahead
  // per CU whatever RT is, X (L2-resident, shared by all workgroups) two or more tiles ahead.
  const bool w_role = wave < 2;
  const int ww = wave & 1;

  const Half* src[DWW > DXX ? DWW : DXX];
  if (w_role) {
#pragma unroll
    for (int j = 0; j < DWW; ++j) {
      int rg = ww + 2 * j;
      rg = rg < NWG ? rg : NWG - 1;
      const int row = rg * 8 + lr;
      int mr;
      if (silu) {
        const int half = row >= HB ? 1 : 0;
        int c = row - half * HB;
        c = c < rows_per_tile ? c : rows_per_tile - 1;
        int col = so.silu_c0 + m0 + c;
        col = col < silu_cols_end(so) ? col : silu_cols_end(so) - 1;
        mr = col + half * so.silu_I;
      } else {
        const int r = row < rows_per_tile ? row : rows_per_tile - 1;
        mr = m0 + r;
        mr = mr < M ? mr : M - 1;
      }
      src[j] = W + (size_t)mr * K + ((ls ^ (row & 7)) << 3);
    }
  } else {
#pragma unroll
    for (int j = 0; j < DXX; ++j) {
      const int row = (ww + 2 * j) * 8 + lr;
      int tr = t0 + row;
      tr = tr < T ? tr : T - 1;
      src[j] = X + (size_t)tr * K + ((ls ^ (row & 7)) << 3);
    }
  }
  f32x4 acc[RT][TJ];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // inline-asm DMAs: a compiler-visible LDS-DMA makes hipcc put s_waitcnt vmcnt(0) in front of every later ds_read
  const uint32_t ws_lds = (uint32_t)(uintptr_t)(lptr_t)&ws[0][0], xs_lds = (uint32_t)(uintptr_t)(lptr_t)&xs[0][0];
  auto stage_w = [&](int buf, int kt) {
#pragma unroll
    for (int j = 0; j < DWW; ++j) {
      int rg = ww + 2 * j;
      rg = rg < NWG ? rg : NWG - 1;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(ws_lds + (uint32_t)(buf * RT * 16 * 8 + rg * 64) * 16u);
      if (so.w_nt)   // one token tile: this workgroup is the only reader of these W bytes (SplitOut::w_nt)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                     :: "v"(src[j] + (size_t)kt * BK), "s"(dst) : "memory", "m0");
      else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src[j] + (size_t)kt * BK), "s"(dst) : "memory", "m0");
    }
  };
  auto stage_x = [&](int buf, int kt) {
#pragma unroll
    for (int j = 0; j < DXX; ++j) {
      const uint32_t dst = __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)(buf * TT * 8 + (ww + 2 * j) * 64) * 16u);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                   :: "v"(src[j] + (size_t)kt * BK), "s"(dst) : "memory", "m0");
    }
  };
  const int nk = K / BK;
  if (w_role) {
#pragma unroll
    for (int p = 0; p < STW - 1; ++p)
      if (p < nk) stage_w(p, p);
    if (nk > STW - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STW - 2) * DWW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
#pragma unroll
    for (int p = 0; p < STX - 1; ++p)
      if (p < nk) stage_x(p, p);
    if (nk > STX - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STX - 2) * DXX) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int wt = wave * (TT / 4);
  int curw = 0, nxtw = STW - 1, curx = 0, nxtx = STX - 1;   // ring slots of tile kt and of the tile staged during kt
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = w_role ? kt + STW - 1 < nk : kt + STX - 1 < nk;
    if (more) { if (w_role) stage_w(nxtw, kt + STW - 1); else stage_x(nxtx, kt + STX - 1); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t b[TJ];
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        b[j] = __builtin_bit_cast(bf16x8_t, xs[curx][lds_slot(wt + j * 16 + l15, ks * 4 + g)]);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ws[curw][lds_slot(i * 16 + l15, ks * 4 + g)]);
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
      }
    }
    // tile kt + 1 of this wave's ring has landed when at most (depth - 2) tiles' worth of its DMAs are outstanding; in the
    // tail (nothing left to stage) the remaining tiles are simply all waited for
    if (w_role) {
      if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STW - 2) * DWW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STX - 2) * DXX) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    curw = curw + 1 == STW ? 0 : curw + 1;
    nxtw = nxtw + 1 == STW ? 0 : nxtw + 1;
    curx = curx + 1 == STX ? 0 : curx + 1;
    nxtx = nxtx + 1 == STX ? 0 : nxtx + 1;
  }
  // C layout (16x16x32): col = lane & 15 -> token, rows (lane >> 4) * 4 + e -> 4 consecutive tile rows
  if (silu) {
    if constexpr (RT >= 2) {
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int t = t0 + wt + j * 16 + l15;
        if (t >= T) continue;
#pragma unroll
        for (int i = 0; i < RT / 2; ++i) {
          const int c = i * 16 + g * 4;
          if (c >= rows_per_tile) continue;                      // rows_per_tile % 4 == 0: a lane's 4 columns are all in or all out
          const int m = so.silu_c0 + m0 + c;
          float r[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {   // the GEMM output is bf16 before SwiGLU (fused_proj.cu:57-62)
            const float sg = silu_f(bf16_round_f(acc[i][j][e]));
            r[e] = (so.silu_round ? bf16_round_f(sg) : sg) * bf16_round_f(acc[i + RT / 2][j][e]);
          }
          if (m + 3 < silu_cols_end(so)) {
            u32x2 o;
            o.x = pack_bf2(r[0], r[1]);
            o.y = pack_bf2(r[2], r[3]);
            *reinterpret_cast<u32x2*>(Y + (size_t)t * so.silu_I + m) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (m + e < silu_cols_end(so)) Y[(size_t)t * so.silu_I + m + e] = f2bf(r[e]);
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int t = t0 + wt + j * 16 + l15;
    if (t >= T) continue;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int rl = i * 16 + g * 4;
      if (rl >= rows_per_tile) continue;
      const int m = m0 + rl;
      Half* dst = Y;
      int ld = M, mm = m, mlim = M;
      if (so.Y1) {
        const int b1 = so.M0 + so.M1, b2 = b1 + so.M2;
        if (m < so.M0) { ld = mlim = so.M0; }
        else if (m < b1) { dst = so.Y1; ld = mlim = so.M1; mm = m - so.M0; }
        else if (m < b2) { dst = so.Y2; ld = mlim = so.M2; mm = m - b1; }
        else { dst = so.Y3; ld = mlim = M - b2; mm = m - b2; }
      }
      if (m + 3 < M && mm + 3 < mlim) {
        u32x2 o;
        o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
        o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
        *reinterpret_cast<u32x2*>(dst + (size_t)t * ld + mm) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (m + e < M && mm + e < mlim) dst[(size_t)t * ld + mm + e] = f2bf(acc[i][j][e]);
      }
    }
  }
}

// Tile plan by shape only: RT x 16 rows so that the row tiles deal onto the CUs in as few, as full rounds as possible.
// rt == 0: not taken (more than 6 blocks per tile: the 128-row kernel's territory, e.g. lm_head).
struct StreamPlan { int rt, rows_per_tile, m_tiles; };
inline StreamPlan stream_plan(int M, int silu_cols /* 0 = plain */) {
  constexpr int kCus = 256;
  if (silu_cols > 0) {
    int c = ceil_div(ceil_div(silu_cols, kCus), 4) * 4;          // activation columns per tile, a multiple of 4
    const int half_blocks = ceil_div(c, 16);
    if (half_blocks > 3) return {0, 0, 0};
    return {2 * half_blocks, c, ceil_div(silu_cols, c)};
  }
  int r = ceil_div(ceil_div(M, kCus), 16) * 16;
  if (r > 96) return {0, 0, 0};
  return {r / 16, r, ceil_div(M, r)};
}

// Ring depths.  W: ~2 us x 24 GB/s per CU = 40-48 KB ahead whatever RT is; X: a few tiles.  Measured: going from 3 X tiles
// to 5-13 changed NOTHING (gate_up 25.2 us, down_proj at RT = 1 38.6 us either way) - the K loop is not latency-bound but
// bound by the CU's total ingest, W + X bytes alike at ~30-40 GB/s per CU (RT = 5: 18 KB per K step in 0.63 us; RT = 1:
// 10 KB in 0.25 us).  What lowers the time is fewer bytes per CU: that is K split over workgroups, not this kernel.
template <int RT, int TT = 64> struct StreamDepth {
  static constexpr int W = RT == 1 ? 20 : RT == 2 ? 12 : RT == 3 ? 8 : RT == 4 ? 6 : 6;
  static constexpr int X64 = RT == 1 ? 13 : RT == 2 ? 9 : RT == 3 ? 7 : RT == 4 ? 6 : 5;
  static constexpr int X = TT == 64 ? X64 : (X64 > 5 ? 5 : X64);   // 128-token tiles: 16 KB per X stage, 160 KB of LDS in all
};
template <int RT, int TT>
inline void stream_gemm_launch_rt(const Half* W, const Half* X, Half* Y, int M, int T, int K, const StreamPlan& pl,
                                  const SplitOut& so, hipStream_t s) {
  constexpr int STW = StreamDepth<RT, TT>::W, STX = StreamDepth<RT, TT>::X;
  static_assert((STW - 2) * ((RT * 2 + 1) / 2) <= 63 && (STX - 2) * (TT / 16) <= 63, "vmcnt immediate");
  constexpr int kLds = (STW * RT * 16 + STX * TT) * 8 * 16;
  static_assert(kLds <= 160 * 1024, "LDS");
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_gemm_kernel<RT, TT, STW, STX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    return true;
  }();
  (void)once;
  const int t_tiles = ceil_div(T, TT);
  SplitOut sk = so;
  sk.w_nt = t_tiles == 1 && weights_nt_on();
  stream_gemm_kernel<RT, TT, STW, STX><<<pl.m_tiles * t_tiles, 256, kLds, s>>>(W, X, Y, M, T, K, pl.m_tiles, t_tiles,
                                                                              pl.rows_per_tile, sk);
}
template <int TT>
inline bool stream_gemm_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, const SplitOut& so, hipStream_t s) {
  const StreamPlan pl = stream_plan(M, so.silu_I > 0 ? silu_cols_end(so) - so.silu_c0 : 0);
  if constexpr (TT == 64) {
    if (pl.rt == 1) { stream_gemm_launch_rt<1, TT>(W, X, Y, M, T, K, pl, so, s); return true; }
    if (pl.rt == 2) { stream_gemm_launch_rt<2, TT>(W, X, Y, M, T, K, pl, so, s); return true; }
  }
  switch (pl.rt) {   // 128-token tiles exist for >= 3 row blocks only (the 16 / 32-row tiles are not routed at all)
    case 3: stream_gemm_launch_rt<3, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    case 4: stream_gemm_launch_rt<4, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    case 5: stream_gemm_launch_rt<5, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    case 6: stream_gemm_launch_rt<6, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    default: return false;
  }
}

}  // namespace pk
