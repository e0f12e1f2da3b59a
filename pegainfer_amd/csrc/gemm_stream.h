// Weight-streaming GEMM for 17..128 token columns (round 4): Y[T, M] = X[T, K] . W[M, K]^T with the row tiles sized to the
// CU COUNT instead of to a power of two.
//
// At these widths the op is bound by how fast the chip INGESTS the weight matrix, and a streaming CU ingests ~24 GB/s: the
// 128-row tiles of mfma_gemm_glds_kernel give gate_up (19 456 rows) 152 workgroups - 152 of 256 CUs carry 655 KB each while
// 104 idle (28 us at 32 columns for 99.6 MB = 3.6 TB/s) - and the small matrices (qkv 48 tiles, o_proj / down_proj 20) had
// to split K over workgroups and pay fp32 partials + a slice-sum launch.  Here a tile is RT x 16 rows with
// RT = ceil(ceil(M / CUs) / 16): gate_up 80 rows -> 244 workgroups, qkv 32 -> 192, every one walking the FULL K (no
// partials, no second launch), all token columns in the workgroup.  o_proj / down_proj (16-row tiles would re-read all of x
// per 16 rows of W) keep their K slices and slice-sum launch, but the GEMM half is this kernel too (part != nullptr):
// 256 / slices row tiles, so tiles x slices fill the chip.
//
// Structure: LDS-DMA rings of K tiles of 64 as in mfma_gemm_glds_kernel (source-side swizzle, counted vmcnt), but
//  * four COMPUTE waves side by side along the token axis (TT / 4 tokens each, all RT row blocks): any RT works;
//  * separate FEEDER waves that only issue the DMAs, the W ring and the X ring each with its own waves and depth: a wave
//    gets one 1 KB DMA instruction through its issue slot per ~93-133 cycles (18-26 GB/s per wave, however many it keeps in
//    flight - tools/probes/ingest_probe) and issues nothing else meanwhile, and its loads retire in order (the L2-resident
//    X tile of the next step must not queue behind W tiles requested ten steps ahead);
//  * G K tiles per barrier (2 or 4): barrier -> LDS read -> dependent MFMAs -> barrier is a latency chain of ~0.28 us
//    whatever the tile holds.
// Per-element K order is that of every un-split tiled kernel (K tiles ascending, two 32-wide MFMA steps each):
// bit-identical to mfma_gemm_glds_kernel<.., false>.
//
// Forms (SplitOut as in linear.hip): plain / row-segmented output (rows_per_tile distinct W rows, a multiple of 4);
// SwiGLU (silu_I > 0): RT is even, the first RT / 2 blocks hold `cols_per_tile` gate rows, the last RT / 2 their up rows
// (rows past cols_per_tile are clamped duplicates: same cache lines, never stored), so 40 + 40 rows run as 3 + 3 blocks.
#pragma once

namespace pk {

// Waves 0..3 compute only; then NLW waves that only feed the W ring and NLX that only feed the X ring.  (The first form of
// this kernel had four waves doing both - gate_up 28.9 us at 32 / 64 columns where this one takes 21.8 / 24.1.)  tools/probes/ingest_probe: a wave gets ~26 GB/s out of its
// DMA issue slot (one 1 KB instruction per ~93 cycles) however many it keeps in flight, and while it sits there it issues
// no MFMA; a CU takes ~125 GB/s when every SIMD has a wave issuing.  With loader waves the K step is bound by what the
// bytes cost (W from HBM at ~23 GB/s per CU, X from L2), not by issue + LDS reads + MFMAs of one wave in series.
// part != nullptr: the K-split form - blockIdx.y = K slice z of nk_slice K tiles (the last one shorter), the workgroup's
// fp32 tile goes to part[z][T][M] and a slice-sum launch adds the slices in z order (linear.hip, splitk_reduce kernels).
template <int RT, int TT, int STW, int STX, int NLW, int NLX, int G>
__global__ __launch_bounds__((4 + NLW + NLX) * 64) void stream_gemm_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                                      Half* __restrict__ Y, int M, int T, int K, int m_tiles,
                                                                      int t_tiles, int rows_per_tile, SplitOut so,
                                                                      float* __restrict__ part, int nk_slice) {
  constexpr int NW_ = NLW;                    // waves feeding the W ring
  constexpr int NX_ = NLX;                    // waves feeding the X ring
  constexpr int TJ = TT / 64;                 // 16-token blocks per compute wave
  constexpr int NWG = RT * 2;                 // 8-row groups of the W tile
  constexpr int NXG = TT / 8;                 // 8-row groups of the X tile
  constexpr int DWW = (NWG + NW_ - 1) / NW_;  // W DMAs per W wave per K tile (groups past NWG re-send the last one)
  constexpr int DXX = (NXG + NX_ - 1) / NX_;  // X DMAs per x wave per K tile (likewise)
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
  // A wave's loads retire IN ORDER, so one wave cannot keep a deep W prefetch and a shallow X prefetch at once: the X tile
  // of the next K step would wait behind W tiles requested for ten steps ahead.  Split by role, each ring has its own
  // depth: W (cold HBM, ~2 us away) 40-48 KB ahead per CU whatever RT is, X (L2-resident, shared by all workgroups) two
  // or more tiles ahead.
  const int li = wave - 4;                    // feeder index (negative = compute wave)
  const bool computes = wave < 4;
  const bool feeds = wave >= 4;
  const bool w_role = feeds && li < NW_;
  const int ww = w_role ? li : li - NW_;      // index among the feeders of this wave's ring

  const int kt_begin = part ? (int)blockIdx.y * nk_slice : 0;
  const size_t kofs = (size_t)kt_begin * BK;
  const Half* src[DWW > DXX ? DWW : DXX];
  if (w_role) {
#pragma unroll
    for (int j = 0; j < DWW; ++j) {
      int rg = ww + NW_ * j;
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
      src[j] = W + (size_t)mr * K + kofs + ((ls ^ (row & 7)) << 3);
    }
  } else if (feeds) {
#pragma unroll
    for (int j = 0; j < DXX; ++j) {
      int rg = ww + NX_ * j;
      rg = rg < NXG ? rg : NXG - 1;
      const int row = rg * 8 + lr;
      int tr = t0 + row;
      tr = tr < T ? tr : T - 1;
      src[j] = X + (size_t)tr * K + kofs + ((ls ^ (row & 7)) << 3);
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
      int rg = ww + NW_ * j;
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
      int rg = ww + NX_ * j;
      rg = rg < NXG ? rg : NXG - 1;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)(buf * TT * 8 + rg * 64) * 16u);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                   :: "v"(src[j] + (size_t)kt * BK), "s"(dst) : "memory", "m0");
    }
  };
  // G K tiles per barrier: barrier -> ds_read latency -> dependent MFMAs -> barrier is ~0.28 us of pure latency whatever
  // the tile holds (down_proj at 16 rows x 64 tokens: 152 K tiles x 0.28 us = 43 us for 2 + 8 KB per step), so the
  // narrow tiles take 4 K tiles per round trip, the wide ones 2.  Ring arithmetic in tiles: before iteration `it` tiles
  // < it G + ST - G have been requested; during it the G slots freed by the previous iteration are refilled; the barrier
  // that ends it needs tiles < (it + 2) G landed, i.e. at most ST - 2 G tiles' worth of this wave's DMAs outstanding.
  static_assert(STW >= 2 * G && STX >= 2 * G, "ring depth");
  const int nk_all = K / BK;
  const int nk = part ? (nk_all - kt_begin < nk_slice ? nk_all - kt_begin : nk_slice) : nk_all;
  if (w_role) {
#pragma unroll
    for (int p = 0; p < STW - G; ++p)
      if (p < nk) stage_w(p, p);
    if (nk >= STW - G) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STW - 2 * G) * DWW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (feeds) {
#pragma unroll
    for (int p = 0; p < STX - G; ++p)
      if (p < nk) stage_x(p, p);
    if (nk >= STX - G) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STX - 2 * G) * DXX) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int wt = (wave & 3) * (TT / 4);
  int curw = 0, nxtw = STW - G, curx = 0, nxtx = STX - G;   // ring slots of tile it G and of the first tile staged during it
  for (int k0 = 0; k0 < nk; k0 += G) {
    const bool full = w_role ? k0 + STW - 1 < nk : k0 + STX - 1 < nk;   // all G tiles this wave would stage exist
    if (feeds) {
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        if (w_role) {
          const int t = k0 + STW - G + gi;
          int slot = nxtw + gi; slot = slot >= STW ? slot - STW : slot;
          if (t < nk) stage_w(slot, t);
        } else {
          const int t = k0 + STX - G + gi;
          int slot = nxtx + gi; slot = slot >= STX ? slot - STX : slot;
          if (t < nk) stage_x(slot, t);
        }
      }
    }
    if (computes) {
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        if (k0 + gi >= nk) break;
        int sw = curw + gi; sw = sw >= STW ? sw - STW : sw;
        int sx = curx + gi; sx = sx >= STX ? sx - STX : sx;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8_t b[TJ];
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            b[j] = __builtin_bit_cast(bf16x8_t, xs[sx][lds_slot(wt + j * 16 + l15, ks * 4 + g)]);
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ws[sw][lds_slot(i * 16 + l15, ks * 4 + g)]);
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    // in the tail (not all G tiles left to stage) everything outstanding is simply waited for
    if (w_role) {
      if (full) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STW - 2 * G) * DWW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (feeds) {
      if (full) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STX - 2 * G) * DXX) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    curw += G; curw = curw >= STW ? curw - STW : curw;
    nxtw += G; nxtw = nxtw >= STW ? nxtw - STW : nxtw;
    curx += G; curx = curx >= STX ? curx - STX : curx;
    nxtx += G; nxtx = nxtx >= STX ? nxtx - STX : nxtx;
  }
  if (!computes) return;
  if (part) {   // fp32 partial tile (M % 4 == 0: a lane's 4 rows are all in or all out)
    float* dst = part + (size_t)blockIdx.y * T * M;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int t = t0 + wt + j * 16 + l15;
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int rl = i * 16 + g * 4, m = m0 + rl;
        if (rl < rows_per_tile && m < M) *reinterpret_cast<f32x4*>(dst + (size_t)t * M + m) = acc[i][j];
      }
    }
    return;
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
  const int kCus = device_cus();
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

// Ring depths and K tiles per barrier.  W: ~2 us x 24 GB/s per CU = 40-48 KB beyond the 2 G tiles the barrier scheme itself
// holds, whatever RT is; X (L2) one or two tiles beyond.  LDS = STW x RT x 2 KB + STX x TT / 8 KB <= 160 KB.
// Measured before the feeder waves: going from 3 X tiles to 5-13 changed nothing - the K loop was bound by DMA issue and the
// per-barrier latency chain, not by request depth.
template <int RT, int TT = 64> struct StreamDepth {
  // TT = 64 (X tile 8 KB):  RT  1: 24 x 2 + 12 x 8 = 144 KB   2: 16 x 4 + 12 x 8 = 160   3: 12 x 6 + 6 x 8 = 120
  //                             4: 10 x 8 + 6 x 8 = 128       5: 10 x 10 + 6 x 8 = 148   6: 9 x 12 + 6 x 8 = 156
  // TT = 128 (X tile 16 KB): RT 3: 12 x 6 + 5 x 16 = 152      4: 9 x 8 + 5 x 16 = 152    5: 8 x 10 + 5 x 16 = 160
  //                             6: 6 x 12 + 5 x 16 = 152 (24-48 KB of W ahead; one tile per barrier with 48-60 KB ahead measured the same)
  static constexpr int G = TT == 64 ? (RT <= 2 ? 4 : 2) : 2;
  static constexpr int W = TT == 64 ? (RT == 1 ? 24 : RT == 2 ? 16 : RT == 3 ? 12 : RT == 4 ? 10 : RT == 5 ? 10 : 9)
                                    : (RT <= 2 ? 16 : RT == 3 ? 12 : RT == 4 ? 9 : RT == 5 ? 8 : 6);   // RT 2: 16 x 4 + 5 x 16 = 144 KB
  static constexpr int X = TT == 64 ? (RT <= 2 ? 12 : 6) : 5;
};

template <int RT, int TT, int NLW, int NLX>
inline void stream_gemm_launch_nl(const Half* W, const Half* X, Half* Y, int M, int T, int K, const StreamPlan& pl,
                                  const SplitOut& so, hipStream_t s) {
  constexpr int STW = StreamDepth<RT, TT>::W, STX = StreamDepth<RT, TT>::X, G = StreamDepth<RT, TT>::G;
  constexpr int NW_ = NLW, NX_ = NLX;
  static_assert((STW - 2 * G) * ((RT * 2 + NW_ - 1) / NW_) <= 63 && (STX - 2 * G) * ((TT / 8 + NX_ - 1) / NX_) <= 63, "vmcnt immediate");
  constexpr int kLds = (STW * RT * 16 + STX * TT) * 8 * 16;
  static_assert(kLds <= 160 * 1024, "LDS");
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_gemm_kernel<RT, TT, STW, STX, NLW, NLX, G>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    return true;
  }();
  (void)once;
  const int t_tiles = ceil_div(T, TT);
  SplitOut sk = so;
  sk.w_nt = t_tiles == 1 && weights_nt_on();
  stream_gemm_kernel<RT, TT, STW, STX, NLW, NLX, G><<<pl.m_tiles * t_tiles, (4 + NLW + NLX) * 64, kLds, s>>>(
      W, X, Y, M, T, K, pl.m_tiles, t_tiles, pl.rows_per_tile, sk, nullptr, 0);
}
// K-split launch (T <= TT): grid (row tiles, K slices), fp32 partials into part[ksplit][T][M]
template <int RT, int TT>
inline void stream_splitk_launch_rt(const Half* W, const Half* X, int M, int T, int K, const StreamPlan& pl, float* part,
                                    int ksplit, int nk_slice, hipStream_t s) {
  constexpr int NLW = RT >= 4 ? 2 : 1, NLX = RT <= 2 ? TT / 16 : TT / 32;
  constexpr int STW = StreamDepth<RT, TT>::W, STX = StreamDepth<RT, TT>::X, G = StreamDepth<RT, TT>::G;
  constexpr int kLds = (STW * RT * 16 + STX * TT) * 8 * 16;
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_gemm_kernel<RT, TT, STW, STX, NLW, NLX, G>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    return true;
  }();
  (void)once;
  SplitOut sk{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, weights_nt_on()};
  stream_gemm_kernel<RT, TT, STW, STX, NLW, NLX, G><<<dim3(pl.m_tiles, ksplit), (4 + NLW + NLX) * 64, kLds, s>>>(
      W, X, nullptr, M, T, K, pl.m_tiles, 1, pl.rows_per_tile, sk, part, nk_slice);
}
// Row tiles of the K-split form: as many as fill the chip together with the K slices (256 / ksplit row tiles; o_proj /
// down_proj with 8 slices: 32 tiles of 80 rows = 256 workgroups where the 128-row tiles made 160).  rt == 0: not taken.
inline StreamPlan stream_splitk_plan(int M, int ksplit) {
  const int want = device_cus() / ksplit;
  if (want < 1) return {0, 0, 0};
  const int r = ceil_div(ceil_div(M, want), 16) * 16;
  if (r > 96 || (M & 3)) return {0, 0, 0};   // taller (stacked qkv at 65..128 tokens, 5 slices): two rounds of 64-row tiles measured 28 us against 24
  return {r / 16, r, ceil_div(M, r)};
}
inline bool stream_splitk_launch(const Half* W, const Half* X, int M, int T, int K, float* part, int ksplit, int nk_slice,
                                 hipStream_t s) {
  const StreamPlan pl = stream_splitk_plan(M, ksplit);
  if (T > 64) {   // one 128-token tile (W read once where the 64-token tiles of the 128-row kernel read it twice)
    switch (T <= 128 ? pl.rt : 0) {
      case 3: stream_splitk_launch_rt<3, 128>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
      case 4: stream_splitk_launch_rt<4, 128>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
      case 5: stream_splitk_launch_rt<5, 128>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
      case 6: stream_splitk_launch_rt<6, 128>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
      default: return false;
    }
  }
  switch (pl.rt) {
    case 1: stream_splitk_launch_rt<1, 64>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
    case 2: stream_splitk_launch_rt<2, 64>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
    case 3: stream_splitk_launch_rt<3, 64>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
    case 4: stream_splitk_launch_rt<4, 64>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
    case 5: stream_splitk_launch_rt<5, 64>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
    case 6: stream_splitk_launch_rt<6, 64>(W, X, M, T, K, pl, part, ksplit, nk_slice, s); return true;
    default: return false;
  }
}

// Feeder waves: a wave moves ~18 GB/s of 8-row x 128-B tile pieces through its DMA issue slot (ingest_probe, tile walk),
// i.e. one 1 KB instruction per ~0.055 us.  A K step must not take longer to ISSUE than its W bytes take to arrive from
// HBM (RT x 2 KB at ~23 GB/s per CU = RT x 0.09 us): W feeders carry <= 6 instructions each, X feeders <= 4.
template <int RT, int TT>
inline void stream_gemm_launch_rt(const Half* W, const Half* X, Half* Y, int M, int T, int K, const StreamPlan& pl,
                                  const SplitOut& so, hipStream_t s) {
  constexpr int NLW = RT >= 4 ? 2 : 1, NLX = RT <= 2 ? TT / 16 : TT / 32;   // X: 2 (x-dominated tiles; 1 measured no better) or 4 instructions per wave and K tile
  stream_gemm_launch_nl<RT, TT, NLW, NLX>(W, X, Y, M, T, K, pl, so, s);
}
template <int TT>
inline bool stream_gemm_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, const SplitOut& so, hipStream_t s) {
  const StreamPlan pl = stream_plan(M, so.silu_I > 0 ? silu_cols_end(so) - so.silu_c0 : 0);
  if constexpr (TT == 64) {
    if (pl.rt == 1) { stream_gemm_launch_rt<1, TT>(W, X, Y, M, T, K, pl, so, s); return true; }
    if (pl.rt == 2) { stream_gemm_launch_rt<2, TT>(W, X, Y, M, T, K, pl, so, s); return true; }
  }
  switch (pl.rt) {   // 128-token tiles exist for >= 2 row blocks (16-row tiles are not routed un-split at all)
    case 2: if constexpr (TT == 128) { stream_gemm_launch_rt<2, TT>(W, X, Y, M, T, K, pl, so, s); return true; } return false;
    case 3: stream_gemm_launch_rt<3, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    case 4: stream_gemm_launch_rt<4, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    case 5: stream_gemm_launch_rt<5, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    case 6: stream_gemm_launch_rt<6, TT>(W, X, Y, M, T, K, pl, so, s); return true;
    default: return false;
  }
}

}  // namespace pk
