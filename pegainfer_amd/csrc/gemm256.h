// Prefill GEMM, long prompts: 256 x 256 x 64 tiles, 8 waves (2 x 4), v_mfma_f32_16x16x32_bf16, 128 KiB of LDS.
//
//   Y[T, M] = X[T, K] . W[M, K]^T,  a workgroup owns 256 W rows x 256 token rows, wave (wr, wc) a 128 x 64 block of it
//   as 8 x 4 MFMA tiles (128 accumulator VGPRs).
//
// The 128 x {128,64} kernel of linear.hip is LDS-read bound (two barriers per K tile, every fragment read serialised
// behind the stage) and reaches 23-28 % of the dense bf16 peak.  This one follows the CDNA4 "8-phase" schedule:
//   * a K tile is FOUR 16 KiB half-tiles - A_lo / A_hi = the rows of the wave's m-tiles 0-3 / 4-7 (both wave rows),
//     B_lo / B_hi = the token rows of n-tiles 0-1 / 2-3 (all four wave columns) - and one phase computes one quadrant
//     (4 m-tiles x 2 n-tiles x K 64 = 16 MFMAs) from one A half and one B half:
//       phase 1: read A_lo + B_lo, quadrant (lo, lo)     phase 2: read B_hi, quadrant (lo, hi)
//       phase 3: read A_hi,        quadrant (hi, hi)     phase 4: read B_lo, quadrant (hi, lo)
//     so a K tile costs 24 ds_read_b128 per wave for 64 MFMAs (the 128-tile kernel: 16 for 32);
//   * half-tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, XOR-swizzled through the SOURCE address),
//     two instructions per thread per half-tile, ONE half-tile staged per phase into the slot whose last reader is two
//     phases back: A_hi / B_lo of the next K tile in phases 1 / 2, A_lo / B_hi of the tile after that in phases 3 / 4.
//     The only wait is a counted `s_waitcnt vmcnt(4)` at phases 4 and 8 (the two youngest half-tiles stay in flight
//     across the barrier); a staged slot is first read one phase after the wait that retires it;
//   * two raw s_barrier per phase, and the wave row wr = 1 runs one barrier behind wr = 0: while one half of the
//     workgroup (one wave per SIMD) issues its 16 MFMAs, the other half does its fragment reads and its share of the
//     DMA - the matrix pipe of every SIMD always has a wave in its MFMA segment.
// Per-element K order (K tile by K tile, k-step 0 then 1) is that of the other tiled kernels: same bits.
#pragma once

#include "common.h"

namespace pk {

constexpr int G256_BM = 256, G256_BT = 256, G256_BK = 64;
constexpr int kG256HalfBytes = 128 * 128;            // 128 rows x 128 B
constexpr int kG256LdsBytes = 8 * kG256HalfBytes;    // 2 parities x {A_lo, A_hi, B_lo, B_hi}
enum { kHalfAlo = 0, kHalfAhi = 1, kHalfBlo = 2, kHalfBhi = 3 };

struct G256Out { Half* Y; int ld; };

template <bool SILU>
__global__ __launch_bounds__(512) void mfma_gemm256_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                           Half* __restrict__ Y, int M, int T, int K, int m_tiles,
                                                           int t_tiles, SplitOut so, float* __restrict__ part = nullptr,
                                                           int nk_slice = 0) {
  extern __shared__ __attribute__((aligned(16))) u32x4 g256_smem[];
  const int ntiles = m_tiles * t_tiles;
  int tile = blockIdx.x, zslice = (int)blockIdx.y;
  if (part && so.xcd_slices) {
    // K-split launches (round 6): one contiguous run of (slice, tile) work items per XCD in SLICE-major order, so an XCD pulls
    // one K slice of X through its L2 instead of all of X (see mfma_gemm128x256_kernel); same work items, same bits
    const int nitems = ntiles * (int)gridDim.y, id = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
    const int q = nitems / 8, r = nitems % 8, xcd = id % 8, idx = id / 8;
    const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    zslice = item / ntiles;
    tile = item - zslice * ntiles;
  } else {  // XCD-aware order (block b runs on XCD b % 8): every XCD walks a contiguous run of tiles, token tile fastest
    const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // (round 6) t_major: row tile fastest inside an XCD's run - for T > M (o_proj / down_proj / qkv at 10 k tokens) an XCD then pulls
  // 1/8 of X and all of W through its L2 instead of all of X and 1/8 of W
  // (round 6) chunk: the slow dimension of the order is walked in chunks of `chunk` tiles, so the ~32 workgroups an XCD runs at a time
  // are a (32 / chunk) x chunk block of tiles sharing 32 / chunk + chunk operand panels through its L2 instead of 1 + 32: gate_up at
  // 10 k tokens (76 x 40 tiles, token tile fastest) fetched 4.2 GB from the memory side per launch for 151 MB of operands - every X
  // panel once per row tile (FETCH_SIZE, profiles/r6c_ctx10000_pmc_fetch.csv)
  int mt, tt;
  if (so.chunk > 0) {
    const int fast_n = so.t_major ? m_tiles : t_tiles, slow_n = so.t_major ? t_tiles : m_tiles;   // fast = the inner dimension of the plain order
    const int per_chunk = slow_n * so.chunk;                  // tiles of one full chunk of the FAST dimension
    const int c = tile / per_chunk, rem = tile - c * per_chunk;
    const int cw = fast_n - c * so.chunk < so.chunk ? fast_n - c * so.chunk : so.chunk;
    const int slow = rem / cw, fast = c * so.chunk + rem - slow * cw;
    mt = so.t_major ? fast : slow;
    tt = so.t_major ? slow : fast;
  } else {
    mt = so.t_major ? tile % m_tiles : tile / t_tiles;
    tt = so.t_major ? tile / m_tiles : tile - mt * t_tiles;
  }
  const int m0 = mt * G256_BM, t0 = tt * G256_BT;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- DMA sources: piece j of a half-tile = LDS positions (j*8 + wave)*64 + lane: half-row hrow, 16-byte slot cpos;
  //      the lane fetches global slot cpos ^ (hrow & 7) so that readers use lds_slot() ----
  const Half* src[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = (j * 8 + wave) * 64 + lane, hrow = p >> 3, cs = (p & 7) ^ (hrow & 7);
    // A halves: hrow = wr' * 64 + r  ->  tile row wr' * 128 (+ 64 for hi) + r
    {
      const int wr_ = hrow >> 6, r = hrow & 63;
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) {
        int row;
        if (SILU) {  // tile = 128 gate rows + their 128 up rows: the wave's m-tiles 0-3 are gate rows, 4-7 the up rows
          int gr = so.silu_c0 + mt * 128 + wr_ * 64 + r;
          gr = gr < silu_cols_end(so) ? gr : silu_cols_end(so) - 1;
          row = gr + (hi ? so.silu_I : 0);
        } else {
          row = m0 + wr_ * 128 + hi * 64 + r;
          row = row < M ? row : M - 1;
        }
        src[hi ? kHalfAhi : kHalfAlo][j] = W + (size_t)row * K + cs * 8;
      }
    }
    // B halves: hrow = wc' * 32 + r  ->  tile token wc' * 64 (+ 32 for hi) + r
    {
      const int wc_ = hrow >> 5, r = hrow & 31;
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) {
        int tr = t0 + wc_ * 64 + hi * 32 + r;
        tr = tr < T ? tr : T - 1;
        src[hi ? kHalfBhi : kHalfBlo][j] = X + (size_t)tr * K + cs * 8;
      }
    }
  }
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&g256_smem[0];
  // split-K form (part != null): blockIdx.y = K slice of nk_slice K tiles (even, like the last, shorter one: the
  // tiles are walked in pairs); the workgroup's fp32 tile goes to part[z][T][M] and a slice-sum launch adds the slices
  // in z order (down_proj at 1-2 k tokens: 40-80 tiles of 256 x 256 for 256 CUs)
  const int kt_begin = part ? zslice * nk_slice : 0;
  const int nk_total = K / G256_BK;
  const int nk = part ? (nk_total - kt_begin < nk_slice ? nk_total - kt_begin : nk_slice) : nk_total;
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int j = 0; j < 2; ++j) src[w][j] += (size_t)kt_begin * G256_BK;
  // stage half-tile `which` of K tile kt (clamped: past the end the slot is dead, the load only keeps vmcnt uniform)
  auto stage = [&](int which, int kt) {
    const int ktc = kt < nk ? kt : nk - 1;
    const uint32_t slot = lds0 + (uint32_t)(((kt & 1) * 4 + which) * kG256HalfBytes);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t dst = __builtin_amdgcn_readfirstlane(slot + (uint32_t)(j * 8 + wave) * 1024u);
      if ((which == kHalfAlo || which == kHalfAhi) && so.w_nt)   // a single token tile: W bytes have one reader (SplitOut::w_nt)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                     :: "v"(src[which][j] + (size_t)ktc * G256_BK), "s"(dst) : "memory", "m0");
      else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src[which][j] + (size_t)ktc * G256_BK), "s"(dst) : "memory", "m0");
    }
  };
  auto half_ptr = [&](int par, int which) { return g256_smem + (size_t)(par * 4 + which) * (kG256HalfBytes / 16); };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t a[4][2], b[2][2];
  auto read_a = [&](int par, int which) {
    const u32x4* h = half_ptr(par, which);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        a[i][ks] = __builtin_bit_cast(bf16x8_t, h[lds_slot(wr * 64 + i * 16 + l15, ks * 4 + g)]);
  };
  auto read_b = [&](int par, int which) {
    const u32x4* h = half_ptr(par, which);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        b[j][ks] = __builtin_bit_cast(bf16x8_t, h[lds_slot(wc * 32 + j * 16 + l15, ks * 4 + g)]);
  };
  auto quad = [&](int ih, int jh) {   // 16 MFMAs: m-tiles ih*4.., n-tiles jh*2..
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[ih * 4 + i][jh * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: K tile 0 complete, A_lo / B_hi of K tile 1 in flight ----
  stage(kHalfAlo, 0); stage(kHalfBlo, 0); stage(kHalfBhi, 0); stage(kHalfAhi, 0);
  stage(kHalfAlo, 1); stage(kHalfBhi, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind the first

  // one K tile = four phases; `kt` has parity `par`, the other slots belong to kt + 1
  auto ktile = [&](int kt, int par) {
    // phase 1: A_lo + B_lo -> (lo, lo); stage A_hi of kt + 1
    read_b(par, kHalfBlo);
    __builtin_amdgcn_sched_barrier(0);
    read_a(par, kHalfAlo);
    stage(kHalfAhi, kt + 1);
    __builtin_amdgcn_s_barrier();
    quad(0, 0);
    __builtin_amdgcn_s_barrier();
    // phase 2: B_hi -> (lo, hi); stage B_lo of kt + 1
    read_b(par, kHalfBhi);
    stage(kHalfBlo, kt + 1);
    __builtin_amdgcn_s_barrier();
    quad(0, 1);
    __builtin_amdgcn_s_barrier();
    // phase 3: A_hi -> (hi, hi); stage A_lo of kt + 2
    read_a(par, kHalfAhi);
    stage(kHalfAlo, kt + 2);
    __builtin_amdgcn_s_barrier();
    quad(1, 1);
    __builtin_amdgcn_s_barrier();
    // phase 4: B_lo -> (hi, lo); stage B_hi of kt + 2; everything but the two youngest half-tiles has landed
    read_b(par, kHalfBlo);
    stage(kHalfBhi, kt + 2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    quad(1, 0);
    __builtin_amdgcn_s_barrier();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(kt, 0);
    ktile(kt + 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail loads must not outlive the workgroup's LDS
  if (wr == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue: lane holds rows m = .. + g*4 + e (4 consecutive) of token t = .. + l15 ----
  if (SILU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + wc * 64 + j * 16 + l15;
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = so.silu_c0 + mt * 128 + wr * 64 + i * 16 + g * 4;
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // the GEMM output is bf16 before SwiGLU (fused_proj.cu:57-62)
          const float sg = silu_f(bf16_round_f(acc[i][j][e]));
          r[e] = (so.silu_round ? bf16_round_f(sg) : sg) * bf16_round_f(acc[i + 4][j][e]);
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
    return;
  }
  if (part) {
    float* pz = part + (size_t)zslice * T * M;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + wc * 64 + j * 16 + l15;
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + g * 4;
        if (m + 3 < M) {
          *reinterpret_cast<f32x4*>(pz + (size_t)t * M + m) = acc[i][j];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (m + e < M) pz[(size_t)t * M + m + e] = acc[i][j][e];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + wc * 64 + j * 16 + l15;
    if (t >= T) continue;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wr * 128 + i * 16 + g * 4;
      Half* dst = Y;
      int ld = M, mm = m, mlim = M;
      if (so.Y1) {
        const int b1 = so.M0 + so.M1, b2 = b1 + so.M2;
        if (m < so.M0) { ld = mlim = so.M0; }
        else if (m < b1) { dst = so.Y1; ld = mlim = so.M1; mm = m - so.M0; }
        else if (m < b2) { dst = so.Y2; ld = mlim = so.M2; mm = m - b1; }
        else { dst = so.Y3; ld = mlim = M - b2; mm = m - b2; }
      }
      if (mm + 3 < mlim) {
        u32x2 o;
        o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
        o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
        *reinterpret_cast<u32x2*>(dst + (size_t)t * ld + mm) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (mm + e < mlim) dst[(size_t)t * ld + mm + e] = f2bf(acc[i][j][e]);
      }
    }
  }
}

// K must be a multiple of 128 (K tiles are processed in pairs); same alignment rules as the other LDS-DMA kernels
inline bool gemm256_ok(int M, int T, int K) { return (K % 128) == 0 && K >= 256 && M >= 256 && T >= 256; }

inline bool gemm_t_major(int M, int T, const SplitOut& so) {   // PEGAINFER_GEMM_T_MAJOR=0: always row-tile-major runs (A/B)
  static const bool on = [] { const char* e = getenv("PEGAINFER_GEMM_T_MAJOR"); return !(e && e[0] == '0'); }();
  return on && so.silu_I == 0 && T > M;
}
inline void gemm256_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, hipStream_t s) {
  const bool silu = so.silu_I > 0;
  const int m_tiles = silu ? ceil_div(silu_cols_end(so) - so.silu_c0, 128) : ceil_div(M, G256_BM), t_tiles = ceil_div(T, G256_BT);
  so.w_nt = t_tiles == 1 && weights_nt_on();
  so.t_major = gemm_t_major(M, T, so);
  {  // PEGAINFER_GEMM_CHUNK=0: the plain order (A/B); N: chunks of N tiles
    static const int chunk = [] { const char* e = getenv("PEGAINFER_GEMM_CHUNK"); return e && *e ? atoi(e) : 8; }();
    // chunks of equal width (17 tiles: 6 + 6 + 5, not 8 + 8 + 1); only for launches of more than one round of tiles.  4 / 6 / 8 measure
    // the same (TTFT(10 000) 94.9-95.4 ms against 98.4 plain, 12: 96.0, 16: 96.8 - profiles/r6_gemm_chunk_sweep.txt)
    const int fast_n = so.t_major ? m_tiles : t_tiles;
    const int nch = chunk > 0 ? ceil_div(fast_n, chunk) : 0;
    so.chunk = nch > 1 && (long)m_tiles * t_tiles > 256 ? ceil_div(fast_n, nch) : 0;
  }
  if (silu) {
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm256_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kG256LdsBytes);
      return true;
    }();
    (void)once;
    mfma_gemm256_kernel<true><<<m_tiles * t_tiles, 512, kG256LdsBytes, s>>>(W, X, Y, M, T, K, m_tiles, t_tiles, so);
  } else {
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm256_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kG256LdsBytes);
      return true;
    }();
    (void)once;
    mfma_gemm256_kernel<false><<<m_tiles * t_tiles, 512, kG256LdsBytes, s>>>(W, X, Y, M, T, K, m_tiles, t_tiles, so);
  }
}

// split-K launch of the plain form: fp32 partials into `part` ([ksplit][T][M]); the caller sums the slices
inline bool splitk_xcd_on() {   // PEGAINFER_SPLITK_XCD=0: the slices of a tile on one XCD, as until round 6 (A/B)
  static const bool v = [] { const char* e = getenv("PEGAINFER_SPLITK_XCD"); return !(e && e[0] == '0'); }();
  return v;
}
inline void gemm256_splitk_launch(const Half* W, const Half* X, int M, int T, int K, float* part, int ksplit, int nk_slice,
                                  hipStream_t s) {
  const int m_tiles = ceil_div(M, G256_BM), t_tiles = ceil_div(T, G256_BT);
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm256_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kG256LdsBytes);
    return true;
  }();
  (void)once;
  mfma_gemm256_kernel<false><<<dim3(m_tiles * t_tiles, ksplit), 512, kG256LdsBytes, s>>>(
      W, X, nullptr, M, T, K, m_tiles, t_tiles,
      SplitOut{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, t_tiles == 1 && weights_nt_on(), ksplit > 1 && splitk_xcd_on(), 0}, part, nk_slice);
}

// ---- stream-K form of the 256 x 256 kernel (round 6) --------------------------------------------------------------------
// The data-parallel launch above pays for tile quantisation: 304 tiles (gate_up at 1024 tokens) are one full round of the
// 256 CUs plus 48 tiles that cost a second full tile time; 152 tiles (512 tokens) leave 104 CUs idle.  Here the launch is
// PERSISTENT - one workgroup per CU - and the work is the linear sequence of (tile, K-tile pair) units, tile-major, dealt in
// equal contiguous ranges: workgroup position p owns units [U p / G, U (p + 1) / G).  A range is cut at tile boundaries into
// segments; each segment runs the 8-phase main loop over its K range and then
//   * covers the whole tile            -> the ordinary epilogue;
//   * starts inside a tile             -> (only ever a workgroup's FIRST segment) the fp32 accumulators go to the workgroup's
//                                         256 KiB slot of the split-K workspace, lane-contiguous, write-through (sc1) stores,
//                                         drained, then flag[p] = 1: published EARLY in the workgroup's timeline, no waiting;
//   * starts a tile but ends inside it -> (only ever its LAST segment) the workgroup OWNS the tile: it takes the partials of
//                                         positions p + 1, p + 2, ... up to the one that holds the tile's last unit, in that
//                                         order (= ascending K), adds them to its accumulators, clears their flags, and runs
//                                         the epilogue.  Those partials were published a whole range ago.
// So the sum over K is taken in a FIXED order that depends on (M, T, K, G) only: deterministic, rerun- and graph-stable, but
// not the K order of the data-parallel kernels - the cross-route bit identity of rounds 1-5 is given up for prefill GEMMs at
// these sizes; the oracle comparison under the derived bar referees.  No deadlock: a publisher never waits,
// an owner waits only for publishers, and all G <= CU-count workgroups are co-resident (one per CU by LDS).  Positions follow
// the XCD-aware order of the other kernels (block b runs on XCD b % 8): neighbours in the unit order share an XCD's L2 except
// at seven seams, so most partials are re-read from the L2 they were written through.
constexpr int kG256SlotFloats = 256 * 256;   // one workgroup's accumulators: 8 x 4 f32x4 per thread x 512 threads
template <bool SILU>
__global__ __launch_bounds__(512) void mfma_gemm256_streamk_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                                   Half* __restrict__ Y, int M, int T, int K, int m_tiles,
                                                                   int t_tiles, SplitOut so, float* __restrict__ part,
                                                                   uint32_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) u32x4 g256_smem[];
  const int G = gridDim.x;
  int pos = blockIdx.x;
  {  // XCD-aware position: the workgroups of one XCD are neighbours in the unit order
    const int q = G / 8, r = G % 8, xcd = pos % 8, idx = pos / 8;
    pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int KP = K / (2 * G256_BK);                       // K-tile pairs per tile
  // TEAMS (second form, same round): the first form dealt the (tile, K pair) units tile-major to single workgroups and LOST
  // (gate_up at 1024 tokens 154 us against 136-140 data-parallel, profiles/r6_streamk_gemm_ab.txt): neighbours in the unit
  // order then sit at K offsets a third of a tile apart, no two workgroups ever want the same W K-tile at the same time, and
  // the 4-fold W re-use the data-parallel order gets from an XCD's L2 (the t_tiles token tiles of one row tile run side by
  // side) is gone.  So the unit is a ROW tile's K pair, dealt to TEAMS of t_tiles workgroups - consecutive positions, one
  // XCD - whose member tt computes token tile tt of every unit of the team's range: the members walk the same W K-tiles in
  // step, exactly the sharing pattern of the data-parallel launch.  A member's partner for the hand-off is the same member
  // of the next team (position + t_tiles).
  const int nteams = G / t_tiles;
  if (pos >= nteams * t_tiles) return;                    // G % t_tiles left-over workgroups
  const int team = pos / t_tiles, tt = pos - team * t_tiles;
  const long U = (long)m_tiles * KP;
  auto range_begin = [&](int tm) { return U * tm / nteams; };
  const long u_begin = range_begin(team), u_end = range_begin(team + 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&g256_smem[0];
  auto half_ptr = [&](int par, int which) { return g256_smem + (size_t)(par * 4 + which) * (kG256HalfBytes / 16); };

  f32x4 acc[8][4];
  bf16x8_t a[4][2], b[2][2];
  auto read_a = [&](int par, int which) {
    const u32x4* h = half_ptr(par, which);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        a[i][ks] = __builtin_bit_cast(bf16x8_t, h[lds_slot(wr * 64 + i * 16 + l15, ks * 4 + g)]);
  };
  auto read_b = [&](int par, int which) {
    const u32x4* h = half_ptr(par, which);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        b[j][ks] = __builtin_bit_cast(bf16x8_t, h[lds_slot(wc * 32 + j * 16 + l15, ks * 4 + g)]);
  };
  auto quad = [&](int ih, int jh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[ih * 4 + i][jh * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  long u = u_begin;
  while (u < u_end) {
    const int mt = (int)(u / KP), kp0 = (int)(u - (long)mt * KP);
    const int kp1 = (long)(KP - kp0) < u_end - u ? KP : kp0 + (int)(u_end - u);
    const int m0 = mt * G256_BM, t0 = tt * G256_BT;
    const int nk = 2 * (kp1 - kp0);
    // ---- DMA sources of this tile (as in the data-parallel kernel), advanced to the segment's first K tile ----
    const Half* src[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = (j * 8 + wave) * 64 + lane, hrow = p >> 3, cs = (p & 7) ^ (hrow & 7);
      {
        const int wr_ = hrow >> 6, r = hrow & 63;
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
          int row;
          if (SILU) {
            int gr = so.silu_c0 + mt * 128 + wr_ * 64 + r;
            gr = gr < silu_cols_end(so) ? gr : silu_cols_end(so) - 1;
            row = gr + (hi ? so.silu_I : 0);
          } else {
            row = m0 + wr_ * 128 + hi * 64 + r;
            row = row < M ? row : M - 1;
          }
          src[hi ? kHalfAhi : kHalfAlo][j] = W + (size_t)row * K + cs * 8 + (size_t)kp0 * 2 * G256_BK;
        }
      }
      {
        const int wc_ = hrow >> 5, r = hrow & 31;
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
          int tr = t0 + wc_ * 64 + hi * 32 + r;
          tr = tr < T ? tr : T - 1;
          src[hi ? kHalfBhi : kHalfBlo][j] = X + (size_t)tr * K + cs * 8 + (size_t)kp0 * 2 * G256_BK;
        }
      }
    }
    auto stage = [&](int which, int kt) {
      const int ktc = kt < nk ? kt : nk - 1;
      const uint32_t slot = lds0 + (uint32_t)(((kt & 1) * 4 + which) * kG256HalfBytes);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(slot + (uint32_t)(j * 8 + wave) * 1024u);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src[which][j] + (size_t)ktc * G256_BK), "s"(dst) : "memory", "m0");
      }
    };
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the 8-phase main loop over K tiles [2 kp0, 2 kp1) ----
    stage(kHalfAlo, 0); stage(kHalfBlo, 0); stage(kHalfBhi, 0); stage(kHalfAhi, 0);
    stage(kHalfAlo, 1); stage(kHalfBhi, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    auto ktile = [&](int kt, int par) {
      read_b(par, kHalfBlo);
      __builtin_amdgcn_sched_barrier(0);
      read_a(par, kHalfAlo);
      stage(kHalfAhi, kt + 1);
      __builtin_amdgcn_s_barrier();
      quad(0, 0);
      __builtin_amdgcn_s_barrier();
      read_b(par, kHalfBhi);
      stage(kHalfBlo, kt + 1);
      __builtin_amdgcn_s_barrier();
      quad(0, 1);
      __builtin_amdgcn_s_barrier();
      read_a(par, kHalfAhi);
      stage(kHalfAlo, kt + 2);
      __builtin_amdgcn_s_barrier();
      quad(1, 1);
      __builtin_amdgcn_s_barrier();
      read_b(par, kHalfBlo);
      stage(kHalfBhi, kt + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      quad(1, 0);
      __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt < nk; kt += 2) {
      ktile(kt, 0);
      ktile(kt + 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();   // both wave rows are level again: the LDS may be re-staged

    if (kp0 != 0) {
      // ---- publish: lane-contiguous write-through stores (one 1 KiB line per wave and instruction), drain, flag ----
      float* slot = part + (size_t)pos * kG256SlotFloats + (size_t)threadIdx.x * 4;
      asm volatile("" : "+v"(slot));   // opaque: keeps the 32 loop-invariant store addresses from being hoisted (and spilled)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slot + (size_t)(i * 4 + j) * 2048), "v"(acc[i][j]) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flags + pos, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (kp1 != KP) {
        // ---- own the tile: add the later K ranges in ascending order ----
        const long need = (long)(mt + 1) * KP;
        long covered = u_end;
        int tm = team;
        for (int p = pos + t_tiles; covered < need; p += t_tiles) {
          if (threadIdx.x == 0)
            while (__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
          __syncthreads();
          const float* slot = part + (size_t)p * kG256SlotFloats + (size_t)threadIdx.x * 4;
          asm volatile("" : "+v"(slot));
#pragma unroll
          for (int qt = 0; qt < 4; ++qt) {   // four batches of eight 16-byte loads: 32 registers next to the 128 accumulators
            f32x4 tmp[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
              asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(tmp[q]) : "v"(slot + (size_t)(qt * 8 + q) * 2048) : "memory");
            // the waitcnt is tied to the loaded registers so that no use can be scheduled in front of it
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(tmp[0]), "+v"(tmp[1]), "+v"(tmp[2]), "+v"(tmp[3]), "+v"(tmp[4]), "+v"(tmp[5]), "+v"(tmp[6]), "+v"(tmp[7])
                         :: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[(qt * 8 + q) >> 2][(qt * 8 + q) & 3] += tmp[q];
          }
          __syncthreads();   // every wave has taken its part of the slot
          if (threadIdx.x == 0) __hip_atomic_store(flags + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tm += 1;
          covered = range_begin(tm + 1);
        }
      }
      // ---- epilogue (the data-parallel kernel's): lane holds rows m = .. + g*4 + e of token t = .. + l15 ----
      if (SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int t = t0 + wc * 64 + j * 16 + l15;
          if (t >= T) continue;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = so.silu_c0 + mt * 128 + wr * 64 + i * 16 + g * 4;
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float sg = silu_f(bf16_round_f(acc[i][j][e]));
              r[e] = (so.silu_round ? bf16_round_f(sg) : sg) * bf16_round_f(acc[i + 4][j][e]);
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
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int t = t0 + wc * 64 + j * 16 + l15;
          if (t >= T) continue;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int m = m0 + wr * 128 + i * 16 + g * 4;
            Half* dst = Y;
            int ld = M, mm = m, mlim = M;
            if (so.Y1) {
              const int b1 = so.M0 + so.M1, b2 = b1 + so.M2;
              if (m < so.M0) { ld = mlim = so.M0; }
              else if (m < b1) { dst = so.Y1; ld = mlim = so.M1; mm = m - so.M0; }
              else if (m < b2) { dst = so.Y2; ld = mlim = so.M2; mm = m - b1; }
              else { dst = so.Y3; ld = mlim = M - b2; mm = m - b2; }
            }
            if (mm + 3 < mlim) {
              u32x2 o;
              o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
              o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
              *reinterpret_cast<u32x2*>(dst + (size_t)t * ld + mm) = o;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (mm + e < mlim) dst[(size_t)t * ld + mm + e] = f2bf(acc[i][j][e]);
            }
          }
        }
      }
    }
    u += kp1 - kp0;
  }
}

// Stream-K plan, by shape and CU count only (the launcher and pegainfer_debug_gemm_route share it): taken - when switched
// on - where a round of 256 x 256 tiles would leave more than `min_waste_pct` % of the CU-rounds idle and the K range of a
// team is long enough to amortise a pipeline fill (>= 4 K-tile pairs).
// OFF by default: measured SLOWER than the data-parallel launch + tail at every shape it applies to (profiles/r6_streamk_*:
// gate_up 512 / 1024 / 2048 tokens 91 / 151 / 265 us against 71 / 137 / 230; TTFT(1024) 11.7 against 10.9 ms) - every
// workgroup publishes and re-reads a 256 KiB fp32 tile at the same moment, 64 MB each way through the fabric with nothing
// to overlap it, which costs more than the idle CU-rounds it removes.  PEGAINFER_STREAMK=1 / pegainfer_debug_streamk(1)
// switch it on (the GPU tests do: the kernel stays correct and deterministic).
extern int g_streamk_override;   // linear.hip: pegainfer_debug_streamk; -1 = the environment decides
inline bool gemm256_streamk_on() {
  static const bool v = [] { const char* e = getenv("PEGAINFER_STREAMK"); return e && e[0] == '1'; }();
  return g_streamk_override >= 0 ? g_streamk_override != 0 : v;
}
inline bool gemm256_streamk_plan(long tiles256, int K, int cus, int t_tiles) {
  static const int min_waste = [] { const char* e = getenv("PEGAINFER_STREAMK_MIN_WASTE"); return e && *e ? atoi(e) : 8; }();
  if (!gemm256_streamk_on() || cus < 8 || tiles256 <= 0 || t_tiles < 1 || t_tiles > cus / 8) return false;   // a team fits an XCD
  const long rounds = (tiles256 + cus - 1) / cus;
  const long waste_pct = 100 - 100 * tiles256 / (rounds * cus);
  const long nteams = cus / t_tiles, row_units = tiles256 / t_tiles * (K / (2 * G256_BK));
  const long idle_pct = 100 - 100 * nteams * t_tiles / cus;             // workgroups that fit no team
  return waste_pct >= min_waste + idle_pct && row_units / nteams >= 4;
}

inline void gemm256_streamk_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, float* part,
                                   uint32_t* flags, int cus, hipStream_t s) {
  const bool silu = so.silu_I > 0;
  const int m_tiles = silu ? ceil_div(silu_cols_end(so) - so.silu_c0, 128) : ceil_div(M, G256_BM), t_tiles = ceil_div(T, G256_BT);
  so.w_nt = 0;
  if (silu) {
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm256_streamk_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kG256LdsBytes);
      return true;
    }();
    (void)once;
    mfma_gemm256_streamk_kernel<true><<<cus, 512, kG256LdsBytes, s>>>(W, X, Y, M, T, K, m_tiles, t_tiles, so, part, flags);
  } else {
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm256_streamk_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kG256LdsBytes);
      return true;
    }();
    (void)once;
    mfma_gemm256_streamk_kernel<false><<<cus, 512, kG256LdsBytes, s>>>(W, X, Y, M, T, K, m_tiles, t_tiles, so, part, flags);
  }
}

// ---- 128 x 256 tiles: the same machinery for matrices with too few 256-row tiles to fill the chip (qkv / o_proj /
// down_proj at ~0.5-2 k tokens: 2560 rows are 10 tiles of 256 but 20 of 128).  8 waves as 2 x 4, wave (wr, wc) owns
// 64 W rows x 64 token rows = 4 x 4 MFMA tiles (64 accumulator VGPRs).  A K tile is THREE 16 KiB half-tiles - A (all 128
// rows), B_lo / B_hi (token rows of n-tiles 0-1 / 2-3 of every wave column) - and two phases:
//     phase 1: read A + B_lo, quadrant lo (16 MFMAs)        phase 2: read B_hi, quadrant hi (16 MFMAs)
// LDS holds a ring of three K tiles (144 KiB): the three half-tiles of K tile kt + 2 are staged during K tile kt into the
// slots K tile kt - 1 was read from, one counted vmcnt(6) per K tile (tile kt + 2 stays in flight across the barrier,
// tile kt + 1 has landed), wave row 1 runs one barrier behind wave row 0 as above.  Per-element K order unchanged: same
// bits as the other tiled kernels.
// Measured (profiles/r3_gemm128x256_ab.txt, 1024 tokens, cold weights): qkv 57 -> 43 us, down_proj 84 -> 67 us, o_proj
// 45 -> 41 us; per-CU efficiency ~40-45 % (the 256 x 256 kernel: 57 %): a 128 x 256 tile moves 48 KiB into LDS per 32
// MFMAs per wave, ~3/4 of a CU's L1 fill rate at the MFMA-bound pace.  A variant with double-buffered fragment registers
// and ONE barrier per K tile (every read a quadrant ahead of its MFMAs, no wave-row skew) was 25 % SLOWER (qkv 55 us):
// without the skew both waves of a SIMD sit in the same segment and nothing feeds the matrix pipe across the barrier.
constexpr int G128_BM = 128;
constexpr int kG128LdsBytes = 9 * kG256HalfBytes;
enum { kG128A = 0, kG128Blo = 1, kG128Bhi = 2 };

// NF = 4 (round 4): four more waves that do nothing but issue the DMAs (12 per K tile each), in the barrier rhythm of a
// wave-row-0 wave; the eight compute waves issue none.  A DMA instruction costs its wave ~133 cycles of issue slot
// (tools/probes/ingest_probe) - six per wave and K tile were 800 cycles next to 32 MFMAs = 512.  159 VGPRs: three waves
// per SIMD fit.  NF = 0: the round-3 form (every wave stages its share), kept for the A/B (PEGAINFER_GEMM128X256_FEED=0).
// MT (round 6) = 16-row MFMA tiles per wave row: 4 = the 128-row tile; 3 = a 96-row tile (plain NF = 4 form only) for
// matrices whose 128-row tiling leaves CUs idle - the stacked qkv at 768 / 1024 tokens: 144 / 192 tiles of 128 rows, 192 / 256
// of 96.  Same per-element K order: same bits.
template <bool SILU, int NF, int MT = 4>
__global__ __launch_bounds__(512 + NF * 64) void mfma_gemm128x256_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                               Half* __restrict__ Y, int M, int T, int K, int m_tiles,
                                                               int t_tiles, SplitOut so, float* __restrict__ part,
                                                               int nk_slice) {
  extern __shared__ __attribute__((aligned(16))) u32x4 g256_smem[];
  const int ntiles = m_tiles * t_tiles;
  int tile = blockIdx.x, zslice = (int)blockIdx.y;
  if (part && so.xcd_slices) {
    // (round 6) K-split launches: the hardware deals workgroups to XCDs by their linear id; give every XCD one contiguous run of
    // (slice, tile) work items in SLICE-major order, so an XCD works on (mostly) one K slice of a few row tiles.  With the slices
    // of a tile on one XCD (the 2-D grid's natural order: gridDim.x is a multiple of 8) every XCD pulled the WHOLE of X through
    // its L2 - down_proj at 1024 tokens fetched 209 MB from the memory side for 70 MB of operands (FETCH_SIZE, profiles/
    // r6b_ctx1024_pmc_fetch.csv).  Same work items, same arithmetic: same bits.
    const int nitems = ntiles * (int)gridDim.y, id = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
    const int q = nitems / 8, r = nitems % 8, xcd = id % 8, idx = id / 8;
    const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    zslice = item / ntiles;
    tile = item - zslice * ntiles;
  } else {
    const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = so.t_major ? tile % m_tiles : tile / t_tiles, tt = so.t_major ? tile / m_tiles : tile - mt * t_tiles;   // see mfma_gemm256_kernel
  constexpr int BMT = 32 * MT;            // W rows per tile
  const int m0 = mt * BMT, t0 = tt * G256_BT;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;

  constexpr int NI = NF ? NF : 8;           // waves that issue DMAs
  constexpr int PPW = 16 / NI;              // 1 KiB pieces per issuing wave and half-tile
  constexpr int PPA = 4 * MT / NI;          // ... of the A half-tile (32 * MT rows = 4 * MT pieces)
  constexpr int HT = MT / 2;                // SwiGLU form: gate m-tiles per wave row (their up rows are the wave row's other half)
  static_assert((4 * MT) % NI == 0 && (MT % 2 == 0 || !SILU), "the SwiGLU form pairs m-tiles: an even count per wave row");
  const bool feeder = NF && wave >= 8;
  const int wi = NF ? wave - 8 : wave;      // index among the issuing waves (compute waves of the NF form: negative, unused)
  const Half* src[3][PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int p = (j * NI + (wi < 0 ? 0 : wi)) * 64 + lane, hrow = p >> 3, cs = (p & 7) ^ (hrow & 7);
    int row;
    if (SILU) {   // a wave row's m-tiles 0 .. HT-1 are 16 * HT gate rows, HT .. MT-1 their up rows: gate and up of an element meet in one lane
      const int hr = hrow < BMT ? hrow : BMT - 1, wrow = hr / (16 * MT), ti = (hr >> 4) - wrow * MT, up = ti >= HT;
      int gr = so.silu_c0 + mt * (16 * MT) + wrow * (16 * HT) + (ti - (up ? HT : 0)) * 16 + (hr & 15);
      gr = gr < silu_cols_end(so) ? gr : silu_cols_end(so) - 1;
      row = gr + (up ? so.silu_I : 0);
    } else {
      row = m0 + (hrow < BMT ? hrow : BMT - 1);
      row = row < M ? row : M - 1;
    }
    src[kG128A][j] = W + (size_t)row * K + cs * 8;
    const int wc_ = hrow >> 5, r = hrow & 31;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      int tr = t0 + wc_ * 64 + hi * 32 + r;
      tr = tr < T ? tr : T - 1;
      src[hi ? kG128Bhi : kG128Blo][j] = X + (size_t)tr * K + cs * 8;
    }
  }
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&g256_smem[0];
  const int kt_begin = part ? zslice * nk_slice : 0;
  const int nk_total = K / G256_BK;
  const int nk = part ? (nk_total - kt_begin < nk_slice ? nk_total - kt_begin : nk_slice) : nk_total;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int j = 0; j < PPW; ++j) src[w][j] += (size_t)kt_begin * G256_BK;
  // stage half-tile `which` of K tile kt into ring slot `ring` (clamped past the end: a dead slot, the load only keeps
  // vmcnt uniform)
  auto stage = [&](int which, int kt, int ring) {
    const int ktc = kt < nk ? kt : nk - 1;
    const uint32_t slot = lds0 + (uint32_t)((ring * 3 + which) * kG256HalfBytes);
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      if (which == kG128A && j >= PPA) break;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(slot + (uint32_t)(j * NI + wi) * 1024u);
      if (which == kG128A && so.w_nt)   // a single token tile: W bytes have one reader (SplitOut::w_nt)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                     :: "v"(src[which][j] + (size_t)ktc * G256_BK), "s"(dst) : "memory", "m0");
      else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src[which][j] + (size_t)ktc * G256_BK), "s"(dst) : "memory", "m0");
    }
  };
  auto half_ptr = [&](int ring, int which) { return g256_smem + (size_t)(ring * 3 + which) * (kG256HalfBytes / 16); };

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t a[MT][2], b[2][2];
  auto read_a = [&](int ring) {
    const u32x4* h = half_ptr(ring, kG128A);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        a[i][ks] = __builtin_bit_cast(bf16x8_t, h[lds_slot(wr * (16 * MT) + i * 16 + l15, ks * 4 + g)]);
  };
  auto read_b = [&](int ring, int which) {
    const u32x4* h = half_ptr(ring, which);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        b[j][ks] = __builtin_bit_cast(bf16x8_t, h[lds_slot(wc * 32 + j * 16 + l15, ks * 4 + g)]);
  };
  auto quad = [&](int jh) {   // 16 MFMAs: all 4 m-tiles, n-tiles jh*2..
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[i][jh * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  constexpr int kTileDmas = PPA + 2 * PPW;  // DMAs per issuing wave and K tile: the counted wait leaves one tile in flight
  if (feeder) {
    // the whole staging side, in the barrier rhythm of a wave-row-0 wave (the EARLIEST any wave of the round-3 form touched
    // a ring slot, so the hazards are the ones argued above): A + B_lo of tile kt + 2 in phase 1, B_hi in phase 2
    stage(kG128A, 0, 0); stage(kG128Blo, 0, 0); stage(kG128Bhi, 0, 0);
    stage(kG128A, 1, 1); stage(kG128Blo, 1, 1); stage(kG128Bhi, 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kTileDmas) : "memory");
    __builtin_amdgcn_s_barrier();
    int ring = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int ring2 = ring + 2 >= 3 ? ring - 1 : ring + 2;
      stage(kG128A, kt + 2, ring2);
      stage(kG128Blo, kt + 2, ring2);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      stage(kG128Bhi, kt + 2, ring2);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kTileDmas) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      ring = ring + 1 == 3 ? 0 : ring + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail loads must not outlive the workgroup's LDS
    __builtin_amdgcn_s_barrier();
    return;
  }
  // ---- prologue: K tiles 0 and 1 requested, tile 0 landed ----
  if constexpr (NF == 0) {
    stage(kG128A, 0, 0); stage(kG128Blo, 0, 0); stage(kG128Bhi, 0, 0);
    stage(kG128A, 1, 1); stage(kG128Blo, 1, 1); stage(kG128Bhi, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind the first

  // one K tile = two phases (fragment reads + DMA issue | barrier | 16 MFMAs | barrier); the counted wait sits in front
  // of the first barrier of the last phase, so it covers both wave rows (see the 256 x 256 kernel)
  auto ktile = [&](int kt, int ring) {
    const int ring2 = (ring + 2) % 3;
    read_b(ring, kG128Blo);
    __builtin_amdgcn_sched_barrier(0);
    read_a(ring);
    if constexpr (NF == 0) {
      stage(kG128A, kt + 2, ring2);
      stage(kG128Blo, kt + 2, ring2);
    }
    __builtin_amdgcn_s_barrier();
    quad(0);
    __builtin_amdgcn_s_barrier();
    read_b(ring, kG128Bhi);
    if constexpr (NF == 0) {
      stage(kG128Bhi, kt + 2, ring2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    quad(1);
    __builtin_amdgcn_s_barrier();
  };
  for (int kt = 0; kt < nk; kt += 3) {
    ktile(kt, 0);
    if (kt + 1 < nk) ktile(kt + 1, 1);
    if (kt + 2 < nk) ktile(kt + 2, 2);
  }
  if constexpr (NF == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail loads must not outlive the workgroup's LDS
  if (wr == 0) __builtin_amdgcn_s_barrier();

  if constexpr (SILU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + wc * 64 + j * 16 + l15;
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < HT; ++i) {
        const int m = so.silu_c0 + mt * (16 * MT) + wr * (16 * HT) + i * 16 + g * 4;
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // the GEMM output is bf16 before SwiGLU (fused_proj.cu:57-62)
          const float sg = silu_f(bf16_round_f(acc[i][j][e]));
          r[e] = (so.silu_round ? bf16_round_f(sg) : sg) * bf16_round_f(acc[i + HT][j][e]);
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
    return;
  }
  if (part) {
    float* pz = part + (size_t)zslice * T * M;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + wc * 64 + j * 16 + l15;
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m0 + wr * (16 * MT) + i * 16 + g * 4;
        if (m + 3 < M) {
          *reinterpret_cast<f32x4*>(pz + (size_t)t * M + m) = acc[i][j];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (m + e < M) pz[(size_t)t * M + m + e] = acc[i][j][e];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + wc * 64 + j * 16 + l15;
    if (t >= T) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wr * (16 * MT) + i * 16 + g * 4;
      Half* dst = Y;
      int ld = M, mm = m, mlim = M;
      if (so.Y1) {
        const int b1 = so.M0 + so.M1, b2 = b1 + so.M2;
        if (m < so.M0) { ld = mlim = so.M0; }
        else if (m < b1) { dst = so.Y1; ld = mlim = so.M1; mm = m - so.M0; }
        else if (m < b2) { dst = so.Y2; ld = mlim = so.M2; mm = m - b1; }
        else { dst = so.Y3; ld = mlim = M - b2; mm = m - b2; }
      }
      if (mm + 3 < mlim) {
        u32x2 o;
        o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
        o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
        *reinterpret_cast<u32x2*>(dst + (size_t)t * ld + mm) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (mm + e < mlim) dst[(size_t)t * ld + mm + e] = f2bf(acc[i][j][e]);
      }
    }
  }
}

inline bool gemm128x256_ok(int M, int T, int K) { return (K % G256_BK) == 0 && K >= 128 && M >= 128 && T > 64; }

// plain (part == nullptr, ksplit == 1) or split-K (fp32 partials into part[ksplit][T][M]; the caller sums the slices)
inline bool gemm128x256_feed_on() {
  static const bool v = [] { const char* e = getenv("PEGAINFER_GEMM128X256_FEED"); return !(e && e[0] == '0'); }();
  return v;
}
template <bool SILU, int NF, int MT = 4>
inline void gemm128x256_launch_nf(const Half* W, const Half* X, Half* Y, int M, int T, int K, const SplitOut& so, float* part,
                                  int ksplit, int nk_slice, int m_tiles, int t_tiles, hipStream_t s) {
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm128x256_kernel<SILU, NF, MT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kG128LdsBytes);
    return true;
  }();
  (void)once;
  mfma_gemm128x256_kernel<SILU, NF, MT><<<dim3(m_tiles * t_tiles, part ? ksplit : 1), 512 + NF * 64, kG128LdsBytes, s>>>(
      W, X, Y, M, T, K, m_tiles, t_tiles, so, part, nk_slice);
}
// 96-row tiles (MT = 3), plain un-split feeder form: see the kernel's header
inline void gemm96x256_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, hipStream_t s) {
  const int t_tiles = ceil_div(T, G256_BT);
  so.w_nt = t_tiles == 1 && weights_nt_on();
  gemm128x256_launch_nf<false, 4, 3>(W, X, Y, M, T, K, so, nullptr, 1, 0, ceil_div(M, 96), t_tiles, s);
}
// SwiGLU form on 64-row tiles (MT = 2: 16 gate rows + their 16 up rows per wave row): the thin tail behind the 256 x 256 round
inline void gemm64x256_silu_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, hipStream_t s) {
  const int t_tiles = ceil_div(T, G256_BT);
  so.w_nt = t_tiles == 1 && weights_nt_on();
  gemm128x256_launch_nf<true, 4, 2>(W, X, Y, M, T, K, so, nullptr, 1, 0, ceil_div(silu_cols_end(so) - so.silu_c0, 32), t_tiles, s);
}
inline void gemm128x256_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, float* part,
                               int ksplit, int nk_slice, hipStream_t s) {
  const int t_tiles = ceil_div(T, G256_BT);
  so.w_nt = t_tiles == 1 && weights_nt_on();
  so.t_major = gemm_t_major(M, T, so);
  so.xcd_slices = part && ksplit > 1 && splitk_xcd_on();
  const bool feed = gemm128x256_feed_on();
  if (so.silu_I > 0) {   // SwiGLU form (un-split only): a tile = 64 gate rows + their 64 up rows
    const int m_tiles = ceil_div(silu_cols_end(so) - so.silu_c0, 64);
    if (feed) gemm128x256_launch_nf<true, 4>(W, X, Y, M, T, K, so, nullptr, 1, 0, m_tiles, t_tiles, s);
    else gemm128x256_launch_nf<true, 0>(W, X, Y, M, T, K, so, nullptr, 1, 0, m_tiles, t_tiles, s);
    return;
  }
  const int m_tiles = ceil_div(M, G128_BM);
  if (feed) gemm128x256_launch_nf<false, 4>(W, X, Y, M, T, K, so, part, ksplit, nk_slice, m_tiles, t_tiles, s);
  else gemm128x256_launch_nf<false, 0>(W, X, Y, M, T, K, so, part, ksplit, nk_slice, m_tiles, t_tiles, s);
}

}  // namespace pk
