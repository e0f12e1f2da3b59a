// GEMM call sites of the pegainfer-kernels ABI (reference csrc/linear.cu:45-75: cublasGemmEx
// OP_T/OP_N, bf16 in, fp32 accumulate, bf16 out) re-designed for MI355X.
//
//   Y[T, M] = X[T, K] . W[M, K]^T     W row-major, X / Y token-major ("column-major [K,N]" in
//                                      the reference's words), T = the reference's N.
//
// Three kernels, chosen by shape only (never by data):
//   gemv_kernel<NT,RPW,KSPLIT>  T <= 16  - decode.  HBM-bound weight streaming: every W element
//       is read exactly once with 16-byte non-temporal loads straight into VGPRs (no LDS round
//       trip for the streamed operand), x staged in LDS in K-tiles and shared by the 4 waves,
//       v_dot2c_f32_bf16 accumulation, wave64 butterfly reduction.  K >= 4096 splits K across
//       the 4 waves of a workgroup (LDS combine, fixed order) so that M = 2560 still yields
//       1280 workgroups for 256 CUs.  The per-(row, token) summation order depends only on
//       (K, KSPLIT), never on T or M: batch decode == single decode bit-for-bit and a row
//       slice of the fused QKV matrix == the fused GEMM's rows (reference relies on the latter,
//       batch_decode.rs:160-163).
//   mfma_gemm_kernel            T > 16   - prefill.  128x128x64 tiles, 4 waves (2x2), each wave
//       4x4 blocks of v_mfma_f32_16x16x32_bf16, register-staged double-buffered LDS with an XOR
//       slot swizzle, XCD-aware tile order (each XCD's L2 keeps one W panel hot).
//   naive_gemm_kernel           any shape the fast paths cannot take (K % 8 != 0, unaligned).
#include <cstdlib>

#include "common.h"
#include "gemv_core.h"
namespace pk { unsigned long long* g_gemv_trace = nullptr; }
#include "gemm_skinny.h"
#include "pegainfer_kernels_ext.h"

namespace pk {

// ------------------------------------------------------------------ naive fallback
__global__ __launch_bounds__(256) void naive_gemm_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                         Half* __restrict__ Y, int M, int T, int K) {
  const long out = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (out >= (long)M * T) return;
  const int t = (int)(out / M), m = (int)(out % M);
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += bf2f(W[(size_t)m * K + k]) * bf2f(X[(size_t)t * K + k]);
  acc = wave_sum(acc);
  if (lane == 0) Y[(size_t)t * M + m] = f2bf(acc);
}

// ------------------------------------------------------------------ prefill MFMA GEMM
constexpr int BM = 128, BT = 128, BK = 64;
// LDS tile [128 rows][64 bf16] = 128 B per row = 8 slots of 16 B; slot' = slot ^ (row & 7)
__device__ __forceinline__ int lds_slot(int row, int slot) { return row * 8 + (slot ^ (row & 7)); }

__global__ __launch_bounds__(256) void mfma_gemm_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                        Half* __restrict__ Y, int M, int T, int K,
                                                        int m_tiles, int t_tiles) {
  __shared__ __attribute__((aligned(16))) u32x4 ws[2][BM * 8];
  __shared__ __attribute__((aligned(16))) u32x4 xs[2][BT * 8];
  // XCD-aware order: hardware places block b on XCD b % 8; give each XCD a contiguous run of
  // tiles, token-tile fastest, so one W panel stays in that XCD's L2 while it is reused.
  const int ntiles = m_tiles * t_tiles;
  int tile = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective remap
  }
  const int mt = tile / t_tiles, tt = tile - mt * t_tiles;
  const int m0 = mt * BM, t0 = tt * BT;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * 64, wt = (wave & 1) * 64;  // wave's 64x64 sub-tile
  const int l15 = lane & 15, g = lane >> 4;

  // staging assignment: thread -> 4 chunks per operand: chunk id = threadIdx.x + j*256 -> (row, slot)
  const Half* wsrc[4];
  const Half* xsrc[4];
  int sdst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cid = threadIdx.x + j * 256, row = cid >> 3, slot = cid & 7;
    int mr = m0 + row; mr = mr < M ? mr : M - 1;
    int tr = t0 + row; tr = tr < T ? tr : T - 1;
    wsrc[j] = W + (size_t)mr * K + slot * 8;
    xsrc[j] = X + (size_t)tr * K + slot * 8;
    sdst[j] = lds_slot(row, slot);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 wreg[4], xreg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { wreg[j] = *reinterpret_cast<const u32x4*>(wsrc[j]); xreg[j] = *reinterpret_cast<const u32x4*>(xsrc[j]); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { ws[0][sdst[j]] = wreg[j]; xs[0][sdst[j]] = xreg[j]; }
  __syncthreads();

  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wreg[j] = *reinterpret_cast<const u32x4*>(wsrc[j] + (size_t)(kt + 1) * BK);
        xreg[j] = *reinterpret_cast<const u32x4*>(xsrc[j] + (size_t)(kt + 1) * BK);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8_t, ws[cur][lds_slot(wm + i * 16 + l15, ks * 4 + g)]);
        b[i] = __builtin_bit_cast(bf16x8_t, xs[cur][lds_slot(wt + i * 16 + l15, ks * 4 + g)]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { ws[cur ^ 1][sdst[j]] = wreg[j]; xs[cur ^ 1][sdst[j]] = xreg[j]; }
    }
    __syncthreads();
  }
  // C layout (16x16x32): col = lane&15 -> token, rows (lane>>4)*4 + i -> 4 consecutive W rows
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + wt + j * 16 + l15;
    if (t >= T) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm + i * 16 + g * 4;
      if (m + 3 < M) {
        u32x2 o;
        o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
        o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
        *reinterpret_cast<u32x2*>(Y + (size_t)t * M + m) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (m + e < M) Y[(size_t)t * M + m + e] = f2bf(acc[i][j][e]);
      }
    }
  }
}

// Same tiling and the same per-element K order (so results are bit-identical to mfma_gemm_kernel), but
// the tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  The
// DMA writes LDS lane-linearly (base + lane*16), so the bank swizzle is applied to each lane's SOURCE
// address: lane l of row-group rg fetches global slot (l%8) ^ (row&7) of row rg*8 + l/8, and readers use
// lds_slot() unchanged.  TT = token-tile width (128, or 64 when M is small and 128-wide tiles would leave
// CUs idle).
// Optional row-segmented output (up to 4 segments): W rows [0,M0) -> Y[T][M0], [M0,M0+M1) -> Y1[T][M1],
// [M0+M1,M0+M1+M2) -> Y2[T][M2], the rest -> Y3 (one GEMM over a stacked weight writing the separate contiguous
// buffers the downstream kernels take).  Y1 == nullptr means a plain [T][M] output; unused trailing segments have
// size 0.  Segment sizes are multiples of 4 (a lane stores 4 consecutive rows).
// silu_I > 0 selects the SwiGLU form instead: W = [gate (I rows); up (I rows)], Y[T][I] = silu_mul_fused(W . x).  A
// workgroup then owns 64 gate rows AND the 64 up rows below them (each wave 32 + 32, so gate and up of an element
// meet in one lane's accumulators); per-element K order is unchanged, so the result equals gemm + silu_mul_fused.
// silu_round: the Qwen3.5 activation bf16(bf16(silu(g)) * u) (elementwise.cu:28-42) instead of one rounding.
// silu_c0 / silu_c1: the activation columns [c0, c1) this launch computes (0, 0 = all of [0, silu_I)): lets two launches
// with different tile shapes share one SwiGLU GEMM (the thin last round of the 256 x 256 tiling, glds_gemm_launch)
// w_nt (set by the launchers, never by callers): the W tiles are requested with the non-temporal policy - each W byte is
// read by ONE workgroup once (a single token tile), so it should not displace the x rows every workgroup re-reads from L2.
struct SplitOut { Half* Y1; Half* Y2; Half* Y3; int M0; int M1; int M2; int silu_I; int silu_round; int silu_c0; int silu_c1; int w_nt; int xcd_slices; int t_major; int chunk; };
// t_major (set by the gemm256.h launchers for plain GEMMs with T > M): row tile fastest inside an XCD's run of tiles
// xcd_slices (set by the K-split launchers of gemm256.h): deal the (slice, tile) work items to the XCDs in slice-major runs.  NOT for
// the 128-row kernel below: at 256 tokens (6-8 slices: 0.8-1.6 KB of every W row per XCD) it measured 1.1 % slower (r6_splitk_xcd_ab2.txt)
__host__ __device__ inline int silu_cols_end(const SplitOut& so) { return so.silu_c1 > 0 ? so.silu_c1 : so.silu_I; }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// PEGAINFER_WEIGHTS_NT=0: A/B switch for SplitOut::w_nt (tools/bench_prefill_gemm.py, tools/batch_sweep.py)
inline bool weights_nt_on() {
  static const bool v = [] { const char* e = getenv("PEGAINFER_WEIGHTS_NT"); return !(e && e[0] == '0'); }();
  return v;
}

}  // namespace pk
#include "gemm256.h"   // 256 x 256 tiles, 8 waves, 8-phase schedule (long prompts); uses SplitOut / lds_slot
#include "gemm_stream.h"   // CU-count-sized row tiles for 17..64 token columns (round 4); uses SplitOut / lds_slot / BK
namespace pk {

// SPLITK: blockIdx.y = K slice z of `ksplit` (each nk_slice K tiles); the workgroup writes its fp32 partial tile to
// part[z][T][M] and splitk_reduce_kernel adds the slices in z order (decode batches of 17..64 columns on matrices
// with < 128 row tiles: more workgroups without re-staging x per 16 rows like the skinny kernel has to).
template <int TT, int ST, bool SPLITK = false>
__global__ __launch_bounds__(256) void mfma_gemm_glds_kernel(const Half* __restrict__ W, const Half* __restrict__ X,
                                                             Half* __restrict__ Y, int M, int T, int K,
                                                             int m_tiles, int t_tiles, SplitOut so,
                                                             float* __restrict__ part = nullptr, int nk_slice = 0) {
  constexpr int TJ = TT / 32;       // 16-token blocks per wave
  constexpr int XG = TT / 32;       // X row-groups (8 rows) staged per wave
  constexpr int NG = 4 + XG;        // LDS-DMA instructions per wave per K tile
  extern __shared__ __attribute__((aligned(16))) u32x4 glds_smem[];   // ST x (W tile | X tile)
  u32x4(*ws)[BM * 8] = reinterpret_cast<u32x4(*)[BM * 8]>(glds_smem);
  u32x4(*xs)[TT * 8] = reinterpret_cast<u32x4(*)[TT * 8]>(glds_smem + ST * BM * 8);
  const int ntiles = m_tiles * t_tiles;
  int tile = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = tile / t_tiles, tt = tile - mt * t_tiles;
  const int m0 = mt * BM, t0 = tt * TT;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * 64, wt = (wave & 1) * (TT / 2);
  const int l15 = lane & 15, g = lane >> 4;

  // per-lane DMA sources: W row-groups wave, wave+4, wave+8, wave+12; X row-groups wave + 4*j
  const Half* wsrc[4];
  const Half* xsrc[XG];
  const int lr = lane >> 3, ls = lane & 7;
  const int kt_begin = SPLITK ? (int)blockIdx.y * nk_slice : 0;
  const size_t kofs = (size_t)kt_begin * BK;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave + 4 * j) * 8 + lr;
    int mr;
    if (so.silu_I > 0) {  // tile row -> (wave group, 16-row block i, r16): blocks 0,1 = gate rows, 2,3 = up rows
      int gr = so.silu_c0 + mt * 64 + (row >> 6) * 32 + ((row >> 4) & 1) * 16 + (row & 15);
      gr = gr < silu_cols_end(so) ? gr : silu_cols_end(so) - 1;
      mr = gr + (((row >> 4) & 2) ? so.silu_I : 0);
    } else {
      mr = m0 + row; mr = mr < M ? mr : M - 1;
    }
    wsrc[j] = W + (size_t)mr * K + kofs + ((ls ^ (row & 7)) << 3);
  }
#pragma unroll
  for (int j = 0; j < XG; ++j) {
    const int row = (wave + 4 * j) * 8 + lr;
    int tr = t0 + row; tr = tr < T ? tr : T - 1;
    xsrc[j] = X + (size_t)tr * K + kofs + ((ls ^ (row & 7)) << 3);
  }
  f32x4 acc[4][TJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The DMA is issued from inline asm on purpose: hipcc orders every later ds_read behind a compiler-visible
  // LDS-DMA with s_waitcnt vmcnt(0) (it cannot tell the two halves of ws/xs apart), which would serialise
  // load and math.  The hand-placed vmcnt(0) below, just before the barrier, is the only wait the DMA needs:
  // buffer cur^1 was last read before the previous barrier and is first read after the next one.
  const uint32_t ws_lds = (uint32_t)(uintptr_t)(lptr_t)&ws[0][0], xs_lds = (uint32_t)(uintptr_t)(lptr_t)&xs[0][0];
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t dst = __builtin_amdgcn_readfirstlane(ws_lds + (uint32_t)(buf * BM * 8 + (wave + 4 * j) * 64) * 16u);
      if (so.w_nt)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                     :: "v"(wsrc[j] + (size_t)kt * BK), "s"(dst) : "memory", "m0");
      else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(wsrc[j] + (size_t)kt * BK), "s"(dst) : "memory", "m0");
    }
#pragma unroll
    for (int j = 0; j < XG; ++j) {
      const uint32_t dst = __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)(buf * TT * 8 + (wave + 4 * j) * 64) * 16u);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                   :: "v"(xsrc[j] + (size_t)kt * BK), "s"(dst) : "memory", "m0");
    }
  };
  // ST-deep ring, prefetch distance ST-1.  Tile kt+ST-1 goes into the buffer tile kt-1 was read from (every wave
  // left that compute before the barrier that ended iteration kt-1).  A wave's vmcnt counts its own DMAs in
  // issue order, so "at most (ST-2)*NG outstanding" == tile kt+1 has landed; the barrier extends that to all waves.
  const int nk_all = K / BK;
  const int nk = SPLITK ? (nk_all - kt_begin < nk_slice ? nk_all - kt_begin : nk_slice) : nk_all;
#pragma unroll
  for (int p = 0; p < ST - 1; ++p)
    if (p < nk) stage(p, p);
  if (nk > ST - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((ST - 2) * NG) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0, nxt = ST - 1;  // ring slots of tile kt and tile kt+ST-1
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + ST - 1 < nk;
    if (more) stage(nxt, kt + ST - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4], b[TJ];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = __builtin_bit_cast(bf16x8_t, ws[cur][lds_slot(wm + i * 16 + l15, ks * 4 + g)]);
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        b[j] = __builtin_bit_cast(bf16x8_t, xs[cur][lds_slot(wt + j * 16 + l15, ks * 4 + g)]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((ST - 2) * NG) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur = cur + 1 == ST ? 0 : cur + 1;
    nxt = nxt + 1 == ST ? 0 : nxt + 1;
  }
  if constexpr (SPLITK) {   // fp32 partial tile (M % 4 == 0: a lane's 4 rows are all in or all out)
    float* dst = part + (size_t)blockIdx.y * T * M;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int t = t0 + wt + j * 16 + l15;
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm + i * 16 + g * 4;
        if (m < M) *reinterpret_cast<f32x4*>(dst + (size_t)t * M + m) = acc[i][j];
      }
    }
    return;
  }
  if (so.silu_I > 0) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int t = t0 + wt + j * 16 + l15;
      if (t >= T) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = so.silu_c0 + mt * 64 + (wave >> 1) * 32 + i * 16 + g * 4;
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)  // the GEMM output is bf16 before SwiGLU (fused_proj.cu:57-62)
        {
          const float sg = silu_f(bf16_round_f(acc[i][j][e]));
          r[e] = (so.silu_round ? bf16_round_f(sg) : sg) * bf16_round_f(acc[i + 2][j][e]);
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
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int t = t0 + wt + j * 16 + l15;
    if (t >= T) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm + i * 16 + g * 4;
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

// Y[t][m] = bf16(sum over slices z (ascending) of part[z][t][m]); 4 rows per thread, row-segmented like the GEMM
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, Half* __restrict__ Y, int M,
                                                            int T, int ksplit, SplitOut so) {
  const int m4 = M >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)T * m4) return;
  const int t = (int)(idx / m4), m = (int)(idx - (long)t * m4) * 4;
  f32x4 pv[8];   // ksplit <= 8: every slice's load leaves before the first add (slice order is kept)
#pragma unroll
  for (int z = 0; z < 8; ++z)
    if (z < ksplit) pv[z] = *reinterpret_cast<const f32x4*>(part + ((size_t)z * T + t) * M + m);
  f32x4 v = pv[0];
#pragma unroll
  for (int z = 1; z < 8; ++z)
    if (z < ksplit) v += pv[z];
  Half* dst = Y;
  int ld = M, mm = m;
  if (so.Y1) {
    const int b1 = so.M0 + so.M1, b2 = b1 + so.M2;
    if (m < so.M0) { ld = so.M0; }
    else if (m < b1) { dst = so.Y1; ld = so.M1; mm = m - so.M0; }
    else if (m < b2) { dst = so.Y2; ld = so.M2; mm = m - b1; }
    else { dst = so.Y3; ld = M - b2; mm = m - b2; }
  }
  u32x2 o;
  o.x = pack_bf2(v[0], v[1]);
  o.y = pack_bf2(v[2], v[3]);
  *reinterpret_cast<u32x2*>(dst + (size_t)t * ld + mm) = o;
}

// (round 6) un-split shapes of pegainfer_gemm_add_then_rms_norm: out = bf16(a + y), normed = rms_norm(out) * w in ONE pass over
// the rows instead of add_cuda + rms_norm_batched_cuda (10 000 tokens: 25 + 22 us and 256 MB -> one launch, 205 MB).  One wave per
// row, the canonical order of norm_core.h over the ROUNDED sum: the same bits as the two calls.  out may alias a.
template <int VPL>   // 0: two passes over the row (any d); 5 / 8: the row in registers (d <= 2560 / 4096), one memory pass
__global__ __launch_bounds__(256) void add_then_rms_norm_vec_kernel(const Half* a, const Half* __restrict__ y,
                                                                    const Half* __restrict__ w, Half* out,
                                                                    Half* __restrict__ normed, int d, int rows, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if constexpr (VPL > 0) {
    wave_add_norm_row_cached<VPL, true>(a + (size_t)row * d, y + (size_t)row * d, w, out + (size_t)row * d,
                                        normed + (size_t)row * d, d, eps, 0.0f);
    return;
  }
  const int lane = threadIdx.x & 63;
  const Half* ar = a + (size_t)row * d;
  const Half* yr = y + (size_t)row * d;
  const float inv = wave_row_inv_rms(ar, yr, d, eps, true);
  const int nvec = d >> 3;
  for (int i = lane; i < nvec; i += 64) {
    const u32x4 r = reinterpret_cast<const u32x4*>(yr)[i];
    u32x4 nh;
    const u32x4 o = norm_scale8(reinterpret_cast<const u32x4*>(ar)[i], &r, reinterpret_cast<const u32x4*>(w)[i], inv, 0.f, &nh, true);
    reinterpret_cast<u32x4*>(out + (size_t)row * d)[i] = nh;
    reinterpret_cast<u32x4*>(normed + (size_t)row * d)[i] = o;
  }
}

// splitk_reduce_kernel + fused_add_rms_norm_batched_cuda in one launch, one workgroup per token row (d = M):
//   r = bf16(sum_z part[z][t][:])  (what the reduce kernel would have stored),  hidden = bf16(hidden + r),
//   out = bf16((hidden + r)_fp32 * inv_rms * w).  All 512 threads build r in LDS; wave 0 then takes the row's sum of
// squares in the canonical one-wave order of norm_core.h, so the result is bit-identical to the two-kernel sequence.
// round_sum = false: FlashInfer's fused add + norm (the norm sees the un-rounded fp32 sum; hidden_in == hidden_out).
// round_sum = true:  add_cuda, then rms_norm on its bf16 output (prefill.rs:183 + the next layer's prefill.rs:89):
//                    hidden_out = bf16(hidden_in + r), out = norm(hidden_out) - the norm sees the ROUNDED sum.
__global__ __launch_bounds__(512) void splitk_reduce_add_norm_kernel(const float* __restrict__ part,
                                                                     const Half* __restrict__ hidden_in,
                                                                     Half* __restrict__ hidden,
                                                                     const Half* __restrict__ w, Half* __restrict__ out,
                                                                     int d, int T, int ksplit, float eps, bool round_sum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rn_smem[];
  u32x4* rrow = reinterpret_cast<u32x4*>(rn_smem);              // [d / 8] bf16x8: the slice sum r
  u32x4* hrow = reinterpret_cast<u32x4*>(rn_smem + (size_t)d * 2);   // [d / 8] bf16x8: the residual row (round 6)
  float* sm_inv = reinterpret_cast<float*>(rn_smem + (size_t)d * 4);
  const int t = blockIdx.x, nvec = d >> 3;
  const Half* hin = hidden_in + (size_t)t * d;
  Half* hr = hidden + (size_t)t * d;
  Half* orow = out + (size_t)t * d;
  // (round 6) ONE memory round trip per thread instead of three dependent ones: the residual row and the norm weight leave
  // with the partials; wave 0's canonical sum of squares reads both rows from LDS (same values, same order: same bits) and the
  // scale pass works from registers.  Rows of more than 512 vectors (d > 4096) keep the reloading loop for their tail.
  auto slice_sum = [&](int i) {
    const float* p = part + (size_t)t * d + i * 8;
    f32x4 pa[8], pb[8];   // ksplit <= 8: every slice's loads leave before the first add (slice order is kept)
#pragma unroll
    for (int z = 0; z < 8; ++z)
      if (z < ksplit) {
        const float* q = p + (size_t)z * T * d;
        pa[z] = *reinterpret_cast<const f32x4*>(q);
        pb[z] = *reinterpret_cast<const f32x4*>(q + 4);
      }
    f32x4 a = pa[0], b = pb[0];
#pragma unroll
    for (int z = 1; z < 8; ++z)
      if (z < ksplit) { a += pa[z]; b += pb[z]; }
    u32x4 r;
    r.x = pack_bf2(a[0], a[1]); r.y = pack_bf2(a[2], a[3]); r.z = pack_bf2(b[0], b[1]); r.w = pack_bf2(b[2], b[3]);
    return r;
  };
  // the thread's first vector stays in registers; further ones (d > 4096) go through LDS / a reload
  const int i0 = threadIdx.x;
  const bool have0 = i0 < nvec;
  u32x4 h0 = {0u, 0u, 0u, 0u}, w0 = {0u, 0u, 0u, 0u}, r0 = {0u, 0u, 0u, 0u};
  if (have0) {
    h0 = reinterpret_cast<const u32x4*>(hin)[i0];
    w0 = reinterpret_cast<const u32x4*>(w)[i0];
    r0 = slice_sum(i0);
    rrow[i0] = r0;
    hrow[i0] = h0;
  }
  for (int i = i0 + 512; i < nvec; i += 512) {
    hrow[i] = reinterpret_cast<const u32x4*>(hin)[i];
    rrow[i] = slice_sum(i);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float inv = wave_row_inv_rms(reinterpret_cast<const Half*>(hrow), reinterpret_cast<const Half*>(rrow), d, eps, round_sum);
    if (threadIdx.x == 0) *sm_inv = inv;
  }
  __syncthreads();
  const float inv = *sm_inv;
  if (have0) {
    u32x4 nh;
    const u32x4 o = norm_scale8(h0, &r0, w0, inv, 0.f, &nh, round_sum);
    reinterpret_cast<u32x4*>(hr)[i0] = nh;
    reinterpret_cast<u32x4*>(orow)[i0] = o;
  }
  for (int i = i0 + 512; i < nvec; i += 512) {
    const u32x4 r = rrow[i];
    u32x4 nh;
    const u32x4 o = norm_scale8(hrow[i], &r, reinterpret_cast<const u32x4*>(w)[i], inv, 0.f, &nh, round_sum);
    reinterpret_cast<u32x4*>(hr)[i] = nh;
    reinterpret_cast<u32x4*>(orow)[i] = o;
  }
}

// slice sum + residual add: out[t][m] = bf16(a[t][m] + bf16(sum_z part[z][t][m])) = splitk_reduce_kernel + add_cuda
__global__ __launch_bounds__(256) void splitk_reduce_add_kernel(const float* __restrict__ part, const Half* __restrict__ a,
                                                                Half* __restrict__ out, int M, int T, int ksplit) {
  const int m4 = M >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)T * m4) return;
  const int t = (int)(idx / m4), m = (int)(idx - (long)t * m4) * 4;
  f32x4 pv[8];
#pragma unroll
  for (int z = 0; z < 8; ++z)
    if (z < ksplit) pv[z] = *reinterpret_cast<const f32x4*>(part + ((size_t)z * T + t) * M + m);
  f32x4 v = pv[0];
#pragma unroll
  for (int z = 1; z < 8; ++z)
    if (z < ksplit) v += pv[z];
  const u32x2 av = *reinterpret_cast<const u32x2*>(a + (size_t)t * M + m);
  u32x2 o;
  o.x = pack_bf2(bf_lo(av.x) + bf16_round_f(v[0]), bf_hi(av.x) + bf16_round_f(v[1]));
  o.y = pack_bf2(bf_lo(av.y) + bf16_round_f(v[2]), bf_hi(av.y) + bf16_round_f(v[3]));
  *reinterpret_cast<u32x2*>(out + (size_t)t * M + m) = o;
}

// Split-K workspace (64 MB): created by cublas_init() (the reference's handles own a 32 MB cuBLAS workspace the same way,
// csrc/linear.cu:14-42), per thread = per GPU rank.  Without it the split-K route is simply not taken.
constexpr size_t kSplitKWorkspaceBytes = 64u << 20;
constexpr size_t kStreamKFlagBytes = 64u << 10;   // behind the workspace: one 32-bit flag per stream-K workgroup, zeroed once
int g_skinny_flush_override = -1;
int g_streamk_override = -1;
static thread_local float* g_splitk_ws = nullptr;
static thread_local uint32_t* g_streamk_flags = nullptr;

static bool glds_gemm_ok(const Half* W, const Half* X, const Half* Y, int M, int K) {
  return (K & 7) == 0 && host_aligned16(W) && host_aligned16(X) && (K % BK) == 0 && (M & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(Y) & 7u) == 0;
}
template <int TT, int ST>
static void glds_gemm_launch_t(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, hipStream_t s) {
  constexpr int kLds = ST * (BM + TT) * 8 * 16;
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm_glds_kernel<TT, ST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    return true;
  }();
  (void)once;
  const int m_tiles = so.silu_I > 0 ? ceil_div(silu_cols_end(so) - so.silu_c0, 64) : ceil_div(M, BM), t_tiles = ceil_div(T, TT);
  so.w_nt = t_tiles == 1 && weights_nt_on();
  mfma_gemm_glds_kernel<TT, ST><<<m_tiles * t_tiles, 256, kLds, s>>>(W, X, Y, M, T, K, m_tiles, t_tiles, so);
}
// Split-K plans (nk_slice == 0: not applicable), both chosen by shape only:
//  * decode batches of 17..64 columns on a matrix with too few 128-row tiles to fill the chip: 2..8 slices from
//    (M, K), so a column's bits do not depend on the batch size;
//  * prefill (T > 64) when the tiling (128x64 tiles up to 256 tokens, 128x128 above) has at most 170 tiles for the 256
//    CUs: floor(512 / tiles) slices (at most 8, at least 6 K tiles each), i.e. up to 512 workgroups = two per CU, all
//    resident.  o_proj / down_proj at 1024 tokens (160 tiles): 3 slices, 45 -> 40 and 100 -> 78 us (2 slices leave a
//    quarter of the CUs with two workgroups: the makespan does not move); at 128 tokens qkv / o / down have 48 / 20 / 20
//    tiles of 128x128 and ran at 95-150 TFLOP/s un-split.
static bool stream_gemm_on() {
  static const bool v = [] { const char* e = getenv("PEGAINFER_STREAM_GEMM"); return !(e && e[0] == '0'); }();
  return v;
}
static bool stream_splitk_on() {   // PEGAINFER_STREAM_SPLITK=0: the 128-row split-K kernel at 17..64 columns (A/B)
  static const bool v = [] { const char* e = getenv("PEGAINFER_STREAM_SPLITK"); return !(e && e[0] == '0'); }();
  return v;
}
static int stream_min_rt() {
  static const int v = [] { const char* e = getenv("PEGAINFER_STREAM_MIN_RT"); return e && *e ? atoi(e) : 2; }();
  return v;
}
static bool gemm128x256_on() {
  static const bool v = [] { const char* e = getenv("PEGAINFER_GEMM128X256"); return !(e && e[0] == '0'); }();
  return v;
}
// (round 6) Plain prefill GEMMs of >= 512 tokens WITHOUT a K-split plan whose 128 x 256 tiling is ONE partly filled round: 1283 = the
// same round on 96 x 256 tiles (feeder kernel, gemm256.h) when those fill more of the chip, 0 = no.  The stacked qkv at 768 / 1024
// tokens is 144 / 192 tiles of 128 rows, 192 / 256 of 96: 35.9 -> 31.5 / 37.8 -> 33.9 us, same bits; TTFT(768 / 1024) 7.90 -> 7.77 /
// 9.85 -> 9.75 ms same-box (profiles/r6_gemm96_ab.txt, r6_small_tiles_ttft_ab2.txt).  PEGAINFER_GEMM_SMALL_TILES=0 switches it off.
// Two wider forms of the rule were measured and NOT kept - both look 10-30 % faster per GEMM with L2-hot weights
// (tools/bench_prefill_gemm.py at 1 weight copy, profiles/r6_gemm_force_sweep.txt) and are slower in the model, where weights come
// from HBM: (a) over the K-split plans of o_proj / down_proj at 1536 .. 3072 tokens (whose slice sum also carries the residual add +
// RMSNorm): TTFT(1536 / 2048) +1 / +3 % (r6_small_tiles_ttft_ab.txt); (b) several rounds of small tiles instead of 1.1-1.5 rounds
// of 256 x 256 (qkv at 3072 / 4096 tokens): TTFT +2.7 / +1.7 % (r6_small_tiles_ttft_ab2.txt).
static int small_tile_unsplit(int M, int T, int K) {
  static const bool on = [] { const char* e = getenv("PEGAINFER_GEMM_SMALL_TILES"); return !(e && e[0] == '0'); }();
  if (!on || !gemm128x256_on() || !gemm128x256_feed_on() || !gemm128x256_ok(M, T, K) || T < 512) return 0;
  const int cus = device_cus(), tt = ceil_div(T, G256_BT);
  const long t96 = (long)ceil_div(M, 96) * tt, t128 = (long)ceil_div(M, G128_BM) * tt;
  return t128 > 128 && t128 <= 256 && t96 <= cus && t96 > t128 ? 1283 : 0;
}
constexpr int kSplitKMaxRows = 16384;   // below this (< 128 row tiles) the 17..64-column GEMM splits K
struct SplitKPlan { int ksplit, nk_slice, tt; };
static SplitKPlan splitk_plan(int M, int T, int K, bool assume_ws = false) {
  static const bool enabled = [] { const char* e = getenv("PEGAINFER_SPLITK"); return !(e && e[0] == '0'); }();
  if (!enabled || (!g_splitk_ws && !assume_ws) || T <= 16) return {0, 0, 0};
  const int m_tiles = ceil_div(M, BM), nk_all = K / BK;
  if (T > 64) {
    // long-K matrices at ~1.3-3 k tokens (down_proj: 50-128 tiles of 256 x 256): the 8-phase kernel over 2-5 K
    // slices, one workgroup per CU (tt == 256 marks this plan; PEGAINFER_SPLITK256=0 switches it off).  Same-box A/B of
    // TTFT: 2048 tokens 23.05 -> 21.4 ms; 1024 tokens (40 tiles, 6 slices: 63 MB of partials each way) 12.60 vs 12.56,
    // 512 tokens (8 slices) 9.17 vs 9.33 - so only above 40 tiles.
    static const bool s256 = [] { const char* e = getenv("PEGAINFER_SPLITK256"); return !(e && e[0] == '0'); }();
    static const int s256_mink = [] { const char* e = getenv("PEGAINFER_SPLITK256_MINK"); return e && *e ? atoi(e) : 4096; }();   // o_proj (K 4096) included: TTFT(2048) 21.72 -> 21.29 ms
    if (s256 && K >= s256_mink && gemm256_ok(M, T, K)) {
      const long tiles256 = (long)ceil_div(M, G256_BM) * ceil_div(T, G256_BT);
      if (tiles256 > 40 && tiles256 <= 128) {
        int want = (int)(256 / tiles256);
        want = want > 8 ? 8 : want;
        int nk_slice = (ceil_div(nk_all, want) + 1) & ~1;   // K tiles are walked in pairs
        nk_slice = nk_slice < 16 ? 16 : nk_slice;
        const int ksplit = ceil_div(nk_all, nk_slice);
        if (ksplit >= 2 && (size_t)ksplit * T * M * 4 <= kSplitKWorkspaceBytes) return {ksplit, nk_slice, 256};
      }
    }
    // 128 x 256 tiles on the 2-phase schedule (gemm256.h), one workgroup per CU: up to 128 tiles, floor(256 / tiles)
    // K slices (tt == 129 marks this plan; PEGAINFER_GEMM128X256=0 switches it off)
    if (gemm128x256_on() && gemm128x256_ok(M, T, K) && T >= 512) {
      const long tiles = (long)ceil_div(M, G128_BM) * ceil_div(T, G256_BT);
      if (tiles <= 128) {
        int want = (int)(256 / tiles);
        want = want > 8 ? 8 : want;
        int nk_slice = ceil_div(nk_all, want);
        nk_slice = nk_slice < 8 ? 8 : nk_slice;
        const int ksplit = ceil_div(nk_all, nk_slice);
        if (ksplit >= 2 && (size_t)ksplit * T * M * 4 <= kSplitKWorkspaceBytes) return {ksplit, nk_slice, 129};
      }
    }
    // 65..128 tokens on a matrix the stream kernel takes un-split on 32-row tiles (the stacked qkv: 192 workgroups, one
    // 128-token tile, W read once): no K-split pair (24 us for 5 slices on 128-row tiles + the slice sum)
    if (T <= 128 && stream_gemm_on() && stream_min_rt() <= 2 && stream_plan(M, 0).rt == 2) return {0, 0, 0};
    const int tt = T <= 256 ? 64 : 128;
    const long tiles = (long)m_tiles * ceil_div(T, tt);
    if (tiles > 170) return {0, 0, 0};
    int want = (int)(512 / tiles);
    want = want > 8 ? 8 : want;
    int nk_slice = ceil_div(nk_all, want);
    nk_slice = nk_slice < 6 ? 6 : nk_slice;
    const int ksplit = ceil_div(nk_all, nk_slice);
    if (ksplit < 2 || (size_t)ksplit * T * M * 4 > kSplitKWorkspaceBytes) return {0, 0, 0};
    return {ksplit, nk_slice, tt};
  }
  if (M >= kSplitKMaxRows) return {0, 0, 0};
  int want = ceil_div(192, m_tiles);
  want = want > 8 ? 8 : want;
  if (want < 2 || nk_all < 8) return {0, 0, 0};
  int nk_slice = ceil_div(nk_all, want);
  nk_slice = nk_slice < 4 ? 4 : nk_slice;
  const int ksplit = ceil_div(nk_all, nk_slice);   // no empty slice
  if (ksplit < 2 || (size_t)ksplit * T * M * 4 > kSplitKWorkspaceBytes) return {0, 0, 0};
  return {ksplit, nk_slice, 64};
}
// the split GEMM itself: fp32 partials into the workspace (128x64 tiles, ring 3 up to 64 columns; 128x128 tiles,
// ring 2 - two workgroups per CU - above), then optionally the plain slice-sum launch
static void glds_splitk_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, SplitKPlan pl,
                               hipStream_t s, bool reduce = true) {
  const int m_tiles = ceil_div(M, BM);
  if (pl.tt == 256) {
    gemm256_splitk_launch(W, X, M, T, K, g_splitk_ws, pl.ksplit, pl.nk_slice, s);
  } else if (pl.tt == 129) {
    gemm128x256_launch(W, X, Y, M, T, K, so, g_splitk_ws, pl.ksplit, pl.nk_slice, s);
  } else if (pl.tt == 64 && T <= 128 && stream_gemm_on() && stream_splitk_on() &&
             stream_splitk_launch(W, X, M, T, K, g_splitk_ws, pl.ksplit, pl.nk_slice, s)) {
    // same K slices, same partial layout, same slice sum: bit-identical to the 128-row kernel below, with row tiles
    // sized so that tiles x slices fill the chip, and feeder waves (gemm_stream.h)
  } else if (pl.tt == 64) {
    constexpr int kLds = 3 * (BM + 64) * 8 * 16;
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm_glds_kernel<64, 3, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      return true;
    }();
    (void)once;
    const int t_tiles = ceil_div(T, 64);
    so.w_nt = t_tiles == 1 && weights_nt_on();
    mfma_gemm_glds_kernel<64, 3, true><<<dim3(m_tiles * t_tiles, pl.ksplit), 256, kLds, s>>>(
        W, X, Y, M, T, K, m_tiles, t_tiles, so, g_splitk_ws, pl.nk_slice);
  } else {
    constexpr int kLds = 2 * (BM + 128) * 8 * 16;
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm_glds_kernel<128, 2, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      return true;
    }();
    (void)once;
    const int t_tiles = ceil_div(T, 128);
    so.w_nt = t_tiles == 1 && weights_nt_on();
    mfma_gemm_glds_kernel<128, 2, true><<<dim3(m_tiles * t_tiles, pl.ksplit), 256, kLds, s>>>(
        W, X, Y, M, T, K, m_tiles, t_tiles, so, g_splitk_ws, pl.nk_slice);
  }
  if (reduce) splitk_reduce_kernel<<<ceil_div((long)T * (M >> 2), 256), 256, 0, s>>>(g_splitk_ws, Y, M, T, pl.ksplit, so);
}

// Which tiled kernel a prefill-shaped GEMM (T > 64, not K-split) takes - by shape only.  kind: 12 / 13 / 22 / 23 = the
// 128-row LDS-DMA kernel with 128- / 64-token tiles and that ring depth, 256 = the 256 x 256 8-phase kernel, 257 = its
// SwiGLU form with the thin last round handed to the 128-row kernel (m_head = activation-column tiles of the head), 1280 /
// 1281 = the 128 x 256 kernel, plain / SwiGLU.  One function for the launcher and for pegainfer_debug_gemm_route (the
// routing table is pinned by a CPU test: a threshold edit must not silently change what the GPU tests exercise).
struct TiledRoute { int kind, m_head; };
static TiledRoute tiled_route(int M, int T, int K, const SplitOut& so, bool assume_ws = false) {
  // PEGAINFER_GEMM_FORCE=256 | 1280 | 1283 (A/B probe only): that un-split kernel for every plain GEMM it accepts
  static const int force = [] { const char* e = getenv("PEGAINFER_GEMM_FORCE"); return e && *e ? atoi(e) : 0; }();
  if (force && so.silu_I == 0 && T > 64) {
    if (force == 256 && gemm256_ok(M, T, K)) return {256, 0};
    if ((force == 1280 || force == 1283) && gemm128x256_ok(M, T, K)) return {force, 0};
  }
  if (so.silu_I == 0 && !force)
    if (const int kind = small_tile_unsplit(M, T, K)) return {kind, 0};
  // long prompts: 256 x 256 tiles on the 8-phase schedule once they give every CU work (PEGAINFER_GEMM256 = 0 never,
  // 1 whenever the shape allows, N = from N tiles on; A/B probe knob)
  // from 128 tiles on (half the CUs): one partly filled round of 256 x 256 tiles still beats the smaller tilings -
  // gate_up at 512 tokens (152 tiles) 100 -> 70 us, qkv at 1536 tokens (144 tiles) 80 -> 64 us, same bits
  // (profiles/r3_gemm256_threshold_ab.txt); at 96 tiles (qkv at 1024 tokens) the 128 x 256 kernel wins (43 vs ~71 us)
  static const int g256_min = [] { const char* e = getenv("PEGAINFER_GEMM256"); return e && *e ? atoi(e) : 128; }();
  if (g256_min > 0 && gemm256_ok(M, T, K)) {
    const int mt256 = so.silu_I > 0 ? ceil_div(so.silu_I, 128) : ceil_div(M, G256_BM), tt256 = ceil_div(T, G256_BT);
    const long tiles256 = (long)mt256 * tt256;
    // (round 6) stream-K, kind 258, OFF unless PEGAINFER_STREAMK=1: a PERSISTENT one-workgroup-per-CU launch over (row tile, K-tile
    // pair) units for shapes whose round of 256 x 256 tiles would leave >= 8 % of the CU-rounds idle (gate_up at 512 / 1024 /
    // 2048 tokens: 152 / 304 / 608 tiles on 256 CUs).  Built, correct, deterministic - and measured slower (gemm256.h).
    static const int sk_min = [] { const char* e = getenv("PEGAINFER_STREAMK_MIN_TILES"); return e && *e ? atoi(e) : 128; }();
    if (tiles256 >= sk_min && (g_splitk_ws || assume_ws) && so.silu_c1 == 0 &&
        (size_t)device_cus() * kG256SlotFloats * 4 <= kSplitKWorkspaceBytes && (size_t)device_cus() * 4 <= kStreamKFlagBytes &&
        gemm256_streamk_plan(tiles256, K, device_cus(), tt256))
      return {258, 0};
    if (tiles256 >= g256_min) {
      // A thin last round: 304 tiles (gate_up at 1024 tokens) are one full round of the 256 CUs plus 48 tiles that
      // cost a second full tile time (127 us for 102 GFLOP).  The SwiGLU form can hand the activation columns of
      // that remainder to the 128-row kernel (192 quarter-size tiles: one short round) - both kernels keep the same
      // per-element K order, so the result does not depend on the cut.  PEGAINFER_GEMM256_TAIL=0 switches it off.
      static const bool tail_on = [] { const char* e = getenv("PEGAINFER_GEMM256_TAIL"); return !(e && *e == '0'); }();
      const int kCus = device_cus();
      const int rem = (int)(tiles256 % kCus);
      const int m_head = (int)((tiles256 / kCus) * kCus / tt256);   // whole activation-column tiles in full rounds
      if (tail_on && so.silu_I > 0 && so.silu_c1 == 0 && rem > 0 && rem * 2 <= kCus && m_head > 0 && m_head < mt256)
        return {257, m_head};
      return {256, 0};
    }
  }
  // 65..128 tokens on a gate_up-sized matrix (round 4): the weight-streaming kernel with 128-token tiles - row tiles sized
  // to the CU count (244 workgroups instead of 152), same per-element K order as every un-split tiled kernel.  kind 3000 + RT.
  if (T <= 128 && so.silu_c1 == 0 && stream_gemm_on()) {
    const StreamPlan sp = stream_plan(M, so.silu_I > 0 ? so.silu_I : 0);
    if (sp.rt >= (stream_min_rt() > 2 ? stream_min_rt() : 2)) return {3000 + sp.rt, 0};   // 128-token tiles exist for 2..6 row blocks
  }
  // SwiGLU GEMM on short prompts (65..256 tokens: 152 tiles of (64 + 64) x 256 for Qwen3-4B, one round) - the 128 x 128
  // kernel runs these at one 4-wave workgroup per CU.  Same per-element K order: bit-identical to gemm + silu_mul
  // (tested).  TTFT(128) 4.85 -> 4.63 ms, TTFT(256) 5.79 -> 5.27 ms same-box (PEGAINFER_GEMM128X256_SILU=0 for the A/B)
  static const bool silu128_on = [] { const char* e = getenv("PEGAINFER_GEMM128X256_SILU"); return !(e && e[0] == '0'); }();
  if (so.silu_I > 0 && so.silu_c1 == 0 && silu128_on && gemm128x256_on() && gemm128x256_ok(M, T, K) && T <= 256) {
    const long tiles = (long)ceil_div(so.silu_I, 64) * ceil_div(T, G256_BT);
    if (tiles > 128 && tiles <= 256) return {1281, 0};
  }
  // 129..256 tiles of 128 x 256: one un-split round of the 2-phase kernel (qkv at 1024 tokens: 192 tiles)
  if (so.silu_I == 0 && gemm128x256_on() && gemm128x256_ok(M, T, K) && T >= 512) {
    const long tiles = (long)ceil_div(M, G128_BM) * ceil_div(T, G256_BT);
    if (tiles > 128 && tiles <= 256) return {1280, 0};
  }
  // Measured on MI355X (tools/bench_prefill_gemm.py, T = 1024): co-resident workgroups hide DMA latency better
  // than a deeper ring, so the ring only goes to 3 when every 64-token tile is resident at once (<= 2 per CU);
  // 128-token tiles (half the LDS reads per MFMA) once they give each CU >= 1.5 workgroups.
  const long m_tiles = ceil_div(M, BM), wide = m_tiles * ceil_div(T, 128), narrow = m_tiles * ceil_div(T, 64);
  // With cold (HBM) weights, long-K small matrices (down_proj: 160 wide tiles, K = 9728) do best as one 128-token
  // tile per CU with the 3-deep ring (102 vs 115 us); measured with tools/bench_prefill_gemm.py 1024 12.
  return {narrow <= 512 ? (K >= 8192 && wide <= 256 ? 13 : 23) : wide >= 384 ? 12 : 22, 0};
}

// variant: 0 = by shape; else 10*tile + stages with tile 1 = 128-token, 2 = 64-token (A/B probe only)
static void glds_gemm_launch(const Half* W, const Half* X, Half* Y, int M, int T, int K, SplitOut so, int variant,
                             hipStream_t s) {
  if (variant == 0) {
    const TiledRoute r = tiled_route(M, T, K, so);
    if (r.kind == 257) {
      SplitOut head = so, tail = so;
      head.silu_c0 = 0; head.silu_c1 = r.m_head * 128;
      tail.silu_c0 = r.m_head * 128; tail.silu_c1 = so.silu_I;
      gemm256_launch(W, X, Y, M, T, K, head, s);
      // the thin tail: 128-token tiles with a 2-deep ring (192 workgroups at 1024 tokens), or - PEGAINFER_GEMM256_TAIL_TT=64, A/B
      // knob - 64-token tiles with a 3-deep ring (384 workgroups, two per CU, all resident)
      // (round 6, same-box A/B profiles/r6_reduce_and_tail_ttft.txt: TTFT(1024) 10.53 -> 10.33 ms, TTFT(2048) 18.25 -> 18.40):
      // 64-token tiles exactly when the 128-token tiling would give the tail fewer tiles than there are CUs
      static const int tail_env = [] { const char* e = getenv("PEGAINFER_GEMM256_TAIL_TT"); return e && *e ? atoi(e) : 0; }();
      const long tail_tiles128 = (long)ceil_div(so.silu_I - r.m_head * 128, 64) * ceil_div(T, 128);
      // (round 6, profiles/r6_tail_feeder_ab.txt) the feeder form of the 128 x 256 kernel takes the tail: (64 + 64)-row SwiGLU
      // tiles (tail_tt 4), or (32 + 32)-row ones (tail_tt 2) while those still fit one round - TTFT(1024) 9.99 -> 9.76 ms (192
      // tiles of 64 rows), TTFT(2048) 17.82 -> 17.24 (192 tiles of 128 rows); 64 / 128 = the round-5 kernels, kept for the A/B
      const long tail_tiles4 = (long)ceil_div(so.silu_I - r.m_head * 128, 64) * ceil_div(T, G256_BT);
      const int tail_feeder = !gemm128x256_on() || !gemm128x256_feed_on() ? 0 : tail_tiles4 * 2 <= device_cus() ? 2 : 4;
      const int tail_tt = tail_env ? tail_env : tail_feeder ? tail_feeder : (tail_tiles128 < device_cus() ? 64 : 128);
      if (tail_tt == 2) gemm64x256_silu_launch(W, X, Y, M, T, K, tail, s);
      else if (tail_tt == 4) gemm128x256_launch(W, X, Y, M, T, K, tail, nullptr, 1, 0, s);
      else if (tail_tt == 64) glds_gemm_launch_t<64, 3>(W, X, Y, M, T, K, tail, s);
      else glds_gemm_launch_t<128, 2>(W, X, Y, M, T, K, tail, s);
      return;
    }
    if (r.kind == 258) { gemm256_streamk_launch(W, X, Y, M, T, K, so, g_splitk_ws, g_streamk_flags, device_cus(), s); return; }
    if (r.kind == 256) { gemm256_launch(W, X, Y, M, T, K, so, s); return; }
    if (r.kind == 1283) { gemm96x256_launch(W, X, Y, M, T, K, so, s); return; }
    if (r.kind == 1280 || r.kind == 1281) { gemm128x256_launch(W, X, Y, M, T, K, so, nullptr, 1, 0, s); return; }
    // the plan said a 128-token stream tile exists; should a future plan / route change make the launcher refuse, fall
    // through to the 128-row kernel (same per-element K order) instead of leaving Y unwritten
    if (r.kind >= 3000 && stream_gemm_launch<128>(W, X, Y, M, T, K, so, s)) return;
    variant = r.kind >= 3000 ? 23 : r.kind;
  }
  switch (variant) {
    case 12: glds_gemm_launch_t<128, 2>(W, X, Y, M, T, K, so, s); break;
    case 13: glds_gemm_launch_t<128, 3>(W, X, Y, M, T, K, so, s); break;
    case 22: glds_gemm_launch_t<64, 2>(W, X, Y, M, T, K, so, s); break;
    case 24: glds_gemm_launch_t<64, 4>(W, X, Y, M, T, K, so, s); break;
    default: glds_gemm_launch_t<64, 3>(W, X, Y, M, T, K, so, s); break;
  }
}

// Decode-shaped GEMM routing (T <= 64 columns, weights streamed once), by shape only (measured on MI355X):
//   T <= 2            -> dot2 GEMV (gemv_core.h)            [PEGAINFER_GEMV_T1=mfma routes everything to MFMA]
//   3 <= T <= 16      -> skinny MFMA GEMM (gemm_skinny.h)   when K % 64 == 0 (layer GEMMs at T = 3 / 4: 52 vs 61 us
//                        dot2); the Qwen3.5 flag forms (T <= 4 only) stay on the dot2 kernel
//   17 <= T <= 64     -> tiled LDS-DMA GEMM, K split over workgroups below 16384 rows (skinny if K is too short)
// Within each family a column's result does not depend on the batch size (bitwise batch invariance).
static bool t1_uses_mfma() {
  static const bool v = [] { const char* e = getenv("PEGAINFER_GEMV_T1"); return e && e[0] == 'm'; }();
  return v;
}
// 17..64 columns with a large weight matrix: the 128-row LDS-DMA tiles beat the 16-row skinny blocks by 2-4x
// (gate_up at T = 64: 21 vs 64 us; x is staged once per 128 rows and the DMA ring keeps the stream going), small
// matrices (o_proj / down_proj: 20 tiles) stay on the skinny kernel whose 8 waves split K.
constexpr int kMidBatchMinRows = 5120;
// 17..64 columns, plain (optionally row-segmented) output: 1 = tiled LDS-DMA GEMM, 2 = its split-K form, 0 = not
// taken (skinny kernel / separate calls).  One decision for gemm_cuda, pegainfer_gemm_split and pegainfer_gemm_silu,
// so that a stacked launch and the plain launch over the same matrix always run the same kernel.
// 3 = the weight-streaming kernel of gemm_stream.h (round 4): row tiles sized to the CU count, full K per workgroup, feeder
// waves.  Taken for tiles of >= 2 row blocks: gate_up 28.9 -> 21.8 us at 32 columns, 28.9 -> 24.1 at 64 (same bits as the
// 128-row kernel); the stacked qkv matrix (32-row tiles, 192 workgroups) 13.8 -> 12.6 / 16.8 -> 13.7 us against its split-K
// pair.  o_proj / down_proj (16-row tiles) keep the split-K route: with full K per workgroup they pull 0.5-1.2 MB of x per CU
// for 130-310 KB of weights (15.3 / 30.5 us against 15.0 / 21.7), profiles/r4_stream_spec_ab.txt.
// PEGAINFER_STREAM_GEMM=0 switches it off, PEGAINFER_STREAM_MIN_RT moves the threshold.
static int mid_batch_route(const Half* W, const Half* X, const Half* Y, int M, int T, int K, int silu_I = 0,
                           bool assume_ws = false) {
  static const int min_rows = [] { const char* e = getenv("PEGAINFER_MID_MIN_ROWS"); return e && *e ? atoi(e) : kMidBatchMinRows; }();
  if (T <= 16 || T > 64 || !glds_gemm_ok(W, X, Y, M, K)) return 0;
  if (stream_gemm_on() && stream_plan(M, silu_I).rt >= stream_min_rt()) return 3;
  if (splitk_plan(M, T, K, assume_ws).nk_slice > 0) return 2;
  return M >= min_rows ? 1 : 0;
}
static bool mid_batch_gemm(const Half* W, const Half* X, Half* Y, int M, int T, int K, const SplitOut& so, hipStream_t s) {
  const int route = mid_batch_route(W, X, Y, M, T, K, so.silu_I);
  if (route == 3) return stream_gemm_launch<64>(W, X, Y, M, T, K, so, s);
  if (route == 2) glds_splitk_launch(W, X, Y, M, T, K, so, splitk_plan(M, T, K), s);
  else if (route == 1) glds_gemm_launch(W, X, Y, M, T, K, so, 23, s);
  return route != 0;
}
template <int EPI>
static bool decode_gemm_dispatch(const GemvFusedArgs& a, hipStream_t s) {
  if (a.T > 16 && a.T <= 64) {
    // the fused prologue / epilogue forms exist for decode batches <= 16 only: above that the host runs the
    // unfused sequence, whose large GEMMs take the tiled kernel (so fused == unfused stays true by construction)
    if (a.norm_w || EPI != kEpiStore || a.flags) return false;
    if (mid_batch_gemm(a.W, a.X, a.Y, a.M, a.T, a.K, SplitOut{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}, s)) return true;
  }
  static const int skinny_min_t = [] { const char* e = getenv("PEGAINFER_SKINNY_MIN_T"); return e && *e ? atoi(e) : 3; }();
  // the skinny kernels take one flag, kGemvRoundSum (the prefill residual chain at 5..16 token columns); the other
  // Qwen3.5 forms exist on the dot2 kernel only (T <= 4, enforced by gemv_fused_impl)
  if ((a.T >= 5 || (a.T >= skinny_min_t && a.flags == 0) || t1_uses_mfma()) && a.T <= 64 && (a.K & 63) == 0 &&
      (a.flags & ~kGemvRoundSum) == 0) {
    if (a.T == 1) { skinny_launch<1, EPI>(a, s); return true; }
    return skinny_dispatch<EPI>(a, s);
  }
  return gemv_dispatch<EPI>(a, s);
}

static void gemm_dispatch(const Half* W, const Half* X, Half* Y, int M, int T, int K, hipStream_t s) {
  if (M <= 0 || T <= 0 || K <= 0) return;
  const bool fast = (K & 7) == 0 && host_aligned16(W) && host_aligned16(X);
  if (fast && T <= 64) {
    GemvFusedArgs a{W, X, Y, M, T, K, nullptr, nullptr, nullptr, 0.f, 0, 0, 0};
    if (decode_gemm_dispatch<kEpiStore>(a, s)) return;
  }
  if (glds_gemm_ok(W, X, Y, M, K)) {
    if (T > 64) {
      const SplitKPlan pl = splitk_plan(M, T, K);
      if (pl.nk_slice > 0) {
        glds_splitk_launch(W, X, Y, M, T, K, SplitOut{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}, pl, s);
        return;
      }
    }
    // PEGAINFER_GEMM=reg keeps the register-staged kernel; w2 w3 n2 n3 n4 force a 128 / 64-token LDS-DMA tile
    // with that ring depth (A/B probes, tools/bench_prefill_gemm.py); default picks by shape.
    static const int mode = [] {
      const char* e = getenv("PEGAINFER_GEMM");
      if (!e) return 0;
      if (e[0] == 'r') return -1;
      if ((e[0] == 'w' || e[0] == 'n') && e[1] >= '2' && e[1] <= '4') return (e[0] == 'w' ? 10 : 20) + (e[1] - '0');
      return 0;
    }();
    if (mode < 0) {
      const int m_tiles = ceil_div(M, BM), t_tiles = ceil_div(T, BT);
      mfma_gemm_kernel<<<m_tiles * t_tiles, 256, 0, s>>>(W, X, Y, M, T, K, m_tiles, t_tiles);
    } else {
      glds_gemm_launch(W, X, Y, M, T, K, SplitOut{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}, mode, s);
    }
    return;
  }
  naive_gemm_kernel<<<ceil_div((long)M * T, 4), 256, 0, s>>>(W, X, Y, M, T, K);
}

}  // namespace pk

extern "C" {

// Debug: device buffer of (workgroups of the launch) * 8 uint64 stamped by every later dot2-GEMV launch with the
// 100 MHz wall clock: [0] entry, [1] x staged, [2] first weight block consumed, [3] last row group's K loop done,
// [4] exit, [5] XCC id; nullptr switches it off (tools/gemv_probe.py).  Not part of the reference ABI.
void pegainfer_debug_gemv_trace(uint64_t* buf) { pk::g_gemv_trace = reinterpret_cast<unsigned long long*>(buf); }
// Debug / test hook: force how skinny_resident_kernel's 8 waves meet per row block (0 two barriers, 1 one barrier, 4 tickets, 5 lazy tickets;
// -1 = back to the launcher's choice).  The forms are bit-identical; tests/test_gpu_ops.py compares them in one process.
void pegainfer_debug_skinny_flush(int32_t mode) { pk::g_skinny_flush_override = mode < 0 ? -1 : (mode & 7); }
// Debug / test hook: 1 routes the shapes of gemm256_streamk_plan to the stream-K launch (kind 258), 0 never, -1 = the
// environment (PEGAINFER_STREAMK, default off: measured slower - gemm256.h).
void pegainfer_debug_streamk(int32_t on) { pk::g_streamk_override = on < 0 ? -1 : (on != 0); }


int32_t cuda_set_device(int32_t device_ordinal) { return static_cast<int32_t>(hipSetDevice(device_ordinal)); }

// The reference creates two thread-local cuBLAS handles + a 32 MB prefill workspace here
// (csrc/linear.cu:14-42).  The HIP GEMMs need no handles; the one thing kept per thread is a 64 MB fp32
// workspace for the split-K route of mid-size decode batches.  Same call contract: per-thread, safe to call
// repeatedly (counted: the workspace goes away with the last destroy).
static thread_local int g_blas_inits = 0;
void cublas_init(void) {
  if (g_blas_inits++ == 0 && !pk::g_splitk_ws) {
    void* p = nullptr;
    if (hipMalloc(&p, pk::kSplitKWorkspaceBytes + pk::kStreamKFlagBytes) == hipSuccess) {
      pk::g_splitk_ws = static_cast<float*>(p);
      pk::g_streamk_flags = reinterpret_cast<uint32_t*>(static_cast<char*>(p) + pk::kSplitKWorkspaceBytes);
      (void)hipMemset(pk::g_streamk_flags, 0, pk::kStreamKFlagBytes);   // owners clear what publishers set: zero between launches
    } else {
      (void)hipGetLastError();   // no device / no memory: the split-K and stream-K routes are skipped
    }
  }
}
void cublas_destroy(void) {
  if (g_blas_inits > 0 && --g_blas_inits == 0 && pk::g_splitk_ws) {
    (void)hipFree(pk::g_splitk_ws);
    pk::g_splitk_ws = nullptr;
    pk::g_streamk_flags = nullptr;
  }
}

void gemm_cuda(const Half* W, const Half* X, Half* Y, int32_t M, int32_t N, int32_t K, pegainfer_stream_t stream) {
  pk::gemm_dispatch(W, X, Y, M, N, K, pk::as_stream(stream));
}

// Same arithmetic; no workspace, no allocation, no sync -> safe under hipGraph capture.
void gemm_graphsafe_cuda(const Half* W, const Half* X, Half* Y, int32_t M, int32_t N, int32_t K,
                         pegainfer_stream_t stream) {
  pk::gemm_dispatch(W, X, Y, M, N, K, pk::as_stream(stream));
}

// Fused decode GEMV (extension, include/pegainfer_kernels_ext.h): optional RMSNorm / add+RMSNorm
// prologue and SwiGLU epilogue around the SAME gemv core that gemm_graphsafe_cuda uses.
static pegainfer_status_t gemv_fused_impl(const Half* W, const Half* X, Half* Y, int32_t M, int32_t T, int32_t K,
                                          const Half* residual, const Half* norm_weight, Half* hidden_out, float eps,
                                          int32_t silu_intermediate, int32_t flags, pegainfer_stream_t stream) {
  using namespace pk;
  if (M <= 0 || T < 1 || T > 64 || K <= 0 || (K & 7) != 0) return (pegainfer_status_t)hipErrorInvalidValue;
  if (!host_aligned16(W) || !host_aligned16(X) || (residual && !host_aligned16(residual)) ||
      (norm_weight && !host_aligned16(norm_weight)) || (hidden_out && !host_aligned16(hidden_out)))
    return (pegainfer_status_t)hipErrorInvalidValue;
  if (residual && (!norm_weight || !hidden_out || hidden_out == X)) return (pegainfer_status_t)hipErrorInvalidValue;
  if (silu_intermediate > 0 && M != 2 * silu_intermediate) return (pegainfer_status_t)hipErrorInvalidValue;
  // (1 + w) norm weights and the rounded silu exist on the dot2 kernel only (T <= 4); "add, then norm" (bit 1) also on
  // the skinny kernels (T <= 16, K % 64 == 0)
  if ((flags & ~kGemvRoundSum) != 0 && (T > 4 || t1_uses_mfma())) return (pegainfer_status_t)hipErrorInvalidValue;
  if (flags == kGemvRoundSum && T > 4 && (T > 16 || (K & 63) != 0)) return (pegainfer_status_t)hipErrorInvalidValue;
  GemvFusedArgs a{W, X, Y, M, T, K, residual, norm_weight, hidden_out, eps, silu_intermediate, 0, flags};
  const bool ok = silu_intermediate > 0 ? decode_gemm_dispatch<kEpiSilu>(a, as_stream(stream))
                                        : decode_gemm_dispatch<kEpiStore>(a, as_stream(stream));
  return ok ? (pegainfer_status_t)hipGetLastError() : (pegainfer_status_t)hipErrorInvalidValue;
}

pegainfer_status_t pegainfer_gemv_fused(const Half* W, const Half* X, Half* Y, int32_t M, int32_t T, int32_t K,
                                        const Half* residual, const Half* norm_weight, Half* hidden_out, float eps,
                                        int32_t silu_intermediate, pegainfer_stream_t stream) {
  return gemv_fused_impl(W, X, Y, M, T, K, residual, norm_weight, hidden_out, eps, silu_intermediate, 0, stream);
}

// Qwen3.5 forms (flags: 1 = (1 + w) norm weight, 2 = residual sum rounded to bf16 before the norm, 4 = silu rounded
// to bf16 before the multiply); T <= 4 only.
pegainfer_status_t pegainfer_gemv_fused_ex(const Half* W, const Half* X, Half* Y, int32_t M, int32_t T, int32_t K,
                                           const Half* residual, const Half* norm_weight, Half* hidden_out, float eps,
                                           int32_t silu_intermediate, int32_t flags, pegainfer_stream_t stream) {
  return gemv_fused_impl(W, X, Y, M, T, K, residual, norm_weight, hidden_out, eps, silu_intermediate, flags, stream);
}

// One GEMM over a row-stacked weight writing 2..4 separate outputs Y_i[T][M_i] (extension): what prefill's
// q_proj / k_proj / v_proj calls (prefill.rs:120-129), Qwen3.5's in_proj_qkv / z / b / a (qwen35 prefill.rs:373-376)
// and gate_proj / up_proj (:187-188) compute, in one launch that fills the chip.  For T > 64 each element is
// bit-identical to the separate gemm_cuda call (same kernel, same K order); for 17..64 columns the separate calls
// on sub-matrices below 5120 rows take the skinny kernel instead, so equality there is within the GEMM tolerance.
pegainfer_status_t pegainfer_gemm_split(const Half* W, const Half* X, int32_t n_out, Half* const* Y, const int32_t* Ms,
                                        int32_t T, int32_t K, pegainfer_stream_t stream) {
  using namespace pk;
  if (n_out < 2 || n_out > 4 || T <= 0 || K <= 0) return (pegainfer_status_t)hipErrorInvalidValue;
  int M = 0, m[4] = {0, 0, 0, 0};
  Half* y[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ok = T > 16;  // 17..64: only worth it (and only taken) when the stacked matrix is mid-batch sized
  for (int i = 0; i < n_out; ++i) {
    if (Ms[i] <= 0 || !Y[i]) return (pegainfer_status_t)hipErrorInvalidValue;
    m[i] = Ms[i]; y[i] = Y[i]; M += Ms[i];
    ok = ok && (Ms[i] & 3) == 0 && (reinterpret_cast<uintptr_t>(Y[i]) & 7u) == 0;
  }
  ok = ok && glds_gemm_ok(W, X, y[0], M, K) && (T > 64 || mid_batch_route(W, X, y[0], M, T, K) != 0);
  if (ok) {
    // the last present segment is "the rest"; with fewer than 4 outputs the unused boundaries collapse onto M
    SplitOut so{y[1], y[2] ? y[2] : y[1], y[3] ? y[3] : (y[2] ? y[2] : y[1]), m[0], m[1], n_out > 2 ? m[2] : 0, 0, 0};
    if (n_out == 2) { so.M1 = m[1]; so.M2 = 0; }
    if (T > 64) {   // same decision as gemm_cuda over the stacked matrix
      const SplitKPlan pl = splitk_plan(M, T, K);
      if (pl.nk_slice > 0) glds_splitk_launch(W, X, y[0], M, T, K, so, pl, as_stream(stream));
      else glds_gemm_launch(W, X, y[0], M, T, K, so, 0, as_stream(stream));
    } else {
      mid_batch_gemm(W, X, y[0], M, T, K, so, as_stream(stream));
    }
  } else {  // shapes the tiled kernel does not take: separate reference-ABI calls
    size_t row = 0;
    for (int i = 0; i < n_out; ++i) {
      gemm_dispatch(W + row * K, X, y[i], m[i], T, K, as_stream(stream));
      row += m[i];
    }
  }
  return (pegainfer_status_t)hipGetLastError();
}

// gate_up GEMM with the SwiGLU activation in its epilogue (extension): Y[T][I] = silu_mul_fused(W[2I][K] . X), what
// prefill's gemm + silu_mul_fused_cuda pair computes (prefill.rs:167-175) without writing / re-reading [T][2I].
// Bit-identical to that pair for T > 16 and 2I >= 5120 (same tiled kernel); falls back to the pair otherwise - the
// caller passes the [T][2I] scratch the pair needs.
static pegainfer_status_t gemm_silu_impl(const Half* W, const Half* X, Half* Y, Half* gate_up_scratch, int32_t I,
                                         int32_t T, int32_t K, int double_round, pegainfer_stream_t stream) {
  using namespace pk;
  if (I <= 0 || T <= 0 || K <= 0) return (pegainfer_status_t)hipErrorInvalidValue;
  const int M = 2 * I;
  // 17..64 columns: fused only where the PLAIN GEMM over the same matrix also runs an un-split tiled kernel (route 1 / 3) -
  // the tile plans differ (rows vs activation columns), and a narrow matrix whose plain form takes the K-split or skinny
  // kernel would otherwise get a different per-element K order fused than unfused
  const int mroute = T <= 64 ? mid_batch_route(W, X, Y, M, T, K, I) : 0;
  const int proute = T <= 64 ? mid_batch_route(W, X, Y, M, T, K, 0) : 0;
  if (T > 16 && (I & 3) == 0 && glds_gemm_ok(W, X, Y, M, K) &&
      (T > 64 ? splitk_plan(M, T, K).nk_slice == 0 : ((mroute == 1 || mroute == 3) && (proute == 1 || proute == 3)))) {
    SplitOut so{nullptr, nullptr, nullptr, 0, 0, 0, I, double_round};
    // 40 + 40 rows per tile: 244 workgroups; a refused launch (cannot happen under today's plans) takes the 128-row kernel
    if (!(mroute == 3 && stream_gemm_launch<64>(W, X, Y, M, T, K, so, as_stream(stream))))
      glds_gemm_launch(W, X, Y, M, T, K, so, T <= 64 ? 23 : 0, as_stream(stream));
    return (pegainfer_status_t)hipGetLastError();
  }
  if (!gate_up_scratch) return (pegainfer_status_t)hipErrorInvalidValue;
  gemm_dispatch(W, X, gate_up_scratch, M, T, K, as_stream(stream));
  if (double_round) {  // the unfused Qwen3.5 pair works on separate gate / up buffers: run it per token row
    for (int t = 0; t < T; ++t) {
      const pegainfer_status_t rc = silu_mul_triton_aot_cuda(gate_up_scratch + (size_t)t * M, gate_up_scratch + (size_t)t * M + I,
                                                             Y + (size_t)t * I, I, stream);
      if (rc) return rc;
    }
  } else {
    silu_mul_fused_cuda(gate_up_scratch, Y, I, T, stream);
  }
  return (pegainfer_status_t)hipGetLastError();
}
pegainfer_status_t pegainfer_gemm_silu(const Half* W, const Half* X, Half* Y, Half* gate_up_scratch, int32_t I, int32_t T,
                                       int32_t K, pegainfer_stream_t stream) {
  return gemm_silu_impl(W, X, Y, gate_up_scratch, I, T, K, 0, stream);
}
// Qwen3.5 form: Y = silu_mul(gate, up) with silu rounded to bf16 first (qwen35 prefill.rs:187-190)
pegainfer_status_t pegainfer_gemm_silu_rounded(const Half* W, const Half* X, Half* Y, Half* gate_up_scratch, int32_t I,
                                               int32_t T, int32_t K, pegainfer_stream_t stream) {
  return gemm_silu_impl(W, X, Y, gate_up_scratch, I, T, K, 1, stream);
}

// Debug / test hook (no device work, callable without a GPU): which kernel a prefill-shaped GEMM of this shape takes.
// silu_I > 0 asks for the SwiGLU form (M is then ignored, the matrix has 2 * silu_I rows).  out[0] = kind, out[1] = K
// slices (1 = un-split), out[2] = K tiles per slice or, for kind 257, the activation-column tiles of the 256 x 256 head.
// kind: 0 = GEMV / skinny family (T <= 16, or a 17..64-column shape the tiled kernels do not take); the TiledRoute kinds (12,
// 13, 22, 23, 256, 257, 258 = stream-K over 256 x 256 tiles, 1280, 1281, 3000 + row blocks of the stream kernel); 1000 + tt for the K-split plans (tt = 64 / 128:
// 128-row kernel, 129: 128 x 256 kernel, 256: 256 x 256 kernel), 2000 + row blocks where the GEMM half of a tt = 64 plan
// runs on the stream kernel (<= 128 tokens).  Assumes the split-K workspace of cublas_init() exists.
pegainfer_status_t pegainfer_debug_gemm_route(int32_t M, int32_t T, int32_t K, int32_t silu_I, int32_t* out) {
  using namespace pk;
  if (!out || T <= 0 || K <= 0 || (silu_I <= 0 && M <= 0)) return (pegainfer_status_t)hipErrorInvalidValue;
  out[0] = 0; out[1] = 1; out[2] = 0;
  if (T <= 16) {
    // 3..16 columns on the resident-x skinny kernel: out[1] = how its 8 waves meet per row block (0 two barriers, 1 one
    // barrier, 4 tickets, 5 lazy tickets), out[2] = partial buffers * 100 + rows per row block; out[1] = -1: another kernel
    // of the family (dot2 GEMV at 1-2 columns, the tiled skinny kernel when x does not fit in LDS)
    out[1] = -1;
    const int nw = silu_I > 0 ? 2 : 1;
    if (T >= 3 && (K & 63) == 0 && skinny_is_resident(nw, T, K)) {
      const SkinnyFlushPlan fp = skinny_flush_plan(nw, 1, T, K, silu_I > 0 ? silu_I : M);
      out[1] = fp.variant & 7;
      out[2] = fp.bufs * 100 + fp.rpb;
    }
    return 0;
  }
  // the K-split pairs whose GEMM half runs on the stream kernel (glds_splitk_launch): same slices, row tiles of rt blocks
  auto split_kind = [&](const SplitKPlan& pl) {
    const StreamPlan sp = stream_splitk_plan(M, pl.ksplit);
    const bool stream = pl.tt == 64 && T <= 128 && stream_gemm_on() && stream_splitk_on() && sp.rt >= (T > 64 ? 3 : 1);
    return stream ? 2000 + sp.rt : 1000 + pl.tt;
  };
  if (T <= 64) {   // the mid-batch family: 3000 + rt stream kernel, 2000 + rt / 1064 K-split pair, 23 the 128-row kernel, 0 skinny
    const int rows = silu_I > 0 ? 2 * silu_I : M;
    const int r = mid_batch_route(nullptr, nullptr, nullptr, rows, T, K, silu_I > 0 ? silu_I : 0, true);
    if (r == 3) out[0] = 3000 + stream_plan(rows, silu_I > 0 ? silu_I : 0).rt;
    else if (r == 2) {
      const SplitKPlan pl = splitk_plan(rows, T, K, true);
      out[0] = split_kind(pl); out[1] = pl.ksplit; out[2] = pl.nk_slice;
    } else if (r == 1) out[0] = 23;
    return 0;
  }
  if (silu_I <= 0) {
    const SplitKPlan pl = splitk_plan(M, T, K, true);
    if (pl.nk_slice > 0) { out[0] = split_kind(pl); out[1] = pl.ksplit; out[2] = pl.nk_slice; return 0; }
  }
  SplitOut so{nullptr, nullptr, nullptr, 0, 0, 0, silu_I > 0 ? silu_I : 0, 0, 0, 0};
  const TiledRoute r = tiled_route(silu_I > 0 ? 2 * silu_I : M, T, K, so, true);
  out[0] = r.kind; out[2] = r.m_head;
  return 0;
}

pegainfer_status_t pegainfer_gemm_split3(const Half* W, const Half* X, Half* Y0, int32_t M0, Half* Y1, int32_t M1,
                                         Half* Y2, int32_t M2, int32_t T, int32_t K, pegainfer_stream_t stream) {
  Half* ys[3] = {Y0, Y1, Y2};
  const int32_t ms[3] = {M0, M1, M2};
  return pegainfer_gemm_split(W, X, 3, ys, ms, T, K, stream);
}

// o_proj / down_proj + residual add + RMSNorm (extension): exactly gemm_cuda(W, X, y_scratch) followed by
// fused_add_rms_norm_batched_cuda(hidden, y_scratch, norm_weight, normed_out) (batch_decode.rs:262-270, 288-296).
// For 17..64 columns on a split-K shape the slice sum, the add and the norm are ONE launch over the fp32 partials
// (y_scratch is then left untouched); every other shape runs the two calls.  Same bits either way.
pegainfer_status_t pegainfer_gemm_add_rms_norm(const Half* W, const Half* X, Half* y_scratch, Half* hidden,
                                               const Half* norm_weight, Half* normed_out, int32_t M, int32_t T,
                                               int32_t K, float eps, pegainfer_stream_t stream) {
  using namespace pk;
  if (M <= 0 || T <= 0 || K <= 0 || !y_scratch || !hidden || !norm_weight || !normed_out)
    return (pegainfer_status_t)hipErrorInvalidValue;
  hipStream_t s = as_stream(stream);
  const bool vec = (M & 7) == 0 && host_aligned16(hidden) && host_aligned16(norm_weight) && host_aligned16(normed_out) &&
                   (size_t)M * 4 + 16 <= 64 * 1024;
  const bool split = glds_gemm_ok(W, X, y_scratch, M, K) &&
                     (T > 64 ? splitk_plan(M, T, K).nk_slice > 0 : mid_batch_route(W, X, y_scratch, M, T, K) == 2);
  if (vec && split) {
    const SplitKPlan pl = splitk_plan(M, T, K);
    glds_splitk_launch(W, X, y_scratch, M, T, K, SplitOut{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}, pl, s, false);
    splitk_reduce_add_norm_kernel<<<T, 512, (size_t)M * 4 + 16, s>>>(g_splitk_ws, hidden, hidden, norm_weight, normed_out,
                                                                    M, T, pl.ksplit, eps, false);
    return (pegainfer_status_t)hipGetLastError();
  }
  gemm_dispatch(W, X, y_scratch, M, T, K, s);
  fused_add_rms_norm_batched_cuda(hidden, y_scratch, norm_weight, normed_out, M, T, eps, stream);
  return (pegainfer_status_t)hipGetLastError();
}

// down_proj + residual add (extension): exactly gemm_cuda(W, X, y_scratch) followed by add_cuda(a, y_scratch, out)
// (prefill.rs:176-185).  On a split-K shape the slice sum and the add are one launch over the fp32 partials
// (y_scratch is then left untouched).  out may alias a.  Same bits either way.
pegainfer_status_t pegainfer_gemm_add(const Half* W, const Half* X, Half* y_scratch, const Half* a, Half* out,
                                      int32_t M, int32_t T, int32_t K, pegainfer_stream_t stream) {
  using namespace pk;
  if (M <= 0 || T <= 0 || K <= 0 || !y_scratch || !a || !out) return (pegainfer_status_t)hipErrorInvalidValue;
  hipStream_t s = as_stream(stream);
  const bool split = glds_gemm_ok(W, X, y_scratch, M, K) && (reinterpret_cast<uintptr_t>(a) & 7u) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 7u) == 0 &&
                     (T > 64 ? splitk_plan(M, T, K).nk_slice > 0 : mid_batch_route(W, X, y_scratch, M, T, K) == 2);
  if (split) {
    const SplitKPlan pl = splitk_plan(M, T, K);
    glds_splitk_launch(W, X, y_scratch, M, T, K, SplitOut{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}, pl, s, false);
    splitk_reduce_add_kernel<<<ceil_div((long)T * (M >> 2), 256), 256, 0, s>>>(g_splitk_ws, a, out, M, T, pl.ksplit);
    return (pegainfer_status_t)hipGetLastError();
  }
  gemm_dispatch(W, X, y_scratch, M, T, K, s);
  return add_cuda(a, y_scratch, out, M * T, stream);
}

// down_proj + residual add + the NEXT layer's input RMSNorm (extension, prefill): exactly gemm_cuda(W, X, y_scratch),
// add_cuda(a, y_scratch, out) and rms_norm_batched_cuda(out, norm_weight, normed_out) (prefill.rs:176-185 and the next
// layer's prefill.rs:89) - the norm sees the bf16-ROUNDED sum, unlike the fused add + norm above.  On a split-K shape the
// slice sum, the add and the norm are one launch over the fp32 partials; every other shape runs the three calls.  out may
// alias a.  Same bits either way.
pegainfer_status_t pegainfer_gemm_add_then_rms_norm(const Half* W, const Half* X, Half* y_scratch, const Half* a, Half* out,
                                                    const Half* norm_weight, Half* normed_out, int32_t M, int32_t T,
                                                    int32_t K, float eps, pegainfer_stream_t stream) {
  using namespace pk;
  if (M <= 0 || T <= 0 || K <= 0 || !y_scratch || !a || !out || !norm_weight || !normed_out)
    return (pegainfer_status_t)hipErrorInvalidValue;
  hipStream_t s = as_stream(stream);
  const bool vec = (M & 7) == 0 && host_aligned16(a) && host_aligned16(out) && host_aligned16(norm_weight) &&
                   host_aligned16(normed_out) && (size_t)M * 4 + 16 <= 64 * 1024;
  const bool split = glds_gemm_ok(W, X, y_scratch, M, K) &&
                     (T > 64 ? splitk_plan(M, T, K).nk_slice > 0 : mid_batch_route(W, X, y_scratch, M, T, K) == 2);
  if (vec && split) {
    const SplitKPlan pl = splitk_plan(M, T, K);
    glds_splitk_launch(W, X, y_scratch, M, T, K, SplitOut{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}, pl, s, false);
    splitk_reduce_add_norm_kernel<<<T, 512, (size_t)M * 4 + 16, s>>>(g_splitk_ws, a, out, norm_weight, normed_out, M, T,
                                                                    pl.ksplit, eps, true);
    return (pegainfer_status_t)hipGetLastError();
  }
  static const bool fuse_rows = [] { const char* e = getenv("PEGAINFER_ADD_THEN_NORM_FUSED"); return !(e && e[0] == '0'); }();
  if (vec && !split && fuse_rows && T > 16 && host_aligned16(y_scratch)) {
    gemm_dispatch(W, X, y_scratch, M, T, K, s);
    static const bool cached = [] { const char* e = getenv("PEGAINFER_NORM_ROWS_CACHED"); return !(e && e[0] == '0'); }();
    if (cached && M <= 8 * 64 * 5)
      add_then_rms_norm_vec_kernel<5><<<ceil_div(T, 4), 256, 0, s>>>(a, y_scratch, norm_weight, out, normed_out, M, T, eps);
    else if (cached && M <= 8 * 64 * 8)
      add_then_rms_norm_vec_kernel<8><<<ceil_div(T, 4), 256, 0, s>>>(a, y_scratch, norm_weight, out, normed_out, M, T, eps);
    else
      add_then_rms_norm_vec_kernel<0><<<ceil_div(T, 4), 256, 0, s>>>(a, y_scratch, norm_weight, out, normed_out, M, T, eps);
    return (pegainfer_status_t)hipGetLastError();
  }
  const pegainfer_status_t rc = pegainfer_gemm_add(W, X, y_scratch, a, out, M, T, K, stream);
  if (rc) return rc;
  rms_norm_batched_cuda(out, norm_weight, normed_out, M, T, eps, stream);
  return (pegainfer_status_t)hipGetLastError();
}

}  // extern "C"
