// Decode GEMV core for gfx950 - shared by the reference-ABI GEMM entry points (linear.hip) and the
// fused decode entry point (pegainfer_gemv_fused).  HBM-bound weight streaming:
//   * every W element is read exactly once with 16-byte non-temporal loads straight into VGPRs
//     (no LDS round trip for the streamed operand); U x RPW loads (16 KB per wave) are in flight
//     and the first group is issued BEFORE the x prologue so HBM latency overlaps it;
//   * x lives in LDS (K-tiled, shared by the 4 waves); optional prologue builds x on the fly as
//     rms_norm(X [+ residual]) * w with the canonical summation order of norm_core.h, so the fused
//     form rounds exactly like "fused_add_rms_norm kernel, then GEMV";
//   * v_dot2c_f32_bf16 accumulation, wave64 butterfly; K >= 4096 deals the 512-element K blocks
//     round-robin to the 4 waves (LDS combine in fixed order) so M = 2560 still gives 1280 workgroups;
//   * optional epilogue applies SwiGLU to (gate row r, up row I + r) pairs computed by the same wave,
//     rounding gate/up to bf16 first exactly like the unfused GEMM -> silu_mul_fused sequence.
// The per-(row, token) summation order depends only on (K, KSPLIT): never on T, M, RPW or the
// prologue/epilogue, which is what makes batch decode == single decode bit-for-bit.
#pragma once

#include "common.h"
#include "norm_core.h"

namespace pk {

// x K-tile: chosen on the host (gemv_pick_kt).  Whenever NT*K*2 bytes fit in 152 KB of LDS the whole x block
// is resident (single tile) and workgroups persist over row groups; otherwise x is streamed in tiles whose
// width is a multiple of 2048 (4 waves x 512) so the KSPLIT block->wave deal is identical for every NT.
constexpr int kGemvMaxLds = 152 * 1024;
inline int gemv_pick_kt(int NT, int K) {
  if ((long)NT * K * 2 <= kGemvMaxLds) return K;
  int kt = (64 * 1024 / (NT * 2)) / 2048 * 2048;
  return kt < 2048 ? 2048 : kt;
}

enum { kEpiStore = 0, kEpiSilu = 1 };

template <int NT>
__host__ __device__ inline int gemv_xs_bytes(int K, int KT) {
  const int kt = K < KT ? K : KT;
  return (NT * kt * 2 + 15) & ~15;
}
template <int NT, int RPW, int KSPLIT, int EPI>
inline int gemv_lds_bytes(int K, int KT, bool residual) {
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;
  // x tile (+ the raw residual rows of the early prologue) + K-split partials + 16 floats: inv_rms per token
  return gemv_xs_bytes<NT>(K, KT) * (residual && NT <= 2 ? 2 : 1) + ((KSPLIT == 1 ? 0 : 4 * NW * RPW * NT) + 16) * 4;
}

struct GemvFusedArgs {
  const Half* W; const Half* X; Half* Y; int M; int T; int K;
  const Half* residual;   // optional: x = norm(X + residual), hidden_out = bf16(X + residual)
  const Half* norm_w;     // optional: non-null enables the RMSNorm prologue
  Half* hidden_out;       // written by workgroup 0 only (must not alias X)
  float eps;
  int I;                  // kEpiSilu: W = [gate(I rows); up(I rows)], Y = [T, I]
  int KT;                 // x tile width (gemv_pick_kt)
  int flags;              // kGemvNormOffset | kGemvRoundSum | kGemvSiluRound (dot2 GEMV path only)
  unsigned long long* trace;  // debug (pegainfer_debug_gemv_trace): 8 wall-clock stamps per workgroup, else null
  int rpb;                // skinny kernels: W rows per row block (0 = 16); set by the launcher (skinny_pick_rpb)
  int variant;            // skinny resident kernel (launcher: skinny_flush_mode): bit 0 one-barrier flush, bit 2 ticket flush with
                          // the ring size in bits 4..7, bit 1 timing probe
};
// process-wide debug hook, defined in linear.hip; copied into the launch arguments by gemv_launch_one
extern unsigned long long* g_gemv_trace;
#define PK_GEMV_STAMP(a, i)                                                                        \
  do {                                                                                             \
    if ((a).trace && threadIdx.x == 0) (a).trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64();   \
  } while (0)
// Qwen3.5 forms of the fused pieces (pegainfer_gemv_fused_ex): (1 + w) norm weight; residual sum rounded to bf16
// before the norm ("add, then norm" instead of FlashInfer's fused add+norm); silu rounded to bf16 before * up
enum { kGemvNormOffset = 1, kGemvRoundSum = 2, kGemvSiluRound = 4 };
constexpr int kGemvProloguePrio = 64;   // launcher-only bit (PEGAINFER_GEMV_PRIO=0 clears it): see the prologue
constexpr int kGemvNormAllWaves = 32;   // launcher-only bit (PEGAINFER_GEMV_NORM1W=0): every wave sums the squares (A/B probe)
// NOTE (round 6): this kernel's code generation is FRAGILE.  Two extra debug stamps inside the norm prologue (a launcher-only
// flag bit, dead at run time) cost the K = 4096 KSPLIT forms 10 % (Qwen3-8B gate_up 33.8 -> 37.2 us, 323.5 -> 304.5 tok/s) and
// the K = 2560 forms 0.9 %, same box, same hour (profiles/r6_libs_ab_*.txt) - at unchanged VGPR counts.  They were removed
// again; the split they measured is in profiles/r6_gemv_prologue_split.txt.  Any edit here needs a same-box A/B of BOTH models
// (python -m pegainfer_amd.build --variant NAME + PEGAINFER_LIB_DIR).

// U = K blocks a wave keeps in flight per row (U*NW*RPW loads of 1 KB).  The launcher picks U = 5 when that covers a
// whole row (K = 2560: 5 blocks; K = 9728 dealt to 4 waves: 5, 5, 5, 4), so a row group needs ONE memory round trip
// instead of "four blocks, then a dependent fifth"; the per-(row, token) block order is the same for every U.
template <int NT, int RPW, int KSPLIT, int EPI, int U = 4>
__global__ __launch_bounds__(256) void gemv_fused_kernel(const GemvFusedArgs a) {
  const int KT = a.KT;
  constexpr int NW = EPI == kEpiSilu ? 2 : 1;  // weight row sets streamed together
  constexpr int ROWS_PER_GROUP = (KSPLIT == 1 ? 4 : 1) * RPW;
  // all LDS is dynamic and sized to the shape (guide G17: one 16-byte aligned carve, no statics): a K = 2560
  // GEMV needs 5 KB, not the 32 KB tile capacity, which is what lets 5+ workgroups share a CU.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
  const int xs_bytes = gemv_xs_bytes<NT>(a.K, a.KT);
  u32x4* rs = reinterpret_cast<u32x4*>(smem_raw + xs_bytes);   // raw residual rows (early prologue), when given
  float* part = reinterpret_cast<float*>(smem_raw + xs_bytes * (a.residual && NT <= 2 ? 2 : 1));
  float* red = part + (KSPLIT == 1 ? 0 : 4 * NW * RPW * NT);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int K = a.K, T = a.T;
  const int pitch = (K < KT ? K : KT) >> 3;  // 16-byte vectors per token row of the x tile
  const int rows_total = EPI == kEpiSilu ? a.I : a.M;
  const int ngroups = (rows_total + ROWS_PER_GROUP - 1) / ROWS_PER_GROUP;

  // Persistent over row groups: the launcher sizes the grid to what is co-resident (gridDim <= ngroups) and a
  // workgroup walks groups g, g + gridDim, ... - x is normalised / staged ONCE per workgroup and the next
  // group's weight loads are already in flight while the current group is reduced and stored.
  const Half* wrow[NW][RPW];
  auto set_rows = [&](int g) {
    const int row0 = (KSPLIT == 1 ? g * 4 + wave : g) * RPW;
#pragma unroll
    for (int s = 0; s < NW; ++s)
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        int row = row0 + r;
        row = row < rows_total ? row : rows_total - 1;  // clamp: loads stay in bounds, stores are masked
        wrow[s][r] = a.W + ((size_t)row + (size_t)s * a.I) * K;
      }
  };
  float acc[NW][RPW][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int s = 0; s < NW; ++s)
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][r][t] = 0.f;
  };

  u32x4 wv[U][NW][RPW];
  auto issue = [&](int k0, int kt, int b0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int kk = (b0 + u * KSPLIT) * 512 + lane * 8;
      kk = kk < kt ? kk : kt - 8;  // clamped loads read valid weights; their x is zeroed below
#pragma unroll
      for (int s = 0; s < NW; ++s)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
          wv[u][s][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[s][r] + k0 + kk));
    }
  };
  auto consume = [&](int kt, int nblk, int b0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int blk = b0 + u * KSPLIT;
      const int kk = blk * 512 + lane * 8;
      const bool live = blk < nblk && kk < kt;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        u32x4 xv = xs[t * pitch + (live ? (kk >> 3) : 0)];
        if (!live) xv = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int s = 0; s < NW; ++s)
#pragma unroll
          for (int r = 0; r < RPW; ++r) acc[s][r][t] = dot8(wv[u][s][r], xv, acc[s][r][t]);
      }
    }
  };

  const int bfirst = KSPLIT == 1 ? 0 : wave;
  const int kt0 = K < KT ? K : KT;
  int g = blockIdx.x;
  PK_GEMV_STAMP(a, 0);
  set_rows(g);
  // ---- prologue.  A wave's loads retire in order, so anything loaded AFTER the weight group waits for the weights
  //      (cold HBM, and the whole matrix is requested at once): with the weights issued first the x vector was staged
  //      only when they had all landed - 5.8 of the 9.0 us of the fused qkv GEMV, 6.1 of gate_up's 21.5 (in-kernel
  //      stamps, tools/gemv_probe.py) - and the dot products started after the stream instead of under it.  The
  //      resident-x form therefore moves the raw x (and residual) rows global -> LDS by LDS-DMA FIRST (1 KiB pieces,
  //      no staging registers: the kernel stays at its 4-workgroups-per-CU register budget), issues the weight group
  //      right behind them, and waits with a COUNTED vmcnt for exactly the DMAs:
  //        raw rows in LDS | barrier | canonical one-wave sum of squares read back from LDS (every wave computes
  //        it, same bits as norm_core.h) | barrier | scale own vectors in place | barrier.
  float inv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) inv[t] = 0.f;
  const float nbias = (a.flags & kGemvNormOffset) ? 1.f : 0.f;
  const bool round_sum = (a.flags & kGemvRoundSum) != 0;
  const int nvec_row = K >> 3;
  const bool early = NT <= 2 && K <= KT && (K & 511) == 0 && nvec_row <= 2 * 256 * (a.norm_w ? 1 : 4);
  // The prologue is a serial chain (loads -> barrier -> sum -> barrier -> scale -> barrier) that runs next to the dot
  // products of the workgroups already streaming on this CU; those are waiting on HBM most of the time, so the
  // prologue's instructions go first.
  const bool prio = (a.flags & kGemvProloguePrio) != 0;
  if (prio) __builtin_amdgcn_s_setprio(3);
  if (early) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const uint32_t xs_lds = (uint32_t)(uintptr_t)(lds_ptr_t)xs, rs_lds = (uint32_t)(uintptr_t)(lds_ptr_t)rs;
    constexpr int XP = 2;
    u32x4 gx[XP];
    // Who does what in the norm prologue.  Measured (stamps, tools/gemv_probe.py sites 8 / 9): the 2.6 us the norm
    // sites need beyond a plain site's 1.4 us are not memory - one hot norm weight for every launch changes nothing -
    // but VALU time: every wave of every co-resident workgroup summed the squares redundantly (4 waves per SIMD x ~200
    // instructions) and wave 0 scaled two vectors per lane.  So ONE wave per token row sums (the canonical one-wave
    // order needs exactly one), its neighbour takes the vectors beyond 256, and the roles rotate with blockIdx / 256 -
    // the workgroups sharing a CU differ in that - so they land on different SIMDs.
    const int w0 = (a.flags & kGemvNormAllWaves) ? 0 : (int)((blockIdx.x >> 8) & 3);
    const int cvec[XP] = {(int)threadIdx.x, 256 + ((((wave - w0 - 1) & 3) << 6) | lane)};
    if (a.norm_w) {
#pragma unroll
      for (int i = 0; i < XP; ++i)
        if (cvec[i] < nvec_row) gx[i] = reinterpret_cast<const u32x4*>(a.norm_w)[cvec[i]];
    }
    const int pieces = K >> 9;   // 1 KiB pieces per row; wave w moves pieces w, w + 4, ...
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= T) continue;
      for (int p = wave; p < pieces; p += 4) {
        const uint32_t off = (uint32_t)(t * pitch * 16 + p * 1024);
        const Half* src = a.X + (size_t)t * K + p * 512 + lane * 8;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(xs_lds + off)) : "memory", "m0");
        if (a.residual) {
          const Half* rsrc = a.residual + (size_t)t * K + p * 512 + lane * 8;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                       :: "v"(rsrc), "s"(__builtin_amdgcn_readfirstlane(rs_lds + off)) : "memory", "m0");
        }
      }
    }
    issue(0, kt0, bfirst);
    // the U * NW * RPW weight loads issued last may stay in flight; everything older (the DMAs) has landed
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U * NW * RPW) : "memory");
    PK_GEMV_STAMP(a, 6);   // wave 0's own x-side loads have landed
    __syncthreads();
    PK_GEMV_STAMP(a, 7);   // every wave's have
    if (a.norm_w) {
      const bool all_waves = (a.flags & kGemvNormAllWaves) != 0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t >= T || !(all_waves || wave == ((w0 + t) & 3))) continue;
        float ss = 0.f;
        if (a.residual && round_sum) {
          for (int j = lane; j < nvec_row; j += 64) add_round_sq8(xs[t * pitch + j], rs[t * pitch + j], ss);
        } else if (a.residual) {
          for (int j = lane; j < nvec_row; j += 64) add_sq8(xs[t * pitch + j], rs[t * pitch + j], ss);
        } else {
          for (int j = lane; j < nvec_row; j += 64) sq8(xs[t * pitch + j], ss);
        }
        ss = wave_sum(ss);
        inv[t] = rsqrtf(__fadd_rn(ss / (float)K, a.eps));
        if (!all_waves && lane == 0) red[t] = inv[t];
      }
      __syncthreads();  // the raw rows have been read; the inverse RMS values are in LDS
      if (!all_waves) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t < T) inv[t] = red[t];
      }
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const int c = cvec[i];
        if (c < nvec_row) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (t >= T) continue;
            const u32x4 h = xs[t * pitch + c];
            u32x4 v;
            if (a.residual) {
              const u32x4 r = rs[t * pitch + c];
              u32x4 nh;
              v = norm_scale8(h, &r, gx[i], inv[t], nbias, &nh, round_sum);
              if (blockIdx.x == 0) reinterpret_cast<u32x4*>(a.hidden_out + (size_t)t * K)[c] = nh;
            } else {
              v = norm_scale8(h, nullptr, gx[i], inv[t], nbias, nullptr);
            }
            xs[t * pitch + c] = v;
          }
        }
      }
      __syncthreads();
    }
  } else {
  issue(0, kt0, bfirst);  // weights first; the x-side loads queue behind them (wide-batch / tiled-x forms)
  // per-token inverse RMS (canonical one-wave-per-row order), only when a norm weight is given:
  // wave w takes tokens t == w (mod 4), results meet in LDS, one barrier
  if (a.norm_w) {
    for (int t = wave; t < T; t += 4) {
      const float v = wave_row_inv_rms(a.X + (size_t)t * K, a.residual ? a.residual + (size_t)t * K : nullptr, K, a.eps,
                                       (a.flags & kGemvRoundSum) != 0);
      if (lane == 0) red[t] = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < T) inv[t] = red[t];
  }
  }
  auto stage = [&](int k0, int kt) {
    const int nvec = (kt + 7) >> 3;  // lanes beyond kt never read their slot (consume() zeroes them)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= T) continue;          // accumulators of absent tokens are never stored
      for (int c = threadIdx.x; c < nvec; c += 256) {
        u32x4 v;
        const size_t off = (size_t)t * K + k0 + c * 8;
        const u32x4 h = *reinterpret_cast<const u32x4*>(a.X + off);
        if (a.norm_w) {
          const u32x4 gw = *reinterpret_cast<const u32x4*>(a.norm_w + k0 + c * 8);
          if (a.residual) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(a.residual + off);
            u32x4 nh;
            v = norm_scale8(h, &r, gw, inv[t], nbias, &nh, round_sum);
            if (blockIdx.x == 0) *reinterpret_cast<u32x4*>(a.hidden_out + off) = nh;
          } else {
            v = norm_scale8(h, nullptr, gw, inv[t], nbias, nullptr);
          }
        } else {
          v = h;
        }
        xs[t * pitch + c] = v;
      }
    }
  };
  if (!early) {
    stage(0, kt0);
    __syncthreads();
  }
  if (prio) __builtin_amdgcn_s_setprio(0);
  PK_GEMV_STAMP(a, 1);
  bool first_group = true;

  for (;;) {
    zero_acc();
    // ---- tile 0 (first block group already in flight) ----
    {
      const int nblk = (kt0 + 511) >> 9;
      consume(kt0, nblk, bfirst);
      if (first_group) { PK_GEMV_STAMP(a, 2); first_group = false; }
      for (int b0 = bfirst + U * KSPLIT; b0 < nblk; b0 += U * KSPLIT) {
        issue(0, kt0, b0);
        consume(kt0, nblk, b0);
      }
    }
    // ---- further K tiles: only when K > KT, in which case the launcher gives every group its own workgroup ----
    for (int k0 = KT; k0 < K; k0 += KT) {
      const int kt = (K - k0) < KT ? (K - k0) : KT;
      __syncthreads();
      stage(k0, kt);
      __syncthreads();
      const int nblk = (kt + 511) >> 9;
      for (int b0 = bfirst; b0 < nblk; b0 += U * KSPLIT) {
        issue(k0, kt, b0);
        consume(kt, nblk, b0);
      }
    }
    const int g_next = g + gridDim.x;
    const bool has_next = g_next < ngroups;
    if (!has_next) PK_GEMV_STAMP(a, 3);
    const int row0 = (KSPLIT == 1 ? g * 4 + wave : g) * RPW;  // rows of the group being finished
    if (has_next) {
      set_rows(g_next);
      issue(0, kt0, bfirst);  // next group's weights stream while this group is reduced
    }

#pragma unroll
    for (int s = 0; s < NW; ++s)
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][r][t] = wave_sum(acc[s][r][t]);

    auto emit = [&](int r, int t, float v0, float v1) {
      const int row = row0 + r;
      if (row >= rows_total || t >= T) return;
      if (EPI == kEpiSilu) {
        const float gt = bf16_round_f(v0), up = bf16_round_f(v1);  // the GEMM output is bf16 before SwiGLU
        const float sg = (a.flags & kGemvSiluRound) ? bf16_round_f(silu_f(gt)) : silu_f(gt);
        a.Y[(size_t)t * a.I + row] = f2bf(sg * up);
      } else {
        a.Y[(size_t)t * a.M + row] = f2bf(v0);
      }
    };
    if (KSPLIT == 1) {
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) emit(r, t, acc[0][r][t], acc[NW - 1][r][t]);
      }
    } else {
      if (lane == 0) {
#pragma unroll
        for (int s = 0; s < NW; ++s)
#pragma unroll
          for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) part[((wave * NW + s) * RPW + r) * NT + t] = acc[s][r][t];
      }
      __syncthreads();
      if (threadIdx.x < RPW * NT) {
        const int r = threadIdx.x / NT, t = threadIdx.x - r * NT;
        float tot[NW];
#pragma unroll
        for (int s = 0; s < NW; ++s) {
          float v = part[((0 * NW + s) * RPW + r) * NT + t];
          v += part[((1 * NW + s) * RPW + r) * NT + t];
          v += part[((2 * NW + s) * RPW + r) * NT + t];
          v += part[((3 * NW + s) * RPW + r) * NT + t];
          tot[s] = v;
        }
        emit(r, t, tot[0], tot[NW - 1]);
      }
      if (has_next) __syncthreads();  // `part` is rewritten by the next group
    }
    if (!has_next) break;
    g = g_next;
  }
  PK_GEMV_STAMP(a, 4);
  if (a.trace && threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.trace[(size_t)blockIdx.x * 8 + 5] = xcc & 0xfu;
  }
}

// Grid sizing: every workgroup should be co-resident (no second, half-empty scheduling round) and all of them
// should walk the same number of row groups.  capacity = occupancy(kernel, lds) x CU count; rounds =
// ceil(ngroups / capacity); grid = ceil(ngroups / rounds).  K > KT (multi-tile x) needs one group per workgroup.
template <int NT, int RPW, int KSPLIT, int EPI, int U = 4>
inline void gemv_launch_one(const GemvFusedArgs& a, hipStream_t s) {
  constexpr int ROWS_PER_GROUP = (KSPLIT == 1 ? 4 : 1) * RPW;
  const int rows = EPI == kEpiSilu ? a.I : a.M;
  const int ngroups = ceil_div(rows, ROWS_PER_GROUP);
  const int lds = gemv_lds_bytes<NT, RPW, KSPLIT, EPI>(a.K, a.KT, a.residual != nullptr);
  auto kern = &gemv_fused_kernel<NT, RPW, KSPLIT, EPI, U>;
  static const bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  static thread_local int cached_lds = -1, cached_cap = 0;
  if (cached_lds != lds) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cached_cap = per_cu * (cus > 0 ? cus : 256);
    cached_lds = lds;
  }
  int grid = ngroups;
  // PEGAINFER_GEMV_PERSIST=0: one row group per workgroup, the hardware dispatcher balances (A/B probe knob)
  static const bool persist = [] { const char* e = getenv("PEGAINFER_GEMV_PERSIST"); return !(e && *e == '0'); }();
  // PEGAINFER_GEMV_GRID_MULT=N: persistent grids are rounded down to a multiple of N (N = CU count: every CU hosts the
  // same number of workgroups, so no CU carries 4/3 of its neighbour's rows - the 23 % tail of the round-2 phase trace
  // was 811 workgroups on 256 CUs); PEGAINFER_GEMV_GRID_PER_CU caps the workgroups per CU below the occupancy
  static const int mult = [] { const char* e = getenv("PEGAINFER_GEMV_GRID_MULT"); return e && *e ? atoi(e) : 256; }();
  // default for the 1-2 column forms: 2 persistent workgroups per CU when a workgroup keeps >= kGemvCapMinKB of weight
  // loads in flight (K = 2560 rows whole: 40 KB; K = 9728 dealt to 4 waves: 40 KB) - 512 workgroups walking their row
  // groups beat 768-1280 there (x is staged by fewer workgroups, every CU carries the same row count): 2.045 -> 2.010 ms
  // per Qwen3-4B step, same box.  Workgroups with little in flight (K = 4096 dealt to 4 waves: 16 KB; the Qwen3-8B
  // family: 3.166 capped vs 3.066 ms uncapped) need the occupancy to cover the latency and keep it.
  static const int per_cu_env = [] { const char* e = getenv("PEGAINFER_GEMV_GRID_PER_CU"); return e && *e ? atoi(e) : -1; }();
  static const int cap_min_kb = [] { const char* e = getenv("PEGAINFER_GEMV_CAP_MIN_KB"); return e && *e ? atoi(e) : 40; }();
  const int blocks_per_wave = KSPLIT == 1 ? ceil_div(a.K, 512) : ceil_div(ceil_div(a.K, 512), KSPLIT);
  constexpr int NW_ = EPI == kEpiSilu ? 2 : 1;
  const int inflight_kb = 4 * (blocks_per_wave < U ? blocks_per_wave : U) * NW_ * RPW;
  const int per_cu_cap = per_cu_env >= 0 ? per_cu_env : (NT <= 2 && inflight_kb >= cap_min_kb ? 2 : 0);
  int cap = cached_cap;
  if (per_cu_cap > 0 && mult > 0 && per_cu_cap * mult < cap) cap = per_cu_cap * mult;
  if (persist && a.K <= a.KT && ngroups > cap) {
    const int rounds = ceil_div(ngroups, cap);
    grid = ceil_div(ngroups, rounds);
    // the CU-count multiple matters when a workgroup walks only a few groups (gate_up: 3); a grid that walks dozens of
    // rounds (an uncapped lm_head: 60) balances by count, and there the even split measured better (Qwen3-8B lm_head
    // 177 vs 185 us)
    if (mult > 0 && cap / mult * mult >= mult && (rounds <= 8 || per_cu_cap > 0)) grid = cap / mult * mult;
  }
  GemvFusedArgs b = a;
  b.trace = g_gemv_trace;
  static const bool norm1w = [] { const char* e = getenv("PEGAINFER_GEMV_NORM1W"); return !(e && *e == '0'); }();
  if (!norm1w) b.flags |= kGemvNormAllWaves;
  static const bool prio = [] { const char* e = getenv("PEGAINFER_GEMV_PRIO"); return !(e && *e == '0'); }();
  if (prio) b.flags |= kGemvProloguePrio;
  kern<<<grid, 256, lds, s>>>(b);
}

template <int NT, int EPI>
inline void gemv_launch_nt(const GemvFusedArgs& a, hipStream_t s) {
  // rows per wave: 2 for the plain store (16 KB of loads per wave with U = 4 ... 8 KB), 1 for SwiGLU (two
  // matrices per wave); both stay <= 96 VGPRs for NT <= 2, i.e. 5 workgroups per CU.  RPW never changes results.
  constexpr int RPW = EPI == kEpiSilu ? 1 : 2;
  // whole rows in flight (U = 5) for the single-request / pair forms when a wave's share of a row is exactly 5 blocks
  // (PEGAINFER_GEMV_U5=0: the four-then-one form, for A/B runs); wider batches keep U = 4 (register budget)
  static const bool u5 = [] { const char* e = getenv("PEGAINFER_GEMV_U5"); return !(e && *e == '0'); }();
  const int nblk = (a.K + 511) >> 9;
  if (a.K >= 4096) {
    if (NT <= 2 && u5 && a.K <= a.KT && (nblk + 3) / 4 == 5) gemv_launch_one<NT, RPW, 4, EPI, (NT <= 2 ? 5 : 4)>(a, s);
    else gemv_launch_one<NT, RPW, 4, EPI>(a, s);
  } else {
    if (NT <= 2 && u5 && a.K <= a.KT && nblk == 5) gemv_launch_one<NT, RPW, 1, EPI, (NT <= 2 ? 5 : 4)>(a, s);
    else gemv_launch_one<NT, RPW, 1, EPI>(a, s);
  }
}

// T <= 16, K % 8 == 0, 16-byte aligned W/X (and residual/norm_w/hidden_out when given)
template <int EPI>
inline bool gemv_dispatch(const GemvFusedArgs& a_in, hipStream_t s) {
  if (a_in.T < 1 || a_in.T > 16 || (a_in.K & 7) != 0) return false;
  GemvFusedArgs a = a_in;
  if (a.T > 8 && (long)16 * a.K * 2 > kGemvMaxLds) {
    // 9..16 tokens but x[16][K] would not be LDS-resident: two resident 8-token passes instead (the second
    // pass streams the weights out of the 256 MB Infinity Cache); per-token arithmetic is unchanged.
    GemvFusedArgs lo = a, hi = a;
    lo.T = 8;
    hi.T = a.T - 8;
    hi.X = a.X + (size_t)8 * a.K;
    if (a.residual) hi.residual = a.residual + (size_t)8 * a.K;
    if (a.hidden_out) hi.hidden_out = a.hidden_out + (size_t)8 * a.K;
    hi.Y = a.Y + (size_t)8 * (EPI == kEpiSilu ? a.I : a.M);
    return gemv_dispatch<EPI>(lo, s) && gemv_dispatch<EPI>(hi, s);
  }
  const int nt = a.T == 1 ? 1 : a.T == 2 ? 2 : a.T <= 4 ? 4 : a.T <= 8 ? 8 : 16;
  a.KT = gemv_pick_kt(nt, a.K);
  if (nt == 1) gemv_launch_nt<1, EPI>(a, s);
  else if (nt == 2) gemv_launch_nt<2, EPI>(a, s);
  else if (nt == 4) gemv_launch_nt<4, EPI>(a, s);
  else if (nt == 8) gemv_launch_nt<8, EPI>(a, s);
  else gemv_launch_nt<16, EPI>(a, s);
  return true;
}

}  // namespace pk
