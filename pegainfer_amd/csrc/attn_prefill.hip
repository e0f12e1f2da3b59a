// Causal GQA prefill attention over the paged KV cache for gfx950 (replaces FlashInfer
// BatchPrefillWithPagedKVCacheDispatched / SinglePrefillWithKVCacheDispatched behind
// batch_prefill_paged_cuda* and single_prefill_cuda, reference csrc/paged_attention.cu:399-608),
// plus the host-side plan helpers (FA2DetermineCtaTileQ restated, :326-397).
//
// MFMA-bound (dense QK^T / PV contractions).  Flash-attention structure for wave64 + MFMA:
//   * one workgroup (4 waves) per plan tile x kv head; a plan tile = cta_tile_q packed rows where
//     packed row r -> (token r / GROUP, head kvh*GROUP + r % GROUP), so the GROUP query heads that
//     share a KV head share every K/V tile staged in LDS.
//   * swapped product S^T = K . Q^T with v_mfma_f32_16x16x32_bf16: the C fragment then holds, per
//     lane, 4 consecutive KV tokens of ONE query row -> the row max / row sum are lane-local plus
//     two cross-row shuffles, and bf16(P^T) is already the B operand of O^T = V^T . P^T with no
//     cross-lane movement (k-slot permutation applied identically to the V^T fragment).
//   * K tile [64 tok][128] in LDS with slot ^= row&15 (conflict-free ds_read_b128 A fragments);
//     V tile transposed on the way into LDS (4 rows -> 8 x ds_write_b64 after an in-register 4x8
//     transpose; row pitch 136 B) so V^T fragments are two ds_read_b64 each.
//   * global->register prefetch of the next KV tile overlaps the MFMAs of the current one.
//   * online softmax in the exp2 domain (sm_scale*log2e folded), fp32 accumulation, bf16 output.
// Round 3 adds the LDS-DMA form (template flag DMA; head_dim 128, power-of-two pages, 64- / 128-row tiles - what the model
// path runs): K and V tiles go global -> LDS by global_load_lds_dwordx4 into two-deep rings with ONE barrier per tile, V
// stays row-major and the V^T fragments are gathered by ds_read_b64_tr_b16, the row max crosses lane groups by
// v_permlane16/32_swap.  Both forms share compute_tile(): identical per-row arithmetic (sha-checked,
// tools/bench_prefill_attn.py).  The kernel is instruction-issue-bound (DESIGN.md "Prefill attention"), so the loop is
// written to spend few instructions: fragment reads in pairs (one s_waitcnt each), interleaved max chains.
// kv_len comes from the page table; the causal rule is kv_idx <= q_idx + (kv_len - qo_len).
#include <type_traits>

#include "common.h"

namespace pk {

constexpr int TKV = 64;           // KV tokens per LDS tile
constexpr int VT_PITCH = 68;      // bf16 elements per V^T row (64 tokens + 4 pad) = 136 B

constexpr int kPgLdsMax = 2048;   // page ids held in LDS at a time (a 32k-token window at page_size 16)

// Row addressing.  row(t) is called 8 times per thread and KV tile, so it must be a handful of integer ops: the
// paged form keeps a window of the request's page ids in LDS and, for a power-of-two page size (shift >= 0),
// splits t with a shift and a mask - reading the page id from global memory behind a 32-bit division per row cost
// 340 VALU instructions per thread and tile, a third of the whole loop.  load_window() is executed by every
// thread of the workgroup between two barriers; covers(t_end) says whether tokens < t_end are inside the window.
template <bool POW2>
struct PagedAddr {
  const int* page_indices; int pbase; int page_size; int stride_page /* elements, < 2^31 (host-checked) */;
  int row_stride; int shift; int* pg; int win0; int n_pages_total;
  int dma_inline;   // LDS-DMA form: issue the next tile's requests between this tile's MFMA clusters (A/B: PEGAINFER_PREFILL_DMA_INLINE=0)
  __device__ __forceinline__ int page_of_token(int t) const { return POW2 ? t >> shift : t / page_size; }
  __device__ __forceinline__ void load_window(int first_token) {
    win0 = page_of_token(first_token);
    const int n = n_pages_total - win0 < kPgLdsMax ? n_pages_total - win0 : kPgLdsMax;
    for (int i = threadIdx.x; i < n; i += blockDim.x) pg[i] = page_indices[pbase + win0 + i];
  }
  __device__ __forceinline__ bool covers(int t_end) const { return page_of_token(t_end - 1) < win0 + kPgLdsMax; }
  // element offset of token t's row inside the cache, without the (uniform) layer / head / K-V offsets
  __device__ __forceinline__ long row(int t) const {
    const int pi = page_of_token(t);
    const int in_page = POW2 ? t & (page_size - 1) : t - pi * page_size;
    return (long)pg[pi - win0] * (long)stride_page + (long)(in_page * row_stride);
  }
};
struct ContigAddr {  // HND: cache[head][pos][dim]; the head base is folded into the buffer pointers
  int D;
  __device__ __forceinline__ void load_window(int) {}
  __device__ __forceinline__ bool covers(int) const { return true; }
  __device__ __forceinline__ long row(int t) const { return (long)(t * D); }
};

__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32, no denormal fix-up
// fmaxf() on MFMA results makes hipcc canonicalise every operand first (a v_max_f32 x, x, x each: 54 extra VALU
// instructions per tile here); the raw three-input instruction does not
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int QBLK, int D, int NW, typename Addr, bool DMA = false>
__device__ __forceinline__ void prefill_tile(const Half* __restrict__ q_base, Half* __restrict__ o_base,
                                             const Half* __restrict__ kbuf, const Half* __restrict__ vbuf,
                                             Addr& addr, int qo_len, int kv_len, int tile_row0,
                                             int tile_rows, int group, long q_stride_n, float scale_log2,
                                             u32x4* ks /*[TKV*D/8]*/, Half* vt /*[D*VT_PITCH]*/) {
  constexpr int KCH = D / 8;               // 16-byte chunks per K/V row
  constexpr int KS = D / 32;               // MFMA k-steps over the head dim
  constexpr int DB = D / 16;               // 16-dim output blocks
  // NW waves x QBLK x 16 packed rows per workgroup.  Measured on MI355X (TTFT 1024 / 8192 tokens, ms): 4 waves x
  // 16 rows 17.0 / 164.5, 2 waves x 32 rows (each LDS fragment feeds two MFMAs) 17.6 / 184.5, 1 wave x 64 rows
  // 19.8 / 195.1 - occupancy, not LDS reads, is what the 64-row tile is short of.
  constexpr int NT = NW * 64;
  constexpr int KPT = TKV * KCH / NT;      // K chunks staged per thread
  constexpr int VPT = 16 * KCH / NT > 0 ? 16 * KCH / NT : 1;   // V (4-token x 8-dim) units staged per thread (register-staged form: NT <= 256)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, g = lane >> 4;
  const int packed_len = qo_len * group;
  const int tile_end = (tile_row0 + tile_rows) < packed_len ? (tile_row0 + tile_rows) : packed_len;
  const int causal_off = kv_len - qo_len;
  const int cta_kv_end_raw = (tile_end - 1) / group + causal_off + 1;
  const int cta_kv_end = cta_kv_end_raw < kv_len ? cta_kv_end_raw : kv_len;

  addr.load_window(0);                     // page ids -> LDS; visible after the barrier in front of the first load

  // per-lane query rows (one per 16-row block)
  int qtok[QBLK];
  bool qok[QBLK];
  bf16x8_t qf[QBLK][KS];
  long o_off[QBLK];
#pragma unroll
  for (int qb = 0; qb < QBLK; ++qb) {
    const int r = tile_row0 + (wave * QBLK + qb) * 16 + l15;
    qok[qb] = r < tile_end;
    const int rc = qok[qb] ? r : tile_row0;
    qtok[qb] = rc / group;
    const long off = (long)qtok[qb] * q_stride_n + (long)(rc % group) * D;
    o_off[qb] = off;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      qf[qb][s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(q_base + off + s * 32 + g * 8));
  }
  // LDS-DMA form: retire the q loads with a wait the COMPILER sees.  The DMAs below are inline asm, invisible to hipcc's
  // wait-count bookkeeping; it would otherwise carry "q fragments still pending" around the loop and put an
  // s_waitcnt vmcnt(0) in front of the first MFMAs of every KV tile - which in hardware waits for the NEXT tile's DMAs.
  if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  // wave-uniform bounds on the KV positions this wave attends to: [0, wave_kv_end) is visible to its last row,
  // [0, wave_kv_full) to every one of its rows (tiles below that line need no causal mask)
  int wave_last_row = tile_row0 + (wave * QBLK + QBLK) * 16 - 1;
  if (wave_last_row >= tile_end) wave_last_row = tile_end - 1;
  const bool wave_active = tile_row0 + wave * QBLK * 16 < tile_end;
  const int wave_kv_end = wave_active ? wave_last_row / group + causal_off + 1 : 0;
  const int wave_kv_full_raw = (tile_row0 + wave * QBLK * 16) / group + causal_off + 1;
  const int wave_kv_full = wave_kv_full_raw < kv_len ? wave_kv_full_raw : kv_len;

  f32x4 acc_o[QBLK][DB];
  float m_run[QBLK], l_run[QBLK];
#pragma unroll
  for (int qb = 0; qb < QBLK; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d) acc_o[qb][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // staging roles
  u32x4 kreg[KPT], vreg[VPT][4];
  // Full tiles of the common layout (head_dim 128, 16-token pages, power-of-two path): thread t stages K rows
  // t/16 + 16 j - row t/16 of page kv0/16 + j - and the four V rows 4 (t/16) .. + 3 of page kv0/16 + t/64, so a tile
  // needs five page ids and two loop-invariant in-page offsets instead of eight general row addresses
  // (100 -> ~30 VALU instructions per tile).
  constexpr bool kFastPaged = std::is_same<Addr, PagedAddr<true>>::value && D == 128 && NT == 256;
  int fast_k_off = 0, fast_v_off = 0, fast_v_pg = 0;
  bool fast_ok = false;
  if constexpr (kFastPaged) {
    fast_ok = addr.page_size == 16;
    const int r0 = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    fast_k_off = r0 * addr.row_stride + c0 * 8;
    fast_v_off = ((r0 & 3) * 4) * addr.row_stride + c0 * 8;
    fast_v_pg = r0 >> 2;
  }
  auto load_tile = [&](int kv0) {
    if constexpr (kFastPaged) {
      if (fast_ok && kv0 + TKV <= kv_len) {   // workgroup-uniform
        const int* pgw = addr.pg + ((kv0 >> 4) - addr.win0);
#pragma unroll
        for (int j = 0; j < KPT; ++j)
          kreg[j] = *reinterpret_cast<const u32x4*>(kbuf + (long)pgw[j] * (long)addr.stride_page + fast_k_off);
        const Half* vp = vbuf + (long)pgw[fast_v_pg] * (long)addr.stride_page + fast_v_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) vreg[0][i] = *reinterpret_cast<const u32x4*>(vp + i * addr.row_stride);
        return;
      }
    }
    const int t_last = kv_len - 1;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int cid = threadIdx.x + j * NT, row = cid / KCH, slot = cid % KCH;
      int t = kv0 + row;
      t = t < t_last ? t : t_last;
      kreg[j] = *reinterpret_cast<const u32x4*>(kbuf + addr.row(t) + slot * 8);
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int uid = threadIdx.x + j * NT, v_tq = uid / KCH, v_dc = uid % KCH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int tv = kv0 + v_tq * 4 + i;
        tv = tv < t_last ? tv : t_last;
        vreg[j][i] = *reinterpret_cast<const u32x4*>(vbuf + addr.row(tv) + v_dc * 8);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int cid = threadIdx.x + j * NT, row = cid / KCH, slot = cid % KCH;
      ks[row * KCH + (slot ^ (row & 15))] = kreg[j];
    }
    // 4 tokens x 8 dims -> 8 rows of V^T, 4 tokens each
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int uid = threadIdx.x + j * NT, v_tq = uid / KCH, v_dc = uid % KCH;
      const uint32_t w[4][4] = {{vreg[j][0].x, vreg[j][0].y, vreg[j][0].z, vreg[j][0].w},
                                {vreg[j][1].x, vreg[j][1].y, vreg[j][1].z, vreg[j][1].w},
                                {vreg[j][2].x, vreg[j][2].y, vreg[j][2].z, vreg[j][2].w},
                                {vreg[j][3].x, vreg[j][3].y, vreg[j][3].z, vreg[j][3].w}};
#pragma unroll
      for (int c = 0; c < 4; ++c) {  // dims 2c, 2c+1
        u32x2 lo, hi;
        lo.x = (w[0][c] & 0xFFFFu) | (w[1][c] << 16);
        lo.y = (w[2][c] & 0xFFFFu) | (w[3][c] << 16);
        hi.x = (w[0][c] >> 16) | (w[1][c] & 0xFFFF0000u);
        hi.y = (w[2][c] >> 16) | (w[3][c] & 0xFFFF0000u);
        // lanes of a wave write rows 8 apart (v_dc * 8): with any 8-byte-aligned pitch those land on only 4 bank
        // groups (4-way conflict, 46 % of the LDS cycles of this kernel by SQ_LDS_BANK_CONFLICT).  Rotating the
        // token columns of each 32-row group by 4 quads spreads the 16 v_dc over all banks (2-way).  Same-box A/B at
        // ctx 4096 (us per launch): no rotation 633, this 568; a 2-quad rotation is conflict-free on paper and by
        // counter (LDS active 5.2M vs 6.3M) yet measured 652 - kept what the clock says.
        const int qcol = ((v_tq + 4 * (v_dc >> 2)) & 15) * 4;
        *reinterpret_cast<u32x2*>(vt + (v_dc * 8 + 2 * c) * VT_PITCH + qcol) = lo;
        *reinterpret_cast<u32x2*>(vt + (v_dc * 8 + 2 * c + 1) * VT_PITCH + qcol) = hi;
      }
    }
  };

  // one KV tile's arithmetic for this wave: kt = the staged K tile, load_vf(db, kb) = the V^T fragment (16 dims x 32 tokens)
  // hook(slot), slot 0..7 in the S loop and 8..15 in the PV loop: the LDS-DMA form issues the NEXT tile's requests from
  // there, one between MFMA clusters (see issue_piece), the register-staged form passes a no-op
  auto compute_tile = [&](int kv0, const u32x4* kt, auto&& load_vf, auto&& hook) {
    // ---- S^T = K . Q^T ----  (raised issue priority around the MFMA clusters: the other resident wave of the SIMD is
    //      in its softmax / staging VALU segment and should not delay these)
    __builtin_amdgcn_s_setprio(1);
    f32x4 sacc[QBLK][4];
#pragma unroll
    for (int qb = 0; qb < QBLK; ++qb)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) sacc[qb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K fragments in PAIRS through a two-pair register ring.  The kernel is instruction-issue-bound and every fragment
    // hipcc sees for the first time costs one s_waitcnt: the pair's YOUNGER read is used first, so one wait covers both.
    // Pair p = dim slice p % KS of the token blocks 2 (p / KS) and 2 (p / KS) + 1: different accumulators, so every
    // accumulator still sees its slices in ascending order (same bits).  Pair p + 2 is requested behind pair p's MFMAs.
    auto load_kf = [&](int tb, int sl) {
      const int row = tb * 16 + l15;
      return __builtin_bit_cast(bf16x8_t, kt[row * KCH + ((sl * 4 + g) ^ (row & 15))]);
    };
    {
      bf16x8_t kr[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        kr[p][0] = load_kf(2 * (p / KS), p % KS);
        kr[p][1] = load_kf(2 * (p / KS) + 1, p % KS);
      }
#pragma unroll
      for (int p = 0; p < 2 * KS; ++p) {
        const int tb0 = 2 * (p / KS), sl = p % KS;
        if constexpr (KS == 4) hook(p);
#pragma unroll
        for (int qb = 0; qb < QBLK; ++qb)
          sacc[qb][tb0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[p & 1][1], qf[qb][sl], sacc[qb][tb0 + 1], 0, 0, 0);
#pragma unroll
        for (int qb = 0; qb < QBLK; ++qb)
          sacc[qb][tb0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[p & 1][0], qf[qb][sl], sacc[qb][tb0], 0, 0, 0);
        if (p + 2 < 2 * KS) {
          kr[p & 1][0] = load_kf(2 * ((p + 2) / KS), (p + 2) % KS);
          kr[p & 1][1] = load_kf(2 * ((p + 2) / KS) + 1, (p + 2) % KS);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- online softmax per query row (= per lane column), exp2 domain with the scale folded into one fma.
    //      The causal / length mask is applied only on the tiles that cross this wave's diagonal or the end of
    //      the sequence (wave-uniform test); interior tiles - nearly all of a long prompt - take no compare at all.
    // the first V^T fragments are requested here: their LDS round trip runs under the softmax arithmetic
    // pair p = token half p % 2 of the dim blocks 2 (p / 2) and 2 (p / 2) + 1
    bf16x8_t vr[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      vr[p][0] = load_vf(2 * (p >> 1), p & 1);
      vr[p][1] = load_vf(2 * (p >> 1) + 1, p & 1);
    }
    const bool need_mask = kv0 + TKV > wave_kv_full;
    bf16x8_t pf[QBLK][2];
#pragma unroll
    for (int qb = 0; qb < QBLK; ++qb) {
      if (need_mask) {
        const int lim_c = qok[qb] ? qtok[qb] + causal_off : -1;  // last visible kv index
        const int limit = lim_c < kv_len - 1 ? lim_c : kv_len - 1;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int kvt = kv0 + tb * 16 + g * 4 + i;
            sacc[qb][tb][i] = kvt <= limit ? sacc[qb][tb][i] : -INFINITY;
          }
      }
      // two interleaved max3 chains (a dependent inline-asm pair costs a hazard s_nop each, and issue slots are what
      // this kernel is short of); the maximum does not depend on the order
      float mxa = max3_raw(sacc[qb][0][0], sacc[qb][0][1], sacc[qb][0][2]);
      float mxb = max3_raw(sacc[qb][2][0], sacc[qb][2][1], sacc[qb][2][2]);
      mxa = max3_raw(mxa, sacc[qb][0][3], sacc[qb][1][0]);
      mxb = max3_raw(mxb, sacc[qb][2][3], sacc[qb][3][0]);
      mxa = max3_raw(mxa, sacc[qb][1][1], sacc[qb][1][2]);
      mxb = max3_raw(mxb, sacc[qb][3][1], sacc[qb][3][2]);
      float mx = max3_raw(mxa, sacc[qb][1][3], sacc[qb][3][3]);
      mx = fmaxf(mx, mxb);
      {  // max over the row's 4 lane groups: v_permlane16/32_swap (gfx950; VALU speed, no LDS round trips as with
         // ds_bpermute): rows [m0 m1 m2 m3] -> [m01 m01 m23 m23] -> all.  Inline asm on purpose: the swap needs two
         // REGISTERS, and __builtin_amdgcn_permlane16_swap with one value for both operands (also behind an opaque
         // copy) makes hipcc drop the max that follows (checked in the ISA; results then differ).  s_nop 1 = the two
         // wait states hipcc itself puts between a VALU write and the swap.
        float a = mx, b = mx;
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        a = fmaxf(a, b); b = a;
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        mx = fmaxf(a, b);
      }
      const float mn = fmaxf(m_run[qb], mx * scale_log2);   // scale_log2 > 0: the max commutes with the scaling
      const float msafe = mn == -INFINITY ? 0.f : mn;
      // rescale only when some row's running max grew (alpha == 1 exactly otherwise): wave-uniform branch
      if (__builtin_amdgcn_ballot_w64(mn > m_run[qb]) != 0ull) {
        const float alpha = exp2_raw(m_run[qb] - msafe);
        l_run[qb] *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d) acc_o[qb][d] *= alpha;
        m_run[qb] = mn;
      }
      float p[4][4];
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[tb][i] = exp2_raw(fmaf(sacc[qb][tb][i], scale_log2, -msafe));
        ps0 += p[tb][0] + p[tb][1];
        ps1 += p[tb][2] + p[tb][3];
      }
      l_run[qb] += ps0 + ps1;  // lane-partial; summed over g at the end
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        u32x4 pk4;
        pk4.x = pack_bf2(p[2 * kb][0], p[2 * kb][1]);
        pk4.y = pack_bf2(p[2 * kb][2], p[2 * kb][3]);
        pk4.z = pack_bf2(p[2 * kb + 1][0], p[2 * kb + 1][1]);
        pk4.w = pack_bf2(p[2 * kb + 1][2], p[2 * kb + 1][3]);
        pf[qb][kb] = __builtin_bit_cast(bf16x8_t, pk4);
      }
    }
    // ---- O^T += V^T . P^T ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int p = 0; p < DB; ++p) {   // fragment pairs as above: an accumulator still sees token half 0 before half 1
      const int db0 = 2 * (p >> 1), kb = p & 1;
      if constexpr (DB == 8) hook(8 + p);
#pragma unroll
      for (int qb = 0; qb < QBLK; ++qb)
        acc_o[qb][db0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[p & 1][1], pf[qb][kb], acc_o[qb][db0 + 1], 0, 0, 0);
#pragma unroll
      for (int qb = 0; qb < QBLK; ++qb)
        acc_o[qb][db0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[p & 1][0], pf[qb][kb], acc_o[qb][db0], 0, 0, 0);
      if (p + 2 < DB) {
        vr[p & 1][0] = load_vf(2 * ((p + 2) >> 1), (p + 2) & 1);
        vr[p & 1][1] = load_vf(2 * ((p + 2) >> 1) + 1, (p + 2) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if constexpr (DMA) {
    // ---- LDS-DMA staging (head_dim 128): K and V tiles go global -> LDS with global_load_lds_dwordx4, no staging
    //      registers, no ds_write pass, no software transpose.  Two-deep ring per operand; one barrier per KV tile.
    //      K image: [64 tok][16 chunks of 16 B], chunk ^= row & 15 (as above).  V image: ROW-major [64 tok][128 dims],
    //      32-byte chunk index ^= tok & 7, and the V^T fragments are gathered by ds_read_b64_tr_b16: inside a 16-lane
    //      group lane 4 j + c supplies the address of 4 consecutive dims (chunk c) of token row j and lane i receives
    //      dim i of the four rows (measured semantics, tools/probes/tr16_probe.hip) - exactly the (4 tokens of lane
    //      group g) x (dim l15) block the swapped product's P^T fragment pairs with.  Both swizzles are applied through
    //      the per-lane SOURCE address, the DMA writes LDS lane-linearly.
    static_assert(!DMA || (D == 128 && TKV == 64), "LDS-DMA staging is laid out for head_dim 128");
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_vptr_t;
    u32x4* kring = ks;                                   // 2 x 1024 chunks
    u32x4* vring = reinterpret_cast<u32x4*>(vt);         // 2 x 1024 chunks
    const uint32_t k_lds = (uint32_t)(uintptr_t)(lds_vptr_t)kring, v_lds = (uint32_t)(uintptr_t)(lds_vptr_t)vring;
    constexpr int PPW = 16 / NW;                         // 1-KiB pieces (4 token rows) per wave, operand and tile
    const int t_last = kv_len - 1;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    // A piece = 4 consecutive token rows x 256 B.  For full tiles every row address is (page id) x stride + a per-lane
    // constant: kv0 is a multiple of 64 and the page size a power of two <= 64 (else the general path), so a row's
    // position inside its page and its page's distance from the tile's first page do not depend on the tile.  That
    // leaves one page-id read (LDS) and three 64-bit adds per piece instead of ~15 integer instructions per row.
    const bool fast_pages = addr.page_size <= TKV;
    int pg_step[PPW], k_lane[PPW], v_lane[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int r = (wave_s * PPW + j) * 4 + (lane >> 4), c = lane & 15;
      const int in_page = r & (addr.page_size - 1);
      pg_step[j] = r >> addr.shift;
      k_lane[j] = in_page * addr.row_stride + ((c ^ (r & 15)) << 3);
      v_lane[j] = in_page * addr.row_stride + ((c ^ ((r & 7) << 1)) << 3);
    }
    auto issue_tile = [&](int kv0, int buf) {
      const bool fast = fast_pages && kv0 + TKV <= kv_len;   // workgroup-uniform
      const int pi0 = (kv0 >> addr.shift) - addr.win0;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int piece = wave_s * PPW + j;
        const Half* ksrc;
        const Half* vsrc;
        if (fast) {
          const long pbase = (long)addr.pg[pi0 + pg_step[j]] * (long)addr.stride_page;
          ksrc = kbuf + pbase + k_lane[j];
          vsrc = vbuf + pbase + v_lane[j];
        } else {
          const int r = piece * 4 + (lane >> 4), c = lane & 15;
          int t = kv0 + r;
          t = t < t_last ? t : t_last;
          const long ro = addr.row(t);
          ksrc = kbuf + ro + ((c ^ (r & 15)) << 3);
          vsrc = vbuf + ro + ((c ^ ((r & 7) << 1)) << 3);
        }
        const uint32_t kd = k_lds + (uint32_t)(buf * 16384 + piece * 1024);
        const uint32_t vd = v_lds + (uint32_t)(buf * 16384 + piece * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(ksrc), "s"(kd) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(vsrc), "s"(vd) : "memory", "m0");
      }
    };
    // per-lane byte offsets of the V^T gathers inside a V tile: token row 4 g + (l15 >> 2) of a 16-token block, chunk l15 & 3
    const int jq = l15 >> 2, cq = l15 & 3, xs = (4 * g + jq) & 7;
    uint32_t voff[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
      voff[db] = (uint32_t)((4 * g + jq) * 256 + ((((db ^ xs) << 1) | (cq >> 1)) << 4) + (cq & 1) * 8);
    const __attribute__((address_space(3))) char* vbase = (const __attribute__((address_space(3))) char*)vring;
    if (cta_kv_end > 0) {
      __syncthreads();  // page-id window staged
      issue_tile(0, 0);
    }
    // A wave gets one DMA instruction through its issue slot per ~130 cycles (tools/probes/ingest_probe: 18 GB/s per issuing
    // wave) and nothing behind it in program order issues meanwhile: eight requests in a row right after the barrier kept the
    // wave off its MFMAs for ~900 cycles per tile (knock-out "DMAs for the first two tiles only": -18 %).  For full tiles the
    // next tile's requests are therefore spread through THIS tile's arithmetic, one in front of every second MFMA cluster
    // (K pieces in the S loop, V pieces in the PV loop); the page bases are wave-uniform (a piece = 4 token rows, pages >= 4
    // rows): their page ids wait in SGPRs.  Waves without arithmetic for the tile, and the ragged last tile, issue theirs in one go.
    const bool inline_ok = fast_pages && addr.page_size >= 4 && addr.dma_inline;
    int it = 0;
    for (int kv0 = 0; kv0 < cta_kv_end; kv0 += TKV, ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile `it` have landed
      __syncthreads();                                   // everyone's have; everyone is done with tile it - 1's buffers
      bool inl = false;
      int pgid[PPW];   // page ids of this wave's pieces of the next tile (SGPRs)
      const int nb = (it + 1) & 1;
      if (kv0 + TKV < cta_kv_end) {
        const int next_end = kv0 + 2 * TKV < kv_len ? kv0 + 2 * TKV : kv_len;
        if (!addr.covers(next_end)) {  // workgroup-uniform, once per kPgLdsMax pages
          addr.load_window(kv0 + TKV);
          __syncthreads();
        }
        inl = inline_ok && kv0 + 2 * TKV <= kv_len;
        if (inl) {
          const int pi0 = ((kv0 + TKV) >> addr.shift) - addr.win0;
#pragma unroll
          for (int j = 0; j < PPW; ++j)
            pgid[j] = __builtin_amdgcn_readfirstlane(addr.pg[pi0 + (((wave_s * PPW + j) * 4) >> addr.shift)]);
        } else {
          issue_tile(kv0 + TKV, nb);  // lands under this tile's MFMAs
        }
      }
      auto issue_piece = [&](int j, bool isv) {
        const Half* src = (isv ? vbuf : kbuf) + (long)pgid[j] * (long)addr.stride_page + (isv ? v_lane[j] : k_lane[j]);
        const uint32_t dst = (isv ? v_lds : k_lds) + (uint32_t)(nb * 16384 + (wave_s * PPW + j) * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
      };
      if (!wave_active || kv0 >= wave_kv_end) {
        if (inl) {
#pragma unroll
          for (int j = 0; j < PPW; ++j) { issue_piece(j, false); issue_piece(j, true); }
        }
        continue;
      }
      const int buf = it & 1;
      const __attribute__((address_space(3))) char* vb = vbase + buf * 16384;
      auto vf_dma = [&](int db, int kb) {
        const v4s_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s_t __attribute__((address_space(3)))*)(vb + voff[db] + (2 * kb) * 4096));
        const v4s_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s_t __attribute__((address_space(3)))*)(vb + voff[db] + (2 * kb + 1) * 4096));
        const u32x2 w0 = __builtin_bit_cast(u32x2, a0), w1 = __builtin_bit_cast(u32x2, a1);
        return __builtin_bit_cast(bf16x8_t, u32x4{w0.x, w0.y, w1.x, w1.y});
      };
      constexpr int kSlotStep = 8 / PPW;   // PPW pieces per operand over the 8 slots of its loop
      compute_tile(kv0, kring + buf * 1024, vf_dma, [&](int slot) {
        if (!inl) return;                  // workgroup-uniform
        if ((slot % kSlotStep) == 0) issue_piece((slot & 7) / kSlotStep, slot >= 8);
      });
    }
  } else {
    if (cta_kv_end > 0) {
      __syncthreads();  // page-id window staged
      load_tile(0);
    }
    for (int kv0 = 0; kv0 < cta_kv_end; kv0 += TKV) {
      __syncthreads();  // previous tile fully consumed
      store_tile();
      __syncthreads();
      if (kv0 + TKV < cta_kv_end) {
        const int next_end = kv0 + 2 * TKV < kv_len ? kv0 + 2 * TKV : kv_len;
        if (!addr.covers(next_end)) {  // workgroup-uniform, once per kPgLdsMax pages
          __syncthreads();
          addr.load_window(kv0 + TKV);
          __syncthreads();
        }
        load_tile(kv0 + TKV);  // prefetch under the MFMAs
      }
      if (!wave_active || kv0 >= wave_kv_end) continue;

      compute_tile(kv0, ks, [&](int db, int kb) {
        const Half* vrow = vt + (db * 16 + l15) * VT_PITCH;
        const int rot = 4 * (db >> 1);  // column rotation of this 32-row group (see store_tile)
        u32x2 a0 = *reinterpret_cast<const u32x2*>(vrow + ((kb * 8 + g + rot) & 15) * 4);
        u32x2 a1 = *reinterpret_cast<const u32x2*>(vrow + ((kb * 8 + g + 4 + rot) & 15) * 4);
        return __builtin_bit_cast(bf16x8_t, u32x4{a0.x, a0.y, a1.x, a1.y});
      }, [](int) {});
    }
  }
  // ---- epilogue: O = acc / l ; lane (row l15, g) holds dims db*16 + g*4 .. +3 ----
#pragma unroll
  for (int qb = 0; qb < QBLK; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16, kWave);
    l += __shfl_xor(l, 32, kWave);
    if (!qok[qb]) continue;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      u32x2 o;
      o.x = pack_bf2(acc_o[qb][db][0] * inv, acc_o[qb][db][1] * inv);
      o.y = pack_bf2(acc_o[qb][db][2] * inv, acc_o[qb][db][3] * inv);
      *reinterpret_cast<u32x2*>(o_base + o_off[qb] + db * 16 + g * 4) = o;
    }
  }
}

// GROUPED = number of consecutive plan tiles one workgroup takes (1 or 2).  The reference model crates plan with
// 64-row tiles (config.rs:5), i.e. 16 packed rows per wave: every K / V^T fragment read from LDS then feeds ONE MFMA
// and a staged KV tile is shared by only 64 rows.  With GROUPED = 2 a workgroup takes plan tiles 2i and 2i + 1; when
// they are adjacent rows of the same request (all but the seams between requests) they run as ONE 2 x cta_tile_q
// tile, 32 rows per wave; otherwise one after the other.  Per-row arithmetic does not depend on the grouping.
constexpr int kPrefillDmaLdsBytes = 4 * 16384 + kPgLdsMax * 4;   // K ring, V ring (2 x 16 KiB each), page-id window
template <int QBLK, int D, bool POW2, int GROUPED = 1, int NW = 4, bool DMA = false>
__global__ __launch_bounds__(NW * 64, D == 128 ? 2 : 1) void batch_prefill_paged_kernel(
    const Half* __restrict__ q, Half* __restrict__ out, const Half* __restrict__ kv, long k_off, long v_off,
    const int* __restrict__ page_indices, const int* __restrict__ page_indptr,
    const int* __restrict__ last_page_len, const int* __restrict__ q_indptr,
    const int* __restrict__ request_indices, const int* __restrict__ qo_tile_indices, int num_qo_heads,
    int num_kv_heads, int page_size, long stride_page, float scale_log2, int cta_tile_q, int page_shift,
    int num_plan_tiles, int xcd_heads) {
  u32x4* ks;
  Half* vt;
  int* lds_pg;
  if constexpr (DMA) {   // LDS-DMA staging: two-deep K / V rings in dynamic LDS (kPrefillDmaLdsBytes)
    extern __shared__ __attribute__((aligned(16))) u32x4 prefill_dyn_lds[];
    ks = prefill_dyn_lds;
    vt = reinterpret_cast<Half*>(prefill_dyn_lds + 2048);
    lds_pg = reinterpret_cast<int*>(prefill_dyn_lds + 4096);
  } else {
    __shared__ __attribute__((aligned(16))) u32x4 ks_s[TKV * D / 8];
    __shared__ __attribute__((aligned(16))) Half vt_s[D * VT_PITCH];
    __shared__ int lds_pg_s[kPgLdsMax];
    ks = ks_s; vt = vt_s; lds_pg = lds_pg_s;
  }
  // longest tiles first: within a request the plan lists tiles by ascending row, i.e. ascending causal KV length;
  // dispatching them in reverse keeps the tail of the launch made of short tiles
  // 1-D grid, kv head fastest: workgroup b runs on XCD b % 8, so with 8 kv heads every XCD keeps ONE head's K / V in its
  // L2 (5 MB at 10 k tokens against a 4 MB L2) instead of all eight streaming through each (xcd_heads = 0 keeps the
  // tile-fastest order for the A/B)
  const int n_slots = gridDim.x / num_kv_heads;
  const int kvh = (xcd_heads & 1) ? blockIdx.x % num_kv_heads : blockIdx.x / n_slots;
  const int slot = (xcd_heads & 1) ? blockIdx.x / num_kv_heads : blockIdx.x % n_slots;
  const int group = num_qo_heads / num_kv_heads;
  const int first = (n_slots - 1 - slot) * GROUPED;
  int req[GROUPED], row0[GROUPED];
#pragma unroll
  for (int k = 0; k < GROUPED; ++k) {
    const int tile = first + k;
    req[k] = tile < num_plan_tiles ? request_indices[tile] : -1;
    row0[k] = tile < num_plan_tiles ? qo_tile_indices[tile] * cta_tile_q : 0;
  }
  // runs of adjacent plan tiles of one request are taken as ONE tile of n x cta_tile_q rows
  for (int k = 0, n = 1; k < GROUPED; k += n) {
    n = 1;
    while (k + n < GROUPED && req[k] >= 0 && req[k + n] == req[k] && row0[k + n] == row0[k] + n * cta_tile_q) ++n;
    const int r = req[k];
    if (r < 0) continue;
    const int q0 = q_indptr[r];
    const int qo_len = q_indptr[r + 1] - q0;
    const int pbase = page_indptr[r];
    const int npages = page_indptr[r + 1] - pbase;
    const int kv_len = npages > 0 ? (npages - 1) * page_size + last_page_len[r] : 0;
    if (row0[k] >= qo_len * group || kv_len <= 0) continue;
    if (GROUPED > 1 && k > 0) __syncthreads();  // the previous tile's LDS (page window, K / V^T) is done with
    PagedAddr<POW2> addr{page_indices, pbase, page_size, (int)stride_page, num_kv_heads * D, page_shift, lds_pg, 0, npages,
                         (xcd_heads & 2) ? 0 : 1};
    const long q_stride_n = (long)num_qo_heads * D;
    const long qo_base = (long)q0 * q_stride_n + (long)kvh * group * D;
    const Half* kvh_base = kv + (long)kvh * D;
    prefill_tile<QBLK, D, NW, PagedAddr<POW2>, DMA>(q + qo_base, out + qo_base, kvh_base + k_off, kvh_base + v_off, addr,
                                                    qo_len, kv_len, row0[k], n * cta_tile_q, group,
                                                    q_stride_n, scale_log2, ks, vt);
  }
}

template <int QBLK>
__global__ __launch_bounds__(256) void single_prefill_kernel(const Half* __restrict__ q, Half* __restrict__ out,
                                                             const Half* __restrict__ k_cache,
                                                             const Half* __restrict__ v_cache, int num_qo_heads,
                                                             int num_kv_heads, int seq_len, int kv_len,
                                                             int max_seq_len, float scale_log2, int cta_tile_q) {
  __shared__ __attribute__((aligned(16))) u32x4 ks[TKV * 16];
  __shared__ __attribute__((aligned(16))) Half vt[128 * VT_PITCH];
  const int kvh = blockIdx.y;
  const int group = num_qo_heads / num_kv_heads;
  const int row0 = (gridDim.x - 1 - blockIdx.x) * cta_tile_q;
  if (row0 >= seq_len * group || kv_len <= 0) return;
  ContigAddr addr{128};
  const long q_stride_n = (long)num_qo_heads * 128;
  const long qo_base = (long)kvh * group * 128;
  const long head_base = (long)kvh * max_seq_len * 128;
  prefill_tile<QBLK, 128, 4>(q + qo_base, out + qo_base, k_cache + head_base, v_cache + head_base, addr, seq_len, kv_len,
                             row0, cta_tile_q, group, q_stride_n, scale_log2, ks, vt);
}

static inline int page_shift_of(int page_size) {  // log2 for powers of two, -1 otherwise (division path)
  if (page_size <= 0 || (page_size & (page_size - 1)) != 0) return -1;
  int s = 0;
  while ((1 << s) < page_size) ++s;
  return s;
}

// ---- host plan helpers: FlashInfer FA2DetermineCtaTileQ restated (utils.cuh, un-vendored) ----
static inline uint32_t fa2_cta_tile_q(int64_t avg_packed_qo_len, int head_dim) {
  if (avg_packed_qo_len > 64 && head_dim < 256) return 128;
  return avg_packed_qo_len > 16 ? 64 : 16;
}
static inline uint32_t resolve_cta_tile_q(int64_t packed, int head_dim, int override_q) {
  if (override_q == 0) return fa2_cta_tile_q(packed, head_dim);
  if (override_q == 16 || override_q == 64 || override_q == 128) return (uint32_t)override_q;
  return 0;
}

}  // namespace pk

using namespace pk;

extern "C" {

int32_t batch_prefill_paged_num_tiles(int32_t seq_len, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim) {
  const int64_t packed = (int64_t)seq_len * (num_qo_heads / num_kv_heads);
  const uint32_t t = fa2_cta_tile_q(packed, head_dim);
  return (int32_t)((packed + t - 1) / t);
}
int32_t batch_prefill_paged_num_tiles_with_cta_tile_q(int32_t seq_len, int32_t num_qo_heads, int32_t num_kv_heads,
                                                      int32_t head_dim, int32_t cta_tile_q_override) {
  const int64_t packed = (int64_t)seq_len * (num_qo_heads / num_kv_heads);
  const uint32_t t = resolve_cta_tile_q(packed, head_dim, cta_tile_q_override);
  if (t == 0) return -1;
  return (int32_t)((packed + t - 1) / t);
}
int32_t batch_prefill_cta_tile_q(int32_t total_seq_len, int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim) {
  return (int32_t)fa2_cta_tile_q((int64_t)total_seq_len * (num_qo_heads / num_kv_heads), head_dim);
}
int32_t batch_prefill_cta_tile_q_with_override(int32_t total_seq_len, int32_t num_qo_heads, int32_t num_kv_heads,
                                               int32_t head_dim, int32_t cta_tile_q_override) {
  return (int32_t)resolve_cta_tile_q((int64_t)total_seq_len * (num_qo_heads / num_kv_heads), head_dim,
                                     cta_tile_q_override);
}

int32_t batch_prefill_paged_cuda_with_cta_tile_q(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* q_indptr, const int32_t* request_indices, const int32_t* qo_tile_indices,
    const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const uint32_t* total_num_rows,
    int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t seq_len,
    int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale,
    int32_t cta_tile_q_override, pegainfer_stream_t stream) {
  (void)kv_tile_indices; (void)kv_chunk_size_ptr; (void)total_num_rows; (void)batch_size;
  if (head_dim != 128 || num_kv_heads <= 0 || num_qo_heads % num_kv_heads != 0)
    return (int32_t)hipErrorInvalidValue;
  const int group = num_qo_heads / num_kv_heads;
  const uint32_t cta = resolve_cta_tile_q((int64_t)seq_len * group, head_dim, cta_tile_q_override);
  if (cta == 0) return -1;
  if (padded_batch_size <= 0) return 0;
  const float scale_log2 = sm_scale * 1.4426950408889634f;
  hipStream_t s = as_stream(stream);
  if (stride_page <= 0 || stride_page > 0x7fffffffLL) return (int32_t)hipErrorInvalidValue;
  const int shift = page_shift_of(page_size);
  // two 64-row plan tiles per workgroup once that still leaves >= 2 workgroups per CU (PEGAINFER_PREFILL_GROUP = 1 | 2
  // forces either form; per-row results are identical)
  static const int group_env = [] { const char* e = getenv("PEGAINFER_PREFILL_GROUP"); return e && *e ? atoi(e) : 0; }();
  static const int xcd_heads = [] {   // bit 0: kv head fastest in the grid; bit 1: burst DMA issue (both A/B switches)
    const char* e = getenv("PEGAINFER_PREFILL_XCD_HEADS");
    const char* d = getenv("PEGAINFER_PREFILL_DMA_INLINE");
    return ((e && *e == '0') ? 0 : 1) | ((d && *d == '0') ? 2 : 0);
  }();
  const bool pair = cta == 64 && (group_env == 2 || (group_env == 0 && (long)padded_batch_size * num_kv_heads >= 1024));
#define PK_PREFILL(QB, P2, G, CTA)                                                                           \
  batch_prefill_paged_kernel<QB, 128, P2, G><<<((padded_batch_size + G - 1) / G) * num_kv_heads, 256, 0, s>>>( \
      q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d, q_indptr, \
      request_indices, qo_tile_indices, num_qo_heads, num_kv_heads, page_size, stride_page, scale_log2, CTA, shift, \
      padded_batch_size, xcd_heads)
  // LDS-DMA staging + ds_read_b64_tr_b16 form of the 32-rows-per-wave tiles (power-of-two page sizes;
  // PEGAINFER_PREFILL_DMA=0 keeps the register-staged form; per-row arithmetic is identical)
  static const bool dma64_on = [] { const char* e = getenv("PEGAINFER_PREFILL_DMA64"); return !(e && *e == '0'); }();
  static const bool dma_on = [] { const char* e = getenv("PEGAINFER_PREFILL_DMA"); return !(e && *e == '0'); }();
#define PK_PREFILL_DMA(QB, G, CTA)                                                                                 \
  do {                                                                                                          \
    static const bool once = [] {                                                                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&batch_prefill_paged_kernel<QB, 128, true, G, 4, true>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kPrefillDmaLdsBytes);              \
      return true;                                                                                              \
    }();                                                                                                        \
    (void)once;                                                                                                 \
    batch_prefill_paged_kernel<QB, 128, true, G, 4, true>                                                       \
        <<<((padded_batch_size + G - 1) / G) * num_kv_heads, 256, kPrefillDmaLdsBytes, s>>>(                    \
            q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d, q_indptr, \
            request_indices, qo_tile_indices, num_qo_heads, num_kv_heads, page_size, stride_page, scale_log2, CTA, shift, \
            padded_batch_size, xcd_heads);                                                                      \
  } while (0)
  if (dma_on && shift >= 0 && (cta == 128 || pair)) {
    if (cta == 128) PK_PREFILL_DMA(2, 1, 128);
    else PK_PREFILL_DMA(2, 2, 64);
  } else if (dma_on && dma64_on && shift >= 0 && cta == 64) {
    PK_PREFILL_DMA(1, 1, 64);   // 64-row tiles, 16 rows per wave (prompts of ~1 k tokens: too few tiles to pair)
  } else if (cta == 128) {
    if (shift >= 0) PK_PREFILL(2, true, 1, 128);
    else PK_PREFILL(2, false, 1, 128);
  } else if (pair) {
    if (shift >= 0) PK_PREFILL(2, true, 2, 64);
    else PK_PREFILL(2, false, 2, 64);
  } else {
    if (shift >= 0) PK_PREFILL(1, true, 1, (int)cta);
    else PK_PREFILL(1, false, 1, (int)cta);
  }
#undef PK_PREFILL
#undef PK_PREFILL_DMA
  return (int32_t)hipGetLastError();
}

int32_t batch_prefill_paged_cuda(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* q_indptr, const int32_t* request_indices, const int32_t* qo_tile_indices,
    const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const uint32_t* total_num_rows,
    int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t seq_len,
    int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale,
    pegainfer_stream_t stream) {
  return batch_prefill_paged_cuda_with_cta_tile_q(q, output, kv_data, k_offset_elems, v_offset_elems,
                                                  page_indices, page_indptr, last_page_len_d, q_indptr,
                                                  request_indices, qo_tile_indices, kv_tile_indices,
                                                  kv_chunk_size_ptr, total_num_rows, num_qo_heads, num_kv_heads,
                                                  head_dim, page_size, seq_len, batch_size, padded_batch_size,
                                                  stride_page, sm_scale, 0, stream);
}

// HEAD_DIM = 256 instantiation for the Qwen3.5 full-attention layers (ffi.rs:1309-1334): tile 64 / 16 only
// (FA2DetermineCtaTileQ never picks 128 at head_dim 256).
int32_t batch_prefill_paged_cuda_hd256(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* q_indptr, const int32_t* request_indices, const int32_t* qo_tile_indices,
    const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr, const uint32_t* total_num_rows,
    int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t seq_len,
    int32_t batch_size, int32_t padded_batch_size, int64_t stride_page, float sm_scale,
    pegainfer_stream_t stream) {
  (void)kv_tile_indices; (void)kv_chunk_size_ptr; (void)total_num_rows; (void)batch_size;
  if (head_dim != 256 || num_kv_heads <= 0 || num_qo_heads % num_kv_heads != 0)
    return (int32_t)hipErrorInvalidValue;
  const int group = num_qo_heads / num_kv_heads;
  const uint32_t cta = fa2_cta_tile_q((int64_t)seq_len * group, head_dim);
  if (padded_batch_size <= 0) return 0;
  const int grid = padded_batch_size * num_kv_heads;
  if (stride_page <= 0 || stride_page > 0x7fffffffLL) return (int32_t)hipErrorInvalidValue;
  const int shift = page_shift_of(page_size);
  const float scale_log2 = sm_scale * 1.4426950408889634f;
  if (shift >= 0)
    batch_prefill_paged_kernel<1, 256, true><<<grid, 256, 0, as_stream(stream)>>>(
        q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d, q_indptr,
        request_indices, qo_tile_indices, num_qo_heads, num_kv_heads, page_size, stride_page, scale_log2, (int)cta, shift,
        padded_batch_size, 1);
  else
    batch_prefill_paged_kernel<1, 256, false><<<grid, 256, 0, as_stream(stream)>>>(
        q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d, q_indptr,
        request_indices, qo_tile_indices, num_qo_heads, num_kv_heads, page_size, stride_page, scale_log2, (int)cta, shift,
        padded_batch_size, 1);
  return (int32_t)hipGetLastError();
}

int32_t single_prefill_cuda(const Half* q, Half* output, const Half* k_cache, const Half* v_cache,
                            int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t seq_len,
                            int32_t kv_len, int32_t max_seq_len, float sm_scale, pegainfer_stream_t stream) {
  if (head_dim != 128 || num_kv_heads <= 0 || num_qo_heads % num_kv_heads != 0)
    return (int32_t)hipErrorInvalidValue;
  if (seq_len <= 0) return 0;
  const int group = num_qo_heads / num_kv_heads;
  const int packed = seq_len * group;
  const float scale_log2 = sm_scale * 1.4426950408889634f;
  dim3 grid(ceil_div(packed, 128), num_kv_heads);
  single_prefill_kernel<2><<<grid, 256, 0, as_stream(stream)>>>(q, output, k_cache, v_cache, num_qo_heads,
                                                                num_kv_heads, seq_len, kv_len, max_seq_len,
                                                                scale_log2, 128);
  return (int32_t)hipGetLastError();
}

}  // extern "C"
