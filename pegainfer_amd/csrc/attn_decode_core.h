// Shared arithmetic core of the paged GQA decode attention kernels of attn_decode.hip (stand-alone, fused with qk-norm +
// RoPE + KV append, fused with the o_proj).  Everything that determines a result bit lives here exactly once: the per-wave
// KV scan with its per-lane-row online-softmax states (AttnScan), the in-workgroup merge of those states + the partial /
// output store (attn_finish_part) and the partition-KV merge (merge_one).  A caller supplies the waves (4- or 8-wave
// workgroups), so every form produces the same bits for the same (request, KV chunk, kv head).
#pragma once

#include <type_traits>

#include "common.h"
#include "rope_core.h"

namespace pk {

template <int LPT>
__device__ __forceinline__ float token_sum(float v) {
  v = row16_sum(v);
  if (LPT == 32) v += __shfl_xor(v, 16, kWave);
  return v;
}

struct DecodeAttnArgs {
  const Half* q; Half* o_out; const Half* kv; long k_off, v_off;
  const int* page_indices; const int* page_indptr; const int* last_page_len; const int* request_indices;
  const int* kv_tile_indices; const int* kv_chunk_size_ptr; const uint8_t* block_valid_mask;
  Half* tmp_v; float* tmp_s; int num_qo_heads, num_kv_heads, page_size; long stride_page; float scale_log2;
  // fused form only: raw qkv rows [bs, (Hq + 2 Hkv) * 128], per-head norm weights, RoPE tables, positions
  const Half* qkv; const Half* q_norm_w; const Half* k_norm_w; const Half* cos_cache; const Half* sin_cache;
  const int* positions; float eps;
  // fused form, optional: one 32-byte record per slot {b, lo, hi, pbase, pos, kv_len, s0, s1} built by the host
  // (lo < 0 = padding slot; s0, s1 = o_indptr[b], o_indptr[b + 1]) - replaces a 4-deep chain of dependent
  // metadata loads by one load
  const int* slot_desc;
  // partition form, optional: when merge_counters is non-null the LAST workgroup of a (request, kv head) to
  // finish merges that head group's partials itself (no merge_states_kernel launch).  One int per
  // (request, kv head), zero before the first launch; the merging workgroup leaves it zero again.
  // The counters are kMergeCtrStride ints apart (a cache line each): at bs 1 the 8 kv heads' counters shared ONE
  // line and their 8 x 17 arrivals serialised on it (the 1.0-1.1 us "ticket" phase of the in-kernel stamps).
  int* merge_counters; const int* o_indptr;
  // fused attention + o_proj launch only (attn_oproj_kernel): the merging workgroup writes its head group's rows
  // write-through and then adds 1 here - the o_proj phase of every workgroup waits for num_kv_heads arrivals per request
  int* done_ctr;
  // 0: every merging workgroup arrives on done_ctr[0] (the o_proj phase waits for num_kv_heads arrivals);
  // kMergeCtrStride: head group kvh arrives on done_ctr[kvh * kMergeCtrStride] - a cache line per group - and every o_proj
  // WAVE waits only for the groups whose K blocks it holds (round 5)
  int done_stride;
  // debug: 8 wall-clock stamps (100 MHz) per (slot, kv head) written by thread 0 (pegainfer_debug_attn_trace)
  unsigned long long* trace;
};
#define PK_ATTN_STAMP(a, slot, kvh, i)                                                                         \
  do {                                                                                                         \
    if ((a).trace && threadIdx.x == 0)                                                                         \
      (a).trace[((size_t)(slot) * (a).num_kv_heads + (kvh)) * 8 + (i)] = wall_clock64();                       \
  } while (0)

constexpr int kMergeCtrStride = 32;   // ints between the merge counters of two (request, kv head) pairs

struct ChunkInfo { int b, pbase, kv_len, lo, hi; };

// One wave merges the partition-KV partials of one (request, q head): lanes first fetch all log2-sum-exps of
// the request's slots in parallel (<= 64 slots), then every lane accumulates its D/64 output dims over the
// slots with the weights broadcast from registers:  out = sum_s 2^(lse_s - M) v_s / sum_s 2^(lse_s - M).
// Shared by merge_states_kernel and the in-kernel merge so both round identically.
// COHERENT: the partials were published write-through by other workgroups of this launch -> agent-scope relaxed
// atomic loads (global_load ... sc1), which are served past this CU's L1.  COHERENT_OUT: the merged row is itself
// handed to other workgroups inside the launch (the engine's o_proj phase) -> write-through stores.
template <int D, bool COHERENT, bool COHERENT_OUT = false>
__device__ __forceinline__ void merge_one(const Half* __restrict__ tmp_v, const float* __restrict__ tmp_s, int s0,
                                          int s1, int head, int num_qo_heads, Half* __restrict__ dst_row) {
  const int lane = threadIdx.x & 63;
  constexpr int EPL = D / 64;  // elements per lane (2 or 4)
  typedef typename std::conditional<EPL == 2, uint32_t, uint64_t>::type word_t;
  float acc[EPL], wsum = 0.f;
#pragma unroll
  for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
  // both split plans cap a request at 64 chunks (batch_decode_buffers.rs:15); launch_decode rejects more
  const int n = s1 - s0 < 64 ? s1 - s0 : 64;
  float lse = -INFINITY;
  if (lane < n) {
    const float* ps = tmp_s + (size_t)(s0 + lane) * num_qo_heads + head;
    if (COHERENT)
      lse = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(ps), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT));
    else
      lse = *ps;
  }
  const float M = wave_max(lse);
  if (M != -INFINITY) {
    const float w_lane = exp2f(lse - M);
    // 16 partials per batch: all loads of a batch are in flight before the first one is consumed (a plain
    // load-then-accumulate loop paid one memory round trip per slot); accumulation order is still slot order
    constexpr int MB = 16;
    for (int j0 = 0; j0 < n; j0 += MB) {
      word_t pv[MB];
#pragma unroll
      for (int u = 0; u < MB; ++u) {
        int j = j0 + u;
        j = j < n ? j : n - 1;  // clamped reload of a valid slot; its weight is dropped below
        const word_t* v = reinterpret_cast<const word_t*>(tmp_v + ((size_t)(s0 + j) * num_qo_heads + head) * D + lane * EPL);
        pv[u] = COHERENT ? __hip_atomic_load(v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *v;
      }
#pragma unroll
      for (int u = 0; u < MB; ++u) {
        if (j0 + u >= n) break;
        const float w = __shfl(w_lane, j0 + u, kWave);
        wsum += w;
        acc[0] = fmaf(w, bf_lo((uint32_t)pv[u]), acc[0]);
        acc[1] = fmaf(w, bf_hi((uint32_t)pv[u]), acc[1]);
        if (EPL == 4) {
          const uint32_t hi = (uint32_t)((uint64_t)pv[u] >> 32);
          acc[2] = fmaf(w, bf_lo(hi), acc[2]);
          acc[3] = fmaf(w, bf_hi(hi), acc[3]);
        }
      }
    }
  }
  Half* dst = dst_row + lane * EPL;
  if (COHERENT_OUT) {
    uint32_t w0 = pack_bf2(wsum > 0.f ? acc[0] / wsum : 0.f, wsum > 0.f ? acc[1] / wsum : 0.f);
    __hip_atomic_store(reinterpret_cast<uint32_t*>(dst), w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (EPL == 4) {
      uint32_t w1 = pack_bf2(wsum > 0.f ? acc[2] / wsum : 0.f, wsum > 0.f ? acc[3] / wsum : 0.f);
      __hip_atomic_store(reinterpret_cast<uint32_t*>(dst) + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
#pragma unroll
    for (int i = 0; i < EPL; ++i) dst[i] = f2bf(wsum > 0.f ? acc[i] / wsum : 0.f);
  }
}

// The KV scan of ONE wave (real or virtual wave `wave` of `NW`) over the chunk [lo, hi): GROUP query heads share every
// loaded K / V row.  A K row (D bf16) is spread over LPT = D/8 lanes with one 16-byte load each, a load instruction
// fetches TPI = 64/LPT complete rows, U of them are in flight per operand.  Each lane row (grp) keeps its own
// online-softmax state over the tokens it sees - tokens t0 + u*TPI + grp of the tiles t0 = align(lo) + wave*TB + k*NW*TB.
template <int D, int GROUP>
struct AttnScan {
  static constexpr int LPT = D / 8;     // lanes per token row
  static constexpr int TPI = 64 / LPT;  // token rows per load instruction
  static constexpr int U = 4;           // load instructions in flight per operand (U = 8 measured 25-45 % slower: 256 VGPRs)
  static constexpr int TB = TPI * U;    // tokens per wave iteration
  float m[GROUP], l[GROUP], o[GROUP][8];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int h = 0; h < GROUP; ++h) {
      m[h] = -INFINITY;
      l[h] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[h][i] = 0.f;
    }
  }

  __device__ __forceinline__ void load_tile(const DecodeAttnArgs& a, const ChunkInfo& ci, int kvh, int lane, int t0,
                                            u32x4 (&kx)[U], u32x4 (&vx)[U], bool (&ok)[U]) const {
    const int sub = lane % LPT, grp = lane / LPT;
    const long head_off = (long)kvh * D + sub * 8;
    const long row_stride = (long)a.num_kv_heads * D;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * TPI + grp;
      ok[u] = t >= ci.lo && t < ci.hi;
      const int tc = ok[u] ? t : ci.lo;  // clamp to a valid token of this chunk (hi > lo here)
      const int page = a.page_indices[ci.pbase + tc / a.page_size];
      const long base = (long)page * a.stride_page + (long)(tc % a.page_size) * row_stride + head_off;
      kx[u] = *reinterpret_cast<const u32x4*>(a.kv + base + a.k_off);
      vx[u] = *reinterpret_cast<const u32x4*>(a.kv + base + a.v_off);
    }
  }

  __device__ __forceinline__ void compute_tile(const u32x4 (&qv)[GROUP], const u32x4 (&kx)[U], const u32x4 (&vx)[U],
                                               const bool (&ok)[U], float scale_log2) {
    float s[GROUP][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int h = 0; h < GROUP; ++h) {
        const float d = __fmul_rn(token_sum<LPT>(dot8(qv[h], kx[u], 0.f)), scale_log2);
        s[h][u] = ok[u] ? d : -INFINITY;
      }
#pragma unroll
    for (int h = 0; h < GROUP; ++h) {
      float mn = m[h];
#pragma unroll
      for (int u = 0; u < U; ++u) mn = fmaxf(mn, s[h][u]);
      if (mn == -INFINITY) continue;  // nothing seen yet by this lane row (uniform per row)
      const float sc = exp2f(m[h] - mn);  // m = -inf -> 0
      float p[U], ps = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        p[u] = exp2f(s[h][u] - mn);  // masked -> 0
        ps += p[u];
      }
      // Every multiply-add below names its fused form: with -ffp-contract=fast a sum of two products
      // (o * sc + p * v) may be contracted either way, and the choice is allowed to differ between the callers this
      // core is inlined into - which would break their bit-equality.
      m[h] = mn;
      l[h] = fmaf(l[h], sc, ps);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[h][i] = __fmul_rn(o[h][i], sc);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t w[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[h][2 * j] = fmaf(p[u], bf_lo(w[j]), o[h][2 * j]);
          o[h][2 * j + 1] = fmaf(p[u], bf_hi(w[j]), o[h][2 * j + 1]);
        }
      }
    }
  }

  // software-pipelined scan: the next tile's 2*U loads are in flight while the current one is reduced
  __device__ __forceinline__ void scan(const DecodeAttnArgs& a, const ChunkInfo& ci, const u32x4 (&qv)[GROUP], int kvh,
                                       int lane, int wave, int NW) {
    const int lo = ci.lo, hi = ci.hi;
    int t0 = (lo / TB) * TB + wave * TB;
    if (lo < hi && t0 < hi) {
      u32x4 kA[U], vA[U], kB[U], vB[U];
      bool okA[U], okB[U];
      load_tile(a, ci, kvh, lane, t0, kA, vA, okA);
      for (;;) {
        int t1 = t0 + NW * TB;
        bool more = t1 < hi;
        if (more) load_tile(a, ci, kvh, lane, t1, kB, vB, okB);
        compute_tile(qv, kA, vA, okA, a.scale_log2);
        if (!more) break;
        t0 = t1 + NW * TB;
        more = t0 < hi;
        if (more) load_tile(a, ci, kvh, lane, t0, kA, vA, okA);
        compute_tile(qv, kB, vB, okB, a.scale_log2);
        if (!more) break;
      }
    }
  }

  // this lane row's state -> the workgroup's partial-state arrays (part = wave * TPI + grp)
  __device__ __forceinline__ void store_state(float* sm_m, float* sm_l, float* sm_o, int part, int lane) const {
    const int sub = lane % LPT;
#pragma unroll
    for (int h = 0; h < GROUP; ++h) {
      if (sub == 0) { sm_m[part * GROUP + h] = m[h]; sm_l[part * GROUP + h] = l[h]; }
      f32x4 x = {o[h][0], o[h][1], o[h][2], o[h][3]}, c = {o[h][4], o[h][5], o[h][6], o[h][7]};
      *reinterpret_cast<f32x4*>(&sm_o[((size_t)part * GROUP + h) * D + sub * 8]) = x;
      *reinterpret_cast<f32x4*>(&sm_o[((size_t)part * GROUP + h) * D + sub * 8 + 4]) = c;
    }
  }
};

// Stage 1 of the in-workgroup merge (round 6).  Until round 5 ONE wave merged all NW x TPI lane-row states: two dependent
// loops of 32 LDS round trips (the running maximum, then the weighted sums) while seven waves waited - most of the 2.4 us
// "publish" phase of the in-kernel stamps.  Now every wave first folds ITS OWN TPI lane-row states of element e = (head, 8
// output dims) into the first of its slots (p = q * TPI), un-normalised - M = max m_p, L = sum l_p 2^(m_p - M),
// O = sum o_p 2^(m_p - M) - and the final merge walks NW entries.  No barrier in front of it: a wave reads only slots its own
// lanes wrote (LDS serves one wave's instructions in order).  Same formulas, another association: new bits for every decode
// attention form at once (they all come through here), fused == unfused and batch == single untouched.
template <int D, int GROUP, int TPI>
__device__ __forceinline__ void attn_fold_wave(float* sm_m, float* sm_l, float* sm_o, int q, int e) {
  const int h = e / (D / 8), d0 = (e - h * (D / 8)) * 8;
  const int p0 = q * TPI;
  float mp[TPI], M = -INFINITY;
#pragma unroll
  for (int i = 0; i < TPI; ++i) {
    mp[i] = sm_m[(p0 + i) * GROUP + h];
    M = fmaxf(M, mp[i]);
  }
  float L = 0.f, O[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) O[i] = 0.f;
  if (M != -INFINITY) {
#pragma unroll
    for (int i = 0; i < TPI; ++i) {
      const float w = exp2f(mp[i] - M);
      L = fmaf(sm_l[(p0 + i) * GROUP + h], w, L);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&sm_o[((size_t)(p0 + i) * GROUP + h) * D + d0]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&sm_o[((size_t)(p0 + i) * GROUP + h) * D + d0 + 4]);
      O[0] = fmaf(x0[0], w, O[0]); O[1] = fmaf(x0[1], w, O[1]); O[2] = fmaf(x0[2], w, O[2]); O[3] = fmaf(x0[3], w, O[3]);
      O[4] = fmaf(x1[0], w, O[4]); O[5] = fmaf(x1[1], w, O[5]); O[6] = fmaf(x1[2], w, O[6]); O[7] = fmaf(x1[3], w, O[7]);
    }
  }
  asm volatile("" ::: "memory");   // every read of the wave's slots is issued before the fold is written over slot p0
  *reinterpret_cast<f32x4*>(&sm_o[((size_t)p0 * GROUP + h) * D + d0]) = f32x4{O[0], O[1], O[2], O[3]};
  *reinterpret_cast<f32x4*>(&sm_o[((size_t)p0 * GROUP + h) * D + d0 + 4]) = f32x4{O[4], O[5], O[6], O[7]};
  if (d0 == 0) { sm_m[p0 * GROUP + h] = M; sm_l[p0 * GROUP + h] = L; }
}

// Merge of the workgroup's partial states (entries p * pstride, p < npart) for element e = (head h, 8 output dims d0) and the store: one thread
// per e.  PARTITION writes a normalised bf16 partial + fp32 log2-sum-exp to tmp_v / tmp_s (write-through when
// `publish`: they are read by another workgroup later in this launch); otherwise the output row of request b
// (`coherent_out`: also write-through - the engine hands it to the o_proj phase of other workgroups).
template <int D, int GROUP, bool PARTITION>
__device__ __forceinline__ void attn_finish_part(const DecodeAttnArgs& a, int b, int slot, int kvh, int e, int npart,
                                                 const float* sm_m, const float* sm_l, const float* sm_o, bool publish,
                                                 bool coherent_out = false, int pstride = 1) {
  const int h = e / (D / 8), d0 = (e - h * (D / 8)) * 8;
  float M = -INFINITY;
  for (int pi = 0; pi < npart; ++pi) M = fmaxf(M, sm_m[pi * pstride * GROUP + h]);
  float L = 0.f, O[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) O[i] = 0.f;
  if (M != -INFINITY) {
    for (int pi = 0; pi < npart; ++pi) {
      const int p = pi * pstride;
      const float w = exp2f(sm_m[p * GROUP + h] - M);
      L = fmaf(sm_l[p * GROUP + h], w, L);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&sm_o[((size_t)p * GROUP + h) * D + d0]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&sm_o[((size_t)p * GROUP + h) * D + d0 + 4]);
      O[0] = fmaf(x0[0], w, O[0]); O[1] = fmaf(x0[1], w, O[1]); O[2] = fmaf(x0[2], w, O[2]); O[3] = fmaf(x0[3], w, O[3]);
      O[4] = fmaf(x1[0], w, O[4]); O[5] = fmaf(x1[1], w, O[5]); O[6] = fmaf(x1[2], w, O[6]); O[7] = fmaf(x1[3], w, O[7]);
    }
  }
  u32x4 pk;
  pk.x = pack_bf2(L > 0.f ? O[0] / L : 0.f, L > 0.f ? O[1] / L : 0.f);
  pk.y = pack_bf2(L > 0.f ? O[2] / L : 0.f, L > 0.f ? O[3] / L : 0.f);
  pk.z = pack_bf2(L > 0.f ? O[4] / L : 0.f, L > 0.f ? O[5] / L : 0.f);
  pk.w = pack_bf2(L > 0.f ? O[6] / L : 0.f, L > 0.f ? O[7] / L : 0.f);
  const int head = kvh * GROUP + h;
  if (PARTITION) {
    Half* pv = a.tmp_v + ((size_t)slot * a.num_qo_heads + head) * D + d0;
    float* ps = a.tmp_s + (size_t)slot * a.num_qo_heads + head;
    const float lse = L > 0.f ? __fadd_rn(M, log2f(L)) : -INFINITY;
    if (publish) {
      asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pv), "v"(pk) : "memory");
      if (d0 == 0) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(ps), "v"(lse) : "memory");
    } else {
      *reinterpret_cast<u32x4*>(pv) = pk;
      if (d0 == 0) *ps = lse;
    }
  } else {
    Half* po = a.o_out + ((size_t)b * a.num_qo_heads + head) * D + d0;
    if (coherent_out) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(po), "v"(pk) : "memory");
    else *reinterpret_cast<u32x4*>(po) = pk;
  }
}

}  // namespace pk
