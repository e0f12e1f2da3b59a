// Paged GQA decode attention for gfx950 (replaces FlashInfer BatchDecodeWithPagedKVCache behind
// paged_attention_decode_cuda / paged_attention_decode_split_kv_cuda, reference
// csrc/paged_attention.cu:77-230).
//
// HBM-bound KV scan.  Grid = (plan slots, kv_heads); a workgroup of 4 waves owns one KV head of one
// (request, KV chunk) and all GROUP query heads that share it, so every K/V byte is read once per
// group, not once per query head.  Inside a wave a K row (head_dim bf16) is spread over
// LPT = head_dim/8 lanes with one 16-byte load each -> a single load instruction fetches
// 64/LPT complete rows, fully coalesced (the page-first NHD cache keeps a head's row contiguous).
// q.k uses v_dot2c_f32_bf16 on the packed bf16 pairs and a DPP butterfly inside the 16-lane row
// (no LDS, no bpermute); each lane row keeps its own online-softmax state (m, l, o[8 dims]) over the
// tokens it sees, so the main loop has no cross-row traffic at all; the 4 x (64/LPT) partial states
// of the workgroup are merged once through LDS.  exp2 with sm_scale*log2(e) folded in, fp32 state.
//
// Partition-KV ("split-K", mandatory on a 256-CU part: bs=1 exposes only kv_heads=8 workgroups
// otherwise) writes a normalised bf16 partial + fp32 log2-sum-exp per (slot, q head) into the
// caller's tmp_v / tmp_s exactly like FlashInfer, and merge_states_kernel combines slots
// o_indptr[b]..o_indptr[b+1].  kv_len always comes from the page table
// ((pages-1)*page_size + last_page_len), kv_chunk_size_ptr[0] is read only when partitioning.
#include <type_traits>

#include "common.h"
#include "pegainfer_kernels_ext.h"
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
  // fused form, optional: one 32-byte record per slot {b, lo, hi, pbase, pos, kv_len, 0, 0} built by the host
  // (lo < 0 = padding slot) - replaces a 4-deep chain of dependent metadata loads by one load
  const int* slot_desc;
  // partition form, optional: when merge_counters is non-null the LAST workgroup of a (request, kv head) to
  // finish merges that head group's partials itself (no merge_states_kernel launch).  One int per
  // (request, kv head), zero before the first launch; the merging workgroup leaves it zero again.
  int* merge_counters; const int* o_indptr;
};

struct ChunkInfo { int b, pbase, kv_len, lo, hi; };

template <bool PARTITION>
__device__ __forceinline__ ChunkInfo decode_chunk(const DecodeAttnArgs& a, int slot) {
  ChunkInfo c;
  c.b = a.request_indices ? a.request_indices[slot] : slot;
  c.pbase = a.page_indptr[c.b];
  const int npages = a.page_indptr[c.b + 1] - c.pbase;
  c.kv_len = npages > 0 ? (npages - 1) * a.page_size + a.last_page_len[c.b] : 0;
  c.lo = 0;
  c.hi = c.kv_len;
  if (PARTITION) {
    const int chunk = a.kv_chunk_size_ptr[0];
    c.lo = a.kv_tile_indices[slot] * chunk;
    c.hi = c.lo + chunk < c.kv_len ? c.lo + chunk : c.kv_len;
    if (c.lo > c.hi) c.lo = c.hi;
  }
  return c;
}

// One wave merges the partition-KV partials of one (request, q head): lanes first fetch all log2-sum-exps of
// the request's slots in parallel (<= 64 slots), then every lane accumulates its D/64 output dims over the
// slots with the weights broadcast from registers:  out = sum_s 2^(lse_s - M) v_s / sum_s 2^(lse_s - M).
// Shared by merge_states_kernel and the in-kernel merge so both round identically.
template <int D, bool COHERENT>
__device__ __forceinline__ void merge_one(const Half* __restrict__ tmp_v, const float* __restrict__ tmp_s, int s0,
                                          int s1, int head, int num_qo_heads, Half* __restrict__ dst_row) {
  // COHERENT: the partials were published write-through by other workgroups of this launch -> agent-scope
  // relaxed atomic loads (global_load ... sc1), which are served past this CU's L1.
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
        acc[0] += w * bf_lo((uint32_t)pv[u]);
        acc[1] += w * bf_hi((uint32_t)pv[u]);
        if (EPL == 4) {
          const uint32_t hi = (uint32_t)((uint64_t)pv[u] >> 32);
          acc[2] += w * bf_lo(hi);
          acc[3] += w * bf_hi(hi);
        }
      }
    }
  }
  Half* dst = dst_row + lane * EPL;
#pragma unroll
  for (int i = 0; i < EPL; ++i) dst[i] = f2bf(wsum > 0.f ? acc[i] / wsum : 0.f);
}

// The KV scan + in-workgroup merge, given the (already normalised / rotated) bf16 q fragments.
template <int D, int GROUP, bool PARTITION, int NW>
__device__ __forceinline__ void decode_attn_body(const DecodeAttnArgs& a, const ChunkInfo& ci, const u32x4 (&qv)[GROUP],
                                                 int slot, int kvh) {
  constexpr int LPT = D / 8;     // lanes per token row
  constexpr int TPI = 64 / LPT;  // token rows per load instruction
  constexpr int U = 4;           // load instructions in flight per operand (U = 8 measured 25-45 % slower: 256 VGPRs)
  constexpr int TB = TPI * U;    // tokens per wave iteration
  constexpr int NPART = NW * TPI; // partial softmax states per workgroup (NW waves)
  __shared__ float sm_m[NPART][GROUP];
  __shared__ float sm_l[NPART][GROUP];
  __shared__ __attribute__((aligned(16))) float sm_o[NPART][GROUP][D];
  const Half* __restrict__ kv = a.kv;
  const int* __restrict__ page_indices = a.page_indices;
  const long k_off = a.k_off, v_off = a.v_off, stride_page = a.stride_page;
  const int page_size = a.page_size, num_qo_heads = a.num_qo_heads, num_kv_heads = a.num_kv_heads;
  const float scale_log2 = a.scale_log2;
  const int b = ci.b, pbase = ci.pbase, lo = ci.lo, hi = ci.hi;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane % LPT, grp = lane / LPT;

  float m[GROUP], l[GROUP], o[GROUP][8];
#pragma unroll
  for (int h = 0; h < GROUP; ++h) {
    m[h] = -INFINITY;
    l[h] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[h][i] = 0.f;
  }
  const long head_off = (long)kvh * D + sub * 8;
  const long row_stride = (long)num_kv_heads * D;

  auto load_tile = [&](int t0, u32x4 (&kx)[U], u32x4 (&vx)[U], bool (&ok)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * TPI + grp;
      ok[u] = t >= lo && t < hi;
      const int tc = ok[u] ? t : lo;  // clamp to a valid token of this chunk (hi > lo here)
      const int page = page_indices[pbase + tc / page_size];
      const long base = (long)page * stride_page + (long)(tc % page_size) * row_stride + head_off;
      kx[u] = *reinterpret_cast<const u32x4*>(kv + base + k_off);
      vx[u] = *reinterpret_cast<const u32x4*>(kv + base + v_off);
    }
  };
  auto compute_tile = [&](const u32x4 (&kx)[U], const u32x4 (&vx)[U], const bool (&ok)[U]) {
    float s[GROUP][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int h = 0; h < GROUP; ++h) {
        const float d = token_sum<LPT>(dot8(qv[h], kx[u], 0.f)) * scale_log2;
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
      m[h] = mn;
      l[h] = l[h] * sc + ps;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[h][i] *= sc;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t w[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[h][2 * j] += p[u] * bf_lo(w[j]);
          o[h][2 * j + 1] += p[u] * bf_hi(w[j]);
        }
      }
    }
  };
  // software-pipelined scan: the next tile's 2*U loads are in flight while the current one is reduced
  {
    int t0 = (lo / TB) * TB + wave * TB;
    if (lo < hi && t0 < hi) {
      u32x4 kA[U], vA[U], kB[U], vB[U];
      bool okA[U], okB[U];
      load_tile(t0, kA, vA, okA);
      for (;;) {
        int t1 = t0 + NW * TB;
        bool more = t1 < hi;
        if (more) load_tile(t1, kB, vB, okB);
        compute_tile(kA, vA, okA);
        if (!more) break;
        t0 = t1 + NW * TB;
        more = t0 < hi;
        if (more) load_tile(t0, kA, vA, okA);
        compute_tile(kB, vB, okB);
        if (!more) break;
      }
    }
  }

  // merge the workgroup's NPART partial states
  const int part = wave * TPI + grp;
#pragma unroll
  for (int h = 0; h < GROUP; ++h) {
    if (sub == 0) { sm_m[part][h] = m[h]; sm_l[part][h] = l[h]; }
    f32x4 a = {o[h][0], o[h][1], o[h][2], o[h][3]}, c = {o[h][4], o[h][5], o[h][6], o[h][7]};
    *reinterpret_cast<f32x4*>(&sm_o[part][h][sub * 8]) = a;
    *reinterpret_cast<f32x4*>(&sm_o[part][h][sub * 8 + 4]) = c;
  }
  __syncthreads();
  // one thread per (head, 8 output dims): 16-byte stores.  With merge_counters the partials are published
  // write-through (sc1): they are read by a workgroup on another XCD later in this same launch.
  const bool publish = PARTITION && a.merge_counters != nullptr;
  for (int e = threadIdx.x; e < GROUP * (D / 8); e += NW * 64) {
    const int h = e / (D / 8), d0 = (e - h * (D / 8)) * 8;
    float M = -INFINITY;
#pragma unroll
    for (int p = 0; p < NPART; ++p) M = fmaxf(M, sm_m[p][h]);
    float L = 0.f, O[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) O[i] = 0.f;
    if (M != -INFINITY) {
#pragma unroll
      for (int p = 0; p < NPART; ++p) {
        const float w = exp2f(sm_m[p][h] - M);
        L += sm_l[p][h] * w;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(&sm_o[p][h][d0]);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&sm_o[p][h][d0 + 4]);
        O[0] += x0[0] * w; O[1] += x0[1] * w; O[2] += x0[2] * w; O[3] += x0[3] * w;
        O[4] += x1[0] * w; O[5] += x1[1] * w; O[6] += x1[2] * w; O[7] += x1[3] * w;
      }
    }
    u32x4 pk;
    pk.x = pack_bf2(L > 0.f ? O[0] / L : 0.f, L > 0.f ? O[1] / L : 0.f);
    pk.y = pack_bf2(L > 0.f ? O[2] / L : 0.f, L > 0.f ? O[3] / L : 0.f);
    pk.z = pack_bf2(L > 0.f ? O[4] / L : 0.f, L > 0.f ? O[5] / L : 0.f);
    pk.w = pack_bf2(L > 0.f ? O[6] / L : 0.f, L > 0.f ? O[7] / L : 0.f);
    const int head = kvh * GROUP + h;
    if (PARTITION) {
      Half* pv = a.tmp_v + ((size_t)slot * num_qo_heads + head) * D + d0;
      float* ps = a.tmp_s + (size_t)slot * num_qo_heads + head;
      const float lse = L > 0.f ? M + log2f(L) : -INFINITY;
      if (publish) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pv), "v"(pk) : "memory");
        if (d0 == 0) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(ps), "v"(lse) : "memory");
      } else {
        *reinterpret_cast<u32x4*>(pv) = pk;
        if (d0 == 0) *ps = lse;
      }
    } else {
      *reinterpret_cast<u32x4*>(a.o_out + ((size_t)b * num_qo_heads + head) * D + d0) = pk;
    }
  }
  if (publish) {
    // "last workgroup done" merge without cache-wide fences (guide: sc1 payload -> vmcnt(0) -> counter; the
    // reader uses sc1 loads): every chunk's partials are write-through, the ticket is a relaxed agent atomic,
    // and the workgroup that draws n-1 merges this head group and re-arms the counter for the next launch.
    __shared__ int sm_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int s0 = a.o_indptr[b], s1 = a.o_indptr[b + 1];
    if (threadIdx.x == 0) {
      int* ctr = a.merge_counters + b * num_kv_heads + kvh;
      const int last = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s1 - s0 - 1;
      if (last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sm_last = last;
    }
    __syncthreads();
    if (sm_last) {
      for (int h = wave; h < GROUP; h += NW) {
        const int head = kvh * GROUP + h;
        merge_one<D, true>(a.tmp_v, a.tmp_s, s0, s1, head, num_qo_heads,
                           a.o_out + ((size_t)b * num_qo_heads + head) * D);
      }
    }
  }
}

// ---- kernels: reference-ABI form (q already normalised + rotated, K/V already in the cache) ----
template <int D, int GROUP, bool PARTITION, int NW>
__global__ __launch_bounds__(NW * 64) void decode_attn_kernel(const DecodeAttnArgs a) {
  const int slot = blockIdx.x, kvh = blockIdx.y;
  if (PARTITION && a.block_valid_mask && !a.block_valid_mask[slot]) return;
  const ChunkInfo ci = decode_chunk<PARTITION>(a, slot);
  const int sub = (threadIdx.x & 63) % (D / 8);
  u32x4 qv[GROUP];
#pragma unroll
  for (int h = 0; h < GROUP; ++h)
    qv[h] = *reinterpret_cast<const u32x4*>(a.q + ((size_t)ci.b * a.num_qo_heads + kvh * GROUP + h) * D + sub * 8);
  decode_attn_body<D, GROUP, PARTITION, NW>(a, ci, qv, slot, kvh);
}

// ---- fused form (head_dim 128): per-head q/k RMSNorm + RoPE and the KV append folded into the prologue.
// Every workgroup normalises + rotates its GROUP query heads from the raw qkv row (cheap, redundant across
// the request's chunks); the one workgroup per (request, kv head) whose chunk contains the new position also
// normalises + rotates the new K row, writes K and V into the page (same bytes paged_kv_scatter_cuda would
// write) and only then scans.  Replaces qk_norm_rope + paged_kv_scatter + decode attention: 3 launches -> 1.
template <int GROUP, bool PARTITION, int NW>
__global__ __launch_bounds__(NW * 64) void fused_decode_attn_kernel(const DecodeAttnArgs a) {
  constexpr int D = 128;
  const int slot = blockIdx.x, kvh = blockIdx.y;
  ChunkInfo ci;
  int pos;
  if (a.slot_desc) {
    const u32x4 d0 = *reinterpret_cast<const u32x4*>(a.slot_desc + 8 * slot);
    const u32x2 d1 = *reinterpret_cast<const u32x2*>(a.slot_desc + 8 * slot + 4);
    ci.b = (int)d0.x; ci.lo = (int)d0.y; ci.hi = (int)d0.z; ci.pbase = (int)d0.w;
    pos = (int)d1.x; ci.kv_len = (int)d1.y;
    if (ci.lo < 0) return;
  } else {
    if (PARTITION && a.block_valid_mask && !a.block_valid_mask[slot]) return;
    ci = decode_chunk<PARTITION>(a, slot);
    pos = a.positions[ci.b];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane & 15, grp = lane >> 4;
  const int q_dim = a.num_qo_heads * D, kv_dim = a.num_kv_heads * D;
  const Half* row = a.qkv + (size_t)ci.b * (q_dim + 2 * kv_dim);
  const Half* crow = a.cos_cache + (size_t)pos * D;
  const Half* srow = a.sin_cache + (size_t)pos * D;
  u32x4 qv[GROUP];
#pragma unroll
  for (int h = 0; h < GROUP; ++h) {
    const u32x4 x = *reinterpret_cast<const u32x4*>(row + (size_t)(kvh * GROUP + h) * D + sub * 8);
    qv[h] = head_norm_rope16(x, a.q_norm_w, crow, srow, sub, a.eps);
  }
  const bool owns_new = pos >= ci.lo && pos < ci.hi;  // workgroup-uniform
  if (owns_new) {
    if (wave == 0 && grp == 0) {
      const u32x4 xk = *reinterpret_cast<const u32x4*>(row + q_dim + (size_t)kvh * D + sub * 8);
      const u32x4 kn = head_norm_rope16(xk, a.k_norm_w, crow, srow, sub, a.eps);
      const u32x4 xv = *reinterpret_cast<const u32x4*>(row + q_dim + kv_dim + (size_t)kvh * D + sub * 8);
      const int page = a.page_indices[ci.pbase + pos / a.page_size];
      const long base = (long)page * a.stride_page + ((long)(pos % a.page_size) * a.num_kv_heads + kvh) * D + sub * 8;
      Half* kvw = const_cast<Half*>(a.kv);
      *reinterpret_cast<u32x4*>(kvw + base + a.k_off) = kn;
      *reinterpret_cast<u32x4*>(kvw + base + a.v_off) = xv;
    }
    __syncthreads();  // workgroup-scope release/acquire: the new row is visible to the scanning waves
  }
  decode_attn_body<D, GROUP, PARTITION, NW>(a, ci, qv, slot, kvh);
}

// merge of the partition-KV partial states: one wave per (request, q head).  Lanes first fetch all
// log2-sum-exps of the request's slots in parallel (<= 64 slots), then every lane accumulates its
// D/64 output dims over the slots with the weights broadcast from registers:
//   out = sum_s 2^(lse_s - M) v_s / sum_s 2^(lse_s - M)
template <int D>
__global__ __launch_bounds__(256) void merge_states_kernel(const Half* __restrict__ tmp_v,
                                                           const float* __restrict__ tmp_s,
                                                           const int* __restrict__ o_indptr,
                                                           Half* __restrict__ out, int batch_size,
                                                           int num_qo_heads) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= batch_size * num_qo_heads) return;
  const int b = unit / num_qo_heads, head = unit - b * num_qo_heads;
  merge_one<D, false>(tmp_v, tmp_s, o_indptr[b], o_indptr[b + 1], head, num_qo_heads,
               out + ((size_t)b * num_qo_heads + head) * D);
}

static void fill_args(DecodeAttnArgs& a, const Half* q, Half* output, const Half* kv, long k_off, long v_off,
                      const int* pi, const int* pip, const int* lpl, const int* ri, const int* kti, const int* kcs,
                      const uint8_t* mask, Half* tmp_v, float* tmp_s, int hq, int hkv, int page_size,
                      long stride_page, float sm_scale) {
  a = DecodeAttnArgs{};
  a.q = q; a.o_out = output; a.kv = kv; a.k_off = k_off; a.v_off = v_off;
  a.page_indices = pi; a.page_indptr = pip; a.last_page_len = lpl; a.request_indices = ri;
  a.kv_tile_indices = kti; a.kv_chunk_size_ptr = kcs; a.block_valid_mask = mask; a.tmp_v = tmp_v; a.tmp_s = tmp_s;
  a.num_qo_heads = hq; a.num_kv_heads = hkv; a.page_size = page_size; a.stride_page = stride_page;
  a.scale_log2 = sm_scale * 1.4426950408889634f;
}

template <int D, bool PARTITION, bool FUSED>
static int launch_decode(const DecodeAttnArgs& a, const int* o_indptr, int batch_size, int slots, hipStream_t s) {
  if (slots <= 0 || a.num_kv_heads <= 0) return 0;
  if (o_indptr && PARTITION && slots > batch_size * 64) return static_cast<int>(hipErrorInvalidValue);
  const int group = a.num_qo_heads / a.num_kv_heads;
  dim3 grid(slots, a.num_kv_heads);
  // 8 waves per workgroup when the launch has at most one workgroup per CU (bs 32 un-split: 4.60 -> 4.42 ms/step,
  // bs 1 at ctx 10000: 2.82 -> 2.67): twice the loads in flight per CU.  With two workgroups per CU the 64 KB of
  // partial states per workgroup cost more than they give (bs 8 / 16: +9 %).  PEGAINFER_ATTN_WAVES = 4 | 8 forces one.
  static const int nw_env = [] { const char* e = getenv("PEGAINFER_ATTN_WAVES"); return e && *e ? atoi(e) : 0; }();
  const bool wide = nw_env == 8 || (nw_env == 0 && (long)slots * a.num_kv_heads <= 256);
#define PK_LAUNCH(G)                                                                                \
  do {                                                                                              \
    if (wide) {                                                                                     \
      if constexpr (FUSED) fused_decode_attn_kernel<G, PARTITION, 8><<<grid, 512, 0, s>>>(a);       \
      else decode_attn_kernel<D, G, PARTITION, 8><<<grid, 512, 0, s>>>(a);                          \
    } else {                                                                                        \
      if constexpr (FUSED) fused_decode_attn_kernel<G, PARTITION, 4><<<grid, 256, 0, s>>>(a);       \
      else decode_attn_kernel<D, G, PARTITION, 4><<<grid, 256, 0, s>>>(a);                          \
    }                                                                                               \
  } while (0)
  switch (group) {
    case 1: PK_LAUNCH(1); break;
    case 2: PK_LAUNCH(2); break;
    case 4: PK_LAUNCH(4); break;
    case 8: PK_LAUNCH(8); break;
    default: return static_cast<int>(hipErrorInvalidValue);
  }
#undef PK_LAUNCH
  if (PARTITION && !a.merge_counters)
    merge_states_kernel<D><<<ceil_div((long)batch_size * a.num_qo_heads, 4), 256, 0, s>>>(
        a.tmp_v, a.tmp_s, o_indptr, a.o_out, batch_size, a.num_qo_heads);
  return static_cast<int>(hipGetLastError());
}

}  // namespace pk

using namespace pk;

extern "C" {

int32_t paged_attention_decode_cuda(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems,
                                    int64_t v_offset_elems, const int32_t* page_indices,
                                    const int32_t* page_indptr, const int32_t* last_page_len_d,
                                    const int32_t* request_indices, const int32_t* kv_tile_indices,
                                    const int32_t* kv_chunk_size_ptr, int32_t num_qo_heads, int32_t num_kv_heads,
                                    int32_t head_dim, int32_t page_size, int32_t batch_size, int64_t stride_page,
                                    float sm_scale, pegainfer_stream_t stream) {
  if (head_dim != 128) return static_cast<int32_t>(hipErrorInvalidValue);  // HEAD_DIM=128 instantiation
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, nullptr, nullptr, nullptr, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  return launch_decode<128, false, false>(a, nullptr, batch_size, batch_size, as_stream(stream));
}

int32_t paged_attention_decode_split_kv_cuda(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr,
    const int32_t* o_indptr, const uint8_t* block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads,
    int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t padded_batch_size,
    int64_t stride_page, float sm_scale, pegainfer_stream_t stream) {
  if (head_dim != 128) return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, block_valid_mask, tmp_v, tmp_s, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  return launch_decode<128, true, false>(a, o_indptr, batch_size, padded_batch_size, as_stream(stream));
}

// HEAD_DIM = 256 instantiation (Qwen3.5 full-attention layers, ffi.rs:1286-1306): 32 lanes per K row,
// 2 rows per load instruction, otherwise the same kernel.
int32_t paged_attention_decode_cuda_hd256(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems,
                                          int64_t v_offset_elems, const int32_t* page_indices,
                                          const int32_t* page_indptr, const int32_t* last_page_len_d,
                                          const int32_t* request_indices, const int32_t* kv_tile_indices,
                                          const int32_t* kv_chunk_size_ptr, int32_t num_qo_heads,
                                          int32_t num_kv_heads, int32_t head_dim, int32_t page_size,
                                          int32_t batch_size, int64_t stride_page, float sm_scale,
                                          pegainfer_stream_t stream) {
  if (head_dim != 256) return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, nullptr, nullptr, nullptr, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  return launch_decode<256, false, false>(a, nullptr, batch_size, batch_size, as_stream(stream));
}

// Extension: partition-KV decode at head_dim 256.  The reference has no split symbol for the Qwen3.5 full-attention
// layers (ffi.rs:1286-1306 is non-partition only), which leaves bs = 1 with num_kv_heads = 4 workgroups on a
// 256-CU part; same plan arrays and scratch contract as paged_attention_decode_split_kv_cuda.
int32_t pegainfer_paged_attention_decode_split_kv_hd256(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr,
    const int32_t* o_indptr, const uint8_t* block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads,
    int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t padded_batch_size,
    int64_t stride_page, float sm_scale, int32_t* merge_counters, pegainfer_stream_t stream) {
  if (head_dim != 256) return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, block_valid_mask, tmp_v, tmp_s, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  a.merge_counters = merge_counters;  // optional in-launch merge, as in pegainfer_fused_decode_attention
  a.o_indptr = o_indptr;
  return launch_decode<256, true, false>(a, o_indptr, batch_size, padded_batch_size, as_stream(stream));
}

// Extension (include/pegainfer_kernels_ext.h): qk_norm_rope_batched_decode_cuda + paged_kv_scatter_cuda +
// paged_attention_decode[_split_kv]_cuda in one launch (+ the merge when partitioned).
int32_t pegainfer_fused_decode_attention(
    const Half* qkv, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* positions, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache,
    const Half* sin_cache, float rms_eps, int32_t use_split, const int32_t* split_request_indices,
    const int32_t* split_kv_tile_indices, const int32_t* split_kv_chunk_size_ptr, const int32_t* split_o_indptr,
    const uint8_t* split_block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads, int32_t num_kv_heads,
    int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t split_slots, int64_t stride_page,
    float sm_scale, const int32_t* slot_desc, int32_t* merge_counters, pegainfer_stream_t stream) {
  if (head_dim != 128 || !host_aligned16(qkv) || !host_aligned16(kv_data) || !host_aligned16(slot_desc))
    return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, nullptr, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            use_split ? split_request_indices : nullptr, split_kv_tile_indices, split_kv_chunk_size_ptr,
            use_split ? split_block_valid_mask : nullptr, tmp_v, tmp_s, num_qo_heads, num_kv_heads, page_size,
            stride_page, sm_scale);
  a.qkv = qkv; a.q_norm_w = q_norm_weight; a.k_norm_w = k_norm_weight; a.cos_cache = cos_cache;
  a.sin_cache = sin_cache; a.positions = positions; a.eps = rms_eps; a.slot_desc = slot_desc;
  if (use_split) {
    a.merge_counters = merge_counters; a.o_indptr = split_o_indptr;
    return launch_decode<128, true, true>(a, split_o_indptr, batch_size, split_slots, as_stream(stream));
  }
  return launch_decode<128, false, true>(a, nullptr, batch_size, batch_size, as_stream(stream));
}

}  // extern "C"
