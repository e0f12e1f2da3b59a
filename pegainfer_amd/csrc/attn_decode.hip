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
#include "common.h"

namespace pk {

template <int LPT>
__device__ __forceinline__ float token_sum(float v) {
  v = row16_sum(v);
  if (LPT == 32) v += __shfl_xor(v, 16, kWave);
  return v;
}

template <int D, int GROUP, bool PARTITION>
__global__ __launch_bounds__(256) void decode_attn_kernel(
    const Half* __restrict__ q, Half* __restrict__ o_out, const Half* __restrict__ kv, long k_off, long v_off,
    const int* __restrict__ page_indices, const int* __restrict__ page_indptr,
    const int* __restrict__ last_page_len, const int* __restrict__ request_indices,
    const int* __restrict__ kv_tile_indices, const int* __restrict__ kv_chunk_size_ptr,
    const uint8_t* __restrict__ block_valid_mask, Half* __restrict__ tmp_v, float* __restrict__ tmp_s,
    int num_qo_heads, int num_kv_heads, int page_size, long stride_page, float scale_log2) {
  constexpr int LPT = D / 8;     // lanes per token row
  constexpr int TPI = 64 / LPT;  // token rows per load instruction
  constexpr int U = 4;           // load instructions in flight per operand
  constexpr int TB = TPI * U;    // tokens per wave iteration
  constexpr int NPART = 4 * TPI; // partial softmax states per workgroup
  __shared__ float sm_m[NPART][GROUP];
  __shared__ float sm_l[NPART][GROUP];
  __shared__ __attribute__((aligned(16))) float sm_o[NPART][GROUP][D];

  const int slot = blockIdx.x, kvh = blockIdx.y;
  if (PARTITION && block_valid_mask && !block_valid_mask[slot]) return;
  const int b = request_indices ? request_indices[slot] : slot;
  const int pbase = page_indptr[b];
  const int npages = page_indptr[b + 1] - pbase;
  const int kv_len = npages > 0 ? (npages - 1) * page_size + last_page_len[b] : 0;
  int lo = 0, hi = kv_len;
  if (PARTITION) {
    const int chunk = kv_chunk_size_ptr[0];
    lo = kv_tile_indices[slot] * chunk;
    hi = lo + chunk < kv_len ? lo + chunk : kv_len;
    if (lo > hi) lo = hi;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane % LPT, grp = lane / LPT;

  u32x4 qv[GROUP];
#pragma unroll
  for (int h = 0; h < GROUP; ++h)
    qv[h] = *reinterpret_cast<const u32x4*>(q + ((size_t)b * num_qo_heads + kvh * GROUP + h) * D + sub * 8);

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

  for (int t0 = (lo / TB) * TB + wave * TB; t0 < hi && lo < hi; t0 += 4 * TB) {
    u32x4 kx[U], vx[U];
    bool ok[U];
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
  for (int e = threadIdx.x; e < GROUP * D; e += 256) {
    const int h = e / D, d = e - h * D;
    float M = -INFINITY;
#pragma unroll
    for (int p = 0; p < NPART; ++p) M = fmaxf(M, sm_m[p][h]);
    float L = 0.f, O = 0.f;
    if (M != -INFINITY) {
#pragma unroll
      for (int p = 0; p < NPART; ++p) {
        const float w = exp2f(sm_m[p][h] - M);
        L += sm_l[p][h] * w;
        O += sm_o[p][h][d] * w;
      }
    }
    const float val = L > 0.f ? O / L : 0.f;
    const int head = kvh * GROUP + h;
    if (PARTITION) {
      tmp_v[((size_t)slot * num_qo_heads + head) * D + d] = f2bf(val);
      if (d == 0) tmp_s[(size_t)slot * num_qo_heads + head] = L > 0.f ? M + log2f(L) : -INFINITY;
    } else {
      o_out[((size_t)b * num_qo_heads + head) * D + d] = f2bf(val);
    }
  }
}

// one wave per (request, q head): out = sum_s 2^(lse_s - M) * v_s / sum_s 2^(lse_s - M)
template <int D>
__global__ __launch_bounds__(256) void merge_states_kernel(const Half* __restrict__ tmp_v,
                                                           const float* __restrict__ tmp_s,
                                                           const int* __restrict__ o_indptr,
                                                           Half* __restrict__ out, int batch_size,
                                                           int num_qo_heads) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= batch_size * num_qo_heads) return;
  const int b = unit / num_qo_heads, head = unit - b * num_qo_heads;
  const int lane = threadIdx.x & 63;
  const int s0 = o_indptr[b], s1 = o_indptr[b + 1];
  constexpr int EPL = D / 64;  // elements per lane (2 or 4)
  float M = -INFINITY;
  for (int s = s0; s < s1; ++s) M = fmaxf(M, tmp_s[(size_t)s * num_qo_heads + head]);
  float acc[EPL], wsum = 0.f;
#pragma unroll
  for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
  if (M != -INFINITY) {
    for (int s = s0; s < s1; ++s) {
      const float w = exp2f(tmp_s[(size_t)s * num_qo_heads + head] - M);
      wsum += w;
      const Half* v = tmp_v + ((size_t)s * num_qo_heads + head) * D + lane * EPL;
#pragma unroll
      for (int i = 0; i < EPL; ++i) acc[i] += w * bf2f(v[i]);
    }
  }
  Half* dst = out + ((size_t)b * num_qo_heads + head) * D + lane * EPL;
#pragma unroll
  for (int i = 0; i < EPL; ++i) dst[i] = f2bf(wsum > 0.f ? acc[i] / wsum : 0.f);
}

template <int D, bool PARTITION>
static int launch_decode(const Half* q, Half* output, const Half* kv, long k_off, long v_off, const int* pi,
                         const int* pip, const int* lpl, const int* ri, const int* kti, const int* kcs,
                         const int* o_indptr, const uint8_t* mask, Half* tmp_v, float* tmp_s, int hq, int hkv,
                         int page_size, int batch_size, int slots, long stride_page, float sm_scale,
                         hipStream_t s) {
  if (slots <= 0 || hkv <= 0) return 0;
  const int group = hq / hkv;
  const float scale_log2 = sm_scale * 1.4426950408889634f;
  dim3 grid(slots, hkv);
#define PK_LAUNCH(G)                                                                                   \
  decode_attn_kernel<D, G, PARTITION><<<grid, 256, 0, s>>>(q, output, kv, k_off, v_off, pi, pip, lpl, ri, kti, \
                                                           kcs, mask, tmp_v, tmp_s, hq, hkv, page_size,       \
                                                           stride_page, scale_log2)
  switch (group) {
    case 1: PK_LAUNCH(1); break;
    case 2: PK_LAUNCH(2); break;
    case 4: PK_LAUNCH(4); break;
    case 8: PK_LAUNCH(8); break;
    default: return static_cast<int>(hipErrorInvalidValue);
  }
#undef PK_LAUNCH
  if (PARTITION)
    merge_states_kernel<D><<<ceil_div((long)batch_size * hq, 4), 256, 0, s>>>(tmp_v, tmp_s, o_indptr, output,
                                                                             batch_size, hq);
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
  return launch_decode<128, false>(q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr,
                                   last_page_len_d, request_indices, kv_tile_indices, kv_chunk_size_ptr, nullptr,
                                   nullptr, nullptr, nullptr, num_qo_heads, num_kv_heads, page_size, batch_size,
                                   batch_size, stride_page, sm_scale, as_stream(stream));
}

int32_t paged_attention_decode_split_kv_cuda(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr,
    const int32_t* o_indptr, const uint8_t* block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads,
    int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t padded_batch_size,
    int64_t stride_page, float sm_scale, pegainfer_stream_t stream) {
  if (head_dim != 128) return static_cast<int32_t>(hipErrorInvalidValue);
  return launch_decode<128, true>(q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr,
                                  last_page_len_d, request_indices, kv_tile_indices, kv_chunk_size_ptr, o_indptr,
                                  block_valid_mask, tmp_v, tmp_s, num_qo_heads, num_kv_heads, page_size,
                                  batch_size, padded_batch_size, stride_page, sm_scale, as_stream(stream));
}

}  // extern "C"
