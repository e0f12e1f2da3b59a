// Qwen3.5 gated delta rule, chunk-wise prefill (chunk = 64 tokens, key_dim = value_dim = 128) for gfx950.
//
// Replaces the seven Triton-AOT kernels behind ffi.rs:1041-1137 (gated_delta_rule_prefill_chunk_{prepare,
// cumsum,a,solve,recompute,state,o}_cuda; source tools/triton/gated_delta_rule_chunkwise_kernels.py) with the
// same stage boundaries, scratch layouts and bf16 rounding points, so the Rust operator
// (pegainfer-qwen35-4b/src/recurrent.rs:368-470) binds unchanged.  Token-major tensors:
//   q, k, w [T, H, 128] bf16   v, u, v_new, out [T, H, 128] bf16   g, beta [T, H] f32
//   a_tril [T, H, 64] f32      a_inv [T, H, 64] bf16               state [H, K, V] f32 (V contiguous)
//   chunk_state [nchunks, H, K, V] f32
// All matrix products run on v_mfma_f32_16x16x32_bf16 (wave64): A operand = row (lane & 15), 8 consecutive k
// at (lane >> 4) * 8; B operand = column (lane & 15), same k; C = column (lane & 15), rows (lane >> 4) * 4 + i.
// Whenever the contraction runs over TOKENS (the strided dimension of a token-major tensor) the operand is
// transposed on its way into LDS so that fragments are single ds_read_b128.
#include "common.h"

namespace pk {

constexpr int GC = 64;    // chunk
constexpr int GK = 128;   // key dim == value dim
constexpr int TP = GC + 8;   // LDS pitch (bf16) of a [*, 64-token] transposed tile: 144 B, 16-byte aligned rows
constexpr int KP = GK + 8;   // LDS pitch (bf16) of a [*, 128-k] tile

__device__ __forceinline__ bf16x8_t ld_frag(const Half* p) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ bf16x8_t zero_frag() { return __builtin_bit_cast(bf16x8_t, u32x4{0u, 0u, 0u, 0u}); }
__device__ __forceinline__ f32x4 mfma(bf16x8_t a, bf16x8_t b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- stage 1: head expansion + L2 normalisation of q/k, raw v, g and beta (kernels.py:29-84) ----
__global__ __launch_bounds__(256) void gdr_prepare_kernel(const Half* __restrict__ qkv, const Half* __restrict__ b_proj,
                                                          const Half* __restrict__ a_proj,
                                                          const Half* __restrict__ dt_bias,
                                                          const float* __restrict__ a_log, Half* __restrict__ q_out,
                                                          Half* __restrict__ k_out, Half* __restrict__ v_out,
                                                          float* __restrict__ g_out, float* __restrict__ beta_out,
                                                          int nkh, int nvh, int qkv_dim, int T) {
  const int t = blockIdx.x, vh = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (vh >= nvh) return;
  const int kh = (vh * nkh) / nvh;
  const Half* row = qkv + (size_t)t * qkv_dim;
  const uint32_t qw = *reinterpret_cast<const uint32_t*>(row + kh * GK + lane * 2);
  const uint32_t kw = *reinterpret_cast<const uint32_t*>(row + nkh * GK + kh * GK + lane * 2);
  const uint32_t vw = *reinterpret_cast<const uint32_t*>(row + 2 * nkh * GK + vh * GK + lane * 2);
  const float q0 = bf_lo(qw), q1 = bf_hi(qw), k0 = bf_lo(kw), k1 = bf_hi(kw);
  const float qs = rsqrtf(wave_sum(q0 * q0 + q1 * q1) + 1e-12f);
  const float ks = rsqrtf(wave_sum(k0 * k0 + k1 * k1) + 1e-12f);
  const size_t o = ((size_t)t * nvh + vh) * GK + lane * 2;
  *reinterpret_cast<uint32_t*>(q_out + o) = pack_bf2(q0 * qs, q1 * qs);
  *reinterpret_cast<uint32_t*>(k_out + o) = pack_bf2(k0 * ks, k1 * ks);
  *reinterpret_cast<uint32_t*>(v_out + o) = vw;
  if (lane == 0) {
    const float x = bf2f(a_proj[(size_t)t * nvh + vh]) + bf2f(dt_bias[vh]);
    const float sp = x > 20.f ? x : logf(1.f + expf(x));
    g_out[(size_t)t * nvh + vh] = -expf(a_log[vh]) * sp;
    beta_out[(size_t)t * nvh + vh] = 1.f / (1.f + expf(-bf2f(b_proj[(size_t)t * nvh + vh])));
  }
}

// ---- stage 2: chunk-local inclusive prefix sum of g (kernels.py:143-159); one wave per (chunk, head) ----
__global__ __launch_bounds__(256) void gdr_cumsum_kernel(const float* g_in, float* g_out, int T, int nvh) {
  const int chunk = blockIdx.x, vh = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (vh >= nvh) return;
  const int t = chunk * GC + lane;
  float v = t < T ? g_in[(size_t)t * nvh + vh] : 0.f;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float up = __shfl_up(v, d, kWave);
    if (lane >= d) v += up;
  }
  if (t < T) g_out[(size_t)t * nvh + vh] = v;
}

// ---- stage 3: A = tril(beta_t exp(g_t - g_j) <k_t, k_j>, -1)  (kernels.py:162-215) ----
__global__ __launch_bounds__(256) void gdr_kkt_kernel(const Half* __restrict__ k, const float* __restrict__ g,
                                                      const float* __restrict__ beta, float* __restrict__ a_tril,
                                                      int T, int nvh) {
  const int c0 = blockIdx.x * GC, vh = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, q4 = lane >> 4;
  const size_t rs = (size_t)nvh * GK;  // token stride
  auto krow = [&](int t) { t = c0 + t; t = t < T ? t : T - 1; return k + (size_t)t * rs + vh * GK; };
  f32x4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const Half* arow = krow(16 * wave + l15);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8_t a = ld_frag(arow + q4 * 8 + 32 * ks);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = mfma(a, ld_frag(krow(16 * nb + l15) + q4 * 8 + 32 * ks), acc[nb]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tl = 16 * wave + q4 * 4 + i, t = c0 + tl;
    if (t >= T) continue;
    const float gt = g[(size_t)t * nvh + vh], bt = beta[(size_t)t * nvh + vh];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int jl = 16 * nb + l15, j = c0 + jl;
      float v = 0.f;
      if (jl < tl && j < T) v = acc[nb][i] * bt * expf(gt - g[(size_t)j * nvh + vh]);
      a_tril[((size_t)t * nvh + vh) * GC + jl] = v;
    }
  }
}

// ---- stage 4: A_inv = (I + A)^-1, bf16 (kernels.py:218-330).  One wave per (chunk, head): lane c owns
//      column c of the inverse, rows by forward substitution in fp32 ----
__global__ __launch_bounds__(64) void gdr_solve_kernel(const float* __restrict__ a_tril, Half* __restrict__ a_inv,
                                                       int T, int nvh) {
  const int c0 = blockIdx.x * GC, vh = blockIdx.y, lane = threadIdx.x;
  // (round 6) RIGHT-looking order with the broadcasts through LDS: as soon as row j is final, every later row takes its A[i][j] x[j]
  // term.  A row still receives its terms in ascending j - the same operations in the same order as "row i = 1 - sum_{j < i}", hence
  // the same bits.  The row-by-row form took A[i][j] with one v_readlane per term; 2016 of them are what the kernel's 34 us were (a
  // v_readlane costs ~28 cycles here: pinning the order and keeping eight in flight with inline asm moved it to 31.5 us, no further).
  // Now A is stored TRANSPOSED in LDS once and a step reads its column as uniform-address (broadcast) vector loads.
  __shared__ __attribute__((aligned(16))) float at[GC][GC + 4];   // at[j][i] = A[i][j]
  float x[GC];
#pragma unroll
  for (int i = 0; i < GC; ++i) {
    const int t = c0 + i, tc = t < T ? t : T - 1;       // clamped, branch-free: rows past the end read a valid row and are zeroed
    const float a = a_tril[((size_t)tc * nvh + vh) * GC + lane];  // A[i][lane]
    at[lane][i] = t < T ? a : 0.f;
    x[i] = lane == i ? 1.f : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < GC - 1; ++j) {
#pragma unroll
    for (int i = j + 1; i < GC; ++i) x[i] = fmaf(-at[j][i], x[j], x[i]);
  }
#pragma unroll
  for (int i = 0; i < GC; ++i) {
    const int t = c0 + i;
    if (t < T) a_inv[((size_t)t * nvh + vh) * GC + lane] = f2bf(x[i]);
  }
}

// stage a [64 tokens][128] bf16 chunk of a token-major tensor into LDS transposed: dst[n][s] (pitch TP), applying
// f(value_fp32, token) and one bf16 rounding per application step done by the caller's functor
template <typename F>
__device__ __forceinline__ void stage_transposed(Half* dst, const Half* src, size_t rs, int c0, int T, F f) {
  // token index fastest across lanes: a wave's 2-byte LDS writes of one element slot are then 64 consecutive
  // tokens of ONE transposed row (conflict-free).  With the 16-byte chunk index fastest instead, lanes wrote rows
  // 8 apart - 2 bank positions for 16 lanes, and SQ_LDS_BANK_CONFLICT was 90 % of the LDS cycles of the state
  // kernel.  The price is 64 different source rows per load instruction, served by L1/L2 (the tile is 16 KB).
  for (int idx = threadIdx.x; idx < GC * (GK / 8); idx += 256) {
    const int s = idx & 63, n8 = idx >> 6, t = c0 + s;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (t < T) v = *reinterpret_cast<const u32x4*>(src + (size_t)t * rs + n8 * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dst[(n8 * 8 + 2 * e) * TP + s] = f(bf_lo(w[e]), s);
      dst[(n8 * 8 + 2 * e + 1) * TP + s] = f(bf_hi(w[e]), s);
    }
  }
}

// ---- stage 5: u = A_inv (v beta), w = A_inv (k beta exp(g))  (kernels.py:333-430) ----
__global__ __launch_bounds__(256) void gdr_recompute_kernel(const Half* __restrict__ k, const Half* __restrict__ v,
                                                            const float* __restrict__ beta, Half* __restrict__ w,
                                                            Half* __restrict__ u, const Half* __restrict__ a_inv,
                                                            const float* __restrict__ g, int T, int nvh) {
  __shared__ __attribute__((aligned(16))) Half sT[GK * TP];
  __shared__ float sb[GC], sg[GC];
  const int c0 = blockIdx.x * GC, vh = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, q4 = lane >> 4;
  const size_t rs = (size_t)nvh * GK;
  if (threadIdx.x < GC) {
    const int t = c0 + threadIdx.x;
    // beta and exp(g) enter the products as bf16 (`.to(k.dtype)` in the reference)
    sb[threadIdx.x] = t < T ? bf16_round_f(beta[(size_t)t * nvh + vh]) : 0.f;
    sg[threadIdx.x] = t < T ? bf16_round_f(expf(g[(size_t)t * nvh + vh])) : 0.f;
  }
  // A_inv fragments of this wave's 16 token rows (K = 64 -> 2 k-steps); rows past seq_len read as zero
  bf16x8_t af[2];
  {
    const int t = c0 + 16 * wave + l15;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      af[ks] = t < T ? ld_frag(a_inv + ((size_t)t * nvh + vh) * GC + q4 * 8 + 32 * ks) : zero_frag();
  }
  __syncthreads();
  auto product = [&](Half* out) {
    f32x4 acc[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
        acc[nb] = mfma(af[ks], ld_frag(sT + (16 * nb + l15) * TP + q4 * 8 + 32 * ks), acc[nb]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = c0 + 16 * wave + q4 * 4 + i;
      if (t >= T) continue;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) out[(size_t)t * rs + vh * GK + 16 * nb + l15] = f2bf(acc[nb][i]);
    }
  };
  stage_transposed(sT, v + vh * GK, rs, c0, T, [&](float x, int s) { return f2bf(x * sb[s]); });
  __syncthreads();
  product(u);
  __syncthreads();
  stage_transposed(sT, k + vh * GK, rs, c0, T, [&](float x, int s) { return f2bf(bf16_round_f(x * sb[s]) * sg[s]); });
  __syncthreads();
  product(w);
}

// ---- stage 6: the serial recurrence over chunks (kernels.py:433-600).  One workgroup per (16-wide V tile,
//      head); h[128 k][16 v] lives in MFMA accumulators (wave w holds k blocks 2w, 2w+1) ----
constexpr int SBV = 16;
__global__ __launch_bounds__(256) void gdr_state_kernel(const Half* __restrict__ k, const Half* __restrict__ w,
                                                        const Half* __restrict__ u, const float* __restrict__ g,
                                                        const float* initial_state, float* __restrict__ chunk_state,
                                                        Half* __restrict__ v_new, float* final_state, int T,
                                                        int nvh) {
  __shared__ __attribute__((aligned(16))) Half hT[SBV * KP];   // bf16(h)^T  [v][k]
  __shared__ __attribute__((aligned(16))) Half vgT[SBV * TP];  // bf16(v_new * gate)^T [v][t]
  __shared__ __attribute__((aligned(16))) Half kT[GK * TP];    // k^T [k][t]
  const int v0 = blockIdx.x * SBV, vh = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, q4 = lane >> 4;
  const size_t rs = (size_t)nvh * GK;
  f32x4 h[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      h[b][i] = initial_state[((size_t)vh * GK + 16 * (2 * wave + b) + q4 * 4 + i) * GK + v0 + l15];
  const int nchunks = (T + GC - 1) / GC;
  // (round 6) everything a chunk reads from global memory - its k tile, this wave's w fragments, its u / g values - is requested one
  // chunk AHEAD, into registers, so the serial recurrence no longer waits for three dependent global round trips per chunk (70 us for
  // the 16 chunks of a 1024-token prompt = 4.4 us per chunk for ~0.1 us of MFMAs).  Same loads, same arithmetic, same bits.
  struct ChunkRegs { u32x4 kraw[4]; bf16x8_t wf[4]; float uval[4], gt[4], g_last; };
  auto fetch = [&](int ci, ChunkRegs& r) {
    const int c0 = ci * GC, n = T - c0 < GC ? T - c0 : GC;
#pragma unroll
    for (int jv = 0; jv < 4; ++jv) {
      const int idx = threadIdx.x + 256 * jv, sidx = idx & 63, n8 = idx >> 6, t = c0 + sidx;
      r.kraw[jv] = u32x4{0u, 0u, 0u, 0u};
      if (t < T) r.kraw[jv] = *reinterpret_cast<const u32x4*>(k + vh * GK + (size_t)t * rs + n8 * 8);
    }
    {
      const int t = c0 + 16 * wave + l15;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) r.wf[ks] = t < T ? ld_frag(w + (size_t)t * rs + vh * GK + q4 * 8 + 32 * ks) : zero_frag();
    }
    r.g_last = g[(size_t)(c0 + n - 1) * nvh + vh];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = c0 + 16 * wave + q4 * 4 + i;
      r.uval[i] = 0.f;
      r.gt[i] = 0.f;
      if (t < T) {
        r.uval[i] = bf2f(u[(size_t)t * rs + vh * GK + v0 + l15]);
        r.gt[i] = g[(size_t)t * nvh + vh];
      }
    }
  };
  ChunkRegs cur, nxt;
  if (nchunks > 0) fetch(0, cur);
  for (int ci = 0; ci < nchunks; ++ci) {
    const int c0 = ci * GC;
    // snapshot + bf16 transpose of h; k^T of this chunk
    float* cs = chunk_state + ((size_t)ci * nvh + vh) * GK * GK;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kb = 16 * (2 * wave + b) + q4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) cs[(size_t)(kb + i) * GK + v0 + l15] = h[b][i];
      u32x2 p;
      p.x = pack_bf2(h[b][0], h[b][1]);
      p.y = pack_bf2(h[b][2], h[b][3]);
      *reinterpret_cast<u32x2*>(hT + l15 * KP + kb) = p;
    }
#pragma unroll
    for (int jv = 0; jv < 4; ++jv) {   // stage_transposed's store half (token index fastest across lanes: conflict-free)
      const int idx = threadIdx.x + 256 * jv, sidx = idx & 63, n8 = idx >> 6;
      const uint32_t wv[4] = {cur.kraw[jv].x, cur.kraw[jv].y, cur.kraw[jv].z, cur.kraw[jv].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kT[(n8 * 8 + 2 * e) * TP + sidx] = f2bf(bf_lo(wv[e]));
        kT[(n8 * 8 + 2 * e + 1) * TP + sidx] = f2bf(bf_hi(wv[e]));
      }
    }
    if (ci + 1 < nchunks) fetch(ci + 1, nxt);   // lands under this chunk's barriers and MFMAs
    __syncthreads();
    // v_new[t][v] = u[t][v] - sum_k w[t][k] bf16(h[k][v]); this wave: t rows 16*wave .. +15
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = mfma(cur.wf[ks], ld_frag(hT + l15 * KP + q4 * 8 + 32 * ks), acc);
    const float g_last = cur.g_last;
    float vg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tl = 16 * wave + q4 * 4 + i, t = c0 + tl;
      float vn = 0.f, gate = 0.f;
      if (t < T) {
        vn = cur.uval[i] - acc[i];
        v_new[(size_t)t * rs + vh * GK + v0 + l15] = f2bf(vn);
        gate = expf(g_last - cur.gt[i]);
      }
      vg[i] = vn * gate;
    }
    {
      u32x2 p;
      p.x = pack_bf2(vg[0], vg[1]);
      p.y = pack_bf2(vg[2], vg[3]);
      *reinterpret_cast<u32x2*>(vgT + l15 * TP + 16 * wave + q4 * 4) = p;
    }
    __syncthreads();
    // h = h * exp(g_last) + k^T (v_new * gate)
    const float decay = expf(g_last);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int i = 0; i < 4; ++i) h[b][i] *= decay;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        h[b] = mfma(ld_frag(kT + (16 * (2 * wave + b) + l15) * TP + q4 * 8 + 32 * ks),
                    ld_frag(vgT + l15 * TP + q4 * 8 + 32 * ks), h[b]);
    }
    __syncthreads();  // kT / vgT / hT are rewritten by the next chunk
    cur = nxt;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      final_state[((size_t)vh * GK + 16 * (2 * wave + b) + q4 * 4 + i) * GK + v0 + l15] = h[b][i];
}

// ---- stage 7: out = (exp(g) (q bf16(h_chunk)) + bf16(tril(q k^T exp(g_i - g_j))) v_new) * scale
//      (kernels.py:603-709).  Workgroup = (64-wide V tile, chunk, head); wave w owns token rows 16w..16w+15 ----
constexpr int OBV = 64;
__global__ __launch_bounds__(256) void gdr_o_kernel(const Half* __restrict__ q, const Half* __restrict__ k,
                                                    const Half* __restrict__ v_new,
                                                    const float* __restrict__ chunk_state,
                                                    const float* __restrict__ g, Half* __restrict__ out, int T,
                                                    int nvh, float scale) {
  __shared__ __attribute__((aligned(16))) Half hT[OBV * KP];  // bf16(h_chunk)^T [v][k]
  __shared__ __attribute__((aligned(16))) Half pS[GC * TP];   // bf16(P) [t][j]
  __shared__ __attribute__((aligned(16))) Half vT[OBV * TP];  // v_new^T [v][j]
  const int v0 = blockIdx.x * OBV, c0 = blockIdx.y * GC, vh = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, q4 = lane >> 4;
  const size_t rs = (size_t)nvh * GK;
  // stage bf16(h)^T: h_chunk [k][v] f32, 4 v per 16-byte load
  const float* hc = chunk_state + ((size_t)blockIdx.y * nvh + vh) * GK * GK;
  for (int idx = threadIdx.x; idx < GK * (OBV / 4); idx += 256) {
    const int kk = idx >> 4, v4 = idx & 15;  // (k-fastest lanes measured slower here: 25.0 vs 22.4 us - the fp32 source
                                             //  rows are 512 B apart, the strided loads cost more than the conflicts)
    const f32x4 x = *reinterpret_cast<const f32x4*>(hc + (size_t)kk * GK + v0 + v4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) hT[(v4 * 4 + e) * KP + kk] = f2bf(x[e]);
  }
  // stage v_new^T [v][j]
  for (int idx = threadIdx.x; idx < GC * (OBV / 8); idx += 256) {
    const int s = idx >> 3, n8 = idx & 7, t = c0 + s;
    u32x4 x = u32x4{0u, 0u, 0u, 0u};
    if (t < T) x = *reinterpret_cast<const u32x4*>(v_new + (size_t)t * rs + vh * GK + v0 + n8 * 8);
    const uint32_t wv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      vT[(n8 * 8 + 2 * e) * TP + s] = (Half)(wv[e] & 0xFFFFu);
      vT[(n8 * 8 + 2 * e + 1) * TP + s] = (Half)(wv[e] >> 16);
    }
  }
  auto row = [&](const Half* base, int tl) { int t = c0 + tl; t = t < T ? t : T - 1; return base + (size_t)t * rs + vh * GK; };
  // S = q k^T (this wave's 16 rows x 64 columns)
  f32x4 sacc[4], oacc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) { sacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  bf16x8_t qf[4];
  {
    const Half* qr = row(q, 16 * wave + l15);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_frag(qr + q4 * 8 + 32 * ks);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) sacc[nb] = mfma(qf[ks], ld_frag(row(k, 16 * nb + l15) + q4 * 8 + 32 * ks), sacc[nb]);
  // gate + causal mask, bf16, into LDS as [t][j]
  float gt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = c0 + 16 * wave + q4 * 4 + i;
    gt[i] = t < T ? g[(size_t)t * nvh + vh] : 0.f;
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int jl = 16 * nb + l15, j = c0 + jl;
    const float gj = j < T ? g[(size_t)j * nvh + vh] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tl = 16 * wave + q4 * 4 + i;
      float p = 0.f;
      if (jl <= tl && c0 + tl < T) p = sacc[nb][i] * expf(gt[i] - gj);
      pS[tl * TP + jl] = f2bf(p);
    }
  }
  __syncthreads();
  // O1 = q bf16(h): K = 128
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) oacc[nb] = mfma(qf[ks], ld_frag(hT + (16 * nb + l15) * KP + q4 * 8 + 32 * ks), oacc[nb]);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[nb][i] *= expf(gt[i]);
  // O2 = P v_new: K = 64 tokens
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t pf = ld_frag(pS + (16 * wave + l15) * TP + q4 * 8 + 32 * ks);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) oacc[nb] = mfma(pf, ld_frag(vT + (16 * nb + l15) * TP + q4 * 8 + 32 * ks), oacc[nb]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = c0 + 16 * wave + q4 * 4 + i;
    if (t >= T) continue;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) out[(size_t)t * rs + vh * GK + v0 + 16 * nb + l15] = f2bf(oacc[nb][i] * scale);
  }
}

}  // namespace pk

using namespace pk;

extern "C" {

pegainfer_status_t gated_delta_rule_prefill_chunk_prepare_cuda(const Half* qkv, const Half* b_proj, const Half* a_proj,
                                                               const Half* dt_bias, const float* a_log, Half* q_out,
                                                               Half* k_out, Half* v_out, float* g_out, float* beta_out,
                                                               int32_t num_key_heads, int32_t num_value_heads,
                                                               int32_t qkv_dim, int32_t seq_len,
                                                               pegainfer_stream_t stream) {
  if (seq_len <= 0) return 0;
  if (num_key_heads <= 0 || num_value_heads <= 0 || qkv_dim != (2 * num_key_heads + num_value_heads) * GK)
    return (pegainfer_status_t)hipErrorInvalidValue;  // fixed key_dim = value_dim = 128 like the reference
  gdr_prepare_kernel<<<dim3(seq_len, ceil_div(num_value_heads, 4)), 256, 0, as_stream(stream)>>>(
      qkv, b_proj, a_proj, dt_bias, a_log, q_out, k_out, v_out, g_out, beta_out, num_key_heads, num_value_heads,
      qkv_dim, seq_len);
  return (pegainfer_status_t)hipGetLastError();
}

pegainfer_status_t gated_delta_rule_prefill_chunk_cumsum_cuda(const float* g_in, float* g_out, int32_t seq_len,
                                                              int32_t num_value_heads, pegainfer_stream_t stream) {
  if (seq_len <= 0) return 0;
  gdr_cumsum_kernel<<<dim3(ceil_div(seq_len, GC), ceil_div(num_value_heads, 4)), 256, 0, as_stream(stream)>>>(
      g_in, g_out, seq_len, num_value_heads);
  return (pegainfer_status_t)hipGetLastError();
}

pegainfer_status_t gated_delta_rule_prefill_chunk_a_cuda(const Half* k, const float* g_cumsum, const float* beta,
                                                         float* a_tril, int32_t seq_len, int32_t num_value_heads,
                                                         pegainfer_stream_t stream) {
  if (seq_len <= 0) return 0;
  gdr_kkt_kernel<<<dim3(ceil_div(seq_len, GC), num_value_heads), 256, 0, as_stream(stream)>>>(
      k, g_cumsum, beta, a_tril, seq_len, num_value_heads);
  return (pegainfer_status_t)hipGetLastError();
}

pegainfer_status_t gated_delta_rule_prefill_chunk_solve_cuda(const float* a_tril, Half* a_inv, int32_t seq_len,
                                                             int32_t num_value_heads, pegainfer_stream_t stream) {
  if (seq_len <= 0) return 0;
  gdr_solve_kernel<<<dim3(ceil_div(seq_len, GC), num_value_heads), 64, 0, as_stream(stream)>>>(a_tril, a_inv, seq_len,
                                                                                             num_value_heads);
  return (pegainfer_status_t)hipGetLastError();
}

pegainfer_status_t gated_delta_rule_prefill_chunk_recompute_cuda(const Half* k, const Half* v, const float* beta,
                                                                 Half* w, Half* u, const Half* a_inv,
                                                                 const float* g_cumsum, int32_t seq_len,
                                                                 int32_t num_value_heads, pegainfer_stream_t stream) {
  if (seq_len <= 0) return 0;
  gdr_recompute_kernel<<<dim3(ceil_div(seq_len, GC), num_value_heads), 256, 0, as_stream(stream)>>>(
      k, v, beta, w, u, a_inv, g_cumsum, seq_len, num_value_heads);
  return (pegainfer_status_t)hipGetLastError();
}

pegainfer_status_t gated_delta_rule_prefill_chunk_state_cuda(const Half* k, const Half* w, const Half* u,
                                                             const float* g_cumsum, const float* initial_state,
                                                             float* chunk_state, Half* v_new, float* final_state,
                                                             int32_t seq_len, int32_t num_value_heads,
                                                             pegainfer_stream_t stream) {
  if (seq_len <= 0) return 0;
  gdr_state_kernel<<<dim3(GK / SBV, num_value_heads), 256, 0, as_stream(stream)>>>(
      k, w, u, g_cumsum, initial_state, chunk_state, v_new, final_state, seq_len, num_value_heads);
  return (pegainfer_status_t)hipGetLastError();
}

pegainfer_status_t gated_delta_rule_prefill_chunk_o_cuda(const Half* q, const Half* k, const Half* v_new,
                                                         const float* chunk_state, const float* g_cumsum,
                                                         Half* output, int32_t seq_len, int32_t num_value_heads,
                                                         float scale, pegainfer_stream_t stream) {
  if (seq_len <= 0) return 0;
  gdr_o_kernel<<<dim3(GK / OBV, ceil_div(seq_len, GC), num_value_heads), 256, 0, as_stream(stream)>>>(
      q, k, v_new, chunk_state, g_cumsum, output, seq_len, num_value_heads, scale);
  return (pegainfer_status_t)hipGetLastError();
}

}  // extern "C"
