// Qwen3.5-4B hybrid extras for gfx950: causal depthwise conv1d, gated-delta-rule recurrent
// decode, and the HD256 full-attention prep / gate kernels.  (rms_norm_offset / rms_norm_gated live in
// norm.hip, silu_mul in elementwise.hip, the HD256 attention kernels in attn_decode.hip / attn_prefill.hip.)
#include "common.h"

namespace pk {

// =====================================================================================================
// conv1d_prefill_cuda (reference csrc/conv1d.cu:18-98): causal depthwise conv (kernel_size <= 5) over
// x_seq[t][c] with the previous K-1 inputs in conv_state[c][K-1]; fp32 taps in order k = 0..K-1, the sum
// rounded to bf16 BEFORE SiLU (conv1d.cu:56-58), bf16 out.  HBM-bound: one lane = 8 channels x 1 token,
// 16-byte loads of the K input rows, weights read as K-strided 16-byte rows.
// The state update runs as a SECOND launch: in the reference the thread at t = seq_len-1 rewrites
// conv_state while threads at t < K-1 of other blocks may still read it; splitting removes that race.
// Also used for decode with seq_len = 1 (qwen35 recurrent.rs:49-79).
// =====================================================================================================
constexpr int kConvMaxK = 5;

// Shapes the vector kernel below does not take (K != 4, C % 8 != 0, misaligned): the decode-step kernel's sliding
// window, walked over a segment of tokens.  One lane owns one channel and kConvSeg consecutive tokens: the window starts
// from the K-1 inputs before the segment (x rows, or conv_state for the rows before token 0), every token appends one
// x element, multiplies the window by the channel's taps in order k = 0..K-1 and shifts.  Neighbouring lanes are
// neighbouring channels, so every x / out access of a wave is one contiguous row piece.
constexpr int kConvSeg = 64;

__global__ __launch_bounds__(256) void conv1d_window_kernel(const Half* __restrict__ x, const Half* __restrict__ w,
                                                            const Half* __restrict__ state, Half* __restrict__ out,
                                                            int C, int T, int K) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int sw = K - 1;
  const int t_begin = blockIdx.y * kConvSeg;
  const int t_end = t_begin + kConvSeg < T ? t_begin + kConvSeg : T;
  float taps[kConvMaxK], win[kConvMaxK];
  for (int k = 0; k < K; ++k) taps[k] = bf2f(w[(size_t)c * K + k]);
  for (int i = 0; i < sw; ++i) {   // the K-1 inputs that precede the segment
    const int src_t = t_begin - sw + i;
    win[i] = src_t >= 0 ? bf2f(x[(size_t)src_t * C + c]) : bf2f(state[(size_t)c * sw + (sw + src_t)]);
  }
  for (int t = t_begin; t < t_end; ++t) {
    win[sw] = bf2f(x[(size_t)t * C + c]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += win[k] * taps[k];
    out[(size_t)t * C + c] = f2bf(silu_f(bf16_round_f(sum)));
    for (int i = 0; i < sw; ++i) win[i] = win[i + 1];
  }
}

// 8 channels x 1 token per lane (C % 8 == 0, K == 4, 16-byte aligned): 16-byte loads of the K source rows and of the
// 8 x 4 weights, one 16-byte store.  Per element the taps are summed in the same order as conv1d_window_kernel.
// TT consecutive tokens of 8 channels per lane (round 6: TT = 8 for prompts of >= 256 tokens): the K - 1 rows in front of a token are
// the previous tokens' rows, so a lane that walks TT tokens loads TT + 3 rows instead of 4 TT and its 8 x 4 taps once - 19 -> ~10 us
// for the 1024 x 8192 tile of a Qwen3.5 layer.  Per output the same taps in the same order: same bits.
template <int TT>
__global__ __launch_bounds__(256) void conv1d_vec4_kernel(const Half* __restrict__ x, const Half* __restrict__ w,
                                                          const Half* __restrict__ state, Half* __restrict__ out,
                                                          int C, int T) {
  constexpr int K = 4, sw = 3;
  const int cvec = C >> 3, tblocks = (T + TT - 1) / TT;
  const long total = (long)cvec * tblocks;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % cvec), t0 = (int)(idx / cvec) * TT, c0 = c8 * 8;
    float taps[8][K];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const u32x2 wv = *reinterpret_cast<const u32x2*>(w + (size_t)(c0 + e) * K);  // 4 taps of this channel
      taps[e][0] = bf_lo(wv.x); taps[e][1] = bf_hi(wv.x); taps[e][2] = bf_lo(wv.y); taps[e][3] = bf_hi(wv.y);
    }
    float xv[TT + sw][8];   // rows t0 - 3 .. t0 + TT - 1
#pragma unroll
    for (int r = 0; r < TT + sw; ++r) {
      const int src_t = t0 - sw + r;
      if (src_t >= 0) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (src_t < T) v = *reinterpret_cast<const u32x4*>(x + (size_t)src_t * C + c0);
        const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[r][2 * e] = bf_lo(ww[e]); xv[r][2 * e + 1] = bf_hi(ww[e]); }
      } else {
        const int si = sw + src_t;  // >= 0 because K - 1 == sw
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[r][e] = bf2f(state[(size_t)(c0 + e) * sw + si]);
      }
    }
#pragma unroll
    for (int tl = 0; tl < TT; ++tl) {
      if (t0 + tl >= T) break;
      uint32_t o[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = 2 * e2 + h;
          float sum = 0.f;
          sum += xv[tl + 0][e] * taps[e][0];
          sum += xv[tl + 1][e] * taps[e][1];
          sum += xv[tl + 2][e] * taps[e][2];
          sum += xv[tl + 3][e] * taps[e][3];
          r[h] = silu_f(bf16_round_f(sum));
        }
        o[e2] = pack_bf2(r[0], r[1]);
      }
      *reinterpret_cast<u32x4*>(out + (size_t)(t0 + tl) * C + c0) = u32x4{o[0], o[1], o[2], o[3]};
    }
  }
}
// T == 1 (decode, recurrent.rs:49-79): output and window shift in one pass, same arithmetic as conv1d_window_kernel
__global__ __launch_bounds__(256) void conv1d_step_kernel(const Half* __restrict__ x, const Half* __restrict__ w,
                                                          Half* __restrict__ state, Half* __restrict__ out, int C,
                                                          int K) {
  const int sw = K - 1;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  Half win[kConvMaxK];
  for (int i = 0; i < sw; ++i) win[i] = state[(size_t)c * sw + i];
  win[sw] = x[c];
  float sum = 0.f;
  for (int k = 0; k < K; ++k) sum += bf2f(win[k]) * bf2f(w[(size_t)c * K + k]);
  out[c] = f2bf(silu_f(bf16_round_f(sum)));
  for (int i = 0; i < sw; ++i) state[(size_t)c * sw + i] = win[i + 1];
}
__global__ __launch_bounds__(256) void conv1d_state_kernel(const Half* __restrict__ x, Half* __restrict__ state,
                                                           int C, int T, int K) {
  const int sw = K - 1;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  Half old[kConvMaxK];
  for (int i = 0; i < sw; ++i) old[i] = state[(size_t)c * sw + i];
  for (int i = 0; i < sw; ++i) {
    const int src_t = T - sw + i;
    Half v;
    if (src_t >= 0) v = x[(size_t)src_t * C + c];
    else { const int si = sw + src_t; v = si >= 0 ? old[si] : (Half)0; }
    state[(size_t)c * sw + i] = v;
  }
}

// =====================================================================================================
// gated_delta_rule_decode_cuda (reference csrc/gated_delta_rule.cu:27-190), per value head h (key head
// kh = h * num_key_heads / num_value_heads), all fp32:
//   q = l2norm(q_kh) / sqrt(key_dim), k = l2norm(k_kh)           (eps 1e-12 inside the rsqrt)
//   g = -exp(A_log_h) * softplus(a_h + dt_bias_h), beta = sigmoid(b_h)
//   S *= exp(g);  kv[v] = sum_j S[j][v] k[j];  delta[v] = (v_h[v] - kv[v]) * beta
//   S[j][v] += k[j] * delta[v];  o[v] = sum_j S[j][v] q[j]      (state S: [key_dim][val_dim] f32, v contiguous)
// The state (64 KB per head, read + written once per token) is the traffic: HBM-bound.  The reference uses
// one block per head (32 blocks); value columns are independent given q and k, so here a head is split into
// VB = val_dim/32 column blocks -> 128 workgroups, and each workgroup keeps its [key_dim x 32] slice in
// registers between the two passes: the state crosses HBM exactly once in each direction (reference: twice).
// =====================================================================================================
constexpr int kGdrCols = 32;  // value columns per workgroup (8 lanes x 16 B per state row)

__global__ __launch_bounds__(256) void gdr_decode_kernel(const Half* __restrict__ qkv, const Half* __restrict__ b_proj,
                                                         const Half* __restrict__ a_proj,
                                                         const Half* __restrict__ dt_bias,
                                                         const float* __restrict__ A_log, float* __restrict__ state,
                                                         Half* __restrict__ output, int num_key_heads,
                                                         int num_value_heads, int key_dim, int val_dim) {
  __shared__ float sq[128], sk[128];
  __shared__ float red[8];
  __shared__ float colred[4][kGdrCols];
  const int vblocks = val_dim / kGdrCols;
  const int vh = blockIdx.x / vblocks, vb = blockIdx.x % vblocks;
  const int kh = vh * num_key_heads / num_value_heads;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q_total = key_dim * num_key_heads;
  // ---- q / k L2 norms (key_dim == 128: threads 0..127 hold one element each) ----
  float qv = 0.f, kv = 0.f;
  if (tid < key_dim) {
    qv = bf2f(qkv[(size_t)kh * key_dim + tid]);
    kv = bf2f(qkv[(size_t)q_total + (size_t)kh * key_dim + tid]);
  }
  float q2 = wave_sum(qv * qv), k2 = wave_sum(kv * kv);
  if (lane == 0) { red[wave] = q2; red[4 + wave] = k2; }
  __syncthreads();
  const float qn = rsqrtf(red[0] + red[1] + red[2] + red[3] + 1e-12f) * rsqrtf((float)key_dim);
  const float kn = rsqrtf(red[4] + red[5] + red[6] + red[7] + 1e-12f);
  if (tid < key_dim) { sq[tid] = qv * qn; sk[tid] = kv * kn; }
  // ---- gate scalars ----
  const float a_val = bf2f(a_proj[vh]), b_val = bf2f(b_proj[vh]), bias = bf2f(dt_bias[vh]);
  const float xs = a_val + bias;
  const float softplus = xs > 20.0f ? xs : logf(1.0f + expf(xs));
  const float exp_g = expf(-expf(A_log[vh]) * softplus);
  const float beta = 1.0f / (1.0f + expf(-b_val));
  __syncthreads();
  // ---- state slice: rows j = it*32 + wave*8 + lane/8, columns vb*32 + (lane%8)*4 .. +3 ----
  const int cg = lane & 7, rsub = lane >> 3;
  const int col0 = vb * kGdrCols + cg * 4;
  float* sbase = state + (size_t)vh * key_dim * val_dim + col0;
  f32x4 s[4];
  f32x4 kvp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int j = it * 32 + wave * 8 + rsub;
    s[it] = *reinterpret_cast<const f32x4*>(sbase + (size_t)j * val_dim);
    s[it] *= exp_g;
    kvp += s[it] * sk[j];
  }
  // column sums: over the 8 row-lanes of the wave (xor 8, 16, 32), then over the 4 waves through LDS
  auto col_reduce = [&](f32x4 v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = v[c];
      x += __shfl_xor(x, 8, kWave);
      x += __shfl_xor(x, 16, kWave);
      x += __shfl_xor(x, 32, kWave);
      v[c] = x;
    }
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) colred[wave][cg * 4 + c] = v[c];
    }
    __syncthreads();
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      r[c] = colred[0][cg * 4 + c] + colred[1][cg * 4 + c] + colred[2][cg * 4 + c] + colred[3][cg * 4 + c];
    return r;
  };
  const f32x4 kvm = col_reduce(kvp);
  const Half* vrow = qkv + (size_t)2 * q_total + (size_t)vh * val_dim + col0;
  f32x4 delta;
#pragma unroll
  for (int c = 0; c < 4; ++c) delta[c] = (bf2f(vrow[c]) - kvm[c]) * beta;
  f32x4 op = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int j = it * 32 + wave * 8 + rsub;
    s[it] += delta * sk[j];
    *reinterpret_cast<f32x4*>(sbase + (size_t)j * val_dim) = s[it];
    op += s[it] * sq[j];
  }
  const f32x4 o = col_reduce(op);
  if (wave == 0 && rsub == 0) {
    Half* dst = output + (size_t)vh * val_dim + col0;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = f2bf(o[c]);
  }
}

// =====================================================================================================
// (round 6) One decode step of a linear-attention layer's token mixer in ONE launch (extension, include/pegainfer_kernels_ext.h):
// conv1d_prefill_cuda(seq_len 1) -> gated_delta_rule_decode_cuda -> rms_norm_gated_cuda, i.e. recurrent.rs:49-79 + the gated norm
// that follows it.  In the bs-1 decode trace the three launches are 4.7 + 4.8 + 4.8 us per layer (24 layers) for ~4.3 MB of state
// traffic: launch latency, not bytes.  One 256-thread workgroup per VALUE head: it walks the four 32-column blocks the stand-alone
// kernel gives to four workgroups, with the same lane -> (row, column) map and the same order inside every sum, and one wave
// normalises the head exactly as rms_norm_gated_kernel does - bit-identical to the three calls (tests/test_gpu_qwen35.py).
// The conv windows: a head's v channels are its own; the q / k channels of its key head are read by all VPK value heads of
// that key head, so their windows are shifted by whichever of those workgroups ARRIVES LAST at the key head's ticket (every one of
// them holds the same old window and the same new input; the ticket resets itself).  A first form with one workgroup per KEY head
// (no ticket, 16 workgroups x 256 KB of state) took 8.7 us - 16 CUs cannot move 4.3 MB any faster - and bought nothing.
// =====================================================================================================
__global__ __launch_bounds__(256) void linattn_decode_fused_kernel(
    const Half* __restrict__ x, const Half* __restrict__ conv_w, Half* __restrict__ conv_state, const Half* __restrict__ b_proj,
    const Half* __restrict__ a_proj, const Half* __restrict__ dt_bias, const float* __restrict__ A_log, float* __restrict__ state,
    const float* __restrict__ norm_w, const Half* __restrict__ gate, Half* __restrict__ out, int num_key_heads, int vpk, int K,
    float eps, int* __restrict__ ticket) {
  constexpr int KD = 128, VD = 128, VB = VD / kGdrCols;
  __shared__ float cv[VD];                 // conv outputs (bf16 values) of this head's v channels
  __shared__ float sq[KD], sk[KD];
  __shared__ float red[8];
  __shared__ float colred[4][VD];
  __shared__ float so[VD];                 // bf16-rounded outputs of the recurrence
  __shared__ int last_flag;
  const int vh = blockIdx.x, kh = vh / vpk, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q_total = KD * num_key_heads;
  const int sw = K - 1;
  // the state slice first: its 64 KB are the long pole, everything below runs under these loads
  const int cg = lane & 7, rsub = lane >> 3;
  float* sbase = state + (size_t)vh * KD * VD + cg * 4;
  f32x4 s[VB][4], acc[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int j = it * 32 + wave * 8 + rsub;
      s[vb][it] = *reinterpret_cast<const f32x4*>(sbase + (size_t)j * VD + vb * kGdrCols);
    }
  // the gate scalars and the norm's gate / weight rows: requested now, used at the very end
  const float a_val = bf2f(a_proj[vh]), b_val = bf2f(b_proj[vh]), bias = bf2f(dt_bias[vh]), a_log = A_log[vh];
  float gpre[2] = {0.f, 0.f}, wpre[2] = {0.f, 0.f};
  if (wave == 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) { gpre[u] = bf2f(gate[(size_t)vh * VD + lane + 64 * u]); wpre[u] = norm_w[lane + 64 * u]; }
  }
  // ---- conv1d step (conv1d_step_kernel's arithmetic): threads 0..127 the key head's q AND k channel of their index, every
  //      thread < 128 also one v channel of this head ----
  auto conv = [&](int c, Half (&win)[kConvMaxK]) {
    for (int k = 0; k < sw; ++k) win[k] = conv_state[(size_t)c * sw + k];
    win[sw] = x[c];
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += bf2f(win[k]) * bf2f(conv_w[(size_t)c * K + k]);
    return bf2f(f2bf(silu_f(bf16_round_f(sum))));
  };
  float qv = 0.f, kv = 0.f;
  Half qwin[kConvMaxK], kwin[kConvMaxK];
  const int qc = kh * KD + tid, kc = q_total + kh * KD + tid;
  if (tid < KD) {
    qv = conv(qc, qwin);
    kv = conv(kc, kwin);
  } else {
    const int i = tid - KD, c = 2 * q_total + vh * VD + i;
    Half vwin[kConvMaxK];
    cv[i] = conv(c, vwin);
    for (int k = 0; k < sw; ++k) conv_state[(size_t)c * sw + k] = vwin[k + 1];
  }
  // every q / k window of this key head has been read by this workgroup (the values were consumed above): arrive
  __syncthreads();
  // (relaxed: the last arriver reads nothing the others wrote - it only must not WRITE before they have read, and their reads were
  // consumed before they arrived; the returned count is looked at after the recurrence, so nobody waits for the round trip)
  int arrived = 0;
  if (tid == 0) arrived = __hip_atomic_fetch_add(ticket + kh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float q2 = wave_sum(qv * qv), k2 = wave_sum(kv * kv);
  if (lane == 0) { red[wave] = q2; red[4 + wave] = k2; }
  __syncthreads();
  const float qn = rsqrtf(red[0] + red[1] + red[2] + red[3] + 1e-12f) * rsqrtf((float)KD);
  const float kn = rsqrtf(red[4] + red[5] + red[6] + red[7] + 1e-12f);
  if (tid < KD) { sq[tid] = qv * qn; sk[tid] = kv * kn; }
  const float xs = a_val + bias;
  const float softplus = xs > 20.0f ? xs : logf(1.0f + expf(xs));
  const float exp_g = expf(-expf(a_log) * softplus);
  const float beta = 1.0f / (1.0f + expf(-b_val));
  __syncthreads();
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    acc[vb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int j = it * 32 + wave * 8 + rsub;
      s[vb][it] *= exp_g;
      acc[vb] += s[vb][it] * sk[j];
    }
  }
  // column sums: over the 8 row-lanes of a wave (xor 8, 16, 32), then over the 4 waves through LDS, in wave order
  auto col_reduce = [&](f32x4 (&v)[VB]) {
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = v[vb][c];
        t += __shfl_xor(t, 8, kWave);
        t += __shfl_xor(t, 16, kWave);
        t += __shfl_xor(t, 32, kWave);
        v[vb][c] = t;
      }
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
      for (int vb = 0; vb < VB; ++vb)
#pragma unroll
        for (int c = 0; c < 4; ++c) colred[wave][vb * kGdrCols + cg * 4 + c] = v[vb][c];
    }
    __syncthreads();
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = vb * kGdrCols + cg * 4 + c;
        v[vb][c] = colred[0][col] + colred[1][col] + colred[2][col] + colred[3][col];
      }
  };
  col_reduce(acc);
  f32x4 delta[VB];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb)
#pragma unroll
    for (int c = 0; c < 4; ++c) delta[vb][c] = (cv[vb * kGdrCols + cg * 4 + c] - acc[vb][c]) * beta;
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    acc[vb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int j = it * 32 + wave * 8 + rsub;
      s[vb][it] += delta[vb] * sk[j];
      *reinterpret_cast<f32x4*>(sbase + (size_t)j * VD + vb * kGdrCols) = s[vb][it];
      acc[vb] += s[vb][it] * sq[j];
    }
  }
  col_reduce(acc);
  if (wave == 0 && rsub == 0) {
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int c = 0; c < 4; ++c) so[vb * kGdrCols + cg * 4 + c] = bf2f(f2bf(acc[vb][c]));
  }
  if (tid == 0) {
    last_flag = arrived == vpk - 1;
    if (arrived == vpk - 1) __hip_atomic_store(ticket + kh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // ---- gated RMSNorm of the head: one wave, lane l holds dims l and l + 64 (rms_norm_gated_kernel's order) ----
  if (wave == 0) {
    float ss = 0.f;
    for (int i = lane; i < VD; i += 64) ss += so[i] * so[i];
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)VD + eps);
    const size_t base = (size_t)vh * VD;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = lane + 64 * u;
      const float normed = so[i] * inv * wpre[u];
      const float g = gpre[u];
      out[base + i] = f2bf(normed * (g / (1.0f + expf(-g))));
    }
  }
  // the last of the key head's value heads shifts the shared q / k windows (last_flag: written in front of the norm's barrier)
  if (last_flag && tid < KD) {
    for (int k = 0; k < sw; ++k) {
      conv_state[(size_t)qc * sw + k] = qwin[k + 1];
      conv_state[(size_t)kc * sw + k] = kwin[k + 1];
    }
  }
}

// =====================================================================================================
// HD256 full-attention prep (reference csrc/prefill_attention_hd256.cu): q comes interleaved with its gate
// (q_full row = [q_h0(256) gate_h0(256) q_h1 ...]); (1+w) RMSNorm with ONE rounding, then partial NeoX RoPE
// on the first rotary_dim dims (pairs d, d + rotary_dim/2; table row = pos*rotary_dim), rest passes through.
// One wave per (head, token): lane l owns dims l*4..l*4+3 (8-byte accesses); rotary_dim = 64 lives in lanes
// 0..15, partner of dim d < 32 is d + 32 = 8 lanes up -> one bpermute per element.
// =====================================================================================================
struct Hd256Src { const Half* x; Half* dst; const Half* w; };

__device__ __forceinline__ void hd256_norm_rope_store(const Half* __restrict__ src, Half* __restrict__ dst,
                                                      const Half* __restrict__ w, const Half* __restrict__ crow,
                                                      const Half* __restrict__ srow, int rotary_dim, float eps,
                                                      int lane) {
  const u32x2 xv = *reinterpret_cast<const u32x2*>(src + lane * 4);
  const u32x2 wv = *reinterpret_cast<const u32x2*>(w + lane * 4);
  float v[4] = {bf_lo(xv.x), bf_hi(xv.x), bf_lo(xv.y), bf_hi(xv.y)};
  const float wf[4] = {bf_lo(wv.x), bf_hi(wv.x), bf_lo(wv.y), bf_hi(wv.y)};
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) ss += v[i] * v[i];
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss / 256.0f + eps);
  float n[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) n[i] = bf16_round_f(v[i] * inv * (1.0f + wf[i]));
  const int half = rotary_dim >> 1;
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane * 4 + i;
    // partner element lives (half/4) lanes away at the same i (half % 4 == 0 for rotary 64)
    const int plane = d < half ? lane + (half >> 2) : lane - (half >> 2);
    const float pn = __shfl(n[i], plane & 63, kWave);
    if (d < half) {
      const float c = bf2f(crow[d]), s = bf2f(srow[d]);
      o[i] = n[i] * c - pn * s;
    } else if (d < rotary_dim) {
      const float c = bf2f(crow[d - half]), s = bf2f(srow[d - half]);
      o[i] = pn * s + n[i] * c;
    } else {
      o[i] = n[i];
    }
  }
  u32x2 r;
  r.x = pack_bf2(o[0], o[1]);
  r.y = pack_bf2(o[2], o[3]);
  *reinterpret_cast<u32x2*>(dst + lane * 4) = r;
}

// grid.x = (num_q_heads + num_kv_heads) * tokens / 4 (4 waves per block)
template <bool PREFILL>
__global__ __launch_bounds__(256) void hd256_prep_kernel(const Half* __restrict__ q_full, Half* __restrict__ k_inout,
                                                         const Half* __restrict__ k_in, const Half* __restrict__ qw,
                                                         const Half* __restrict__ kw, const Half* __restrict__ cosc,
                                                         const Half* __restrict__ sinc, Half* __restrict__ q_out,
                                                         Half* __restrict__ k_cache, int hq, int hkv, int tokens,
                                                         const int* __restrict__ pos_src, int rotary_dim, float eps,
                                                         int max_seq_len) {
  const int heads = hq + hkv;
  const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= (long)tokens * heads) return;
  const int token = (int)(unit / heads), hg = (int)(unit % heads);
  const int lane = threadIdx.x & 63;
  const bool is_q = hg < hq;
  const int h = is_q ? hg : hg - hq;
  const int pos = PREFILL ? pos_src[0] + token : pos_src[token];
  const Half* crow = cosc + (size_t)pos * rotary_dim;
  const Half* srow = sinc + (size_t)pos * rotary_dim;
  if (is_q) {
    hd256_norm_rope_store(q_full + ((size_t)token * hq * 2 + (size_t)h * 2) * 256,
                          q_out + ((size_t)token * hq + h) * 256, qw, crow, srow, rotary_dim, eps, lane);
  } else if (PREFILL) {  // K goes straight into the contiguous HND scratch cache[head][pos][256]
    hd256_norm_rope_store(k_in + ((size_t)token * hkv + h) * 256,
                          k_cache + ((size_t)h * max_seq_len + pos) * 256, kw, crow, srow, rotary_dim, eps, lane);
  } else {               // decode: K normalised + rotated in place
    Half* kp = k_inout + ((size_t)token * hkv + h) * 256;
    hd256_norm_rope_store(kp, kp, kw, crow, srow, rotary_dim, eps, lane);
  }
}

__global__ __launch_bounds__(256) void hd256_v_write_kernel(const Half* __restrict__ v, Half* __restrict__ v_cache,
                                                            int hkv, int tokens, const int* __restrict__ start_pos,
                                                            int max_seq_len) {
  const long nvec = (long)tokens * hkv * 32;  // 16-byte chunks
  const int sp = start_pos[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const int c = (int)(i & 31);
    const long th = i >> 5;
    const int h = (int)(th % hkv), t = (int)(th / hkv);
    reinterpret_cast<u32x4*>(v_cache + ((size_t)h * max_seq_len + sp + t) * 256)[c] =
        reinterpret_cast<const u32x4*>(v + ((size_t)t * hkv + h) * 256)[c];
  }
}

// attn_out *= sigmoid(gate), gate = the second half of each q_full head (prefill_attention_hd256.cu:135-157)
__global__ __launch_bounds__(256) void hd256_gate_kernel(const Half* __restrict__ q_full, Half* __restrict__ attn,
                                                         int hq, int tokens) {
  const long nvec = (long)tokens * hq * 32;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const int c = (int)(i & 31);
    const long th = i >> 5;  // token * hq + head
    const u32x4 g = reinterpret_cast<const u32x4*>(q_full + ((size_t)th * 2 + 1) * 256)[c];
    u32x4* ap = reinterpret_cast<u32x4*>(attn + (size_t)th * 256) + c;
    const u32x4 a = *ap;
    auto f = [](uint32_t aw, uint32_t gw) {
      return pack_bf2(bf_lo(aw) * (1.0f / (1.0f + expf(-bf_lo(gw)))), bf_hi(aw) * (1.0f / (1.0f + expf(-bf_hi(gw)))));
    };
    u32x4 o;
    o.x = f(a.x, g.x); o.y = f(a.y, g.y); o.z = f(a.z, g.z); o.w = f(a.w, g.w);
    *ap = o;
  }
}

static inline int grid_cap(long work, int per_block) {
  long g = (work + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace pk

using namespace pk;

extern "C" {

void conv1d_prefill_cuda(const Half* x_seq, const Half* conv_weight, Half* conv_state, Half* out_seq,
                         int32_t num_channels, int32_t seq_len, int32_t kernel_size, pegainfer_stream_t stream) {
  if (num_channels <= 0 || seq_len <= 0 || kernel_size < 1 || kernel_size > kConvMaxK) return;
  hipStream_t s = as_stream(stream);
  if (seq_len == 1 && kernel_size > 1) {  // decode step: a channel's window is private to one lane -> one launch
    conv1d_step_kernel<<<ceil_div(num_channels, 256), 256, 0, s>>>(x_seq, conv_weight, conv_state, out_seq,
                                                                   num_channels, kernel_size);
    return;
  }
  if (kernel_size == 4 && (num_channels & 7) == 0 && host_aligned16(x_seq) && host_aligned16(out_seq) &&
      (reinterpret_cast<uintptr_t>(conv_weight) & 7u) == 0)
    if (seq_len >= 256)
      conv1d_vec4_kernel<8><<<grid_cap((long)(num_channels >> 3) * ((seq_len + 7) / 8), 256), 256, 0, s>>>(
          x_seq, conv_weight, conv_state, out_seq, num_channels, seq_len);
    else
      conv1d_vec4_kernel<1><<<grid_cap((long)(num_channels >> 3) * seq_len, 256), 256, 0, s>>>(
          x_seq, conv_weight, conv_state, out_seq, num_channels, seq_len);
  else
    conv1d_window_kernel<<<dim3(ceil_div(num_channels, 256), ceil_div(seq_len, kConvSeg)), 256, 0, s>>>(
        x_seq, conv_weight, conv_state, out_seq, num_channels, seq_len, kernel_size);
  if (kernel_size > 1)
    conv1d_state_kernel<<<ceil_div(num_channels, 256), 256, 0, s>>>(x_seq, conv_state, num_channels, seq_len,
                                                                    kernel_size);
}

void gated_delta_rule_decode_cuda(const Half* qkv, const Half* b_proj, const Half* a_proj, const Half* dt_bias,
                                  const float* A_log, float* state, Half* output, int32_t num_key_heads,
                                  int32_t num_value_heads, int32_t key_dim, int32_t val_dim,
                                  pegainfer_stream_t stream) {
  if (key_dim != 128 || val_dim % kGdrCols != 0 || num_value_heads <= 0) return;  // GDR_KEY_DIM = 128 in the reference
  gdr_decode_kernel<<<num_value_heads * (val_dim / kGdrCols), 256, 0, as_stream(stream)>>>(
      qkv, b_proj, a_proj, dt_bias, A_log, state, output, num_key_heads, num_value_heads, key_dim, val_dim);
}

// extension (include/pegainfer_kernels_ext.h): conv1d_prefill_cuda(seq_len 1) + gated_delta_rule_decode_cuda + rms_norm_gated_cuda
// of ONE request's linear-attention layer in one launch, same bits.  hipErrorInvalidValue = a shape this kernel does not take
// (the caller then runs the three calls).
int32_t pegainfer_linear_attn_decode_fused(const Half* x_qkv, const Half* conv_weight, Half* conv_state, const Half* b_proj,
                                           const Half* a_proj, const Half* dt_bias, const float* A_log, float* state,
                                           const float* norm_weight, const Half* gate, Half* out, int32_t num_key_heads,
                                           int32_t num_value_heads, int32_t key_dim, int32_t val_dim, int32_t kernel_size,
                                           float eps, int32_t* tickets, pegainfer_stream_t stream) {
  if (!x_qkv || !conv_weight || !conv_state || !b_proj || !a_proj || !dt_bias || !A_log || !state || !norm_weight || !gate ||
      !out || !tickets || key_dim != 128 || val_dim != 128 || num_key_heads <= 0 || num_value_heads <= 0 ||
      num_value_heads % num_key_heads != 0 || kernel_size < 2 || kernel_size > kConvMaxK ||
      (reinterpret_cast<uintptr_t>(state) & 15u))
    return static_cast<int32_t>(hipErrorInvalidValue);
  linattn_decode_fused_kernel<<<num_value_heads, 256, 0, as_stream(stream)>>>(
      x_qkv, conv_weight, conv_state, b_proj, a_proj, dt_bias, A_log, state, norm_weight, gate, out, num_key_heads,
      num_value_heads / num_key_heads, kernel_size, eps, tickets);
  return static_cast<int32_t>(hipGetLastError());
}

void prefill_attention_hd256_prep_cuda(const Half* q_full_batch, const Half* k_batch, const Half* v_batch,
                                       const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache,
                                       const Half* sin_cache, Half* q_batch_out, Half* k_cache, Half* v_cache,
                                       int32_t num_q_heads, int32_t num_kv_heads, int32_t seq_len,
                                       const int32_t* start_pos_ptr, int32_t rotary_dim, float rms_eps,
                                       int32_t max_seq_len, pegainfer_stream_t stream) {
  if (seq_len <= 0) return;
  hipStream_t s = as_stream(stream);
  const long units = (long)(num_q_heads + num_kv_heads) * seq_len;
  hd256_prep_kernel<true><<<ceil_div(units, 4), 256, 0, s>>>(q_full_batch, nullptr, k_batch, q_norm_weight,
                                                             k_norm_weight, cos_cache, sin_cache, q_batch_out, k_cache,
                                                             num_q_heads, num_kv_heads, seq_len, start_pos_ptr,
                                                             rotary_dim, rms_eps, max_seq_len);
  hd256_v_write_kernel<<<grid_cap((long)seq_len * num_kv_heads * 32, 256), 256, 0, s>>>(
      v_batch, v_cache, num_kv_heads, seq_len, start_pos_ptr, max_seq_len);
}

void qk_norm_partial_rope_batched_decode_hd256_cuda(const Half* q_full_batch, Half* k_batch,
                                                    const Half* q_norm_weight, const Half* k_norm_weight,
                                                    const Half* cos_cache, const Half* sin_cache,
                                                    const int32_t* positions, Half* q_batch_out, int32_t num_q_heads,
                                                    int32_t num_kv_heads, int32_t batch_size, int32_t rotary_dim,
                                                    float rms_eps, pegainfer_stream_t stream) {
  if (batch_size <= 0) return;
  const long units = (long)(num_q_heads + num_kv_heads) * batch_size;
  hd256_prep_kernel<false><<<ceil_div(units, 4), 256, 0, as_stream(stream)>>>(
      q_full_batch, k_batch, nullptr, q_norm_weight, k_norm_weight, cos_cache, sin_cache, q_batch_out, nullptr,
      num_q_heads, num_kv_heads, batch_size, positions, rotary_dim, rms_eps, 0);
}

void attention_gate_batch_hd256_cuda(const Half* q_full_batch, Half* attn_out, int32_t num_q_heads, int32_t seq_len,
                                     pegainfer_stream_t stream) {
  if (seq_len <= 0) return;
  hd256_gate_kernel<<<grid_cap((long)seq_len * num_q_heads * 32, 256), 256, 0, as_stream(stream)>>>(
      q_full_batch, attn_out, num_q_heads, seq_len);
}

}  // extern "C"
