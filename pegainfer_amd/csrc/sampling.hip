// Sampling for gfx950: argmax, greedy top-1, temperature/top-k/top-p multinomial.
//
//   argmax_cuda            1 workgroup x 1024 lanes, 16-byte loads, lowest index wins ties
//                          (reference csrc/argmax.cu:18,29-31).
//   flashinfer_top1_cuda   64 workgroups; each folds its slice into a packed
//                          (orderable bf16 key << 32 | ~index) word and device-scope atomic-maxes it
//                          into the caller's row_states scratch; the last arriver (ticket counter,
//                          acq_rel) writes the token + top value and re-zeroes both words, so the
//                          scratch keeps the "zero-initialised once" contract of the reference
//                          (ops/sampling.rs:7).  Replaces FlashInfer TopKDispatch(k=1); ties ->
//                          lowest index.
//   gpu_sample_flashinfer_cuda  softmax(logits*inv_T) -> fp32 probs (reference
//                          csrc/flashinfer_sampling.cu:13-70), then joint top-k / top-p filter on the
//                          original distribution (FlashInfer TopKTopPSamplingFromProb semantics,
//                          restated) by a 3-pass radix select over the fp32 bit patterns, then one
//                          inverse-CDF draw in index order.  RNG = splitmix64(seed); the reference's
//                          Philox stream lives in un-vendored FlashInfer -> distribution parity only.
#include "common.h"
#include "pegainfer_kernels_ext.h"

namespace pk {

__device__ __forceinline__ uint32_t bf16_order_key(uint16_t h) {
  if ((h & 0x7FFF) > 0x7F80) return 0;           // NaN never wins (reference: comparisons false)
  if (h == 0x8000) h = 0;                        // -0 == +0
  return (h & 0x8000) ? (uint32_t)(uint16_t)~h : (uint32_t)(h | 0x8000);
}
__device__ __forceinline__ unsigned long long pack_key(uint16_t h, int idx) {
  return ((unsigned long long)bf16_order_key(h) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)idx);
}
__device__ __forceinline__ unsigned long long u64_max(unsigned long long a, unsigned long long b) {
  return a > b ? a : b;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, kWave);
    v = u64_max(v, o);
  }
  return v;
}

// local best over x[begin:end) strided by the whole grid of threads, 8 elements per 16-byte load
__device__ __forceinline__ unsigned long long scan_best(const Half* __restrict__ x, int n, int tid, int nthreads) {
  unsigned long long best = 0;
  if (aligned16(x)) {
    const int nvec = n >> 3;
    for (int i = tid; i < nvec; i += nthreads) {
      u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        best = u64_max(best, pack_key((uint16_t)(w[j] & 0xFFFF), i * 8 + 2 * j));
        best = u64_max(best, pack_key((uint16_t)(w[j] >> 16), i * 8 + 2 * j + 1));
      }
    }
    for (int i = (nvec << 3) + tid; i < n; i += nthreads) best = u64_max(best, pack_key(x[i], i));
  } else {
    for (int i = tid; i < n; i += nthreads) best = u64_max(best, pack_key(x[i], i));
  }
  return best;
}

template <int WAVES>
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long* smem) {
  v = wave_max_u64(v);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = smem[0];
#pragma unroll
  for (int i = 1; i < WAVES; ++i) r = u64_max(r, smem[i]);
  return r;
}

__global__ __launch_bounds__(1024) void argmax_kernel(const Half* __restrict__ x, int* __restrict__ out, int n) {
  __shared__ unsigned long long red[16];
  unsigned long long best = scan_best(x, n, threadIdx.x, 1024);
  best = block_max_u64<16>(best, red);
  if (threadIdx.x == 0) out[0] = n > 0 ? (int)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFu)) : 0;
}

constexpr int kTop1Blocks = 64;
__global__ __launch_bounds__(256) void top1_kernel(const Half* __restrict__ logits, Half* __restrict__ top_value,
                                                   unsigned long long* __restrict__ state, int* __restrict__ out,
                                                   int n) {
  __shared__ unsigned long long red[4];
  unsigned long long best = scan_best(logits, n, blockIdx.x * 256 + threadIdx.x, kTop1Blocks * 256);
  best = block_max_u64<4>(best, red);
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_max(&state[0], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long ticket =
        __hip_atomic_fetch_add(&state[1], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == kTop1Blocks - 1) {
      const unsigned long long win = __hip_atomic_exchange(&state[0], 0ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[1], 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const int idx = (int)(0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFu));
      out[0] = idx;
      if (top_value) top_value[0] = logits[idx];
    }
  }
}

// Batched greedy: grid (kTop1Blocks, rows); row r uses state words [2r, 2r+1] (self-resetting).
__global__ __launch_bounds__(256) void batched_top1_kernel(const Half* __restrict__ logits, long row_stride,
                                                           unsigned long long* __restrict__ state,
                                                           int* __restrict__ out, int n) {
  __shared__ unsigned long long red[4];
  const int row = blockIdx.y;
  const Half* x = logits + (size_t)row * row_stride;
  unsigned long long* st = state + 2 * (size_t)row;
  unsigned long long best = scan_best(x, n, blockIdx.x * 256 + threadIdx.x, kTop1Blocks * 256);
  best = block_max_u64<4>(best, red);
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_max(&st[0], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long ticket = __hip_atomic_fetch_add(&st[1], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == kTop1Blocks - 1) {
      const unsigned long long win = __hip_atomic_exchange(&st[0], 0ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&st[1], 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      out[row] = (int)(0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFu));
    }
  }
}

// ---------------------------------------------------------------- temperature / top-k / top-p
constexpr int kSampleBlock = 1024;

__device__ __forceinline__ float block_reduce_f(float v, bool is_max, float* smem16) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem16[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = smem16[0];
  for (int i = 1; i < kSampleBlock / 64; ++i) r = is_max ? fmaxf(r, smem16[i]) : r + smem16[i];
  return r;
}

// Largest fp32 bit pattern v such that  weight(p >= v) >= target  (weight = count or mass).
__device__ uint32_t radix_threshold(const float* __restrict__ probs, int n, float target, bool by_mass,
                                    float* hist /* 2048 */, uint32_t* sh_prefix, float* sh_above) {
  if (threadIdx.x == 0) { *sh_prefix = 0u; *sh_above = 0.f; }
  int shift = 32;
  for (int pass = 0; pass < 3; ++pass) {
    const int bits = pass < 2 ? 11 : 10;
    const int hi_shift = shift;  // bits above this pass
    shift -= bits;
    const int nb = 1 << bits;
    for (int i = threadIdx.x; i < nb; i += kSampleBlock) hist[i] = 0.f;
    __syncthreads();
    const uint32_t prefix = *sh_prefix;
    for (int i = threadIdx.x; i < n; i += kSampleBlock) {
      const float p = probs[i];
      const uint32_t u = __builtin_bit_cast(uint32_t, p);
      const bool match = pass == 0 || (u >> hi_shift) == (prefix >> hi_shift);
      if (match) atomicAdd(&hist[(u >> shift) & (nb - 1)], by_mass ? p : 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float acc = *sh_above;
      int chosen = 0;
      for (int b = nb - 1; b >= 0; --b) {
        const float h = hist[b];
        if (acc + h >= target) { chosen = b; break; }
        acc += h;
      }
      *sh_above = acc;
      *sh_prefix = prefix | ((uint32_t)chosen << shift);
    }
    __syncthreads();
  }
  return *sh_prefix;
}

__device__ __forceinline__ float uniform01(uint64_t seed) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0,1)
}

__global__ __launch_bounds__(kSampleBlock) void sample_kernel(const Half* __restrict__ logits,
                                                               float* __restrict__ probs,
                                                               uint8_t* __restrict__ valid, int* __restrict__ out,
                                                               int n, float inv_t, int top_k, float top_p,
                                                               uint64_t seed) {
  __shared__ float red[16];
  __shared__ float hist[2048];
  __shared__ uint32_t sh_prefix;
  __shared__ float sh_above;
  __shared__ float sh_run;
  __shared__ int sh_pick;
  const int tid = threadIdx.x;
  // softmax (reference logits_to_probs_kernel)
  float m = -INFINITY;
  for (int i = tid; i < n; i += kSampleBlock) {
    const float v = bf2f(logits[i]) * inv_t;
    probs[i] = v;
    m = fmaxf(m, v);
  }
  m = block_reduce_f(m, true, red);
  float s = 0.f;
  for (int i = tid; i < n; i += kSampleBlock) {
    const float e = expf(probs[i] - m);
    probs[i] = e;
    s += e;
  }
  s = block_reduce_f(s, false, red);
  const float inv_sum = 1.0f / s;
  for (int i = tid; i < n; i += kSampleBlock) probs[i] *= inv_sum;
  __syncthreads();

  // joint filter thresholds on the ORIGINAL distribution
  uint32_t thr = 0u;
  if (top_k > 0 && top_k < n) thr = radix_threshold(probs, n, (float)top_k, false, hist, &sh_prefix, &sh_above);
  if (top_p < 1.0f) {
    const uint32_t tp = radix_threshold(probs, n, top_p, true, hist, &sh_prefix, &sh_above);
    thr = tp > thr ? tp : thr;
  }
  const float thr_f = __builtin_bit_cast(float, thr);

  float kept = 0.f;
  for (int i = tid; i < n; i += kSampleBlock) kept += probs[i] >= thr_f ? probs[i] : 0.f;
  kept = block_reduce_f(kept, false, red);
  const float target = uniform01(seed) * kept;

  // inverse CDF in index order: chunks of 1024 consecutive tokens, wave scan + wave offsets
  if (tid == 0) { sh_run = 0.f; sh_pick = 0x7FFFFFFF; }
  __syncthreads();
  int last_kept = -1;
  for (int base = 0; base < n; base += kSampleBlock) {
    const int i = base + tid;
    const float p = (i < n && probs[i] >= thr_f) ? probs[i] : 0.f;
    if (p > 0.f) last_kept = i;
    float incl = p;  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(incl, off, kWave);
      if ((tid & 63) >= off) incl += o;
    }
    if ((tid & 63) == 63) red[tid >> 6] = incl;
    __syncthreads();
    float woff = sh_run;
    for (int w = 0; w < (tid >> 6); ++w) woff += red[w];
    const float cum = woff + incl;
    if (p > 0.f && cum > target && cum - p <= target) atomicMin(&sh_pick, i);
    __syncthreads();
    if (sh_pick != 0x7FFFFFFF) break;
    if (tid == 0) {
      float t = sh_run;
      for (int w = 0; w < kSampleBlock / 64; ++w) t += red[w];
      sh_run = t;
    }
    __syncthreads();
  }
  if (sh_pick == 0x7FFFFFFF) {  // rounding left the target above the running sum: take the last kept token
    __shared__ int sh_last;
    if (tid == 0) sh_last = -1;
    __syncthreads();
    if (last_kept >= 0) atomicMax(&sh_last, last_kept);
    __syncthreads();
    if (tid == 0) sh_pick = sh_last < 0 ? 0 : sh_last;
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = sh_pick;
    if (valid) valid[0] = 1;
  }
}

}  // namespace pk

using namespace pk;

extern "C" {

void argmax_cuda(const Half* x, int32_t* out, int32_t n, pegainfer_stream_t stream) {
  argmax_kernel<<<1, 1024, 0, as_stream(stream)>>>(x, out, n);
}

void flashinfer_top1_cuda(const Half* logits, Half* top1_value_scratch, uint8_t* row_states_scratch,
                          int32_t* output, int32_t vocab_size, pegainfer_stream_t stream) {
  top1_kernel<<<kTop1Blocks, 256, 0, as_stream(stream)>>>(
      logits, top1_value_scratch, reinterpret_cast<unsigned long long*>(row_states_scratch), output, vocab_size);
}

pegainfer_status_t pegainfer_batched_top1(const Half* logits, int32_t vocab_size, int32_t rows, int64_t row_stride,
                                          uint8_t* state_scratch, int32_t* out_tokens, pegainfer_stream_t stream) {
  if (rows <= 0) return 0;
  batched_top1_kernel<<<dim3(kTop1Blocks, rows), 256, 0, as_stream(stream)>>>(
      logits, row_stride, reinterpret_cast<unsigned long long*>(state_scratch), out_tokens, vocab_size);
  return static_cast<pegainfer_status_t>(hipGetLastError());
}

void gpu_sample_flashinfer_cuda(const Half* logits, float* probs_scratch, uint8_t* valid_scratch,
                                int32_t* output, int32_t vocab_size, float inv_temperature, int32_t top_k,
                                float top_p, uint64_t seed, pegainfer_stream_t stream) {
  sample_kernel<<<1, kSampleBlock, 0, as_stream(stream)>>>(logits, probs_scratch, valid_scratch, output,
                                                           vocab_size, inv_temperature, top_k, top_p, seed);
}

}  // extern "C"
