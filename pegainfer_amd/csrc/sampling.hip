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
// Publish a workgroup's best key and draw its ticket.  Only atomics touch the two state words and agent-scope
// atomics are performed at the device coherence point, so no cache-wide release / acquire is needed (an ACQ_REL
// ticket cost an L2 write-back + invalidate per workgroup: 114 us for 64 rows x 64 workgroups).  What must hold is
// that this workgroup's max is performed before its ticket: the RETURNING fetch_max is waited for (vmcnt) first.
__device__ __forceinline__ unsigned long long top1_publish(unsigned long long* st, unsigned long long best) {
  const unsigned long long prev = __hip_atomic_fetch_max(&st[0], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(prev) : "memory");   // uses prev: the returning form, completed
  return __hip_atomic_fetch_add(&st[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void top1_kernel(const Half* __restrict__ logits, Half* __restrict__ top_value,
                                                   unsigned long long* __restrict__ state, int* __restrict__ out,
                                                   int n) {
  __shared__ unsigned long long red[4];
  unsigned long long best = scan_best(logits, n, blockIdx.x * 256 + threadIdx.x, kTop1Blocks * 256);
  best = block_max_u64<4>(best, red);
  if (threadIdx.x == 0) {
    const unsigned long long ticket = top1_publish(state, best);
    if (ticket == kTop1Blocks - 1) {
      const unsigned long long win = __hip_atomic_exchange(&state[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const unsigned long long ticket = top1_publish(st, best);
    if (ticket == kTop1Blocks - 1) {
      const unsigned long long win = __hip_atomic_exchange(&st[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&st[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out[row] = (int)(0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFu));
    }
  }
}

// ---------------------------------------------------------------- temperature / top-k / top-p
// One workgroup of 1024 lanes, 7 vectorised sweeps of the vocabulary (L2-resident after the first):
//   softmax: max, sum, write  |  3 radix passes that build the top-k COUNT and the top-p MASS histograms together
//   |  one sweep that folds the kept mass into 256-token chunk sums, then a block scan over the chunks and a wave
//   scan inside the chosen chunk give the inverse-CDF draw in index order.
// Histograms are integer (counts u32, masses as floor(p * 2^44) in u64): native LDS atomics, and an
// order-independent sum, so the thresholds - hence the support set - are reproducible run to run.  The bin that
// crosses the target is found with a block-wide suffix scan, not a serial walk.
constexpr int kSampleBlock = 1024;
constexpr int kSampleWaves = kSampleBlock / 64;
constexpr float kMassScale = 17592186044416.0f;  // 2^44

__device__ __forceinline__ float block_reduce_f(float v, bool is_max, float* smem16) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem16[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = smem16[0];
  for (int i = 1; i < kSampleWaves; ++i) r = is_max ? fmaxf(r, smem16[i]) : r + smem16[i];
  return r;
}

// inclusive scan of one value per thread in thread order; *total = sum over the block
template <typename T>
__device__ __forceinline__ T block_scan_incl(T v, T* wave_tot /* 16 */, T* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const T o = __shfl_up(v, off, kWave);
    if (lane >= off) v += o;
  }
  __syncthreads();
  if (lane == 63) wave_tot[w] = v;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kSampleWaves; ++i) {
    const T x = wave_tot[i];
    if (i < w) base += x;
    tot += x;
  }
  *total = tot;
  return base + v;
}

// Walk the histogram from the top bin down: the bin b with  above + S(b+1) < target <= above + S(b)
// (S = suffix sum) is written to *chosen and *above += S(b+1).  Thread t owns bins nb-1-t*per .. (descending).
template <typename T>
__device__ __forceinline__ void select_bin(const T* hist, int nb, T target_total, int* chosen, T* above, T* wave_tot) {
  const int t = threadIdx.x, per = nb / kSampleBlock;  // 2 (11-bit pass) or 1 (10-bit pass)
  const T prev = *above;
  const T target = target_total > prev ? target_total - prev : 0;
  const T h0 = hist[nb - 1 - t * per], h1 = per == 2 ? hist[nb - 2 - t * per] : (T)0;
  if (t == 0) *chosen = 0;  // target beyond the total (rounding): lowest bin, keeps everything below
  T total;
  const T incl = block_scan_incl<T>(h0 + h1, wave_tot, &total);
  const T excl = incl - (h0 + h1);
  if (target == 0) {
    if (t == 0) *chosen = nb - 1;
  } else if (excl < target && target <= excl + h0) {
    *chosen = nb - 1 - t * per;
    *above = prev + excl;
  } else if (per == 2 && excl + h0 < target && target <= incl) {
    *chosen = nb - 2 - t * per;
    *above = prev + excl + h0;
  } else if (t == 0 && target > total) {
    *above = prev + total - hist[0];
  }
  __syncthreads();
}

__device__ __forceinline__ float uniform01(uint64_t seed) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0,1)
}

// f(value, index) over logits*inv_t, 8 per 16-byte load when aligned
template <typename F>
__device__ __forceinline__ void for_logits(const Half* __restrict__ logits, int n, float inv_t, F f) {
  const int tid = threadIdx.x;
  int done = 0;
  if (aligned16(logits)) {
    const int nvec = n >> 3;
    for (int i = tid; i < nvec; i += kSampleBlock) {
      const u32x4 v = reinterpret_cast<const u32x4*>(logits)[i];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f(bf_lo(w[j]) * inv_t, i * 8 + 2 * j);
        f(bf_hi(w[j]) * inv_t, i * 8 + 2 * j + 1);
      }
    }
    done = nvec << 3;
  }
  for (int i = done + tid; i < n; i += kSampleBlock) f(bf2f(logits[i]) * inv_t, i);
}
// f(prob) over probs, 4 per 16-byte load when aligned (order of visits is irrelevant to the callers)
template <typename F>
__device__ __forceinline__ void for_probs(const float* __restrict__ probs, int n, F f) {
  const int tid = threadIdx.x;
  int done = 0;
  if (aligned16(probs)) {
    const int nvec = n >> 2;
    for (int i = tid; i < nvec; i += kSampleBlock) {
      const f32x4 v = reinterpret_cast<const f32x4*>(probs)[i];
      f(v[0]); f(v[1]); f(v[2]); f(v[3]);
    }
    done = nvec << 2;
  }
  for (int i = done + tid; i < n; i += kSampleBlock) f(probs[i]);
}

__global__ __launch_bounds__(kSampleBlock) void sample_kernel(const Half* __restrict__ logits,
                                                               float* __restrict__ probs,
                                                               uint8_t* __restrict__ valid, int* __restrict__ out,
                                                               int n, float inv_t, int top_k, float top_p,
                                                               uint64_t seed) {
  __shared__ float red[kSampleWaves];
  __shared__ uint32_t cnt[2048];
  __shared__ unsigned long long mass[2048];
  __shared__ uint32_t wt32[kSampleWaves];
  __shared__ unsigned long long wt64[kSampleWaves];
  __shared__ float wtf[kSampleWaves];
  __shared__ float chunk_sum[kSampleBlock];
  __shared__ uint32_t above_k;
  __shared__ unsigned long long above_p;
  __shared__ int chosen_k, chosen_p, sh_chunk, sh_pick, sh_last;
  __shared__ float sh_rem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- softmax (reference logits_to_probs_kernel): p = expf(v - max) / sum ----
  float m = -INFINITY;
  for_logits(logits, n, inv_t, [&](float v, int) { m = fmaxf(m, v); });
  m = block_reduce_f(m, true, red);
  float s = 0.f;
  for_logits(logits, n, inv_t, [&](float v, int) { s += expf(v - m); });
  s = block_reduce_f(s, false, red);
  const float inv_sum = 1.0f / s;
  for_logits(logits, n, inv_t, [&](float v, int i) { probs[i] = expf(v - m) * inv_sum; });
  __syncthreads();

  // ---- joint filter thresholds on the ORIGINAL distribution: largest bit pattern v with count(p >= v) >= top_k,
  //      resp. mass(p >= v) >= top_p; 11 + 11 + 10 bit radix passes, both histograms per sweep ----
  const bool do_k = top_k > 0 && top_k < n, do_p = top_p < 1.0f;
  uint32_t pk = 0u, pp = 0u;
  if (do_k || do_p) {
    if (tid == 0) { above_k = 0u; above_p = 0ull; }
    const unsigned long long target_p = (unsigned long long)((double)top_p * (double)kMassScale);
    int shift = 32;
    for (int pass = 0; pass < 3; ++pass) {
      const int bits = pass < 2 ? 11 : 10, hi = shift, nb = 1 << bits;
      shift -= bits;
      for (int i = tid; i < nb; i += kSampleBlock) { cnt[i] = 0u; mass[i] = 0ull; }
      __syncthreads();
      for_probs(probs, n, [&](float p) {
        const uint32_t u = __builtin_bit_cast(uint32_t, p);
        const int bin = (u >> shift) & (nb - 1);
        if (do_k && (pass == 0 || (u >> hi) == (pk >> hi))) atomicAdd(&cnt[bin], 1u);
        if (do_p && (pass == 0 || (u >> hi) == (pp >> hi))) atomicAdd(&mass[bin], (unsigned long long)(p * kMassScale));
      });
      __syncthreads();
      if (do_k) { select_bin<uint32_t>(cnt, nb, (uint32_t)top_k, &chosen_k, &above_k, wt32); pk |= (uint32_t)chosen_k << shift; }
      if (do_p) { select_bin<unsigned long long>(mass, nb, target_p, &chosen_p, &above_p, wt64); pp |= (uint32_t)chosen_p << shift; }
    }
  }
  const uint32_t thr = (do_k ? pk : 0u) > (do_p ? pp : 0u) ? (do_k ? pk : 0u) : (do_p ? pp : 0u);
  const float thr_f = __builtin_bit_cast(float, thr);

  // ---- kept mass per chunk of cw consecutive tokens (wave w owns chunks w, w+16, ...) ----
  const int cw = 256 * ((n + 256 * kSampleBlock - 1) / (256 * kSampleBlock));
  const int nchunks = (n + cw - 1) / cw;
  const bool vec = aligned16(probs);
  auto chunk_vals = [&](int c, int j, float (&a)[4]) {  // 4 consecutive kept-filtered probs of this lane
    const int i0 = c * cw + j * 256 + lane * 4;
    if (vec && i0 + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(probs + i0);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = v[e] >= thr_f ? v[e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = (i0 + e < n && probs[i0 + e] >= thr_f) ? probs[i0 + e] : 0.f;
    }
    return i0;
  };
  for (int c = wave; c < nchunks; c += kSampleWaves) {
    float t = 0.f;
    for (int j = 0; j < cw / 256; ++j) {
      float a[4];
      chunk_vals(c, j, a);
      t += (a[0] + a[1]) + (a[2] + a[3]);
    }
    t = wave_sum(t);
    if (lane == 0) chunk_sum[c] = t;
  }
  if (tid == 0) { sh_chunk = -1; sh_pick = 0x7FFFFFFF; sh_last = -1; }
  __syncthreads();
  const float mine = tid < nchunks ? chunk_sum[tid] : 0.f;
  float kept;
  const float incl = block_scan_incl<float>(mine, wtf, &kept);
  const float target = uniform01(seed) * kept;
  if (mine > 0.f) {
    if (incl > target && incl - mine <= target) { sh_chunk = tid; sh_rem = target - (incl - mine); }
    atomicMax(&sh_last, tid);  // last chunk holding kept mass (fallback when rounding leaves target >= kept)
  }
  __syncthreads();
  const bool fallback = sh_chunk < 0;
  const int c = fallback ? sh_last : sh_chunk;
  if (c >= 0 && wave == 0) {  // inverse CDF inside the chunk, index order
    const float rem = sh_rem;
    float run = 0.f;
    int last_kept = -1;
    for (int j = 0; j < cw / 256; ++j) {
      float a[4];
      const int i0 = chunk_vals(c, j, a);
      const float lt = (a[0] + a[1]) + (a[2] + a[3]);
      float inc = lt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += o;
      }
      float cum = run + inc - lt;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a[e] > 0.f) {
          last_kept = i0 + e;
          if (!fallback && cum + a[e] > rem && cum <= rem) atomicMin(&sh_pick, i0 + e);
        }
        cum += a[e];
      }
      run += __shfl(inc, 63, kWave);
    }
    // not found inside the chunk (summation order differs from the chunk total by an ulp) or fallback: last kept
    int lk = last_kept;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(lk, off, kWave); lk = lk > o ? lk : o; }
    if (lane == 0 && sh_pick == 0x7FFFFFFF) sh_pick = lk < 0 ? 0 : lk;
  }
  __syncthreads();
  if (tid == 0) {
    out[0] = sh_pick == 0x7FFFFFFF ? 0 : sh_pick;
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
