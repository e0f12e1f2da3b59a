// Issue-overlap probe for gfx950: how much VALU work hides beside MFMAs, in one wave's stream and across the two waves of
// a SIMD.  One workgroup, 4 or 8 waves (1 or 2 per SIMD); every wave runs ITER iterations of
//   [M x v_mfma_f32_32x32x16_bf16 or 2M x v_mfma_f32_16x16x32_bf16]  +  [N x v_fma_f32]
// on independent registers and reports shader cycles per iteration (s_memtime).
//   mode 0: every wave runs MFMAs and VALU interleaved in its own stream
//   mode 1: waves 0-3 run only the MFMAs, waves 4-7 only the VALU (cross-wave overlap; needs 8 waves)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N, bool SMALL>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int mode, int iters) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  f32x4 d[8] = {};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * threadIdx.x + i;
  const bool do_mfma = mode == 0 || wave < 4, do_valu = mode == 0 || wave >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
      if constexpr (SMALL) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d[k], 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
    }
    if (do_valu) {
#pragma unroll
      for (int k = 0; k < N; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k & 7]) : "v"(v[(k + 3) & 7]));
    }
    if constexpr (!SMALL) {
      // 4 MFMAs x 32 cyc = 128 cyc of matrix pipe per iteration; interleave hint: 1 MFMA then N/4 VALU
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, N / 4, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, N / 8, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sink = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + c0[0] + c1[1] + c2[2] + c3[3];
  for (int k = 0; k < 8; ++k) sink += d[k][0];
  if (sink == 123.456f) out[63] = 1;
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int N, bool SMALL>
static void run(int waves, int mode, const char* tag) {
  unsigned long long *d, h[64] = {};
  (void)hipMalloc(&d, sizeof h);
  (void)hipMemset(d, 0, sizeof h);
  const int iters = 2000;
  probe<N, SMALL><<<1, waves * 64>>>(d, mode, iters);
  probe<N, SMALL><<<1, waves * 64>>>(d, mode, iters);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%-9s N=%3d valu/iter  %d waves  mode %d:", tag, N, waves, mode);
  for (int w = 0; w < waves; w += (waves == 8 ? 4 : 4)) printf("  wave %d: %6.1f cyc/iter", w, (double)h[w] / iters);
  printf("\n");
  (void)hipFree(d);
}

int main() {
  printf("per iteration: 4 x mfma 32x32x16 (128 cyc of matrix pipe) or 8 x mfma 16x16x32 (~136 cyc) + N x v_fma_f32\n");
  run<0, false>(4, 0, "32x32x16"); run<8, false>(4, 0, "32x32x16"); run<16, false>(4, 0, "32x32x16"); run<24, false>(4, 0, "32x32x16"); run<32, false>(4, 0, "32x32x16"); run<48, false>(4, 0, "32x32x16");
  run<0, true>(4, 0, "16x16x32"); run<8, true>(4, 0, "16x16x32"); run<16, true>(4, 0, "16x16x32"); run<24, true>(4, 0, "16x16x32"); run<32, true>(4, 0, "16x16x32"); run<48, true>(4, 0, "16x16x32");
  run<16, false>(8, 0, "32x32x16"); run<32, false>(8, 0, "32x32x16");
  run<16, true>(8, 0, "16x16x32"); run<32, true>(8, 0, "16x16x32");
  run<16, false>(8, 1, "32x32x16"); run<32, false>(8, 1, "32x32x16"); run<48, false>(8, 1, "32x32x16");
  run<16, true>(8, 1, "16x16x32"); run<32, true>(8, 1, "16x16x32"); run<48, true>(8, 1, "16x16x32");
  return 0;
}
