// Device helpers shared by the gfx950 kernels: bf16 bit casts, wave64 reductions
// (DPP inside a 16-lane row, bpermute across rows), 16-byte vector types.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pegainfer_kernels.h"

namespace pk {

constexpr int kWave = 64;  // gfx950 wavefront width; never 32.

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float bf2f(uint16_t h) {
  return __builtin_bit_cast(float, static_cast<uint32_t>(h) << 16);
}
// round-to-nearest-even, same as the reference's __float2bfloat16
__device__ __forceinline__ uint16_t f2bf(float f) {
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
// two roundings in ONE v_cvt_pk_bf16_f32 (the scalar form costs a cvt per value plus a merge)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_cv;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_cv{lo, hi}, bf16x2_cv));
}
__device__ __forceinline__ float bf16_round_f(float f) { return bf2f(f2bf(f)); }
// SiLU exactly as the reference writes it: g / (1 + expf(-g))  (csrc/fused_proj.cu:57-62)
__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }

// a.lo*b.lo + a.hi*b.hi + c on packed bf16 pairs (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                         __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ float dot8(const u32x4& a, const u32x4& b, float c) {
  c = dot2(a.x, b.x, c);
  c = dot2(a.y, b.y, c);
  c = dot2(a.z, b.z, c);
  c = dot2(a.w, b.w, c);
  return c;
}

// ---- DPP helpers: every lane of a 16-lane row ends with the row's reduction ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;  // lane i <-> 7-i within 8
constexpr int kDppRowMirror = 0x140;      // lane i <-> 15-i within 16

__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<kDppQuadXor1>(v);
  v += dpp_mov<kDppQuadXor2>(v);
  v += dpp_mov<kDppRowHalfMirror>(v);
  v += dpp_mov<kDppRowMirror>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<kDppQuadXor1>(v));
  v = fmaxf(v, dpp_mov<kDppQuadXor2>(v));
  v = fmaxf(v, dpp_mov<kDppRowHalfMirror>(v));
  v = fmaxf(v, dpp_mov<kDppRowMirror>(v));
  return v;
}
// full-wave butterfly: rows first (DPP), then the 4 rows via bpermute
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, kWave);
  v += __shfl_xor(v, 32, kWave);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, kWave));
  v = fmaxf(v, __shfl_xor(v, 32, kWave));
  return v;
}

// block-wide sum for blockDim.x == WAVES*64; result broadcast to every thread.
template <int WAVES>
__device__ __forceinline__ float block_sum(float v, float* smem /* >= WAVES floats */) {
  v = wave_sum(v);
  if constexpr (WAVES == 1) return v;
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // protect smem reuse across consecutive calls
  if ((threadIdx.x & 63) == 0) smem[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < WAVES; ++i) t += smem[i];
  return t;
}

__device__ __forceinline__ bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}
inline bool host_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline hipStream_t as_stream(pegainfer_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(long a, long b) { return static_cast<int>((a + b - 1) / b); }

// CU count of the current device, cached (256 on an MI355X in SPX mode; a CPX partition has 32).  Every "deal the tiles
// evenly onto the CUs" plan reads it here; without a device (CPU-side routing tests) it is the MI355X figure.
inline int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return cus;
}

}  // namespace pk
