// How fast can ONE CU ingest bytes, by path?  (round 4: every tiled kernel here ends up at ~30-40 GB/s per CU whatever the
// ring depth - is that the L2, or the LDS-DMA path?)   Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -o ingest_probe tools/probes/ingest_probe.hip && ./ingest_probe
// One workgroup per CU (256), NW waves each.  Source: `shared` = every workgroup walks the SAME 512 KB window again and
// again (an x operand: L2-resident after the first pass) or `private` = every workgroup walks its own 8 MB of a 2 GB
// buffer once (a weight stream: HBM).  Paths:
//   dma   global_load_lds_dwordx4 into an LDS ring, counted vmcnt (DEPTH instructions of 1 KB in flight per wave)
//   reg   global_load_dwordx4 into registers, U in flight per lane, XOR-folded (nothing written)
// Prints GB/s per CU and for the chip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void* lptr_t;

// every wave walks `iters` chunks of 1 KB (64 lanes x 16 B), chunk c of wave w of workgroup b at
// base + ((c * nwaves + w) % window_chunks) * 1 KB
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const u32x4* __restrict__ src, long wg_stride_vec, int window_chunks, int iters,
                                                  uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const u32x4* base = src + (long)blockIdx.x * wg_stride_vec;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&smem[0] + (uint32_t)wave * (DEPTH * 1024u);
  int c = wave;
  for (int i = 0; i < iters; ++i) {
    const u32x4* p = base + (long)(c & (window_chunks - 1)) * 64 + lane;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(i % DEPTH) * 1024u);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
    c += nw;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[0].x;
}

template <int U>
__global__ __launch_bounds__(512) void reg_kernel(const u32x4* __restrict__ src, long wg_stride_vec, int window_chunks, int iters,
                                                  uint32_t* __restrict__ sink) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const u32x4* base = src + (long)blockIdx.x * wg_stride_vec;
  u32x4 acc = {0, 0, 0, 0};
  int c = wave;
  for (int i = 0; i < iters; i += U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = base[(long)(c & (window_chunks - 1)) * 64 + lane];
      c += nw;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc ^= v[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;
}

static double run(void (*launch)(hipStream_t), int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(0); launch(0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int r = 0; r < reps; ++r) launch(0);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / reps;
}

static const u32x4* g_src;
static uint32_t* g_sink;
static long g_stride;
static int g_window, g_iters, g_nw;

template <int DEPTH> static void l_dma(hipStream_t s) {
  dma_kernel<DEPTH><<<256, g_nw * 64, g_nw * DEPTH * 1024, s>>>(g_src, g_stride, g_window, g_iters, g_sink);
}
template <int U> static void l_reg(hipStream_t s) {
  reg_kernel<U><<<256, g_nw * 64, 0, s>>>(g_src, g_stride, g_window, g_iters, g_sink);
}


// The access pattern of a tiled GEMM's operand: one instruction = 8 rows x 128 B, rows `stride_b` bytes apart; the
// workgroup walks K tile by K tile (kt = 0..39), 16 row groups per tile (a 128-row tile of a [128, 2560] bf16 operand),
// every workgroup the SAME lines at about the same time.  stride 5120 = the dense layout; + 128 / + 256 = padded rows.
template <int DEPTH>
__global__ __launch_bounds__(512) void tile_kernel(const char* __restrict__ src, int stride_b, int iters, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&smem[0] + (uint32_t)wave * (DEPTH * 1024u);
  const int row = lane >> 3, piece = lane & 7;
  int c = wave;
  for (int i = 0; i < iters; ++i) {
    const int rg = c & 15, kt = (c >> 4) % 40;
    const char* p = src + (long)(rg * 8 + row) * stride_b + kt * 128 + piece * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(i % DEPTH) * 1024u);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
    c += nw;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[0].x;
}
static int g_tile_stride;
static void l_tile(hipStream_t s) {
  tile_kernel<8><<<256, g_nw * 64, g_nw * 8 * 1024, s>>>(reinterpret_cast<const char*>(g_src), g_tile_stride, g_iters, g_sink);
}

// A weight stream read the way a tiled GEMM reads it, from HBM: every workgroup owns row blocks of 96 rows x 5120 B (its
// private 8 MB of the buffer) and walks each block K tile by K tile, a tile being SEG bytes of every row: one DMA
// instruction covers 1024 / SEG rows.  SEG = 128 is a K tile of 64 bf16 (what the kernels here use), 256 / 512 = 128 / 256.
template <int SEG>
__global__ __launch_bounds__(512) void wtile_kernel(const char* __restrict__ src, int iters, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  constexpr int DEPTH = 8, LPR = SEG / 16, RPI = 64 / LPR, IPT = 96 / RPI, TILES = 5120 / SEG;   // instr per tile, tiles per block
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&smem[0] + (uint32_t)wave * (DEPTH * 1024u);
  const char* base = src + (long)blockIdx.x * (8l << 20);
  const int row = lane / LPR, piece = lane % LPR;
  int c = wave;
  for (int i = 0; i < iters; ++i) {
    const int blk = c / (IPT * TILES), r = c % (IPT * TILES), kt = r / IPT, rg = r % IPT;
    const char* p = base + (long)blk * (96 * 5120) + (long)(rg * RPI + row) * 5120 + kt * SEG + piece * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(i % DEPTH) * 1024u);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(p), "s"(dst) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
    c += nw;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[0].x;
}
template <int SEG> static void l_wtile(hipStream_t s) {
  wtile_kernel<SEG><<<256, g_nw * 64, g_nw * 8 * 1024, s>>>(reinterpret_cast<const char*>(g_src), g_iters, g_sink);
}

int main() {
  const size_t bytes = 2ull << 30;
  u32x4* buf;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  CK(hipMalloc(&g_sink, 4096));
  g_src = buf;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wtile_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int stride : {5120, 5120 + 128, 5120 + 256, 8192, 8192 + 256, 19456, 19456 + 256})
    for (int nw : {2, 4, 8}) {
      const long per_wg = 8l << 20;
      g_nw = nw; g_tile_stride = stride; g_iters = (int)(per_wg / 1024 / nw);
      const double us = run(l_tile, 10);
      const double gbs = per_wg / us / 1e3;
      printf("tile walk (8 rows x 128 B per DMA) row stride %5d B  waves %d  %8.1f us  %6.1f GB/s per CU  %6.2f TB/s chip\n", stride, nw,
             us, gbs, gbs * 256 / 1e3);
    }
  {
    struct { int seg; void (*fn)(hipStream_t); } w[] = {{128, l_wtile<128>}, {256, l_wtile<256>}, {512, l_wtile<512>}, {1024, l_wtile<1024>}};
    for (auto& k : w)
      for (int nw : {2, 4}) {
        const long per_wg = 17l * 96 * 5120;                // 17 row blocks of the workgroup's 8 MB
        g_nw = nw; g_iters = (int)(per_wg / 1024 / nw);
        const double us = run(k.fn, 10);
        const double gbs = per_wg / us / 1e3;
        printf("weight tile walk from HBM, %4d B of every row per K tile  waves %d  %8.1f us  %6.1f GB/s per CU  %6.2f TB/s chip\n", k.seg,
               nw, us, gbs, gbs * 256 / 1e3);
      }
  }
  for (int shared = 1; shared >= 0; --shared) {
    const long per_wg = 8l << 20;                       // bytes every workgroup ingests per launch
    g_stride = shared ? 0 : per_wg / 16;
    g_window = shared ? 512 : (int)(per_wg / 1024);     // chunks of 1 KB
    for (int nw : {1, 2, 4, 8}) {
      g_nw = nw;
      g_iters = (int)(per_wg / 1024 / nw);
      struct { const char* name; void (*fn)(hipStream_t); } v[] = {
          {"dma depth 4 ", l_dma<4>}, {"dma depth 8 ", l_dma<8>}, {"dma depth 16", l_dma<16>},
          {"reg U 4     ", l_reg<4>}, {"reg U 8     ", l_reg<8>}};
      for (auto& k : v) {
        if (nw * 16 * 1024 > 160 * 1024 && k.fn == (void (*)(hipStream_t))l_dma<16>) continue;
        const double us = run(k.fn, 10);
        const double gbs = per_wg / us / 1e3;
        printf("%-8s waves %d  %s  %8.1f us  %6.1f GB/s per CU  %6.2f TB/s chip\n", shared ? "shared" : "private", nw, k.name, us,
               gbs, gbs * 256 / 1e3);
      }
    }
  }
  return 0;
}
