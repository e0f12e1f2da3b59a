// Per-CU HBM ingest with only a SUBSET of the CUs streaming.
//
// DESIGN.md bounds the o_proj half of the fused attention + o_proj launch ("120 CUs ingest 21 MB in 7-8 us"), long-context
// decode attention and the GEMVs by "a CU ingests ~24 GB/s from HBM".  Every line of profiles/r4_ingest_probe.txt came from
// 256-workgroup launches, where 23-28 GB/s per CU IS 5.9-7.2 TB/s / 256 - the chip limit, not a per-CU one.  This probe
// separates the two: G of the 256 CUs stream PRIVATE data from HBM (8 MB each, read once, LDS-DMA ring like the kernels
// here), the others (a) are idle or (b) spin on L2-resident loads (what attention workgroups waiting in round trips do).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/ingest_subset_probe tools/probes/ingest_subset_probe.hip
// One workgroup per CU is forced by >= 96 KB of dynamic LDS per workgroup; the placement is VERIFIED from the hardware id
// registers (distinct (xcc, se, cu) triples are printed).  Streamers are dealt evenly over the 8 XCDs (role by
// blockIdx / 8, the XCD being blockIdx % 8) or, `packed`, into as few XCDs as possible - each XCD has its own L2 and its own
// port to the fabric, so the second form asks whether the limit is per CU or per XCD.
// Per-streamer rate = its bytes / (its own last-load-landed minus first-issue, s_memrealtime at 100 MHz); chip rate = all
// streamed bytes / (last end - first start).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void* lptr_t;

struct Rec { uint64_t t0, t1; uint32_t hw, xcc; uint32_t role, pad; };

__device__ inline uint64_t now() { return __builtin_amdgcn_s_memrealtime(); }

// role 0 = exit at once, 1 = stream `chunks` KB of private data, 2 = spin on the shared L2 window until `done` reaches n_stream
template <int DEPTH, bool NT>
__global__ __launch_bounds__(512) void probe_kernel(const u32x4* __restrict__ src, const u32x4* __restrict__ shared_win, int n_stream,
                                                    int packed, int spin, int chunks, int* __restrict__ done, Rec* __restrict__ rec,
                                                    uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  // even: the first n_stream / 8 slots of EVERY xcd stream.  packed: xcds 0 .. ceil(n_stream / 32) - 1 stream with all 32 slots
  int sidx;   // index among the streamers, or -1
  if (!packed) sidx = slot < (n_stream + 7) / 8 && slot * 8 + xcd < n_stream ? slot * 8 + xcd : -1;
  else sidx = xcd * 32 + slot < n_stream ? xcd * 32 + slot : -1;
  const int role = sidx >= 0 ? 1 : (spin ? 2 : 0);
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  uint64_t t0 = now(), t1 = t0;
  if (role == 1) {
    const u32x4* base = src + (long)sidx * (8l << 20) / 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&smem[0] + (uint32_t)wave * (DEPTH * 1024u);
    int c = wave;
    const int iters = chunks / nw;
    for (int i = 0; i < iters; ++i) {
      const u32x4* p = base + (long)c * 64 + lane;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(i % DEPTH) * 1024u);
      if (NT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(p), "s"(dst) : "memory", "m0");
      else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
      c += nw;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    t1 = now();
    if (threadIdx.x == 0) {
      sink[b] = smem[0].x;
      __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (role == 2) {
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < 200000; ++it) {                    // bounded: ~1 us per sweep
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= shared_win[((it * 8 + u) * 512 + threadIdx.x) & 4095];
      if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_stream) break;
    }
    t1 = now();
    if ((acc.x ^ acc.y) == 0x12345u) sink[b] = 1;
  }
  if (threadIdx.x == 0) rec[b] = Rec{t0, t1, hw, xcc, (uint32_t)role, 0};
}

template <int DEPTH, bool NT>
static void run_one(const u32x4* buf, const u32x4* win, int* done, Rec* rec_d, uint32_t* sink, int G, int packed, int spin, int nw,
                    const char* tag) {
  auto k = probe_kernel<DEPTH, NT>;
  const size_t lds = std::max<size_t>((size_t)nw * DEPTH * 1024, 96 * 1024);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int chunks = 8 * 1024;   // 8 MB per streamer
  std::vector<Rec> rec(256);
  double best_chip = 0, mean_cu = 0, min_cu = 0, max_cu = 0, span_us = 0;
  size_t cus = 0, xcds = 0;
  for (int rep = 0; rep < 4; ++rep) {   // rep 0 warms the code path; the buffer is 2 GB, so nothing stays in the caches
    CK(hipMemset(done, 0, 4));
    k<<<256, nw * 64, lds, 0>>>(buf, win, G, packed, spin, chunks, done, rec_d, sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(rec.data(), rec_d, sizeof(Rec) * 256, hipMemcpyDeviceToHost));
    uint64_t first = ~0ull, last = 0;
    double sum = 0, mn = 1e9, mx = 0;
    std::set<uint32_t> where, xs;
    int n = 0;
    for (auto& r : rec) {
      if (r.role != 1) continue;
      first = std::min(first, r.t0); last = std::max(last, r.t1);
      const double us = (double)(r.t1 - r.t0) / 100.0, gbs = 8.0 * 1048576 / us / 1e3;
      sum += gbs; mn = std::min(mn, gbs); mx = std::max(mx, gbs); ++n;
      where.insert((r.xcc & 0xf) << 16 | (r.hw & 0x0003ff00) >> 4 | ((r.hw >> 8) & 0xf));   // xcc | se, sh | cu
      xs.insert(r.xcc & 0xf);
    }
    const double chip = (double)n * 8.0 * 1048576 / ((double)(last - first) / 100.0) / 1e6;   // TB/s
    if (rep > 0 && chip > best_chip) {
      best_chip = chip; mean_cu = sum / n; min_cu = mn; max_cu = mx; span_us = (double)(last - first) / 100.0;
      cus = where.size(); xcds = xs.size();
    }
  }
  printf("%-22s G %3d %-6s others %-5s waves %d x %2d KB  | per CU mean %5.1f min %5.1f max %5.1f GB/s | chip %5.2f TB/s | span %7.1f us | %zu distinct CUs on %zu XCDs\n",
         tag, G, packed ? "packed" : "even", spin ? "spin" : "idle", nw, DEPTH, mean_cu, min_cu, max_cu, best_chip, span_us, cus, xcds);
}

int main() {
  const size_t bytes = 2ull << 30;   // 256 x 8 MB
  u32x4 *buf, *win;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  CK(hipMalloc(&win, 64 * 1024));
  CK(hipMemset(win, 2, 64 * 1024));
  int* done; Rec* rec; uint32_t* sink;
  CK(hipMalloc(&done, 64)); CK(hipMalloc(&rec, sizeof(Rec) * 256)); CK(hipMalloc(&sink, 4096));
  for (int spin = 0; spin <= 1; ++spin)
    for (int G : {8, 32, 64, 120, 128, 192, 256}) {
      run_one<16, true>(buf, win, done, rec, sink, G, 0, spin, 4, "dma nt 4 waves x 16");
      run_one<16, true>(buf, win, done, rec, sink, G, 0, spin, 8, "dma nt 8 waves x 16");
    }
  // in-flight depth at G = 120 and 256: 1 ... 8 issuing waves, 16 ... 128 KB in flight per CU
  for (int G : {120, 256}) {
    run_one<16, true>(buf, win, done, rec, sink, G, 0, 0, 1, "depth sweep");
    run_one<16, true>(buf, win, done, rec, sink, G, 0, 0, 2, "depth sweep");
    run_one<8, true>(buf, win, done, rec, sink, G, 0, 0, 4, "depth sweep");
    run_one<32, true>(buf, win, done, rec, sink, G, 0, 0, 4, "depth sweep");
    run_one<16, false>(buf, win, done, rec, sink, G, 0, 0, 8, "depth sweep, no nt");
  }
  // per CU or per XCD?  the same streamer counts packed into as few XCDs as possible
  for (int G : {32, 64, 128})
    for (int spin = 0; spin <= 1; ++spin) run_one<16, true>(buf, win, done, rec, sink, G, 1, spin, 8, "packed into XCDs");
  return 0;
}
