// Feasibility probe for cross-launch overlap of weight-streaming kernels, stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -o overlap_probe tools/probes/overlap_probe.hip && ./overlap_probe
// A "layer" is four streaming kernels of 31.5 / 21 / 99.6 / 49.8 MB (the Qwen3-4B GEMV byte counts) whose inputs
// depend on the previous kernel (all-to-all).  Mode S: one stream, kernel boundaries (what decode_mode 1 does).
// Mode D: the kernels alternate between TWO streams with no stream dependency; a consumer's workgroups become
// resident while the producer still runs, request their first weight group, then wait on XCD-sharded arrival
// counters written by the producer's workgroups (sc1 payload, vmcnt(0), relaxed agent atomic).  Every kernel uses
// at most half of the workgroup slots, so producer and consumer always fit together (no co-residency deadlock);
// every spin is bounded.  Prints us per layer for both modes and the hand-off stamps.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct Args {
  const u32x4* w;            // weights of this kernel
  long vecs_per_wg;          // 16-byte vectors each workgroup streams
  const uint32_t* wait_ctr;  // 8 counters, 16 words apart (a line each); null = no wait
  uint32_t wait_target;
  uint32_t* arrive_ctr;      // 8 counters of this kernel
  const uint32_t* x_in;      // 2560 words written by the producer (sc1), read after the flag
  uint32_t* x_out;           // this kernel's output: each workgroup writes one word (sc1)
  unsigned long long* stamps;  // [grid][4]: entry, first group requested, flag seen, exit
  uint32_t* status;
  uint32_t* cuid;            // [grid]: (xcc << 16) | HW_ID[15:0] of wave 0
  int prefetch_first;        // 1: request the first weight group before waiting for the flag
  int mode;                  // 0: 8 XCD-sharded counters, every consumer sums them; 1: ONE counter (returning atomic), the last
                             //    arriver stores a done word, consumers poll that word from one lane
  int sleep;                 // s_sleep argument between polls (x 64 cycles)
  uint32_t grid;
};

__device__ __forceinline__ void nap(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);   // 8 x 64 cycles ~ 0.2 us per step
}
template <int U, int NTH>
__global__ __launch_bounds__(NTH) void stream_kernel(const Args a) {
  const int lane = threadIdx.x;
  const u32x4* base = a.w + (long)blockIdx.x * a.vecs_per_wg;
  const long n = a.vecs_per_wg;   // multiple of NTH * U
  __shared__ uint32_t s_ok;
  if (threadIdx.x == 0) {
    a.stamps[blockIdx.x * 4 + 0] = wall_clock64();
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    a.cuid[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xff00);   // se / sh / cu bits, wave + simd masked off
  }
  extern __shared__ unsigned char pad_lds[];   // occupancy limiter only
  u32x4 v[U];
  long i = lane;
  if (a.prefetch_first) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(base + i + (long)u * NTH);
  }
  if (threadIdx.x == 0) a.stamps[blockIdx.x * 4 + 1] = wall_clock64();
  uint32_t xsum = 0;
  if (a.wait_ctr) {
    if (threadIdx.x < 64) {
      const unsigned long long t0 = wall_clock64();
      uint32_t ok = 0;
      for (;;) {
        uint32_t c = 0;
        if (a.mode == 0) {
          if (lane < 8) c = __hip_atomic_load(a.wait_ctr + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int o = 4; o; o >>= 1) c += __shfl_xor(c, o);
          c = __shfl(c, 0);
        } else {
          if (lane == 0) c = __hip_atomic_load(a.wait_ctr + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // done word
          c = __shfl(c, 0) ? a.wait_target : 0;
        }
        if (c >= a.wait_target) { ok = 1; break; }
        if (wall_clock64() - t0 > 20000000ull) break;   // 200 ms
        nap(a.sleep);
      }
      if (threadIdx.x == 0) { s_ok = ok; if (!ok) a.status[0] = 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) a.stamps[blockIdx.x * 4 + 2] = wall_clock64();
    // the producer's output, written through by other XCDs: cache-bypassing loads
    for (int j = lane; j < 2560; j += NTH) xsum += __hip_atomic_load(a.x_in + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (threadIdx.x == 0) {
    a.stamps[blockIdx.x * 4 + 2] = wall_clock64();
  }
  if (!a.prefetch_first) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(base + i + (long)u * NTH);
  }
  uint32_t acc = xsum;
  for (i = lane + (long)U * NTH; i < n; i += (long)U * NTH) {
    u32x4 nv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nv[u] = __builtin_nontemporal_load(base + i + (long)u * NTH);
#pragma unroll
    for (int u = 0; u < U; ++u) { acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w; v[u] = nv[u]; }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  for (int o = 32; o; o >>= 1) acc ^= __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(a.x_out + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % 2560)), "v"(acc) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (a.mode == 0) {
      __hip_atomic_fetch_add(a.arrive_ctr + (xcc & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const uint32_t old = __hip_atomic_fetch_add(a.arrive_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == a.grid - 1) __hip_atomic_store(a.arrive_ctr + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    a.stamps[blockIdx.x * 4 + 3] = wall_clock64();
  }
}

// a kernel that only SITS on the chip for `ticks` (100 MHz): wave 0 naps (and optionally polls a word), the other waves
// are parked at the barrier - what a consumer looks like while it waits
__global__ void idle_kernel(const uint32_t* word, unsigned long long ticks, int poll, int naps) {
  if (threadIdx.x < 64) {
    const unsigned long long t0 = wall_clock64();
    uint32_t c = 0;
    while (wall_clock64() - t0 < ticks) {
      if (poll && threadIdx.x == 0) c += __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nap(naps);
    }
    if (c == 0xdeadbeef) __builtin_trap();
  }
  __syncthreads();
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 12;
  const double mb[4] = {31.457, 20.972, 99.615, 49.807};
  CK(hipSetDevice(0));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int nk = layers * 4;
  // weights: every kernel its own buffer (cold in MALL: 202 MB per layer x layers)
  std::vector<u32x4*> w(nk);
  std::vector<long> vecs(nk);
  for (int k = 0; k < nk; ++k) {
    const long bytes = (long)(mb[k % 4] * 1e6);
    vecs[k] = bytes / 16;
    CK(hipMalloc((void**)&w[k], bytes + (1 << 20)));
    CK(hipMemset(w[k], 1 + k % 7, bytes + (1 << 20)));
  }
  uint32_t *ctr, *x, *status, *cuid;
  unsigned long long* stamps;
  CK(hipMalloc((void**)&ctr, (size_t)(nk + 1) * 128 * 4));
  CK(hipMalloc((void**)&x, 2 * 2560 * 4));
  CK(hipMalloc((void**)&status, 64));
  CK(hipMalloc((void**)&cuid, (size_t)nk * 1024 * 4));
  std::vector<uint32_t> hc((size_t)nk * 1024);
  CK(hipMalloc((void**)&stamps, (size_t)nk * 1024 * 4 * 8));
  CK(hipMemset(x, 0, 2 * 2560 * 4));
  hipEvent_t e0, e1, ef, ej;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  std::vector<unsigned long long> hs((size_t)nk * 1024 * 4);

  auto run = [&](const char* name, bool dual, int grid, int U, int prefetch, int mode = 0, int sleep = 1, int nth = 256, int lds = 0, int idle_grid = 0, int idle_nth = 256, int idle_poll = 0) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(ctr, 0, (size_t)(nk + 1) * 128 * 4, sa));
      CK(hipMemsetAsync(status, 0, 64, sa));
      CK(hipMemsetAsync(stamps, 0, (size_t)nk * 1024 * 4 * 8, sa));
      CK(hipStreamSynchronize(sa));
      CK(hipEventRecord(e0, sa));
      if (dual) { CK(hipEventRecord(ef, sa)); CK(hipStreamWaitEvent(sb, ef, 0)); }
      if (idle_grid) {   // a resident bystander on the other queue for the whole chain (single-stream chain)
        idle_kernel<<<idle_grid, idle_nth, lds, sb>>>(ctr, 80000ull, idle_poll, 3);
        idle_kernel<<<1, 64, 0, sa>>>(ctr, 2000ull, 0, 1);   // 20 us head start so the bystander is resident
        CK(hipEventRecord(e0, sa));
      }
      for (int k = 0; k < nk; ++k) {
        Args a{};
        a.w = w[k];
        const long per = vecs[k] / grid / (nth * U) * (nth * U);
        a.mode = mode; a.sleep = sleep; a.grid = grid;
        a.vecs_per_wg = per;
        a.wait_ctr = (dual && k > 0) ? ctr + (size_t)k * 128 : nullptr;     // counters of kernel k-1 live in slot k
        a.wait_target = grid;
        a.arrive_ctr = ctr + (size_t)(k + 1) * 128;
        a.x_in = x + (k & 1) * 2560;
        a.x_out = x + ((k + 1) & 1) * 2560;
        a.stamps = stamps + (size_t)k * 1024 * 4;
        a.status = status;
        a.cuid = cuid + (size_t)k * 1024;
        a.prefetch_first = prefetch;
        hipStream_t s = (dual && (k & 1)) ? sb : sa;
        if (nth == 512) stream_kernel<8, 512><<<grid, 512, lds, s>>>(a);
        else if (U == 4) stream_kernel<4, 256><<<grid, 256, lds, s>>>(a);
        else stream_kernel<8, 256><<<grid, 256, lds, s>>>(a);
      }
      if (dual) { CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0)); }
      CK(hipEventRecord(e1, sa));
      CK(hipStreamSynchronize(sa));
      CK(hipStreamSynchronize(sb));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    uint32_t st = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), cuid, hc.size() * 4, hipMemcpyDeviceToHost));
    double avg_cus = 0; int max_per_cu = 0;
    for (int k = 0; k < nk; ++k) {
      std::vector<uint32_t> ids(hc.begin() + (size_t)k * 1024, hc.begin() + (size_t)k * 1024 + grid);
      std::sort(ids.begin(), ids.end());
      int distinct = 0, run = 0;
      for (size_t i = 0; i < ids.size(); ++i) {
        if (i == 0 || ids[i] != ids[i - 1]) { ++distinct; run = 1; } else ++run;
        if (run > max_per_cu) max_per_cu = run;
      }
      avg_cus += distinct;
    }
    avg_cus /= nk;
    // per kernel (last repetition): first entry, last exit, and for the consumer the flag-seen time after the producer's last exit
    double sum_span = 0, sum_gap = 0, sum_early = 0;
    int cnt = 0;
    unsigned long long prev_exit = 0;
    for (int k = 0; k < nk; ++k) {
      unsigned long long en = ~0ull, ex = 0, fl = 0;
      for (int b = 0; b < grid; ++b) {
        const unsigned long long* t = &hs[((size_t)k * 1024 + b) * 4];
        if (t[0] && t[0] < en) en = t[0];
        if (t[3] > ex) ex = t[3];
        if (t[2] > fl) fl = t[2];
      }
      sum_span += (ex - en) * 0.01;
      if (k > 0) { sum_gap += ((double)fl - (double)prev_exit) * 0.01; sum_early += ((double)prev_exit - (double)en) * 0.01; ++cnt; }
      prev_exit = ex;
    }
    double total_mb = 0;
    for (int k = 0; k < nk; ++k) total_mb += mb[k % 4];
    printf("%-28s mode %d sleep %d nth %d grid %4d U %d prefetch %d: %.3f ms total, %.2f us/layer, %.2f TB/s, status %u | avg kernel span %.2f us, "
           "consumer entered %.2f us before producer's last exit, last flag seen %.2f us after it | distinct CUs per kernel %.0f, max WGs of one kernel on a CU %d, lds %d\n",
           name, mode, sleep, nth, grid, U, prefetch, best, best * 1e3 / layers, total_mb / 1e6 / (best * 1e-3), st, sum_span / nk,
           cnt ? sum_early / cnt : 0.0, cnt ? sum_gap / cnt : 0.0, avg_cus, max_per_cu, lds);
  };
  hipFuncSetAttribute((const void*)stream_kernel<8, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)stream_kernel<8, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)stream_kernel<4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  run("single stream", false, 512, 8, 1);
  run("single + 512x256 idle", false, 512, 8, 1, 0, 1, 256, 0, 512, 256, 0);
  run("single + 512x256 idle poll", false, 512, 8, 1, 0, 1, 256, 0, 512, 256, 1);
  run("single + 256x512 idle poll", false, 512, 8, 1, 0, 1, 256, 0, 256, 512, 1);
  run("single + 256x64 idle poll", false, 512, 8, 1, 0, 1, 256, 0, 256, 64, 1);
  run("single + 1024x256 idle", false, 512, 8, 1, 0, 1, 256, 0, 1024, 256, 0);
  run("dual 256x512 sharded", true, 256, 8, 1, 0, 3, 512, 0);
  run("dual 256x512 done", true, 256, 8, 1, 1, 3, 512, 0);
  run("dual 256x512 done lds80", true, 256, 8, 1, 1, 3, 512, 80 * 1024);
  return 0;
}
