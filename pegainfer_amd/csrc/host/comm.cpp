// Collective layer of the host library (include/pegainfer_comm.h): the DeepSeek-V4 MP8 verbs and the expert-parallel
// dispatch / combine of the reference, MI355X-first: one process per GPU, RCCL over the xGMI mesh, every verb on the
// caller's HIP stream, a separate comm stream with event fences for the overlapped MoE exchange.
//
//   collectives.rs:8-287 (verbs + fused casts)   moe.rs:1327-1461 (comm stream)   ep_backend.rs:213-331 (dispatch / combine)
//
// The reference reaches its peers through NCCL (cudarc) and, for expert parallelism, through a pplx RDMA worker
// thread; here both ride on RCCL send / recv groups (xGMI is point-to-point: a grouped send/recv to the 7 peers uses
// the 7 links concurrently).  Routing, packing and the weighted combine are small HIP kernels in this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pegainfer_comm.h"

extern "C" {
int32_t deepseek_bf16_to_f32_cuda(const Half* input, float* output, int32_t n, pegainfer_stream_t stream);
int32_t deepseek_f32_to_bf16_cuda(const float* input, Half* output, int32_t n, pegainfer_stream_t stream);
}

namespace pc {

// ---- one-shot small-message all-reduce over peer access (the regime of the DSV4 decode
// collectives, moe-tilelang-review.md:12: ~107 f32 all-reduces of 16 KB per token, and of Qwen3 TP decode: 72 x 5 KB) ----
constexpr int kOsMaxWorld = 8;                 // one xGMI node
constexpr int kOsMaxBytes = 64 * 1024;         // payloads above this go to RCCL
constexpr int kOsSeg = 16 * 1024;              // one 1024-thread workgroup moves one 16-byte vector per thread
constexpr int kOsMaxWg = kOsMaxBytes / kOsSeg;
constexpr size_t kOsDataBytes = (size_t)2 * kOsMaxWorld * kOsMaxBytes;   // [parity][source rank][payload]
constexpr size_t kOsFlagStride = 64;                                      // one flag per cache line
constexpr size_t kOsFlagBytes = (size_t)2 * kOsMaxWorld * kOsMaxWg * kOsFlagStride;
constexpr size_t kOsSlabBytes = kOsDataBytes + kOsFlagBytes;
struct OsPeers { unsigned char* slab[kOsMaxWorld]; };

struct Comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  float* scratch = nullptr;
  size_t scratch_n = 0;
  // one-shot path: this rank's slab (fine-grained device memory, IPC-exported), the peers' slabs as mapped here,
  // per-workgroup epoch counters and a status word (device-local)
  unsigned char* os_slab = nullptr;
  OsPeers os_peers{};
  bool os_opened[kOsMaxWorld] = {};
  uint32_t *os_epoch = nullptr, *os_status = nullptr;
  unsigned char* os_stage = nullptr;   // 64 KB: unaligned / ragged payloads are staged through it
  unsigned char* os_xchg = nullptr;    // RCCL communicators: the handle / flag exchange buffer of oneshot_enable, allocated
                                       // at creation so that enable itself cannot fail before its collectives
  hipStream_t os_stage_stream = nullptr;   // the stream of the last STAGED one-shot call (single-stream contract, enforced)
  bool os_stage_used = false;
  bool os_active = false;
  unsigned long long os_timeout_ticks = 0;
  std::string err;
};

#define PC_HIP(c, expr)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) { (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return -1; } \
  } while (0)
#define PC_NCCL(c, expr)                                                                         \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess) { (c)->err = std::string(#expr) + ": " + ncclGetErrorString(r_); return -1; } \
  } while (0)

static inline hipStream_t st(pegainfer_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static int reserve_f32(Comm* c, size_t n) {
  if (c->scratch_n >= n) return 0;
  if (c->scratch) (void)hipFree(c->scratch);
  c->scratch = nullptr;
  c->scratch_n = 0;
  PC_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->scratch), n * sizeof(float)));
  c->scratch_n = n;
  return 0;
}

static int cast_to_f32(Comm* c, const Half* in, float* out, int64_t n, hipStream_t s) {
  for (int64_t o = 0; o < n; o += (1 << 30)) {
    const int32_t m = static_cast<int32_t>(std::min<int64_t>(n - o, 1 << 30));
    if (deepseek_bf16_to_f32_cuda(in + o, out + o, m, s)) { c->err = "bf16 -> f32 cast launch failed"; return -1; }
  }
  return 0;
}
static int cast_to_bf16(Comm* c, const float* in, Half* out, int64_t n, hipStream_t s) {
  for (int64_t o = 0; o < n; o += (1 << 30)) {
    const int32_t m = static_cast<int32_t>(std::min<int64_t>(n - o, 1 << 30));
    if (deepseek_f32_to_bf16_cuda(in + o, out + o, m, s)) { c->err = "f32 -> bf16 cast launch failed"; return -1; }
  }
  return 0;
}

// ---- one-shot all-reduce kernel ------------------------------------------------------------------------------------
// Push model (xGMI writes are posted, reads are round trips): every rank stores its payload segment into EVERY rank's
// slab (slot [parity][my rank]) with write-through system-scope stores, drains them (s_waitcnt vmcnt(0)), then stores
// the epoch into flag [parity][my rank][segment] of every rank; a rank that has seen all `world` flags of a segment
// reads the `world` copies from ITS OWN slab (cache-bypassing loads) and sums them in rank order in f32 - the same
// order on every rank, so all ranks hold bit-identical results.  Epochs count up per segment index in device memory
// (incremented by the kernel itself: hipGraph replays need no host help); parity = epoch & 1 double-buffers the slab:
// a peer can only start epoch e + 1 after it has seen MY flag of epoch e, which I send after I finished reading e - 1.
// Every spin is bounded; an expired wait sets the status word and the launch returns (the result is then undefined).
typedef __attribute__((ext_vector_type(4))) uint32_t os_u4;
__device__ __forceinline__ void os_st16(void* p, uint4 v) {
  const os_u4 w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void os_st4(void* p, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// cache-bypassing (system-scope) loads through the compiler's own waitcnt bookkeeping: the `world` copies of a vector
// are requested back to back and waited for once (an asm load + wait per copy would serialise 8 round trips)
__device__ __forceinline__ uint4 os_ld16(const void* p) {
  const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ uint32_t os_ld4(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ size_t os_data_off(int parity, int src) { return ((size_t)parity * kOsMaxWorld + src) * kOsMaxBytes; }
__device__ __forceinline__ size_t os_flag_off(int parity, int src, int seg) {
  return kOsDataBytes + (((size_t)parity * kOsMaxWorld + src) * kOsMaxWg + seg) * kOsFlagStride;
}
__device__ __forceinline__ float os_bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float os_bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t os_pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));
}

template <bool BF16>
__global__ __launch_bounds__(1024) void oneshot_all_reduce_kernel(OsPeers peers, int rank, int world, const unsigned char* in,
                                                                  unsigned char* out, int nbytes, uint32_t* epoch_ctr,
                                                                  uint32_t* status, unsigned long long timeout_ticks) {
  __shared__ uint32_t s_epoch, s_fail;
  const int seg = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { s_epoch = epoch_ctr[seg] + 1; epoch_ctr[seg] = s_epoch; s_fail = 0; }   // launches are stream-ordered
  __syncthreads();
  const uint32_t ep = s_epoch;
  const int par = ep & 1;
  const int seg_off = seg * kOsSeg;
  const bool act = seg_off + tid * 16 < nbytes;
  // 1. push my segment everywhere (own slab included), then drain the stores
  if (act) {
    const uint4 mine = *reinterpret_cast<const uint4*>(in + seg_off + tid * 16);
    for (int r = 0; r < world; ++r) os_st16(peers.slab[r] + os_data_off(par, rank) + seg_off + tid * 16, mine);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // 2. signal every rank, 3. wait for every rank's signal in my own slab
  if (tid < world) os_st4(peers.slab[tid] + os_flag_off(par, rank, seg), ep);
  if (tid < world) {
    const unsigned char* f = peers.slab[rank] + os_flag_off(par, tid, seg);
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(os_ld4(f) - ep) < 0) {
      if (wall_clock64() - t0 > timeout_ticks) { atomicOr(&s_fail, 1u << tid); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (s_fail) {
    if (tid == 0) { status[0] = 0x100u | s_fail; status[1] = ep; status[2] = (uint32_t)seg; }
    return;
  }
  // 4. reduce the `world` copies in rank order, f32 accumulation, one rounding
  if (act) {
    const unsigned char* base = peers.slab[rank] + seg_off + tid * 16;
    if (BF16) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      uint4 cp[kOsMaxWorld];
#pragma unroll
      for (int r = 0; r < kOsMaxWorld; ++r) cp[r] = r < world ? os_ld16(base + os_data_off(par, r)) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < kOsMaxWorld; ++r) {
        if (r >= world) break;
        const uint4 v = cp[r];
        a[0] += os_bf_lo(v.x); a[1] += os_bf_hi(v.x); a[2] += os_bf_lo(v.y); a[3] += os_bf_hi(v.y);
        a[4] += os_bf_lo(v.z); a[5] += os_bf_hi(v.z); a[6] += os_bf_lo(v.w); a[7] += os_bf_hi(v.w);
      }
      uint4 o;
      o.x = os_pack2(a[0], a[1]); o.y = os_pack2(a[2], a[3]); o.z = os_pack2(a[4], a[5]); o.w = os_pack2(a[6], a[7]);
      *reinterpret_cast<uint4*>(out + seg_off + tid * 16) = o;
    } else {
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      uint4 cp[kOsMaxWorld];
#pragma unroll
      for (int r = 0; r < kOsMaxWorld; ++r) cp[r] = r < world ? os_ld16(base + os_data_off(par, r)) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < kOsMaxWorld; ++r) {
        if (r >= world) break;
        const uint4 v = cp[r];
        a[0] += __builtin_bit_cast(float, v.x); a[1] += __builtin_bit_cast(float, v.y);
        a[2] += __builtin_bit_cast(float, v.z); a[3] += __builtin_bit_cast(float, v.w);
      }
      *reinterpret_cast<float4*>(out + seg_off + tid * 16) = make_float4(a[0], a[1], a[2], a[3]);
    }
  }
}

__global__ void os_zero_tail_kernel(unsigned char* p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}

// One launch over an aligned, 16-byte-sized payload of at most 64 KB.
static int oneshot_launch(Comm* c, unsigned char* data, int bytes, int elem, hipStream_t s) {
  const int wgs = (bytes + kOsSeg - 1) / kOsSeg;
  if (elem == 2)
    oneshot_all_reduce_kernel<true><<<wgs, 1024, 0, s>>>(c->os_peers, c->rank, c->world, data, data, bytes, c->os_epoch,
                                                         c->os_status, c->os_timeout_ticks);
  else
    oneshot_all_reduce_kernel<false><<<wgs, 1024, 0, s>>>(c->os_peers, c->rank, c->world, data, data, bytes, c->os_epoch,
                                                          c->os_status, c->os_timeout_ticks);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("one-shot all-reduce launch: ") + hipGetErrorString(e); return -1; }
  return 0;
}
// 1 = taken by the one-shot path, 0 = not eligible (caller falls through to RCCL), -1 = launch error.
// Whether a payload is eligible depends on its BYTE COUNT only (every rank passes the same count, so every rank takes
// the same route - an address-dependent rule could send one rank to RCCL while its peers spin on flags): <= 64 KB on an
// RCCL communicator, any size on a peer-only one (64 KB pieces, launch after launch on the caller's stream; the
// parity double-buffer argument holds across launches because a peer starts epoch e + 1 only after it saw my flag of
// e).  A payload whose address or size is not a multiple of 16 travels through the handle's 64 KB staging buffer
// (device-to-device copies on the same stream, zero-padded tail - graph-capturable).
// SINGLE-STREAM CONTRACT: the epochs are plain per-segment counters advanced by the launches in stream order; all
// one-shot all-reduces of a communicator must be ordered with respect to each other (one stream, or streams joined by
// events / host synchronisation between calls).  The comm stream of this handle is for the RCCL verbs of the
// overlapped MoE exchange (all-gather / reduce-scatter), never for all-reduces issued next to the compute stream's.
static int oneshot_all_reduce(Comm* c, void* data, int64_t n, int elem, hipStream_t s) {
  const int64_t bytes = n * elem;
  if (!c->os_active || (c->nccl && bytes > kOsMaxBytes)) return 0;
  unsigned char* p = static_cast<unsigned char*>(data);
  for (int64_t off = 0; off < bytes; off += kOsMaxBytes) {
    const int piece = (int)std::min<int64_t>(kOsMaxBytes, bytes - off);
    const int padded = (piece + 15) & ~15;
    if (padded != piece || (reinterpret_cast<uintptr_t>(p + off) & 15)) {
      if (!c->os_stage) { c->err = "one-shot all-reduce: no staging buffer"; return -1; }
      // ONE staging buffer per communicator: two staged all-reduces on different, un-joined streams would race on it.
      // A stream change is only accepted while nothing is being captured and after the previous stream drained.
      if (c->os_stage_used && c->os_stage_stream != s) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cs);
        if (cs != hipStreamCaptureStatusNone || hipStreamQuery(c->os_stage_stream) != hipSuccess) {
          c->err = "one-shot all-reduce: staged payloads of one communicator must stay on one stream (or the previous "
                   "stream must have drained): single-stream contract";
          return -1;
        }
      }
      c->os_stage_stream = s;
      c->os_stage_used = true;
      // the <= 15 pad bytes are zeroed by a KERNEL: a hipMemsetAsync node inside a replayed hipGraph stops taking effect
      // after an eager kernel launch between two replays (round 5, csrc/elementwise.hip: pegainfer_zero_words)
      if (padded != piece) os_zero_tail_kernel<<<1, 64, 0, s>>>(c->os_stage + piece, padded - piece);
      if (hipMemcpyAsync(c->os_stage, p + off, (size_t)piece, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
      if (oneshot_launch(c, c->os_stage, padded, elem, s)) return -1;
      if (hipMemcpyAsync(p + off, c->os_stage, (size_t)piece, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
    } else if (oneshot_launch(c, p + off, piece, elem, s)) {
      return -1;
    }
  }
  return 1;
}

// ---- expert-parallel kernels ----------------------------------------------------------------------------------
// Argument meaning follows the reference's a2a kernels (pegainfer-comm-a2a-kernels/src/a2a/*.cu) so EpBackend binds 1:1:
// byte strides for the row buffers, element strides for indices / weights / scales / out_tokens, an optional DEVICE
// token bound (`bound_m_ptr ? *bound_m_ptr : num_tokens`, a2a_dispatch_send.cu:172, a2a_combine_recv.cu:52).

__device__ __forceinline__ int ep_bound(int num_tokens, const int32_t* bound_m) {
  if (!bound_m) return num_tokens;
  const int b = *bound_m;
  return b < 0 ? 0 : (b > num_tokens ? num_tokens : b);
}
// one row of `nbytes` by the whole workgroup: 16-byte vectors when everything is 16-byte aligned, words or bytes otherwise
__device__ __forceinline__ void ep_copy_row(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int nbytes) {
  const uintptr_t m = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)nbytes;
  if ((m & 15) == 0) {
    for (int i = threadIdx.x; i < nbytes / 16; i += blockDim.x) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  } else if ((m & 3) == 0) {
    for (int i = threadIdx.x; i < nbytes / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
  } else {
    for (int i = threadIdx.x; i < nbytes; i += blockDim.x) dst[i] = src[i];
  }
}

// Stable rank of every (token, k) pair inside its expert bucket: wave w scans all pairs for expert w with ballots,
// so a pair's position is the number of EARLIER pairs routed to the same expert - deterministic, no atomics (the
// reference draws the offset with an atomicAdd, a2a_dispatch_send.cu:186: its order inside an expert is unspecified).
__global__ __launch_bounds__(256) void ep_rank_kernel(const int32_t* __restrict__ indices, long indices_stride, int topk,
                                                      int num_tokens, const int32_t* __restrict__ bound_m, int num_experts,
                                                      int32_t* __restrict__ rank_in_bucket, int32_t* __restrict__ counts) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= num_experts) return;
  const int n_pairs = ep_bound(num_tokens, bound_m) * topk;
  int base = 0;
  for (int p0 = 0; p0 < n_pairs; p0 += 64) {
    const int p = p0 + lane;
    const bool hit = p < n_pairs && indices[(long)(p / topk) * indices_stride + (p % topk)] == e;
    const unsigned long long m = __ballot(hit);
    if (hit) rank_in_bucket[p] = base + __popcll(m & ((1ull << lane) - 1ull));
    base += __popcll(m);
  }
  if (lane == 0) counts[e] = base;
}

// bucket offsets (expert order = (destination rank, local expert) order) by one workgroup
__global__ __launch_bounds__(256) void ep_offsets_kernel(const int32_t* __restrict__ counts, int num_experts,
                                                         int32_t* __restrict__ offsets) {
  __shared__ int32_t part[256];
  const int per = (num_experts + 255) / 256;
  int s = 0;
  for (int i = 0; i < per; ++i) {
    const int e = threadIdx.x * per + i;
    if (e < num_experts) s += counts[e];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < (int)threadIdx.x; ++i) base += part[i];
  for (int i = 0; i < per; ++i) {
    const int e = threadIdx.x * per + i;
    if (e < num_experts) { offsets[e] = base; base += counts[e]; }
  }
}
// the pack: one workgroup per pair copies row x[t] (+ its scale plane, gathered with the caller's element stride) to
// its slot of the wire buffer and records the slot for the combine.  Wire row = [payload, padded to 16 B][scales f32]
__global__ __launch_bounds__(256) void ep_pack_kernel(const unsigned char* __restrict__ x, long x_stride_bytes, int payload_bytes,
                                                      const float* __restrict__ x_scale, long scale_stride_elem,
                                                      long scale_stride_token, int hidden_scale, int scale_off,
                                                      const int32_t* __restrict__ indices, long indices_stride,
                                                      const int32_t* __restrict__ rank_in_bucket,
                                                      const int32_t* __restrict__ offsets, int topk, int num_tokens,
                                                      const int32_t* __restrict__ bound_m, int num_experts, int row_bytes,
                                                      unsigned char* __restrict__ send, int32_t* __restrict__ slot) {
  const int p = blockIdx.x, t = p / topk;
  int ex = -1;
  if (t < ep_bound(num_tokens, bound_m)) ex = indices[(long)t * indices_stride + (p % topk)];
  if ((unsigned)ex >= (unsigned)num_experts) {   // beyond the bound, or not an expert id: routed nowhere, combines as zero
    if (threadIdx.x == 0) slot[p] = -1;
    return;
  }
  const int pos = offsets[ex] + rank_in_bucket[p];
  if (threadIdx.x == 0) slot[p] = pos;
  unsigned char* dst = send + (long)pos * row_bytes;
  ep_copy_row(x + (long)t * x_stride_bytes, dst, payload_bytes);
  if (x_scale)
    for (int i = threadIdx.x; i < hidden_scale; i += blockDim.x)
      reinterpret_cast<float*>(dst + scale_off)[i] = x_scale[(long)t * scale_stride_token + (long)i * scale_stride_elem];
}
// received wire row r -> caller row map[r] (the expert-major, expert_padding-aligned layout), payload + scale plane
__global__ __launch_bounds__(256) void ep_unpack_kernel(const unsigned char* __restrict__ recv, int row_bytes, int payload_bytes,
                                                        int hidden_scale, int scale_off, const int32_t* __restrict__ map,
                                                        unsigned char* __restrict__ out_x, long out_x_stride_bytes,
                                                        float* __restrict__ out_scale, long scale_stride_elem,
                                                        long scale_stride_token) {
  const int r = blockIdx.x, m = map[r];
  const unsigned char* src = recv + (long)r * row_bytes;
  ep_copy_row(src, out_x + (long)m * out_x_stride_bytes, payload_bytes);
  if (out_scale)
    for (int i = threadIdx.x; i < hidden_scale; i += blockDim.x)
      out_scale[(long)m * scale_stride_token + (long)i * scale_stride_elem] = reinterpret_cast<const float*>(src + scale_off)[i];
}
// expert output row map[r] of the caller's (padded, expert-major) buffer -> combine wire row r (source-major)
__global__ __launch_bounds__(256) void ep_gather_kernel(const unsigned char* __restrict__ expert_x, long stride_bytes,
                                                        int payload_bytes, const int32_t* __restrict__ map, int row_bytes,
                                                        unsigned char* __restrict__ back) {
  const int r = blockIdx.x;
  ep_copy_row(expert_x + (long)map[r] * stride_bytes, back + (long)r * row_bytes, payload_bytes);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));
}
template <typename T> struct EpVec;   // 8 elements <-> 8 floats
template <> struct EpVec<Half> {
  static __device__ __forceinline__ void load(const Half* p, float* v) {
    const uint4 o = *reinterpret_cast<const uint4*>(p);
    v[0] = bf_lo(o.x); v[1] = bf_hi(o.x); v[2] = bf_lo(o.y); v[3] = bf_hi(o.y);
    v[4] = bf_lo(o.z); v[5] = bf_hi(o.z); v[6] = bf_lo(o.w); v[7] = bf_hi(o.w);
  }
  static __device__ __forceinline__ void store(Half* p, const float* v) {
    uint4 r;
    r.x = pack2(v[0], v[1]); r.y = pack2(v[2], v[3]); r.z = pack2(v[4], v[5]); r.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = r;
  }
};
template <> struct EpVec<float> {
  static __device__ __forceinline__ void load(const float* p, float* v) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
};
// out[t] (+)= sum_k w[t][k] * back[slot[t][k]]: f32 accumulation in k order, one rounding to the output type
// (a2a_combine_recv.cu:96-140 + core/combine_utils.cuh Acc::add = fma chain from the destination or from zero)
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void ep_combine_kernel(const unsigned char* __restrict__ back, int row_bytes,
                                                         const int32_t* __restrict__ slot, const float* __restrict__ weights,
                                                         long weights_stride, int topk, int hidden, int num_tokens,
                                                         const int32_t* __restrict__ bound_m, TOUT* __restrict__ out,
                                                         long out_stride, int accumulate) {
  const int t = blockIdx.x;
  if (t >= ep_bound(num_tokens, bound_m)) return;
  TOUT* dst = out + (long)t * out_stride;
  for (int i = threadIdx.x * 8; i < hidden; i += 256 * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (accumulate) EpVec<TOUT>::load(dst + i, acc);
    for (int k = 0; k < topk; ++k) {
      const int sl = slot[t * topk + k];
      if (sl < 0) continue;   // pair with an invalid expert id (ep_pack_kernel)
      const float w = weights[(long)t * weights_stride + k];
      float v[8];
      EpVec<TIN>::load(reinterpret_cast<const TIN*>(back + (long)sl * row_bytes) + i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, v[j], acc[j]);
    }
    EpVec<TOUT>::store(dst + i, acc);
  }
}

struct Ep;
struct EpHub { int world = 0; std::vector<Ep*> eps; };

struct Ep {
  Comm* comm = nullptr;
  EpHub* hub = nullptr;
  int rank = 0, world = 1, hidden = 0, hidden_scale = 0, max_tokens = 0, max_recv = 0, num_experts = 0, topk = 0, epr = 0;
  int pad = 1, in_elem = 2, out_elem = 2, out_dtype = PEGAINFER_SCALAR_BF16;
  int pay_bytes = 0, scale_off = 0, wrow = 0;   // dispatch wire row: payload bytes, offset of the scale plane, row bytes
  int cpay_bytes = 0, crow = 0;                 // combine wire row
  // device
  int32_t *d_rank_in_bucket = nullptr, *d_counts = nullptr, *d_offsets = nullptr, *d_slot = nullptr;
  int32_t *d_counts_all = nullptr, *d_row_map = nullptr;
  uint32_t* d_tokens_per_expert = nullptr;
  unsigned char *d_send = nullptr, *d_recv = nullptr, *d_back_send = nullptr, *d_back_recv = nullptr;
  // pinned host mirrors
  int32_t *h_counts = nullptr, *h_counts_recv = nullptr, *h_counts_all = nullptr, *h_row_map = nullptr;
  uint32_t* h_tokens_per_expert = nullptr;
  std::vector<int64_t> send_off, send_tot, recv_off, recv_tot;   // rows per peer
  int num_tokens = 0, num_recv = 0, num_recv_padded = 0;
  bool sent = false, received = false, combine_sent = false;
  std::string err;
};

#define PE_HIP(e, expr)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) { (e)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return -1; } \
  } while (0)
#define PE_NCCL(e, expr)                                                                         \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess) { (e)->err = std::string(#expr) + ": " + ncclGetErrorString(r_); return -1; } \
  } while (0)

static inline int round16(int64_t n) { return static_cast<int>((n + 15) / 16 * 16); }

static void ep_free(Ep* e) {
  void* dev[] = {e->d_rank_in_bucket, e->d_counts, e->d_offsets, e->d_slot, e->d_counts_all, e->d_row_map,
                 e->d_tokens_per_expert, e->d_send, e->d_recv, e->d_back_send, e->d_back_recv};
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = {e->h_counts, e->h_counts_recv, e->h_counts_all, e->h_row_map, e->h_tokens_per_expert};
  for (void* p : host) if (p) (void)hipHostFree(p);
}

// rows a rank receives for the (source-major) count table `recv` [world][epr], with every local expert's group padded
// to a multiple of expert_padding (a2a_worker.rs:598-606)
static int64_t ep_padded_rows(const Ep* e, const int32_t* recv) {
  int64_t tot = 0;
  for (int le = 0; le < e->epr; ++le) {
    int64_t n = 0;
    for (int r = 0; r < e->world; ++r) n += recv[r * e->epr + le];
    tot += (n + e->pad - 1) / e->pad * e->pad;
  }
  return tot;
}

// per-peer row totals and offsets from the (rank, local expert) count tables
static void ep_layout(Ep* e) {
  int64_t so = 0, ro = 0;
  for (int r = 0; r < e->world; ++r) {
    int64_t s = 0, v = 0;
    for (int le = 0; le < e->epr; ++le) { s += e->h_counts[r * e->epr + le]; v += e->h_counts_recv[r * e->epr + le]; }
    e->send_off[r] = so; e->send_tot[r] = s; so += s;
    e->recv_off[r] = ro; e->recv_tot[r] = v; ro += v;
  }
  e->num_recv = static_cast<int>(ro);
  e->num_recv_padded = static_cast<int>(ep_padded_rows(e, e->h_counts_recv));
}

// destination row of every received (source-major) wire row in the caller's buffers: expert-major, each expert's group
// starting at a multiple of expert_padding, inside a group by source rank then source order; + tokens per local expert
static int ep_build_row_map(Ep* e, hipStream_t s) {
  if (e->num_recv_padded > e->max_recv) { e->err = "dispatch: received rows (expert groups padded) exceed max_recv_tokens"; return -1; }
  int64_t base = 0;
  for (int le = 0; le < e->epr; ++le) {
    uint32_t n_le = 0;
    for (int r = 0; r < e->world; ++r) {
      int64_t within = 0;
      for (int l2 = 0; l2 < le; ++l2) within += e->h_counts_recv[r * e->epr + l2];
      const int n = e->h_counts_recv[r * e->epr + le];
      for (int i = 0; i < n; ++i) e->h_row_map[e->recv_off[r] + within + i] = static_cast<int32_t>(base + n_le + i);
      n_le += static_cast<uint32_t>(n);
    }
    e->h_tokens_per_expert[le] = n_le;
    base += ((int64_t)n_le + e->pad - 1) / e->pad * e->pad;
  }
  if (e->num_recv > 0)
    PE_HIP(e, hipMemcpyAsync(e->d_row_map, e->h_row_map, (size_t)e->num_recv * 4, hipMemcpyHostToDevice, s));
  PE_HIP(e, hipMemcpyAsync(e->d_tokens_per_expert, e->h_tokens_per_expert, (size_t)e->epr * 4, hipMemcpyHostToDevice, s));
  return 0;
}

// RCCL transport: rows [off[r], off[r] + tot[r]) of `send` go to rank r, `recv` is filled peer by peer
static int ep_exchange_rows(Ep* e, int row_bytes, const unsigned char* send, const std::vector<int64_t>& soff,
                            const std::vector<int64_t>& stot, unsigned char* recv, const std::vector<int64_t>& roff,
                            const std::vector<int64_t>& rtot, hipStream_t s) {
  const size_t row = (size_t)row_bytes;
  if (e->world == 1) {
    if (stot[0] > 0) PE_HIP(e, hipMemcpyAsync(recv, send, (size_t)stot[0] * row, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  PE_NCCL(e, ncclGroupStart());
  for (int r = 0; r < e->world; ++r) {
    if (stot[r] > 0) PE_NCCL(e, ncclSend(send + soff[r] * row, (size_t)stot[r] * row, ncclInt8, r, e->comm->nccl, s));
    if (rtot[r] > 0) PE_NCCL(e, ncclRecv(recv + roff[r] * row, (size_t)rtot[r] * row, ncclInt8, r, e->comm->nccl, s));
  }
  PE_NCCL(e, ncclGroupEnd());
  return 0;
}

}  // namespace pc

using namespace pc;

extern "C" {

int32_t pegainfer_comm_unique_id(void* out_128_bytes) {
  ncclUniqueId id;
  if (!out_128_bytes || ncclGetUniqueId(&id) != ncclSuccess) return -1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(out_128_bytes, &id, 128);
  return 0;
}

static pegainfer_comm_t comm_create(int32_t device_ordinal, int32_t rank, int32_t world, const void* unique_id_128, bool rccl) {
  if (world < 1 || rank < 0 || rank >= world || (rccl && world > 1 && !unique_id_128)) return nullptr;
  if (hipSetDevice(device_ordinal) != hipSuccess) return nullptr;
  Comm* c = new Comm();
  c->device = device_ordinal; c->rank = rank; c->world = world;
  const char* to = getenv("PEGAINFER_ONESHOT_TIMEOUT_MS");
  c->os_timeout_ticks = (unsigned long long)(to && *to ? atoll(to) : 10000) * 100000ull;   // wall_clock64: 100 MHz
  bool ok = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) == hipSuccess;
  // before the (collective) RCCL init: a rank that cannot allocate 1 KB fails here, like any rank that dies before init
  if (ok && rccl && world > 1) ok = hipMalloc(reinterpret_cast<void**>(&c->os_xchg), (size_t)(kOsMaxWorld + 1) * 64 + 64) == hipSuccess;
  if (ok && rccl && world > 1) {
    ncclUniqueId id;
    std::memcpy(&id, unique_id_128, 128);
    ok = ncclCommInitRank(&c->nccl, world, id, rank) == ncclSuccess;
  }
  if (!ok) { pegainfer_comm_destroy(c); return nullptr; }
  return c;
}
pegainfer_comm_t pegainfer_comm_create(int32_t device_ordinal, int32_t rank, int32_t world, const void* unique_id_128) {
  return comm_create(device_ordinal, rank, world, unique_id_128, true);
}
pegainfer_comm_t pegainfer_comm_create_peer_only(int32_t device_ordinal, int32_t rank, int32_t world) {
  return world > kOsMaxWorld ? nullptr : comm_create(device_ordinal, rank, world, nullptr, false);
}

// ---- one-shot path: slab export / attach ----
int32_t pegainfer_comm_oneshot_handle(pegainfer_comm_t h, void* out_64_bytes) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !out_64_bytes) return -1;
  if (c->world > kOsMaxWorld) { c->err = "one-shot all-reduce: at most 8 ranks (one xGMI node)"; return -1; }
  PC_HIP(c, hipSetDevice(c->device));
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t ipc;
  if (!c->os_slab) {
    // fine-grained device memory: peers' stores over xGMI and this device's cache-bypassing loads meet in HBM.  If the
    // runtime cannot allocate or export that kind, fall back to a plain hipMalloc slab (every access to it is a
    // system-scope write-through store or cache-bypassing load anyway)
    for (int attempt = 0; attempt < 2 && !c->os_slab; ++attempt) {
      unsigned char* p = nullptr;
      hipError_t e = attempt == 0 ? hipExtMallocWithFlags(reinterpret_cast<void**>(&p), kOsSlabBytes, hipDeviceMallocFinegrained)
                                  : hipMalloc(reinterpret_cast<void**>(&p), kOsSlabBytes);
      if (e == hipSuccess && hipIpcGetMemHandle(&ipc, p) == hipSuccess) { c->os_slab = p; break; }
      (void)hipGetLastError();
      if (p) (void)hipFree(p);
    }
    if (!c->os_slab) { c->err = "one-shot all-reduce: hipIpcGetMemHandle failed for the slab (fine-grained and plain)"; return -1; }
    PC_HIP(c, hipMemset(c->os_slab, 0, kOsSlabBytes));
    PC_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->os_epoch), (kOsMaxWg + 4) * sizeof(uint32_t)));
    PC_HIP(c, hipMemset(c->os_epoch, 0, (kOsMaxWg + 4) * sizeof(uint32_t)));
    c->os_status = c->os_epoch + kOsMaxWg;
    PC_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->os_stage), kOsMaxBytes));
    PC_HIP(c, hipDeviceSynchronize());
  }
  PC_HIP(c, hipIpcGetMemHandle(&ipc, c->os_slab));
  std::memcpy(out_64_bytes, &ipc, 64);
  return 0;
}
int32_t pegainfer_comm_oneshot_attach(pegainfer_comm_t h, const void* handles_world_x_64) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !handles_world_x_64) return -1;
  if (!c->os_slab) { c->err = "oneshot_attach before oneshot_handle"; return -1; }
  PC_HIP(c, hipSetDevice(c->device));
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) { c->os_peers.slab[r] = c->os_slab; continue; }
    if (c->os_opened[r]) continue;
    hipIpcMemHandle_t ipc;
    std::memcpy(&ipc, static_cast<const unsigned char*>(handles_world_x_64) + (size_t)r * 64, 64);
    void* p = nullptr;
    PC_HIP(c, hipIpcOpenMemHandle(&p, ipc, hipIpcMemLazyEnablePeerAccess));
    c->os_peers.slab[r] = static_cast<unsigned char*>(p);
    c->os_opened[r] = true;
  }
  c->os_active = c->world > 1;
  return 0;
}
// RCCL communicators: export, all-gather the handles over RCCL itself, attach.  Call outside graph capture.
// COLLECTIVE: every rank of the communicator must call it, and every rank takes part in both RCCL exchanges whatever
// happened locally - a rank whose export or mapping failed sends a zero handle / a zero success flag instead of
// returning early (its peers would otherwise block in the all-gather forever).  The path is switched on only when the
// min over the ranks' success flags is 1; otherwise it stays off on EVERY rank and the all-reduces keep using RCCL.
// PEGAINFER_TP_ONESHOT / any other opt-out must therefore be decided identically on all ranks BEFORE calling this.
int32_t pegainfer_comm_oneshot_enable(pegainfer_comm_t h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return -1;
  if (c->world == 1) return 0;
  if (!c->nccl) { c->err = "oneshot_enable needs an RCCL communicator (peer-only ones exchange handles out of band)"; return -1; }
  unsigned char mine[64] = {};
  int32_t ok = pegainfer_comm_oneshot_handle(h, mine) == 0 ? 1 : 0;
  std::string local_err = ok ? std::string() : c->err;
  if (!ok) std::memset(mine, 0, sizeof(mine));
  std::vector<unsigned char> all((size_t)c->world * 64);
  // exchange 1: the handles, through the buffer allocated with the communicator - from here on every rank enters BOTH
  // collectives whatever happens locally
  unsigned char* d = c->os_xchg;
  if (!d || c->world > kOsMaxWorld) { c->err = "oneshot_enable: communicator has no exchange buffer (world too large?)"; ok = 0; }
  if (!d) return -1;   // cannot happen for a communicator built by pegainfer_comm_create with world > 1
  bool xfer = hipMemcpy(d, mine, 64, hipMemcpyHostToDevice) == hipSuccess;
  xfer = (ncclAllGather(d, d + 64, 64, ncclInt8, c->nccl, c->comm_stream) == ncclSuccess) && xfer;
  xfer = (hipStreamSynchronize(c->comm_stream) == hipSuccess) && xfer;
  xfer = xfer && hipMemcpy(all.data(), d + 64, all.size(), hipMemcpyDeviceToHost) == hipSuccess;
  if (!xfer) { ok = 0; if (local_err.empty()) local_err = "oneshot_enable: handle exchange over RCCL failed"; }
  if (ok) {
    static const unsigned char zero[64] = {};
    for (int r = 0; r < c->world && ok; ++r)
      if (std::memcmp(all.data() + (size_t)r * 64, zero, 64) == 0) { ok = 0; local_err = "oneshot_enable: rank " + std::to_string(r) + " could not export its slab"; }
  }
  if (ok && pegainfer_comm_oneshot_attach(h, all.data())) { ok = 0; local_err = c->err; }
  c->os_active = false;   // attach switched it on locally; the world decides below
  // exchange 2: min over the success flags.  It is also the barrier the protocol needs: nobody may push into a slab
  // before its owner zeroed it and everyone mapped it
  int32_t* flag = reinterpret_cast<int32_t*>(d + (size_t)(c->world + 1) * 64);
  bool red = hipMemcpy(flag, &ok, 4, hipMemcpyHostToDevice) == hipSuccess;
  red = (ncclAllReduce(flag, flag, 1, ncclInt32, ncclMin, c->nccl, c->comm_stream) == ncclSuccess) && red;
  red = (hipStreamSynchronize(c->comm_stream) == hipSuccess) && red;
  int32_t all_ok = 0;
  red = red && hipMemcpy(&all_ok, flag, 4, hipMemcpyDeviceToHost) == hipSuccess;
  if (!red) { c->err = "oneshot_enable: closing all-reduce failed"; return -1; }
  if (all_ok != 1) {
    c->err = local_err.empty() ? "oneshot_enable: a peer could not export or map the slabs; all ranks stay on RCCL" : local_err;
    return -1;
  }
  c->os_active = true;
  return 0;
}
int32_t pegainfer_comm_oneshot_active(pegainfer_comm_t h) { return h && static_cast<Comm*>(h)->os_active ? 1 : 0; }
// 0 = every bounded wait so far completed; else 0x100 | bitmask of the ranks whose flag never arrived (synchronises)
int32_t pegainfer_comm_oneshot_status(pegainfer_comm_t h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return -1;
  if (!c->os_status) return 0;
  uint32_t st3[3] = {0, 0, 0};
  if (hipMemcpy(st3, c->os_status, sizeof(st3), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (st3[0]) c->err = "one-shot all-reduce: wait expired, mask " + std::to_string(st3[0] & 0xff) + " epoch " +
                      std::to_string(st3[1]) + " segment " + std::to_string(st3[2]);
  return (int32_t)st3[0];
}

// device address of the 3-word status block {0x100 | mask, epoch, segment} (NULL before the slab exists): a runtime that
// issues one-shot all-reduces inside its own captured step copies it back together with the step's results
const uint32_t* pegainfer_comm_oneshot_status_ptr(pegainfer_comm_t h) { return h ? static_cast<Comm*>(h)->os_status : nullptr; }

void pegainfer_comm_destroy(pegainfer_comm_t h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return;
  for (int r = 0; r < kOsMaxWorld; ++r)
    if (c->os_opened[r]) (void)hipIpcCloseMemHandle(c->os_peers.slab[r]);
  if (c->os_slab) (void)hipFree(c->os_slab);
  if (c->os_epoch) (void)hipFree(c->os_epoch);
  if (c->os_stage) (void)hipFree(c->os_stage);
  if (c->os_xchg) (void)hipFree(c->os_xchg);
  if (c->nccl) ncclCommDestroy(c->nccl);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  delete c;
}
const char* pegainfer_comm_last_error(pegainfer_comm_t h) { return h ? static_cast<Comm*>(h)->err.c_str() : "null comm"; }
int32_t pegainfer_comm_rank(pegainfer_comm_t h) { return h ? static_cast<Comm*>(h)->rank : -1; }
int32_t pegainfer_comm_world(pegainfer_comm_t h) { return h ? static_cast<Comm*>(h)->world : -1; }

int32_t pegainfer_comm_all_reduce_bf16(pegainfer_comm_t h, Half* data, int64_t n, pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !data || n < 0) return -1;
  if (c->world == 1 || n == 0) return 0;
  if (const int r = oneshot_all_reduce(c, data, n, 2, st(stream))) return r < 0 ? -1 : 0;   // <= 64 KB: peer-access path
  if (!c->nccl) { c->err = "peer-only communicator: the one-shot path is not attached (oneshot_handle / oneshot_attach first)"; return -1; }
  PC_NCCL(c, ncclAllReduce(data, data, (size_t)n, ncclBfloat16, ncclSum, c->nccl, st(stream)));
  return 0;
}
int32_t pegainfer_comm_all_reduce_f32(pegainfer_comm_t h, float* data, int64_t n, pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !data || n < 0) return -1;
  if (c->world == 1 || n == 0) return 0;
  if (const int r = oneshot_all_reduce(c, data, n, 4, st(stream))) return r < 0 ? -1 : 0;
  if (!c->nccl) { c->err = "peer-only communicator: the one-shot path is not attached (oneshot_handle / oneshot_attach first)"; return -1; }
  PC_NCCL(c, ncclAllReduce(data, data, (size_t)n, ncclFloat32, ncclSum, c->nccl, st(stream)));
  return 0;
}
int32_t pegainfer_comm_all_reduce_bf16_to_f32(pegainfer_comm_t h, const Half* in, float* out_f32, int64_t n,
                                              pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !in || !out_f32 || n < 0) return -1;
  if (n == 0) return 0;
  if (cast_to_f32(c, in, out_f32, n, st(stream))) return -1;
  if (c->world > 1) return pegainfer_comm_all_reduce_f32(h, out_f32, n, stream);   // size-dispatched like the plain verb
  return 0;
}
int32_t pegainfer_comm_all_reduce_bf16_via_f32(pegainfer_comm_t h, Half* data, int64_t n, pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !data || n < 0) return -1;
  if (n == 0) return 0;
  // the scratch grows outside graph capture only: call once with the largest n before capturing
  if (reserve_f32(c, (size_t)n)) return -1;
  if (pegainfer_comm_all_reduce_bf16_to_f32(h, data, c->scratch, n, stream)) return -1;
  return cast_to_bf16(c, c->scratch, data, n, st(stream));
}
int32_t pegainfer_comm_all_gather(pegainfer_comm_t h, const void* local, void* gathered, int64_t n_local,
                                  int32_t elem_bytes, pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !local || !gathered || n_local < 0 || (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)) return -1;
  if (n_local == 0) return 0;
  const size_t bytes = (size_t)n_local * elem_bytes;
  if (c->world == 1) {
    if (local != gathered) PC_HIP(c, hipMemcpyAsync(gathered, local, bytes, hipMemcpyDeviceToDevice, st(stream)));
    return 0;
  }
  PC_NCCL(c, ncclAllGather(local, gathered, bytes, ncclInt8, c->nccl, st(stream)));
  return 0;
}
int32_t pegainfer_comm_reduce_scatter_f32(pegainfer_comm_t h, const float* global, float* local, int64_t n_local,
                                          pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !global || !local || n_local < 0) return -1;
  if (n_local == 0) return 0;
  if (c->world == 1) {
    if (global != local) PC_HIP(c, hipMemcpyAsync(local, global, (size_t)n_local * 4, hipMemcpyDeviceToDevice, st(stream)));
    return 0;
  }
  PC_NCCL(c, ncclReduceScatter(global, local, (size_t)n_local, ncclFloat32, ncclSum, c->nccl, st(stream)));
  return 0;
}
int32_t pegainfer_comm_all_to_all(pegainfer_comm_t h, const void* send, void* recv, int64_t n_per_rank, int32_t elem_bytes,
                                  pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !send || !recv || n_per_rank < 0 || elem_bytes < 1) return -1;
  const size_t bytes = (size_t)n_per_rank * elem_bytes;
  if (bytes == 0) return 0;
  if (c->world == 1) {
    if (send != recv) PC_HIP(c, hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st(stream)));
    return 0;
  }
  PC_NCCL(c, ncclGroupStart());
  for (int r = 0; r < c->world; ++r) {
    PC_NCCL(c, ncclSend(static_cast<const char*>(send) + r * bytes, bytes, ncclInt8, r, c->nccl, st(stream)));
    PC_NCCL(c, ncclRecv(static_cast<char*>(recv) + r * bytes, bytes, ncclInt8, r, c->nccl, st(stream)));
  }
  PC_NCCL(c, ncclGroupEnd());
  return 0;
}
int32_t pegainfer_comm_all_to_allv(pegainfer_comm_t h, const void* send, const int64_t* send_counts,
                                   const int64_t* send_offsets, void* recv, const int64_t* recv_counts,
                                   const int64_t* recv_offsets, int32_t elem_bytes, pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !send_counts || !send_offsets || !recv_counts || !recv_offsets || elem_bytes < 1) return -1;
  for (int r = 0; r < c->world; ++r)
    if (send_counts[r] < 0 || recv_counts[r] < 0 || send_offsets[r] < 0 || recv_offsets[r] < 0) return -1;
  if (c->world == 1) {
    if (send_counts[0] != recv_counts[0]) { c->err = "all_to_allv: world 1 needs send_counts[0] == recv_counts[0]"; return -1; }
    if (send_counts[0] > 0)
      PC_HIP(c, hipMemcpyAsync(static_cast<char*>(recv) + recv_offsets[0] * elem_bytes,
                               static_cast<const char*>(send) + send_offsets[0] * elem_bytes,
                               (size_t)send_counts[0] * elem_bytes, hipMemcpyDeviceToDevice, st(stream)));
    return 0;
  }
  PC_NCCL(c, ncclGroupStart());
  for (int r = 0; r < c->world; ++r) {
    if (send_counts[r] > 0)
      PC_NCCL(c, ncclSend(static_cast<const char*>(send) + send_offsets[r] * elem_bytes, (size_t)send_counts[r] * elem_bytes,
                          ncclInt8, r, c->nccl, st(stream)));
    if (recv_counts[r] > 0)
      PC_NCCL(c, ncclRecv(static_cast<char*>(recv) + recv_offsets[r] * elem_bytes, (size_t)recv_counts[r] * elem_bytes,
                          ncclInt8, r, c->nccl, st(stream)));
  }
  PC_NCCL(c, ncclGroupEnd());
  return 0;
}

pegainfer_stream_t pegainfer_comm_stream(pegainfer_comm_t h) { return h ? static_cast<Comm*>(h)->comm_stream : nullptr; }
int32_t pegainfer_comm_fence_in(pegainfer_comm_t h, pegainfer_stream_t compute_stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return -1;
  PC_HIP(c, hipEventRecord(c->ev_in, st(compute_stream)));
  PC_HIP(c, hipStreamWaitEvent(c->comm_stream, c->ev_in, 0));
  return 0;
}
int32_t pegainfer_comm_fence_out(pegainfer_comm_t h, pegainfer_stream_t compute_stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return -1;
  PC_HIP(c, hipEventRecord(c->ev_out, c->comm_stream));
  PC_HIP(c, hipStreamWaitEvent(st(compute_stream), c->ev_out, 0));
  return 0;
}

// ---- expert parallel ----
pegainfer_ep_hub_t pegainfer_ep_hub_create(int32_t world) {
  if (world < 1) return nullptr;
  EpHub* hub = new EpHub();
  hub->world = world;
  hub->eps.assign(world, nullptr);
  return hub;
}
void pegainfer_ep_hub_destroy(pegainfer_ep_hub_t h) { delete static_cast<EpHub*>(h); }

static int ep_elem_of(int32_t dtype) {
  return dtype == PEGAINFER_SCALAR_BF16 || dtype == PEGAINFER_SCALAR_F16 ? 2 : dtype == PEGAINFER_SCALAR_F32 ? 4 : 0;
}

pegainfer_ep_t pegainfer_ep_create(pegainfer_comm_t comm, pegainfer_ep_hub_t hub_h, const pegainfer_ep_topology_t* t,
                                   const pegainfer_ep_dtypes_t* d) {
  Comm* c = static_cast<Comm*>(comm);
  EpHub* hub = static_cast<EpHub*>(hub_h);
  if ((c == nullptr) == (hub == nullptr) || !t || !d) return nullptr;
  const int world = c ? c->world : hub->world;
  const int rank = c ? c->rank : static_cast<int>(t->rank);
  // what this port carries: pure EP (dp_size 1) inside one xGMI node; payload rows are opaque bytes of 1 / 2 / 4-byte
  // elements, scale planes are f32 (the reference reads them through float*, a2a_dispatch_send.cu:270); the combine
  // sums bf16 or f32 expert outputs into bf16 or f32 tokens
  if ((int)t->world_size != world || (int)t->rank != rank || t->dp_size != 1 || t->hidden_dim == 0 || t->hidden_dim % 8 ||
      t->max_num_tokens == 0 || t->max_recv_tokens == 0 || t->num_experts == 0 || t->num_experts % world ||
      t->num_experts_per_token == 0 || t->expert_padding == 0 || t->hidden_dim > (1u << 24) || t->num_experts > (1u << 20))
    return nullptr;
  if ((d->in_elemsize != 1 && d->in_elemsize != 2 && d->in_elemsize != 4) || (d->out_elemsize != 2 && d->out_elemsize != 4) ||
      ep_elem_of(d->out_dtype) == 0 || (t->hidden_dim_scale != 0 && d->scale_elemsize != 4) || d->out_dtype == PEGAINFER_SCALAR_F16)
    return nullptr;
  Ep* e = new Ep();
  e->comm = c; e->hub = hub; e->rank = rank; e->world = world;
  e->hidden = (int)t->hidden_dim; e->hidden_scale = (int)t->hidden_dim_scale; e->max_tokens = (int)t->max_num_tokens;
  e->max_recv = (int)t->max_recv_tokens; e->num_experts = (int)t->num_experts; e->topk = (int)t->num_experts_per_token;
  e->epr = e->num_experts / world; e->pad = (int)t->expert_padding;
  e->in_elem = (int)d->in_elemsize; e->out_elem = (int)d->out_elemsize; e->out_dtype = d->out_dtype;
  e->pay_bytes = e->hidden * e->in_elem; e->scale_off = round16(e->pay_bytes);
  e->wrow = e->scale_off + round16((int64_t)e->hidden_scale * 4);
  e->cpay_bytes = e->hidden * e->out_elem; e->crow = round16(e->cpay_bytes);
  const size_t pairs = (size_t)e->max_tokens * e->topk, ne = (size_t)e->num_experts;
  bool ok = hipMalloc((void**)&e->d_rank_in_bucket, pairs * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_counts, ne * 4) == hipSuccess && hipMalloc((void**)&e->d_offsets, ne * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_slot, pairs * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_counts_all, ne * world * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_row_map, (size_t)e->max_recv * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_tokens_per_expert, (size_t)e->epr * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_send, pairs * e->wrow) == hipSuccess &&
            hipMalloc((void**)&e->d_recv, (size_t)e->max_recv * e->wrow) == hipSuccess &&
            hipMalloc((void**)&e->d_back_send, (size_t)e->max_recv * e->crow) == hipSuccess &&
            hipMalloc((void**)&e->d_back_recv, pairs * e->crow) == hipSuccess &&
            hipHostMalloc((void**)&e->h_counts, ne * 4) == hipSuccess &&
            hipHostMalloc((void**)&e->h_counts_recv, ne * 4) == hipSuccess &&
            hipHostMalloc((void**)&e->h_counts_all, ne * world * 4) == hipSuccess &&
            hipHostMalloc((void**)&e->h_row_map, (size_t)e->max_recv * 4) == hipSuccess &&
            hipHostMalloc((void**)&e->h_tokens_per_expert, (size_t)e->epr * 4) == hipSuccess;
  if (!ok) { ep_free(e); delete e; return nullptr; }
  e->send_off.assign(world, 0); e->send_tot.assign(world, 0); e->recv_off.assign(world, 0); e->recv_tot.assign(world, 0);
  if (hub) hub->eps[rank] = e;
  return e;
}
void pegainfer_ep_destroy(pegainfer_ep_t h) {
  Ep* e = static_cast<Ep*>(h);
  if (!e) return;
  if (e->hub && e->hub->eps[e->rank] == e) e->hub->eps[e->rank] = nullptr;
  ep_free(e);
  delete e;
}
const char* pegainfer_ep_last_error(pegainfer_ep_t h) { return h ? static_cast<Ep*>(h)->err.c_str() : "null ep"; }
const uint32_t* pegainfer_ep_tokens_per_expert_ptr(pegainfer_ep_t h) { return h ? static_cast<Ep*>(h)->d_tokens_per_expert : nullptr; }
int32_t pegainfer_ep_tokens_per_expert_host(pegainfer_ep_t h, uint32_t* out, int32_t n) {
  Ep* e = static_cast<Ep*>(h);
  if (!e || !out || n != e->epr) return -1;
  std::memcpy(out, e->h_tokens_per_expert, (size_t)n * 4);
  return 0;
}
int32_t pegainfer_ep_num_recv_tokens(pegainfer_ep_t h) { return h ? static_cast<Ep*>(h)->num_recv : -1; }
int32_t pegainfer_ep_num_padded_recv_tokens(pegainfer_ep_t h) { return h ? static_cast<Ep*>(h)->num_recv_padded : -1; }

int32_t pegainfer_ep_dispatch_send(pegainfer_ep_t h, size_t num_tokens, const void* x_ptr, size_t x_stride,
                                   const void* x_scale_ptr, size_t x_scale_stride_elem, size_t x_scale_stride_token,
                                   const int32_t* indices, size_t indices_stride, const float* weights,
                                   size_t weights_stride, const int32_t* bound_m_ptr, pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  (void)weights; (void)weights_stride;  // the weights stay with the source rank: they are applied in combine_recv
  if (!e) return -1;
  if (num_tokens > (size_t)e->max_tokens) { e->err = "dispatch_send: num_tokens exceeds max_num_tokens"; return -1; }
  if (num_tokens > 0 && (!x_ptr || !indices || x_stride < (size_t)e->pay_bytes || indices_stride < (size_t)e->topk)) {
    e->err = "dispatch_send: null x / indices or a stride shorter than a row";
    return -1;
  }
  if (x_scale_ptr && e->hidden_scale == 0) { e->err = "dispatch_send: scale plane given but hidden_dim_scale == 0"; return -1; }
  hipStream_t s = st(stream);
  e->num_tokens = (int)num_tokens;
  e->sent = e->received = e->combine_sent = false;
  const int pairs = e->num_tokens * e->topk;
  // route: stable rank inside each expert bucket, bucket offsets, pack
  ep_rank_kernel<<<(e->num_experts + 3) / 4, 256, 0, s>>>(indices, (long)indices_stride, e->topk, e->num_tokens, bound_m_ptr,
                                                          e->num_experts, e->d_rank_in_bucket, e->d_counts);
  ep_offsets_kernel<<<1, 256, 0, s>>>(e->d_counts, e->num_experts, e->d_offsets);
  if (pairs > 0)
    ep_pack_kernel<<<pairs, 256, 0, s>>>(static_cast<const unsigned char*>(x_ptr), (long)x_stride, e->pay_bytes,
                                         static_cast<const float*>(x_scale_ptr), (long)x_scale_stride_elem,
                                         (long)x_scale_stride_token, e->hidden_scale, e->scale_off, indices,
                                         (long)indices_stride, e->d_rank_in_bucket, e->d_offsets, e->topk, e->num_tokens,
                                         bound_m_ptr, e->num_experts, e->wrow, e->d_send, e->d_slot);
  PE_HIP(e, hipGetLastError());
  if (e->comm) {
    // route counts: every rank learns the WHOLE (source rank, expert) table with one all-gather, so the overflow decision
    // below is the same on every rank and nobody is left waiting in a send / recv group for a peer that bailed out
    const size_t ne = (size_t)e->num_experts;
    if (e->world == 1) {
      PE_HIP(e, hipMemcpyAsync(e->h_counts_all, e->d_counts, ne * 4, hipMemcpyDeviceToHost, s));
    } else {
      PE_NCCL(e, ncclAllGather(e->d_counts, e->d_counts_all, ne, ncclInt32, e->comm->nccl, s));
      PE_HIP(e, hipMemcpyAsync(e->h_counts_all, e->d_counts_all, ne * e->world * 4, hipMemcpyDeviceToHost, s));
    }
    PE_HIP(e, hipStreamSynchronize(s));   // the per-peer row counts size the exchange (reference: worker-thread wait)
    std::memcpy(e->h_counts, e->h_counts_all + (size_t)e->rank * ne, ne * 4);
    std::vector<int32_t> tmp((size_t)e->world * e->epr);
    int overflow_rank = -1;
    for (int dst = 0; dst < e->world; ++dst) {
      for (int r = 0; r < e->world; ++r)
        for (int le = 0; le < e->epr; ++le) tmp[(size_t)r * e->epr + le] = e->h_counts_all[(size_t)r * ne + (size_t)dst * e->epr + le];
      if (dst == e->rank) std::memcpy(e->h_counts_recv, tmp.data(), tmp.size() * 4);
      if (overflow_rank < 0 && ep_padded_rows(e, tmp.data()) > e->max_recv) overflow_rank = dst;
    }
    ep_layout(e);
    if (overflow_rank >= 0) {   // collective: every rank sees the same table and returns here, before any row moves
      e->err = "dispatch: rank " + std::to_string(overflow_rank) + " would receive more rows than max_recv_tokens";
      return -1;
    }
    if (ep_exchange_rows(e, e->wrow, e->d_send, e->send_off, e->send_tot, e->d_recv, e->recv_off, e->recv_tot, s)) return -1;
  } else {
    PE_HIP(e, hipMemcpyAsync(e->h_counts, e->d_counts, (size_t)e->num_experts * 4, hipMemcpyDeviceToHost, s));
    PE_HIP(e, hipStreamSynchronize(s));   // loopback: peers pull from d_send in their dispatch_recv
  }
  e->sent = true;
  return 0;
}

int32_t pegainfer_ep_dispatch_recv(pegainfer_ep_t h, int32_t* out_num_tokens_ptr, void* out_x_ptr, size_t out_x_stride,
                                   void* out_x_scale_ptr, size_t out_x_scale_stride_elem, size_t out_x_scale_stride_token,
                                   pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  if (!e) return -1;
  if (!e->sent) { e->err = "dispatch_recv before dispatch_send"; return -1; }
  if (!out_x_ptr || out_x_stride < (size_t)e->pay_bytes) { e->err = "dispatch_recv: null out_x or a stride shorter than a row"; return -1; }
  if (out_x_scale_ptr && e->hidden_scale == 0) { e->err = "dispatch_recv: scale plane given but hidden_dim_scale == 0"; return -1; }
  hipStream_t s = st(stream);
  if (e->hub) {
    // loopback transport: pull what every virtual rank routed to this one
    for (int r = 0; r < e->world; ++r) {
      Ep* p = e->hub->eps[r];
      if (!p || !p->sent) { e->err = "loopback dispatch_recv before every virtual rank's dispatch_send"; return -1; }
      for (int le = 0; le < e->epr; ++le) e->h_counts_recv[r * e->epr + le] = p->h_counts[e->rank * e->epr + le];
    }
    ep_layout(e);   // my own send layout (needed by the combine) and the receive layout
    if (e->num_recv_padded > e->max_recv) { e->err = "dispatch: received rows (expert groups padded) exceed max_recv_tokens"; return -1; }
    for (int r = 0; r < e->world; ++r) {
      Ep* p = e->hub->eps[r];
      int64_t off = 0;
      for (int i = 0; i < e->rank * e->epr; ++i) off += p->h_counts[i];
      if (e->recv_tot[r] > 0)
        PE_HIP(e, hipMemcpyAsync(e->d_recv + e->recv_off[r] * e->wrow, p->d_send + off * e->wrow,
                                 (size_t)e->recv_tot[r] * e->wrow, hipMemcpyDeviceToDevice, s));
    }
  }
  if (ep_build_row_map(e, s)) return -1;
  if (e->num_recv > 0)
    ep_unpack_kernel<<<e->num_recv, 256, 0, s>>>(e->d_recv, e->wrow, e->pay_bytes, e->hidden_scale, e->scale_off, e->d_row_map,
                                                 static_cast<unsigned char*>(out_x_ptr), (long)out_x_stride,
                                                 static_cast<float*>(out_x_scale_ptr), (long)out_x_scale_stride_elem,
                                                 (long)out_x_scale_stride_token);
  PE_HIP(e, hipGetLastError());
  // the per-local-expert counts go to the caller's DEVICE array (a2a_dispatch_recv.cu:220-224)
  if (out_num_tokens_ptr)
    PE_HIP(e, hipMemcpyAsync(out_num_tokens_ptr, e->d_tokens_per_expert, (size_t)e->epr * 4, hipMemcpyDeviceToDevice, s));
  e->received = true;
  return 0;
}

int32_t pegainfer_ep_combine_send(pegainfer_ep_t h, const void* expert_x_ptr, size_t expert_x_stride, pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  if (!e) return -1;
  if (!e->received) { e->err = "combine_send before dispatch_recv"; return -1; }
  if (e->num_recv > 0 && (!expert_x_ptr || expert_x_stride < (size_t)e->cpay_bytes)) {
    e->err = "combine_send: null expert_x or a stride shorter than a row";
    return -1;
  }
  hipStream_t s = st(stream);
  // the caller's padded expert-major rows back into the source-major order they arrived in, then home
  if (e->num_recv > 0)
    ep_gather_kernel<<<e->num_recv, 256, 0, s>>>(static_cast<const unsigned char*>(expert_x_ptr), (long)expert_x_stride,
                                                 e->cpay_bytes, e->d_row_map, e->crow, e->d_back_send);
  PE_HIP(e, hipGetLastError());
  if (e->comm) {
    if (ep_exchange_rows(e, e->crow, e->d_back_send, e->recv_off, e->recv_tot, e->d_back_recv, e->send_off, e->send_tot, s)) return -1;
  } else {
    PE_HIP(e, hipStreamSynchronize(s));   // loopback: peers pull from d_back_send in their combine_recv
  }
  e->combine_sent = true;
  return 0;
}

int32_t pegainfer_ep_combine_recv(pegainfer_ep_t h, size_t num_tokens, size_t num_recv_tokens, int32_t in_dtype,
                                  void* out_tokens_ptr, size_t out_tokens_stride, const int32_t* indices_ptr,
                                  size_t indices_stride, const float* weights_ptr, size_t weights_stride,
                                  const int32_t* bound_m_ptr, int32_t accumulate, pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  (void)num_recv_tokens;                  // "currently ignored by the a2a combine_recv kernel" (moe_pplx.rs:238)
  (void)indices_ptr; (void)indices_stride;  // the route is remembered from dispatch_send (slot of every (token, k) pair)
  if (!e) return -1;
  if (!e->combine_sent) { e->err = "combine_recv before combine_send"; return -1; }
  if ((int)num_tokens != e->num_tokens) { e->err = "combine_recv: num_tokens differs from the dispatch"; return -1; }
  if (ep_elem_of(in_dtype) != e->out_elem || in_dtype == PEGAINFER_SCALAR_F16) {
    e->err = "combine_recv: in_dtype does not match out_elemsize (bf16 or f32 expert outputs)";
    return -1;
  }
  const int oel = ep_elem_of(e->out_dtype);
  if (num_tokens > 0 && (!out_tokens_ptr || !weights_ptr || out_tokens_stride < (size_t)e->hidden || weights_stride < (size_t)e->topk ||
                         (reinterpret_cast<uintptr_t>(out_tokens_ptr) & 15u) || (out_tokens_stride * oel) % 16)) {
    e->err = "combine_recv: out_tokens rows must be 16-byte aligned, strides at least a row";
    return -1;
  }
  hipStream_t s = st(stream);
  if (e->hub) {
    for (int r = 0; r < e->world; ++r) {
      Ep* p = e->hub->eps[r];
      if (!p || !p->combine_sent) { e->err = "loopback combine_recv before every virtual rank's combine_send"; return -1; }
      // rank r holds my rows at its source-major offset recv_off[me]
      if (e->send_tot[r] > 0)
        PE_HIP(e, hipMemcpyAsync(e->d_back_recv + e->send_off[r] * e->crow, p->d_back_send + p->recv_off[e->rank] * e->crow,
                                 (size_t)e->send_tot[r] * e->crow, hipMemcpyDeviceToDevice, s));
    }
  }
  if (num_tokens > 0) {
    const int nt = e->num_tokens;
#define EP_COMBINE(TIN, TOUT)                                                                                              \
  ep_combine_kernel<TIN, TOUT><<<nt, 256, 0, s>>>(e->d_back_recv, e->crow, e->d_slot, weights_ptr, (long)weights_stride,   \
                                                  e->topk, e->hidden, nt, bound_m_ptr, static_cast<TOUT*>(out_tokens_ptr), \
                                                  (long)out_tokens_stride, accumulate)
    const bool in32 = in_dtype == PEGAINFER_SCALAR_F32, out32 = e->out_dtype == PEGAINFER_SCALAR_F32;
    if (in32 && out32) EP_COMBINE(float, float);
    else if (in32) EP_COMBINE(float, Half);
    else if (out32) EP_COMBINE(Half, float);
    else EP_COMBINE(Half, Half);
#undef EP_COMBINE
  }
  PE_HIP(e, hipGetLastError());
  return 0;
}

}  // extern "C"
