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
#include <cstring>
#include <string>
#include <vector>

#include "pegainfer_comm.h"

extern "C" {
int32_t deepseek_bf16_to_f32_cuda(const Half* input, float* output, int32_t n, pegainfer_stream_t stream);
int32_t deepseek_f32_to_bf16_cuda(const float* input, Half* output, int32_t n, pegainfer_stream_t stream);
}

namespace pc {

struct Comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  float* scratch = nullptr;
  size_t scratch_n = 0;
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

// ---- expert-parallel kernels ----------------------------------------------------------------------------------

// Stable rank of every (token, k) pair inside its expert bucket: wave w scans all pairs for expert w with ballots,
// so a pair's position is the number of EARLIER pairs routed to the same expert - deterministic, no atomics.
__global__ __launch_bounds__(256) void ep_rank_kernel(const int32_t* __restrict__ indices, int n_pairs, int num_experts,
                                                      int32_t* __restrict__ rank_in_bucket, int32_t* __restrict__ counts) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= num_experts) return;
  int base = 0;
  for (int p0 = 0; p0 < n_pairs; p0 += 64) {
    const int p = p0 + lane;
    const bool hit = p < n_pairs && indices[p] == e;
    const unsigned long long m = __ballot(hit);
    if (hit) rank_in_bucket[p] = base + __popcll(m & ((1ull << lane) - 1ull));
    base += __popcll(m);
  }
  if (lane == 0) counts[e] = base;
}

// bucket offsets (expert order = (destination rank, local expert) order) by one workgroup, then the pack: one
// workgroup per pair copies row x[t] to its slot of the send buffer and records the slot for the combine
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
__global__ __launch_bounds__(256) void ep_pack_kernel(const Half* __restrict__ x, long x_stride,
                                                      const int32_t* __restrict__ indices,
                                                      const int32_t* __restrict__ rank_in_bucket,
                                                      const int32_t* __restrict__ offsets, int topk, int hidden,
                                                      int num_experts, Half* __restrict__ send,
                                                      int32_t* __restrict__ slot) {
  const int p = blockIdx.x, t = p / topk;
  const int ex = indices[p];
  if ((unsigned)ex >= (unsigned)num_experts) {   // not an expert id: the pair is routed nowhere and combines as zero
    if (threadIdx.x == 0) slot[p] = -1;
    return;
  }
  const int pos = offsets[ex] + rank_in_bucket[p];
  if (threadIdx.x == 0) slot[p] = pos;
  const uint4* src = reinterpret_cast<const uint4*>(x + (long)t * x_stride);
  uint4* dst = reinterpret_cast<uint4*>(send + (long)pos * hidden);
  for (int i = threadIdx.x; i < hidden / 8; i += 256) dst[i] = src[i];
}
// out row r = in row map[r] (gather) or out row map[r] = in row r (scatter)
template <bool SCATTER>
__global__ __launch_bounds__(256) void ep_rows_kernel(const Half* __restrict__ in, long in_stride, Half* __restrict__ out,
                                                      long out_stride, const int32_t* __restrict__ map, int hidden) {
  const int r = blockIdx.x, m = map[r];
  const uint4* src = reinterpret_cast<const uint4*>(in + (long)(SCATTER ? r : m) * in_stride);
  uint4* dst = reinterpret_cast<uint4*>(out + (long)(SCATTER ? m : r) * out_stride);
  for (int i = threadIdx.x; i < hidden / 8; i += 256) dst[i] = src[i];
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));
}
// out[t] (+)= sum_k w[t][k] * back[slot[t][k]], f32 accumulation in k order, one bf16 rounding
__global__ __launch_bounds__(256) void ep_combine_kernel(const Half* __restrict__ back, const int32_t* __restrict__ slot,
                                                         const float* __restrict__ weights, int topk, int hidden,
                                                         Half* __restrict__ out, long out_stride, int accumulate) {
  const int t = blockIdx.x;
  uint4* dst = reinterpret_cast<uint4*>(out + (long)t * out_stride);
  for (int i = threadIdx.x; i < hidden / 8; i += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (accumulate) {
      const uint4 o = dst[i];
      acc[0] = bf_lo(o.x); acc[1] = bf_hi(o.x); acc[2] = bf_lo(o.y); acc[3] = bf_hi(o.y);
      acc[4] = bf_lo(o.z); acc[5] = bf_hi(o.z); acc[6] = bf_lo(o.w); acc[7] = bf_hi(o.w);
    }
    for (int k = 0; k < topk; ++k) {
      const float w = weights[t * topk + k];
      const int sl = slot[t * topk + k];
      if (sl < 0) continue;   // pair with an invalid expert id (ep_pack_kernel)
      const uint4 v = reinterpret_cast<const uint4*>(back + (long)sl * hidden)[i];
      acc[0] = fmaf(w, bf_lo(v.x), acc[0]); acc[1] = fmaf(w, bf_hi(v.x), acc[1]);
      acc[2] = fmaf(w, bf_lo(v.y), acc[2]); acc[3] = fmaf(w, bf_hi(v.y), acc[3]);
      acc[4] = fmaf(w, bf_lo(v.z), acc[4]); acc[5] = fmaf(w, bf_hi(v.z), acc[5]);
      acc[6] = fmaf(w, bf_lo(v.w), acc[6]); acc[7] = fmaf(w, bf_hi(v.w), acc[7]);
    }
    uint4 r;
    r.x = pack2(acc[0], acc[1]); r.y = pack2(acc[2], acc[3]); r.z = pack2(acc[4], acc[5]); r.w = pack2(acc[6], acc[7]);
    dst[i] = r;
  }
}

struct Ep;
struct EpHub { int world = 0; std::vector<Ep*> eps; };

struct Ep {
  Comm* comm = nullptr;
  EpHub* hub = nullptr;
  int rank = 0, world = 1, hidden = 0, max_tokens = 0, max_recv = 0, num_experts = 0, topk = 0, epr = 0;
  // device
  int32_t *d_rank_in_bucket = nullptr, *d_counts = nullptr, *d_offsets = nullptr, *d_slot = nullptr;
  int32_t *d_counts_recv = nullptr, *d_row_map = nullptr;
  uint32_t* d_tokens_per_expert = nullptr;
  Half *d_send = nullptr, *d_recv = nullptr, *d_back_send = nullptr, *d_back_recv = nullptr;
  // pinned host mirrors
  int32_t *h_counts = nullptr, *h_counts_recv = nullptr, *h_row_map = nullptr;
  uint32_t* h_tokens_per_expert = nullptr;
  std::vector<int64_t> send_off, send_tot, recv_off, recv_tot;   // rows per peer
  int num_pairs = 0, num_recv = 0;
  bool sent = false;
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

// the row kernels move 16-byte vectors: row starts must be 16-byte aligned
static bool rows_aligned(const void* p, int64_t stride_elems) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (stride_elems & 7) == 0;
}

static void ep_free(Ep* e) {
  void* dev[] = {e->d_rank_in_bucket, e->d_counts, e->d_offsets, e->d_slot, e->d_counts_recv, e->d_row_map,
                 e->d_tokens_per_expert, e->d_send, e->d_recv, e->d_back_send, e->d_back_recv};
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = {e->h_counts, e->h_counts_recv, e->h_row_map, e->h_tokens_per_expert};
  for (void* p : host) if (p) (void)hipHostFree(p);
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
}

// expert-major row order of the received (source-major) rows + tokens per local expert
static int ep_build_row_map(Ep* e, hipStream_t s) {
  if (e->num_recv > e->max_recv) { e->err = "dispatch: received rows exceed max_recv_tokens"; return -1; }
  int dst = 0;
  for (int le = 0; le < e->epr; ++le) {
    uint32_t n_le = 0;
    for (int r = 0; r < e->world; ++r) {
      int64_t within = 0;
      for (int l2 = 0; l2 < le; ++l2) within += e->h_counts_recv[r * e->epr + l2];
      const int n = e->h_counts_recv[r * e->epr + le];
      for (int i = 0; i < n; ++i) e->h_row_map[dst++] = static_cast<int32_t>(e->recv_off[r] + within + i);
      n_le += static_cast<uint32_t>(n);
    }
    e->h_tokens_per_expert[le] = n_le;
  }
  if (e->num_recv > 0)
    PE_HIP(e, hipMemcpyAsync(e->d_row_map, e->h_row_map, (size_t)e->num_recv * 4, hipMemcpyHostToDevice, s));
  PE_HIP(e, hipMemcpyAsync(e->d_tokens_per_expert, e->h_tokens_per_expert, (size_t)e->epr * 4, hipMemcpyHostToDevice, s));
  return 0;
}

// RCCL transport: rows [off[r], off[r] + tot[r]) of `send` go to rank r, `recv` is filled peer by peer
static int ep_exchange_rows(Ep* e, const Half* send, const std::vector<int64_t>& soff, const std::vector<int64_t>& stot,
                            Half* recv, const std::vector<int64_t>& roff, const std::vector<int64_t>& rtot, hipStream_t s) {
  const size_t row = (size_t)e->hidden;
  if (e->world == 1) {
    if (stot[0] > 0) PE_HIP(e, hipMemcpyAsync(recv, send, (size_t)stot[0] * row * 2, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  PE_NCCL(e, ncclGroupStart());
  for (int r = 0; r < e->world; ++r) {
    if (stot[r] > 0) PE_NCCL(e, ncclSend(send + soff[r] * row, (size_t)stot[r] * row, ncclBfloat16, r, e->comm->nccl, s));
    if (rtot[r] > 0) PE_NCCL(e, ncclRecv(recv + roff[r] * row, (size_t)rtot[r] * row, ncclBfloat16, r, e->comm->nccl, s));
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

pegainfer_comm_t pegainfer_comm_create(int32_t device_ordinal, int32_t rank, int32_t world, const void* unique_id_128) {
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !unique_id_128)) return nullptr;
  if (hipSetDevice(device_ordinal) != hipSuccess) return nullptr;
  Comm* c = new Comm();
  c->device = device_ordinal; c->rank = rank; c->world = world;
  bool ok = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) == hipSuccess;
  if (ok && world > 1) {
    ncclUniqueId id;
    std::memcpy(&id, unique_id_128, 128);
    ok = ncclCommInitRank(&c->nccl, world, id, rank) == ncclSuccess;
  }
  if (!ok) { pegainfer_comm_destroy(c); return nullptr; }
  return c;
}

void pegainfer_comm_destroy(pegainfer_comm_t h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return;
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
  PC_NCCL(c, ncclAllReduce(data, data, (size_t)n, ncclBfloat16, ncclSum, c->nccl, st(stream)));
  return 0;
}
int32_t pegainfer_comm_all_reduce_f32(pegainfer_comm_t h, float* data, int64_t n, pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !data || n < 0) return -1;
  if (c->world == 1 || n == 0) return 0;
  PC_NCCL(c, ncclAllReduce(data, data, (size_t)n, ncclFloat32, ncclSum, c->nccl, st(stream)));
  return 0;
}
int32_t pegainfer_comm_all_reduce_bf16_to_f32(pegainfer_comm_t h, const Half* in, float* out_f32, int64_t n,
                                              pegainfer_stream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (!c || !in || !out_f32 || n < 0) return -1;
  if (n == 0) return 0;
  if (cast_to_f32(c, in, out_f32, n, st(stream))) return -1;
  if (c->world > 1) PC_NCCL(c, ncclAllReduce(out_f32, out_f32, (size_t)n, ncclFloat32, ncclSum, c->nccl, st(stream)));
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

pegainfer_ep_t pegainfer_ep_create(pegainfer_comm_t comm, pegainfer_ep_hub_t hub_h, int32_t rank, int32_t hidden_dim,
                                   int32_t max_num_tokens, int32_t max_recv_tokens, int32_t num_experts,
                                   int32_t num_experts_per_token) {
  Comm* c = static_cast<Comm*>(comm);
  EpHub* hub = static_cast<EpHub*>(hub_h);
  if ((c == nullptr) == (hub == nullptr)) return nullptr;
  const int world = c ? c->world : hub->world;
  if (c) rank = c->rank;
  if (rank < 0 || rank >= world || hidden_dim <= 0 || hidden_dim % 8 || max_num_tokens <= 0 || max_recv_tokens <= 0 ||
      num_experts <= 0 || num_experts % world || num_experts_per_token <= 0)
    return nullptr;
  Ep* e = new Ep();
  e->comm = c; e->hub = hub; e->rank = rank; e->world = world; e->hidden = hidden_dim; e->max_tokens = max_num_tokens;
  e->max_recv = max_recv_tokens; e->num_experts = num_experts; e->topk = num_experts_per_token; e->epr = num_experts / world;
  const size_t pairs = (size_t)max_num_tokens * num_experts_per_token, row = (size_t)hidden_dim * 2;
  bool ok = hipMalloc((void**)&e->d_rank_in_bucket, pairs * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_counts, (size_t)num_experts * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_offsets, (size_t)num_experts * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_slot, pairs * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_counts_recv, (size_t)num_experts * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_row_map, (size_t)max_recv_tokens * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_tokens_per_expert, (size_t)e->epr * 4) == hipSuccess &&
            hipMalloc((void**)&e->d_send, pairs * row) == hipSuccess &&
            hipMalloc((void**)&e->d_recv, (size_t)max_recv_tokens * row) == hipSuccess &&
            hipMalloc((void**)&e->d_back_send, (size_t)max_recv_tokens * row) == hipSuccess &&
            hipMalloc((void**)&e->d_back_recv, pairs * row) == hipSuccess &&
            hipHostMalloc((void**)&e->h_counts, (size_t)num_experts * 4) == hipSuccess &&
            hipHostMalloc((void**)&e->h_counts_recv, (size_t)num_experts * 4) == hipSuccess &&
            hipHostMalloc((void**)&e->h_row_map, (size_t)max_recv_tokens * 4) == hipSuccess &&
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
const uint32_t* pegainfer_ep_tokens_per_expert(pegainfer_ep_t h) { return h ? static_cast<Ep*>(h)->d_tokens_per_expert : nullptr; }
int32_t pegainfer_ep_tokens_per_expert_host(pegainfer_ep_t h, uint32_t* out, int32_t n) {
  Ep* e = static_cast<Ep*>(h);
  if (!e || !out || n != e->epr) return -1;
  std::memcpy(out, e->h_tokens_per_expert, (size_t)n * 4);
  return 0;
}
int32_t pegainfer_ep_num_recv_tokens(pegainfer_ep_t h) { return h ? static_cast<Ep*>(h)->num_recv : -1; }

int32_t pegainfer_ep_dispatch_send(pegainfer_ep_t h, int32_t num_tokens, const Half* x, int64_t x_stride_elems,
                                   const int32_t* indices, const float* weights, pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  (void)weights;  // the weights stay with the source rank: they are applied in combine_recv
  if (!e || num_tokens < 0 || num_tokens > e->max_tokens || (num_tokens > 0 && (!x || !indices)) || x_stride_elems < e->hidden)
    return -1;
  if (num_tokens > 0 && !rows_aligned(x, x_stride_elems)) { e->err = "dispatch_send: x rows must be 16-byte aligned"; return -1; }
  hipStream_t s = st(stream);
  e->num_pairs = num_tokens * e->topk;
  e->sent = false;
  // route: stable rank inside each expert bucket, bucket offsets, pack
  ep_rank_kernel<<<(e->num_experts + 3) / 4, 256, 0, s>>>(indices, e->num_pairs, e->num_experts, e->d_rank_in_bucket, e->d_counts);
  ep_offsets_kernel<<<1, 256, 0, s>>>(e->d_counts, e->num_experts, e->d_offsets);
  if (e->num_pairs > 0)
    ep_pack_kernel<<<e->num_pairs, 256, 0, s>>>(x, x_stride_elems, indices, e->d_rank_in_bucket, e->d_offsets, e->topk,
                                                e->hidden, e->num_experts, e->d_send, e->d_slot);
  PE_HIP(e, hipGetLastError());
  PE_HIP(e, hipMemcpyAsync(e->h_counts, e->d_counts, (size_t)e->num_experts * 4, hipMemcpyDeviceToHost, s));
  if (e->comm) {
    // route counts: epr ints to / from every peer, then the payload rows they announce
    if (e->world == 1) {
      PE_HIP(e, hipMemcpyAsync(e->h_counts_recv, e->d_counts, (size_t)e->num_experts * 4, hipMemcpyDeviceToHost, s));
    } else {
      PE_NCCL(e, ncclGroupStart());
      for (int r = 0; r < e->world; ++r) {
        PE_NCCL(e, ncclSend(e->d_counts + r * e->epr, e->epr, ncclInt32, r, e->comm->nccl, s));
        PE_NCCL(e, ncclRecv(e->d_counts_recv + r * e->epr, e->epr, ncclInt32, r, e->comm->nccl, s));
      }
      PE_NCCL(e, ncclGroupEnd());
      PE_HIP(e, hipMemcpyAsync(e->h_counts_recv, e->d_counts_recv, (size_t)e->num_experts * 4, hipMemcpyDeviceToHost, s));
    }
    PE_HIP(e, hipStreamSynchronize(s));   // the per-peer row counts size the exchange (reference: worker-thread wait)
    ep_layout(e);
    if (e->num_recv > e->max_recv) { e->err = "dispatch: received rows exceed max_recv_tokens"; return -1; }
    if (ep_exchange_rows(e, e->d_send, e->send_off, e->send_tot, e->d_recv, e->recv_off, e->recv_tot, s)) return -1;
  } else {
    PE_HIP(e, hipStreamSynchronize(s));   // loopback: peers pull from d_send in their dispatch_recv
  }
  e->sent = true;
  return 0;
}

int32_t pegainfer_ep_dispatch_recv(pegainfer_ep_t h, int32_t* out_num_tokens, Half* out_x, int64_t out_x_stride_elems,
                                   pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  if (!e || !e->sent || !out_x || out_x_stride_elems < e->hidden) return -1;
  if (!rows_aligned(out_x, out_x_stride_elems)) { e->err = "dispatch_recv: out_x rows must be 16-byte aligned"; return -1; }
  hipStream_t s = st(stream);
  if (e->hub) {
    // loopback transport: pull what every virtual rank routed to this one
    for (int r = 0; r < e->world; ++r) {
      Ep* p = e->hub->eps[r];
      if (!p || !p->sent) { e->err = "loopback dispatch_recv before every virtual rank's dispatch_send"; return -1; }
      for (int le = 0; le < e->epr; ++le) e->h_counts_recv[r * e->epr + le] = p->h_counts[e->rank * e->epr + le];
    }
    ep_layout(e);   // my own send layout (needed by the combine) and the receive layout
    if (e->num_recv > e->max_recv) { e->err = "dispatch: received rows exceed max_recv_tokens"; return -1; }
    for (int r = 0; r < e->world; ++r) {
      Ep* p = e->hub->eps[r];
      int64_t off = 0;
      for (int i = 0; i < e->rank * e->epr; ++i) off += p->h_counts[i];
      if (e->recv_tot[r] > 0)
        PE_HIP(e, hipMemcpyAsync(e->d_recv + e->recv_off[r] * e->hidden, p->d_send + off * e->hidden,
                                 (size_t)e->recv_tot[r] * e->hidden * 2, hipMemcpyDeviceToDevice, s));
    }
  }
  if (ep_build_row_map(e, s)) return -1;
  if (e->num_recv > 0)
    ep_rows_kernel<false><<<e->num_recv, 256, 0, s>>>(e->d_recv, e->hidden, out_x, out_x_stride_elems, e->d_row_map, e->hidden);
  PE_HIP(e, hipGetLastError());
  if (out_num_tokens) *out_num_tokens = e->num_recv;
  return 0;
}

int32_t pegainfer_ep_combine_send(pegainfer_ep_t h, const Half* expert_x, int64_t expert_x_stride_elems,
                                  pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  if (!e || !e->sent || (e->num_recv > 0 && !expert_x) || expert_x_stride_elems < e->hidden) return -1;
  if (e->num_recv > 0 && !rows_aligned(expert_x, expert_x_stride_elems)) {
    e->err = "combine_send: expert_x rows must be 16-byte aligned";
    return -1;
  }
  hipStream_t s = st(stream);
  // expert-major rows back into the source-major order they arrived in, then home
  if (e->num_recv > 0)
    ep_rows_kernel<true><<<e->num_recv, 256, 0, s>>>(expert_x, expert_x_stride_elems, e->d_back_send, e->hidden, e->d_row_map, e->hidden);
  PE_HIP(e, hipGetLastError());
  if (e->comm) return ep_exchange_rows(e, e->d_back_send, e->recv_off, e->recv_tot, e->d_back_recv, e->send_off, e->send_tot, s);
  PE_HIP(e, hipStreamSynchronize(s));   // loopback: peers pull from d_back_send in their combine_recv
  return 0;
}

int32_t pegainfer_ep_combine_recv(pegainfer_ep_t h, int32_t num_tokens, Half* out_tokens, int64_t out_stride_elems,
                                  const int32_t* indices, const float* weights, int32_t accumulate,
                                  pegainfer_stream_t stream) {
  Ep* e = static_cast<Ep*>(h);
  (void)indices;  // the route is remembered from dispatch_send (slot of every (token, k) pair)
  if (!e || !e->sent || num_tokens * e->topk != e->num_pairs || (num_tokens > 0 && (!out_tokens || !weights)) ||
      out_stride_elems < e->hidden)
    return -1;
  if (num_tokens > 0 && !rows_aligned(out_tokens, out_stride_elems)) {
    e->err = "combine_recv: out_tokens rows must be 16-byte aligned";
    return -1;
  }
  hipStream_t s = st(stream);
  if (e->hub) {
    for (int r = 0; r < e->world; ++r) {
      Ep* p = e->hub->eps[r];
      if (!p) { e->err = "loopback combine_recv: missing virtual rank"; return -1; }
      // rank r holds my rows at its source-major offset recv_off[me]
      if (e->send_tot[r] > 0)
        PE_HIP(e, hipMemcpyAsync(e->d_back_recv + e->send_off[r] * e->hidden, p->d_back_send + p->recv_off[e->rank] * e->hidden,
                                 (size_t)e->send_tot[r] * e->hidden * 2, hipMemcpyDeviceToDevice, s));
    }
  }
  if (num_tokens > 0)
    ep_combine_kernel<<<num_tokens, 256, 0, s>>>(e->d_back_recv, e->d_slot, weights, e->topk, e->hidden, out_tokens,
                                                 out_stride_elems, accumulate);
  PE_HIP(e, hipGetLastError());
  return 0;
}

}  // extern "C"
