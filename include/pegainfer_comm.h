/*
 * pegainfer_comm.h - C ABI of the collective layer (part of libpegainfer_qwen3.so): the DeepSeek-V4 MP8 verbs of
 * the reference re-expressed over RCCL on the xGMI mesh of one MI355X node, one process per GPU.
 *
 *   pegainfer-deepseek-v4/src/runtime/collectives.rs:8-121   -> all_reduce_{bf16,f32}, all_gather, reduce_scatter_f32
 *   pegainfer-deepseek-v4/src/runtime/collectives.rs:123-184 -> all_reduce_bf16_via_f32 (cast, f32 sum, cast back)
 *   pegainfer-deepseek-v4/src/runtime/collectives.rs:186-287 -> all_reduce_bf16_to_f32 (cast + f32 sum, the f32 view
 *                                                               is handed to the caller's fused post kernel)
 *   pegainfer-deepseek-v4/src/runtime/moe.rs:1327-1461        -> comm stream + fence_in / fence_out (all-gather and
 *                                                               reduce-scatter overlapped with the shared expert)
 *   pegainfer-deepseek-v4/src/runtime/core.rs:560-609         -> all_gather (vocab-sharded logits)
 *   pegainfer-comm/src/ep_backend.rs:213-331                  -> pegainfer_ep_* : dispatch_send / dispatch_recv /
 *                                                               combine_send / combine_recv, tokens_per_expert
 *
 * Every verb is enqueued on the caller's stream (graph-capturable: no host synchronisation) unless stated.  Counts
 * are ELEMENTS.  All functions return 0 on success, a negative value on error (pegainfer_comm_last_error).
 *
 * A communicator of world size 1 needs no unique id (NULL): the verbs degenerate to copies / casts, which is what a
 * single-GPU box can test.  The expert-parallel endpoints additionally accept a LOOPBACK hub: `world` virtual ranks
 * inside one process on one GPU, exchanging through device memcpy instead of RCCL - the routing / packing / combine
 * kernels are identical, only the transport differs, so multi-rank semantics are testable on one GPU.
 */
#ifndef PEGAINFER_COMM_H
#define PEGAINFER_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pegainfer_comm_t;
typedef void* pegainfer_ep_hub_t;
typedef void* pegainfer_ep_t;
typedef void* pegainfer_stream_t;
typedef uint16_t Half;

/* ---- communicator (cudarc Comm::from_rank: executor.rs:580-588 / state.rs:129) ---- */
int32_t pegainfer_comm_unique_id(void* out_128_bytes);
/* unique_id_128: NULL only when world == 1.  Creates the RCCL communicator for (rank, world) on device_ordinal plus
 * a non-blocking comm stream and two fence events. */
pegainfer_comm_t pegainfer_comm_create(int32_t device_ordinal, int32_t rank, int32_t world, const void* unique_id_128);
void pegainfer_comm_destroy(pegainfer_comm_t c);
const char* pegainfer_comm_last_error(pegainfer_comm_t c);

/* ---- one-shot small-message all-reduce over peer access (xGMI) ----
 * The decode collectives are latency-bound: DSV4 issues ~107 f32 all-reduces of 16 KB per token
 * (docs/models/deepseek-v4/moe-tilelang-review.md:12), Qwen3 TP 72 bf16 all-reduces of 5 KB (weights.rs:396-405).
 * Once enabled, pegainfer_comm_all_reduce_{bf16,f32} (and the cast-fused forms built on them) take payloads of at
 * most 64 KB (decided by the byte count alone, so every rank takes the same route; an address or size that is not a
 * multiple of 16 is staged through a 64 KB buffer of the handle) through ONE kernel launch: every rank pushes its payload
 * into every rank's slab over xGMI (write-through stores), signals with an epoch flag, waits for the world's flags and
 * sums the copies in rank order in f32 - bit-identical results on all ranks, graph-capturable (epochs live in device
 * memory), every wait bounded (PEGAINFER_ONESHOT_TIMEOUT_MS, default 10 000; expiry -> pegainfer_comm_oneshot_status).
 * Larger payloads go to RCCL as before; a peer-only communicator (no RCCL behind it) cuts them into 64 KB pieces, one
 * launch each.  At most 8 ranks (one node).  All one-shot all-reduces of a communicator must be ordered with respect to
 * each other (one stream, or streams joined by events): the epochs advance in launch order.
 * pegainfer_comm_oneshot_enable is COLLECTIVE and never strands a peer: a rank whose export / mapping failed still takes
 * part in both exchanges, and the path is switched on only if every rank succeeded (otherwise off on every rank).
 *   RCCL communicators:   pegainfer_comm_oneshot_enable(c)            (handles travel over RCCL itself)
 *   peer-only (no RCCL):  create_peer_only -> oneshot_handle -> exchange the 64-byte handles out of band, rank-major ->
 *                         oneshot_attach -> caller barrier.  Only the all-reduces work on such a communicator;
 *                         this is also how two processes sharing ONE GPU exercise the protocol on a single-GPU box. */
pegainfer_comm_t pegainfer_comm_create_peer_only(int32_t device_ordinal, int32_t rank, int32_t world);
int32_t pegainfer_comm_oneshot_handle(pegainfer_comm_t c, void* out_64_bytes);
int32_t pegainfer_comm_oneshot_attach(pegainfer_comm_t c, const void* handles_world_x_64);
int32_t pegainfer_comm_oneshot_enable(pegainfer_comm_t c);
int32_t pegainfer_comm_oneshot_active(pegainfer_comm_t c);
int32_t pegainfer_comm_oneshot_status(pegainfer_comm_t c);
/* device address of the status block behind oneshot_status ({0x100 | mask of missing ranks, epoch, segment}, uint32 x 3;
 * NULL before the slab exists): a runtime whose captured step contains one-shot all-reduces copies it back with the
 * step's results instead of paying a synchronising call (csrc/host/qwen3_runtime.cpp does) */
const uint32_t* pegainfer_comm_oneshot_status_ptr(pegainfer_comm_t c);
int32_t pegainfer_comm_rank(pegainfer_comm_t c);
int32_t pegainfer_comm_world(pegainfer_comm_t c);

/* ---- MP8 verbs ---- */
int32_t pegainfer_comm_all_reduce_bf16(pegainfer_comm_t c, Half* data, int64_t n, pegainfer_stream_t stream);
int32_t pegainfer_comm_all_reduce_f32(pegainfer_comm_t c, float* data, int64_t n, pegainfer_stream_t stream);
/* bf16 in/out, summed in f32: one cast kernel, the f32 all-reduce, one cast kernel; scratch owned by the handle */
int32_t pegainfer_comm_all_reduce_bf16_via_f32(pegainfer_comm_t c, Half* data, int64_t n, pegainfer_stream_t stream);
/* bf16 in, f32 sum out (out_f32 holds n floats): the consumer fuses its own epilogue on the f32 view */
int32_t pegainfer_comm_all_reduce_bf16_to_f32(pegainfer_comm_t c, const Half* in, float* out_f32, int64_t n, pegainfer_stream_t stream);
/* local: n_local elements of elem_bytes (2 = bf16, 4 = u32 / f32) -> gathered: world * n_local, rank-major */
int32_t pegainfer_comm_all_gather(pegainfer_comm_t c, const void* local, void* gathered, int64_t n_local, int32_t elem_bytes, pegainfer_stream_t stream);
/* global: world * n_local f32 -> local: this rank's summed n_local */
int32_t pegainfer_comm_reduce_scatter_f32(pegainfer_comm_t c, const float* global, float* local, int64_t n_local, pegainfer_stream_t stream);
/* equal slabs: slab r (n_per_rank elements) of send goes to rank r; recv slab r came from rank r */
int32_t pegainfer_comm_all_to_all(pegainfer_comm_t c, const void* send, void* recv, int64_t n_per_rank, int32_t elem_bytes, pegainfer_stream_t stream);
/* ragged: host arrays of `world` element counts / element offsets */
int32_t pegainfer_comm_all_to_allv(pegainfer_comm_t c, const void* send, const int64_t* send_counts, const int64_t* send_offsets, void* recv, const int64_t* recv_counts, const int64_t* recv_offsets, int32_t elem_bytes, pegainfer_stream_t stream);

/* ---- comm stream with event fences (moe.rs:1327-1461): work enqueued on pegainfer_comm_stream() between
 *      fence_in(compute) and fence_out(compute) overlaps with what the compute stream does in between ---- */
pegainfer_stream_t pegainfer_comm_stream(pegainfer_comm_t c);
int32_t pegainfer_comm_fence_in(pegainfer_comm_t c, pegainfer_stream_t compute_stream);   /* comm stream waits for compute */
int32_t pegainfer_comm_fence_out(pegainfer_comm_t c, pegainfer_stream_t compute_stream);  /* compute waits for comm stream */

/* ---- expert-parallel dispatch / combine: the EpBackend surface (pegainfer-comm/src/ep_backend.rs), argument for
 *      argument, so `impl EpBackend` binds each method to the function of the same name ----
 * Experts are dealt to ranks in contiguous blocks (expert e lives on rank e / (num_experts / world_size),
 * a2a_dispatch_send.cu:168-170).  Strides follow the reference kernels: BYTES for the row buffers x / out_x / expert_x
 * (`(uint4*)(x_ptr + token * x_stride)`, moe_pplx.rs:131-133), ELEMENTS for indices, weights, the f32 scale planes and
 * out_tokens.  bound_m_ptr, when not NULL, is a DEVICE i32: the number of leading tokens that take part
 * (`bound_m_ptr ? *bound_m_ptr : num_tokens`), read by the kernels, never by the host.
 *   dispatch_send : every (token, k) pair with token < bound travels to the rank that owns expert indices[t][k]; the
 *                   payload row (hidden_dim * in_elemsize opaque bytes) and, when x_scale_ptr != NULL, its
 *                   hidden_dim_scale f32 scales ride in one wire row
 *   dispatch_recv : out_x rows grouped by LOCAL expert, each expert's group starting at a multiple of expert_padding
 *                   (a2a_worker.rs:598-606, 669: padded_offset[expert] + running count; inside a group by source rank,
 *                   then source order - the reference leaves that order to its atomics); out_num_tokens_ptr = DEVICE
 *                   i32[num_experts / world_size] rows per local expert (a2a_dispatch_recv.cu:220-224); rows in the
 *                   padding gaps are not written
 *   combine_send  : expert_x rows in the SAME padded layout travel back to the ranks they came from
 *                   (hidden_dim * out_elemsize bytes each)
 *   combine_recv  : out_tokens[t] (+)= sum_k weights[t][k] * y(t, k) for t < bound, f32 fma chain in k order starting
 *                   from the destination (accumulate) or zero, one rounding to out_dtype (core/combine_utils.cuh);
 *                   in_dtype = what the experts produced (bf16 | f32, element size == out_elemsize)
 * Not carried from EpTopology: dp_size > 1 (rejected), node_size and max_private_tokens (RDMA staging knobs: accepted,
 * unused - one xGMI node, RCCL or loopback transport).  A pair whose index is not in [0, num_experts) is routed nowhere
 * and contributes zero.  dispatch_send synchronises the calling stream once with the host (the route counts size the
 * exchange), exactly where the reference's worker thread waits for them; with RCCL the whole (rank, expert) count table
 * is all-gathered so an overflow of ANY rank's max_recv_tokens fails the call on EVERY rank before a row moves. */
enum { PEGAINFER_SCALAR_BF16 = 0, PEGAINFER_SCALAR_F16 = 1, PEGAINFER_SCALAR_F32 = 2 };   /* p2p_all_to_all::ScalarType subset */
typedef struct {   /* EpTopology, ep_backend.rs:24-51, field for field */
  size_t world_size, rank, node_size, dp_size, num_experts, num_experts_per_token, hidden_dim, hidden_dim_scale,
         max_num_tokens, max_recv_tokens, max_private_tokens, expert_padding;
} pegainfer_ep_topology_t;
typedef struct {   /* EpDtypes, ep_backend.rs:53-66 */
  size_t in_elemsize, out_elemsize;
  int32_t out_dtype;
  size_t scale_elemsize;
} pegainfer_ep_dtypes_t;
pegainfer_ep_hub_t pegainfer_ep_hub_create(int32_t world);   /* loopback transport: `world` virtual ranks, one process */
void pegainfer_ep_hub_destroy(pegainfer_ep_hub_t hub);
/* EpBackend::new: exactly one of (comm, hub) is non-NULL; with a hub, topology->rank is the virtual rank */
pegainfer_ep_t pegainfer_ep_create(pegainfer_comm_t comm, pegainfer_ep_hub_t hub, const pegainfer_ep_topology_t* topology, const pegainfer_ep_dtypes_t* dtypes);
void pegainfer_ep_destroy(pegainfer_ep_t ep);
const char* pegainfer_ep_last_error(pegainfer_ep_t ep);
/* ep_backend.rs:213-246 */
int32_t pegainfer_ep_dispatch_send(pegainfer_ep_t ep, size_t num_tokens, const void* x_ptr, size_t x_stride, const void* x_scale_ptr, size_t x_scale_stride_elem, size_t x_scale_stride_token, const int32_t* indices, size_t indices_stride, const float* weights, size_t weights_stride, const int32_t* bound_m_ptr, pegainfer_stream_t stream);
/* ep_backend.rs:252-274 */
int32_t pegainfer_ep_dispatch_recv(pegainfer_ep_t ep, int32_t* out_num_tokens_ptr, void* out_x_ptr, size_t out_x_stride, void* out_x_scale_ptr, size_t out_x_scale_stride_elem, size_t out_x_scale_stride_token, pegainfer_stream_t stream);
/* ep_backend.rs:277-286 */
int32_t pegainfer_ep_combine_send(pegainfer_ep_t ep, const void* expert_x_ptr, size_t expert_x_stride, pegainfer_stream_t stream);
/* ep_backend.rs:303-331 */
int32_t pegainfer_ep_combine_recv(pegainfer_ep_t ep, size_t num_tokens, size_t num_recv_tokens, int32_t in_dtype, void* out_tokens_ptr, size_t out_tokens_stride, const int32_t* indices_ptr, size_t indices_stride, const float* weights_ptr, size_t weights_stride, const int32_t* bound_m_ptr, int32_t accumulate, pegainfer_stream_t stream);
/* ep_backend.rs:292-294: device u32[num_experts / world_size], populated by dispatch_recv */
const uint32_t* pegainfer_ep_tokens_per_expert_ptr(pegainfer_ep_t ep);
/* host mirrors of the last dispatch (the reference's caller does a D2H copy of the counter) */
int32_t pegainfer_ep_tokens_per_expert_host(pegainfer_ep_t ep, uint32_t* out, int32_t n);
int32_t pegainfer_ep_num_recv_tokens(pegainfer_ep_t ep);          /* rows received (unpadded) */
int32_t pegainfer_ep_num_padded_recv_tokens(pegainfer_ep_t ep);   /* extent of the padded expert-major layout in out_x */

#ifdef __cplusplus
}
#endif
#endif
