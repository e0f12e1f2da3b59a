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

/* ---- expert-parallel dispatch / combine (ep_backend.rs:213-331) ----
 * Topology as EpTopology: experts are dealt to ranks in contiguous blocks (expert e lives on rank
 * e / (num_experts / world)).  bf16 payload rows of hidden_dim elements.
 *   dispatch_send : x [num_tokens, hidden] (x_stride_elems between rows), indices / weights [num_tokens, topk] ->
 *                   every (token, k) pair is sent to the rank that owns expert indices[t][k]
 *   dispatch_recv : out_x [<= max_recv_tokens, hidden] = the received rows grouped by LOCAL expert (expert-major,
 *                   within an expert by source rank then source order); out_num_tokens[0] = row count;
 *                   tokens_per_expert()[e] = rows of local expert e (device u32, num_experts / world entries)
 *   combine_send  : expert_x rows in the SAME order as out_x travel back to the ranks they came from
 *   combine_recv  : out_tokens[t] (+)= sum_k weights[t][k] * y(t, k), f32 accumulation in k order, bf16 result
 * A dispatch_send synchronises the calling stream once with the host (the per-peer row counts size the exchange),
 * exactly where the reference's worker thread waits for the route counts.
 * Row buffers (x, out_x, expert_x, out_tokens) are moved as 16-byte vectors: base pointers 16-byte aligned, strides a
 * multiple of 8 elements (else -1 + last_error).  A pair whose index is not in [0, num_experts) is routed nowhere and
 * contributes zero to its token's combine. */
pegainfer_ep_hub_t pegainfer_ep_hub_create(int32_t world);   /* loopback transport: `world` virtual ranks, one process */
void pegainfer_ep_hub_destroy(pegainfer_ep_hub_t hub);
/* exactly one of (comm, hub) is non-NULL; with a hub, `rank` is the virtual rank of this endpoint */
pegainfer_ep_t pegainfer_ep_create(pegainfer_comm_t comm, pegainfer_ep_hub_t hub, int32_t rank, int32_t hidden_dim, int32_t max_num_tokens, int32_t max_recv_tokens, int32_t num_experts, int32_t num_experts_per_token);
void pegainfer_ep_destroy(pegainfer_ep_t ep);
const char* pegainfer_ep_last_error(pegainfer_ep_t ep);
int32_t pegainfer_ep_dispatch_send(pegainfer_ep_t ep, int32_t num_tokens, const Half* x, int64_t x_stride_elems, const int32_t* indices, const float* weights, pegainfer_stream_t stream);
int32_t pegainfer_ep_dispatch_recv(pegainfer_ep_t ep, int32_t* out_num_tokens, Half* out_x, int64_t out_x_stride_elems, pegainfer_stream_t stream);
int32_t pegainfer_ep_combine_send(pegainfer_ep_t ep, const Half* expert_x, int64_t expert_x_stride_elems, pegainfer_stream_t stream);
int32_t pegainfer_ep_combine_recv(pegainfer_ep_t ep, int32_t num_tokens, Half* out_tokens, int64_t out_stride_elems, const int32_t* indices, const float* weights, int32_t accumulate, pegainfer_stream_t stream);
const uint32_t* pegainfer_ep_tokens_per_expert(pegainfer_ep_t ep);
/* host mirror of the same counters after dispatch_recv (n = num_experts / world entries) */
int32_t pegainfer_ep_tokens_per_expert_host(pegainfer_ep_t ep, uint32_t* out, int32_t n);
int32_t pegainfer_ep_num_recv_tokens(pegainfer_ep_t ep);   /* host copy of the last dispatch's received row count */

#ifdef __cplusplus
}
#endif
#endif
