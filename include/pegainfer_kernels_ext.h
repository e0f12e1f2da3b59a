/*
 * pegainfer_kernels_ext.h - entry points of libpegainfer_kernels_hip.so that have NO counterpart in
 * the reference's ffi.rs.  They exist for the MI355X host DAG (libpegainfer_qwen3.so): at ~1 ms per
 * decode step the reference's 14 launches per layer and per-request sampling sync are first-order
 * costs, so the host may call these fused / batched forms instead.  Each one is
 * REQUIRED to produce bit-identical results to the sequence of reference-named calls it replaces;
 * tests/test_gpu_fused.py checks exactly that.
 */
#ifndef PEGAINFER_KERNELS_EXT_H
#define PEGAINFER_KERNELS_EXT_H

#include "pegainfer_kernels.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Greedy token for every row of logits[rows, vocab] in one launch (replaces, per request,
 * extract_vec + flashinfer_top1_cuda + sync + D2H: executor.rs:324-328, ops/sampling.rs:122-168).
 * state_scratch: rows*16 bytes, zeroed once by the caller, left zeroed.  Ties -> lowest index. */
pegainfer_status_t pegainfer_batched_top1(const Half* logits, int32_t vocab_size, int32_t rows, int64_t row_stride, uint8_t* state_scratch, int32_t* out_tokens, pegainfer_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
