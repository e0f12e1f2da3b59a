// Paged KV append for gfx950 (replaces FlashInfer AppendPagedKVCache behind
// paged_kv_scatter_cuda, reference csrc/paged_attention.cu:274-311).
//
// Cache layout (pegainfer-core/src/kv_pool.rs:66-75): [page][layer][K,V][slot][kv_head][head_dim];
// the caller passes the layer's K/V element offsets and the page stride.  Token i of the source
// goes to page = page_indices[page_indptr[batch_indices[i]] + positions[i] / page_size],
// slot = positions[i] % page_size.  Pure byte movement: one 16-byte chunk per lane, K and V in
// the same launch.
#include "common.h"

namespace pk {

template <bool VEC>
__global__ __launch_bounds__(256) void paged_kv_scatter_kernel(
    Half* __restrict__ kv, long k_off, long v_off, const int* __restrict__ page_indices,
    const int* __restrict__ page_indptr, const Half* __restrict__ src_k, const Half* __restrict__ src_v,
    const int* __restrict__ batch_indices, const int* __restrict__ positions, int nnz, int num_kv_heads,
    int head_dim, int page_size, long stride_page, long src_stride_n, long src_stride_h) {
  const int per_head = VEC ? (head_dim >> 3) : head_dim;  // work items per (token, head, K|V)
  const long total = (long)nnz * num_kv_heads * per_head * 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    long r = idx;
    const int c = (int)(r % per_head); r /= per_head;
    const int h = (int)(r % num_kv_heads); r /= num_kv_heads;
    const int is_v = (int)(r & 1); r >>= 1;
    const int i = (int)r;
    const int b = batch_indices[i];
    const int pos = positions[i];
    const int page = page_indices[page_indptr[b] + pos / page_size];
    const int slot = pos % page_size;
    const long dst = (long)page * stride_page + (is_v ? v_off : k_off) +
                     ((long)slot * num_kv_heads + h) * head_dim;
    const Half* src = (is_v ? src_v : src_k) + (long)i * src_stride_n + (long)h * src_stride_h;
    if (VEC) reinterpret_cast<u32x4*>(kv + dst)[c] = reinterpret_cast<const u32x4*>(src)[c];
    else kv[dst + c] = src[c];
  }
}

}  // namespace pk

extern "C" int32_t paged_kv_scatter_cuda(const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
                                         const int32_t* page_indices, const int32_t* page_indptr,
                                         const int32_t* last_page_len_d, const Half* src_k, const Half* src_v,
                                         const int32_t* batch_indices, const int32_t* positions, int32_t nnz,
                                         int32_t num_kv_heads, int32_t head_dim, int32_t page_size,
                                         int64_t stride_page, int64_t src_stride_n, int64_t src_stride_h,
                                         pegainfer_stream_t stream) {
  (void)last_page_len_d;
  if (nnz <= 0) return 0;
  using namespace pk;
  Half* kv = const_cast<Half*>(kv_data);  // the reference's signature is *const; the cache is written
  const bool vec = (head_dim & 7) == 0 && (src_stride_n & 7) == 0 && (src_stride_h & 7) == 0 &&
                   (stride_page & 7) == 0 && (k_offset_elems & 7) == 0 && (v_offset_elems & 7) == 0 &&
                   host_aligned16(kv) && host_aligned16(src_k) && host_aligned16(src_v);
  const long total = (long)nnz * num_kv_heads * (vec ? head_dim >> 3 : head_dim) * 2;
  int grid = ceil_div(total, 256);
  if (grid > 2048) grid = 2048;
  if (vec)
    paged_kv_scatter_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(
        kv, k_offset_elems, v_offset_elems, page_indices, page_indptr, src_k, src_v, batch_indices, positions,
        nnz, num_kv_heads, head_dim, page_size, stride_page, src_stride_n, src_stride_h);
  else
    paged_kv_scatter_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(
        kv, k_offset_elems, v_offset_elems, page_indices, page_indptr, src_k, src_v, batch_indices, positions,
        nnz, num_kv_heads, head_dim, page_size, stride_page, src_stride_n, src_stride_h);
  return static_cast<int32_t>(hipGetLastError());
}
