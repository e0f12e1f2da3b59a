"""Safe op wrappers over the C ABI - the Python mirror of ``pegainfer_kernels::ops``
(reference pegainfer-kernels/src/ops/{norm,elementwise,embedding,linear,attention,sampling}.rs):
same names, same argument meaning, same shape assertions, same error behaviour
(``void`` ops surface errors at the next sync; status-returning ops raise).

Tensors are torch CUDA tensors used purely as device memory (bf16 = ``torch.bfloat16``);
``HiddenStates [d, T]`` is a contiguous ``[T, d]`` tensor, ``DeviceMatrix [rows, cols]`` a
contiguous ``[rows, cols]`` tensor.  All work is enqueued on torch's current stream.
No CPU fallback exists: without the HIP library every call raises ImportError.
"""
import math

import torch

from . import ffi


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with error {code}")


def _bf16(*ts):
    for t in ts:
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.is_cuda, "expected contiguous CUDA bf16"


class PagedKvLayout:
    """pegainfer-kernels/src/paged_kv.rs:5-34."""

    def __init__(self, num_layers, num_kv_heads, head_dim, page_size):
        self.page_size, self.num_layers = page_size, num_layers
        self.num_kv_heads, self.head_dim = num_kv_heads, head_dim
        self.kv_block_len = page_size * num_kv_heads * head_dim
        self.layer_stride = 2 * self.kv_block_len
        self.page_stride = num_layers * self.layer_stride


# ------------------------------------------------------------------ norms (ops/norm.rs)
def rms_norm_into(x, weight, eps, out):
    assert x.numel() == out.numel()
    _bf16(x, weight, out)
    ffi.lib().rms_norm_cuda(_p(x), _p(weight), _p(out), x.numel(), eps, _stream())


def rms_norm_batch_into(x, weight, eps, out):
    assert weight.numel() == x.shape[1] and out.shape == x.shape
    _bf16(x, weight, out)
    ffi.lib().rms_norm_batched_cuda(_p(x), _p(weight), _p(out), x.shape[1], x.shape[0], eps, _stream())


def rms_norm_offset_into(x, weight, eps, out):
    assert x.numel() == out.numel()
    _bf16(x, weight, out)
    ffi.lib().rms_norm_offset_cuda(_p(x), _p(weight), _p(out), x.numel(), eps, _stream())


def rms_norm_batch_offset_into(x, weight, eps, out):
    assert weight.numel() == x.shape[1] and out.shape == x.shape
    _bf16(x, weight, out)
    ffi.lib().rms_norm_batched_offset_cuda(_p(x), _p(weight), _p(out), x.shape[1], x.shape[0], eps, _stream())


def fused_add_rms_norm_into(hidden, residual, weight, eps, out):
    assert hidden.numel() == residual.numel() == out.numel()
    _bf16(hidden, residual, weight, out)
    ffi.lib().fused_add_rms_norm_cuda(_p(hidden), _p(residual), _p(weight), _p(out), hidden.numel(), eps, _stream())


def fused_add_rms_norm_batch_into(hidden, residual, weight, eps, out):
    assert hidden.shape == residual.shape == out.shape and weight.numel() == hidden.shape[1]
    _bf16(hidden, residual, weight, out)
    ffi.lib().fused_add_rms_norm_batched_cuda(_p(hidden), _p(residual), _p(weight), _p(out), hidden.shape[1],
                                              hidden.shape[0], eps, _stream())


def rms_norm_gated_batch_into(x, weight_f32, gate, out, num_heads, head_dim, eps):
    assert x.shape[1] == num_heads * head_dim and gate.shape == x.shape and out.shape == x.shape
    assert weight_f32.dtype == torch.float32
    ffi.lib().rms_norm_gated_cuda(_p(x), _p(weight_f32), _p(gate), _p(out), x.shape[0] * num_heads, head_dim, eps,
                                  _stream())


# ------------------------------------------------------------------ elementwise (ops/elementwise.rs)
def add_batch_into(a, b, out):
    assert a.shape == b.shape == out.shape
    _bf16(a, b, out)
    _chk(ffi.lib().add_cuda(_p(a), _p(b), _p(out), a.numel(), _stream()), "add_cuda")


def silu_mul_batch_into(gate, up, out):
    assert gate.shape == up.shape == out.shape
    _bf16(gate, up, out)
    _chk(ffi.lib().silu_mul_triton_aot_cuda(_p(gate), _p(up), _p(out), gate.numel(), _stream()), "silu_mul")


def silu_mul_fused_batch_into(gate_up, out):
    assert gate_up.shape[1] == 2 * out.shape[1] and gate_up.shape[0] == out.shape[0]
    _bf16(gate_up, out)
    ffi.lib().silu_mul_fused_cuda(_p(gate_up), _p(out), out.shape[1], out.shape[0], _stream())


def extract_vec(hidden_states, idx):
    """D2D column copy (ops/elementwise.rs:120-163)."""
    return hidden_states[idx].clone()


# ------------------------------------------------------------------ embedding (ops/embedding.rs)
def embedding_batch(embed, token_ids_u32, out):
    assert out.shape == (token_ids_u32.numel(), embed.shape[1])
    _bf16(embed, out)
    _chk(ffi.lib().embedding_batched_cuda(_p(embed), _p(token_ids_u32), _p(out), embed.shape[1],
                                          token_ids_u32.numel(), _stream()), "embedding_batched_cuda")


def embedding_decode_into(embed, token_id_u32, out):
    _chk(ffi.lib().embedding_decode_cuda(_p(embed), _p(token_id_u32), _p(out), embed.shape[1], _stream()),
         "embedding_decode_cuda")


def embedding_batch_vocab_shard(embed_shard, token_ids_u32, out, vocab_start):
    _chk(ffi.lib().embedding_batched_vocab_shard_cuda(_p(embed_shard), _p(token_ids_u32), _p(out),
                                                      embed_shard.shape[1], token_ids_u32.numel(), vocab_start,
                                                      embed_shard.shape[0], _stream()),
         "embedding_batched_vocab_shard_cuda")


# ------------------------------------------------------------------ linear (ops/linear.rs)
def gemm_rows_into(weight, row_offset, num_rows, x, out):
    assert row_offset + num_rows <= weight.shape[0] and weight.shape[1] == x.shape[1]
    assert out.shape == (x.shape[0], num_rows)
    gemm_into(weight[row_offset:row_offset + num_rows], x, out)


def gemm_into(weight, x, out):
    assert weight.shape[1] == x.shape[1], f"weight cols {weight.shape[1]} != hidden_dim {x.shape[1]}"
    assert out.shape == (x.shape[0], weight.shape[0])
    _bf16(weight, x, out)
    fn = ffi.lib().gemm_graphsafe_cuda if x.shape[0] == 1 else ffi.lib().gemm_cuda  # ops/linear.rs:27,120
    fn(_p(weight), _p(x), _p(out), weight.shape[0], x.shape[0], weight.shape[1], _stream())


def gemm_split3_into(weight, x, out0, out1, out2):
    """One GEMM over the row-stacked weight, three outputs (pegainfer_kernels_ext.h: pegainfer_gemm_split3)."""
    m0, m1, m2 = out0.shape[1], out1.shape[1], out2.shape[1]
    assert weight.shape == (m0 + m1 + m2, x.shape[1])
    assert out0.shape[0] == out1.shape[0] == out2.shape[0] == x.shape[0]
    _bf16(weight, x, out0, out1, out2)
    _chk(ffi.lib().pegainfer_gemm_split3(_p(weight), _p(x), _p(out0), m0, _p(out1), m1, _p(out2), m2, x.shape[0],
                                         weight.shape[1], _stream()), "pegainfer_gemm_split3")


def gemm_silu_into(gate_up_weight, x, out, scratch=None):
    """Y[T, I] = silu_mul_fused(W[2I, K] . X) in one launch (pegainfer_kernels_ext.h: pegainfer_gemm_silu)."""
    inter = out.shape[1]
    assert gate_up_weight.shape == (2 * inter, x.shape[1]) and out.shape[0] == x.shape[0]
    _bf16(gate_up_weight, x, out)
    if scratch is None:
        scratch = torch.empty((x.shape[0], 2 * inter), dtype=torch.bfloat16, device=x.device)
    _chk(ffi.lib().pegainfer_gemm_silu(_p(gate_up_weight), _p(x), _p(out), _p(scratch), inter, x.shape[0],
                                       gate_up_weight.shape[1], _stream()), "pegainfer_gemm_silu")


def gemm_add_rms_norm_into(weight, x, hidden, norm_weight, eps, normed_out, scratch=None):
    """hidden += W . x; normed_out = rms_norm(hidden) * norm_weight - gemm_cuda + fused_add_rms_norm_batched_cuda in
    one call (pegainfer_kernels_ext.h: pegainfer_gemm_add_rms_norm)."""
    T, M = x.shape[0], weight.shape[0]
    assert weight.shape[1] == x.shape[1] and hidden.shape == normed_out.shape == (T, M) and norm_weight.numel() == M
    _bf16(weight, x, hidden, norm_weight, normed_out)
    if scratch is None:
        scratch = torch.empty((T, M), dtype=torch.bfloat16, device=x.device)
    _chk(ffi.lib().pegainfer_gemm_add_rms_norm(_p(weight), _p(x), _p(scratch), _p(hidden), _p(norm_weight),
                                               _p(normed_out), M, T, weight.shape[1], eps, _stream()),
         "pegainfer_gemm_add_rms_norm")


def gemm_add_into(weight, x, a, out, scratch=None):
    """out = a + W . x - gemm_cuda + add_cuda in one call (pegainfer_kernels_ext.h: pegainfer_gemm_add)."""
    T, M = x.shape[0], weight.shape[0]
    assert weight.shape[1] == x.shape[1] and a.shape == out.shape == (T, M)
    _bf16(weight, x, a, out)
    if scratch is None:
        scratch = torch.empty((T, M), dtype=torch.bfloat16, device=x.device)
    _chk(ffi.lib().pegainfer_gemm_add(_p(weight), _p(x), _p(scratch), _p(a), _p(out), M, T, weight.shape[1], _stream()),
         "pegainfer_gemm_add")


def gemm(weight, x):
    out = torch.empty((x.shape[0], weight.shape[0]), dtype=torch.bfloat16, device=x.device)
    gemm_into(weight, x, out)
    return out


def linear(x_vec, weight):
    return gemm(weight, x_vec.view(1, -1)).view(-1)


# ------------------------------------------------------------------ attention (ops/attention.rs)
def qk_norm_rope_batch_decode_into(q, k, q_norm, k_norm, cos, sin, positions_i32, num_q_heads, num_kv_heads,
                                   head_dim, eps):
    assert q.shape[0] == k.shape[0] == positions_i32.numel()
    _bf16(q, k, q_norm, k_norm, cos, sin)
    ffi.lib().qk_norm_rope_batched_decode_cuda(_p(q), _p(k), _p(q_norm), _p(k_norm), _p(cos), _p(sin),
                                               _p(positions_i32), num_q_heads, num_kv_heads, head_dim,
                                               q.shape[0], eps, _stream())


def prefill_qk_norm_rope_only(q, k, q_norm, k_norm, cos, sin, num_q_heads, num_kv_heads, head_dim, start_pos, eps):
    _bf16(q, k, q_norm, k_norm, cos, sin)
    ffi.lib().prefill_qk_norm_rope_only_cuda(_p(q), _p(k), _p(q_norm), _p(k_norm), _p(cos), _p(sin), num_q_heads,
                                             num_kv_heads, head_dim, q.shape[0], start_pos, eps, _stream())


def paged_kv_scatter(kv_buffer, layout, layer, page_indices, page_indptr, last_page_len, k, v, batch_indices,
                     positions):
    nkv, hd = layout.num_kv_heads, layout.head_dim
    _chk(ffi.lib().paged_kv_scatter_cuda(
        _p(kv_buffer), layer * layout.layer_stride, layer * layout.layer_stride + layout.kv_block_len,
        _p(page_indices), _p(page_indptr), _p(last_page_len), _p(k), _p(v), _p(batch_indices), _p(positions),
        positions.numel(), nkv, hd, layout.page_size, layout.page_stride, nkv * hd, hd, _stream()),
        "paged_kv_scatter_cuda")


def paged_attention_batch_decode_into(q, k, v, kv_buffer, layout, layer, page_indices, page_indptr, last_page_len,
                                      positions, request_indices, kv_tile_indices, kv_chunk_size, output,
                                      num_qo_heads, batch_size):
    """ops/attention.rs:572-674: scatter K/V then non-partition decode."""
    paged_kv_scatter(kv_buffer, layout, layer, page_indices, page_indptr, last_page_len, k, v, request_indices,
                     positions)
    _chk(ffi.lib().paged_attention_decode_cuda(
        _p(q), _p(output), _p(kv_buffer), layer * layout.layer_stride,
        layer * layout.layer_stride + layout.kv_block_len, _p(page_indices), _p(page_indptr), _p(last_page_len),
        _p(request_indices), _p(kv_tile_indices), _p(kv_chunk_size), num_qo_heads, layout.num_kv_heads,
        layout.head_dim, layout.page_size, batch_size, layout.page_stride,
        1.0 / math.sqrt(layout.head_dim), _stream()), "paged_attention_decode_cuda")


def paged_attention_batch_decode_split_kv_into(q, k, v, kv_buffer, layout, layer, page_indices, page_indptr,
                                               last_page_len, positions, request_indices, split_request_indices,
                                               split_kv_tile_indices, split_kv_chunk_size, split_o_indptr,
                                               split_block_valid_mask, split_tmp_v, split_tmp_s, split_padded_slots,
                                               output, num_qo_heads, batch_size):
    """ops/attention.rs:681-796."""
    paged_kv_scatter(kv_buffer, layout, layer, page_indices, page_indptr, last_page_len, k, v, request_indices,
                     positions)
    _chk(ffi.lib().paged_attention_decode_split_kv_cuda(
        _p(q), _p(output), _p(kv_buffer), layer * layout.layer_stride,
        layer * layout.layer_stride + layout.kv_block_len, _p(page_indices), _p(page_indptr), _p(last_page_len),
        _p(split_request_indices), _p(split_kv_tile_indices), _p(split_kv_chunk_size), _p(split_o_indptr),
        _p(split_block_valid_mask), _p(split_tmp_v), _p(split_tmp_s), num_qo_heads, layout.num_kv_heads,
        layout.head_dim, layout.page_size, batch_size, split_padded_slots, layout.page_stride,
        1.0 / math.sqrt(layout.head_dim), _stream()), "paged_attention_decode_split_kv_cuda")


class PrefillPagedPlan:
    """ops/attention.rs:17-303 - host plan + 11 device arrays; same accessor surface."""

    def __init__(self, page_indices, last_page_lens, start_positions, seq_lens, num_q_heads, num_kv_heads,
                 head_dim, cta_tile_q_override=0, device="cuda"):
        L = ffi.lib()
        group = num_q_heads // num_kv_heads
        total = int(sum(seq_lens))
        cta = L.batch_prefill_cta_tile_q_with_override(total, num_q_heads, num_kv_heads, head_dim,
                                                       cta_tile_q_override)
        if cta <= 0:
            raise ValueError(f"invalid prefill CTA tile override {cta_tile_q_override}")
        pages, indptr, kv_chunk, bidx, pos, q_indptr = [], [0], [], [], [], [0]
        req, qo_tile, kv_tile = [], [], []
        for i, pg in enumerate(page_indices):
            pages.extend(int(x) for x in pg)
            indptr.append(len(pages))
            kv_chunk.append(int(start_positions[i] + seq_lens[i]))
            bidx.extend([i] * seq_lens[i])
            pos.extend(range(start_positions[i], start_positions[i] + seq_lens[i]))
            q_indptr.append(q_indptr[-1] + seq_lens[i])
            for t in range(-(-seq_lens[i] * group // cta)):
                req.append(i)
                qo_tile.append(t)
                kv_tile.append(0)
        i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=device)
        self.page_indices_d, self.page_indptr_d = i32(pages), i32(indptr)
        self.last_page_len_d = i32([int(x) for x in last_page_lens])
        self.batch_indices_d, self.positions_d, self.q_indptr_d = i32(bidx), i32(pos), i32(q_indptr)
        self.request_indices_d, self.qo_tile_indices_d, self.kv_tile_indices_d = i32(req), i32(qo_tile), i32(kv_tile)
        self.kv_chunk_size_d = i32(kv_chunk)
        self.total_num_rows_d = torch.tensor([total], dtype=torch.int32, device=device)
        self.num_tiles, self.batch_size, self.total_tokens, self.cta_tile_q = len(req), len(seq_lens), total, cta


def prefill_attention_paged_into(q_batch, k_batch, v_batch, q_norm, k_norm, cos, sin, kv_buffer, layout, layer,
                                 plan, output, num_q_heads, num_kv_heads, head_dim, start_pos, rms_eps):
    """ops/attention.rs:310-458: qk_norm_rope -> scatter -> causal batch prefill."""
    if plan.batch_size == 1:
        prefill_qk_norm_rope_only(q_batch, k_batch, q_norm, k_norm, cos, sin, num_q_heads, num_kv_heads, head_dim,
                                  start_pos, rms_eps)
    else:
        qk_norm_rope_batch_decode_into(q_batch, k_batch, q_norm, k_norm, cos, sin, plan.positions_d, num_q_heads,
                                       num_kv_heads, head_dim, rms_eps)
    paged_kv_scatter(kv_buffer, layout, layer, plan.page_indices_d, plan.page_indptr_d, plan.last_page_len_d,
                     k_batch, v_batch, plan.batch_indices_d, plan.positions_d)
    _chk(ffi.lib().batch_prefill_paged_cuda_with_cta_tile_q(
        _p(q_batch), _p(output), _p(kv_buffer), layer * layout.layer_stride,
        layer * layout.layer_stride + layout.kv_block_len, _p(plan.page_indices_d), _p(plan.page_indptr_d),
        _p(plan.last_page_len_d), _p(plan.q_indptr_d), _p(plan.request_indices_d), _p(plan.qo_tile_indices_d),
        _p(plan.kv_tile_indices_d), _p(plan.kv_chunk_size_d), _p(plan.total_num_rows_d), num_q_heads,
        num_kv_heads, head_dim, layout.page_size, plan.total_tokens, plan.batch_size, plan.num_tiles,
        layout.page_stride, 1.0 / math.sqrt(head_dim), plan.cta_tile_q, _stream()),
        "batch_prefill_paged_cuda")


# ------------------------------------------------------------------ sampling (ops/sampling.rs)
FLASHINFER_TOPK_ROW_STATES_BYTES = 1024 * 1024


def flashinfer_topk_row_states_bytes():
    return FLASHINFER_TOPK_ROW_STATES_BYTES


def argmax(x):
    out = torch.zeros(1, dtype=torch.int32, device=x.device)
    ffi.lib().argmax_cuda(_p(x), _p(out), x.numel(), _stream())
    torch.cuda.current_stream().synchronize()
    return int(out.item())


def batched_top1(logits, state, out=None):
    """Greedy token of every row of logits [rows, V] in one launch (pegainfer_kernels_ext.h: pegainfer_batched_top1);
    `state` = zero-initialised uint8 scratch of >= 16 * rows bytes, left zero by the kernel."""
    rows, vocab = logits.shape
    _bf16(logits)
    if out is None:
        out = torch.zeros(rows, dtype=torch.int32, device=logits.device)
    _chk(ffi.lib().pegainfer_batched_top1(_p(logits), vocab, rows, logits.stride(0), _p(state), _p(out), _stream()),
         "pegainfer_batched_top1")
    return out


def gpu_sample_into(logits, probs_scratch, top1_value_scratch, row_states_scratch, valid_scratch, out,
                    temperature, top_k, top_p, random_val):
    """ops/sampling.rs:109-170: greedy branch iff (T<=0 or top_k==1) and top_p>=1; syncs and
    returns the token."""
    import struct
    if (temperature <= 0.0 or top_k == 1) and top_p >= 1.0:
        ffi.lib().flashinfer_top1_cuda(_p(logits), _p(top1_value_scratch), _p(row_states_scratch), _p(out),
                                       logits.numel(), _stream())
    else:
        seed = struct.unpack("<I", struct.pack("<f", random_val))[0]
        ffi.lib().gpu_sample_flashinfer_cuda(_p(logits), _p(probs_scratch), _p(valid_scratch), _p(out),
                                             logits.numel(), 1.0 / temperature, top_k, top_p, seed, _stream())
    torch.cuda.current_stream().synchronize()
    return int(out.item())


def gpu_sample(logits, probs_scratch, top1_value_scratch, row_states_scratch, temperature, top_k, top_p,
               random_val):
    valid = torch.zeros(1, dtype=torch.uint8, device=logits.device)
    out = torch.zeros(1, dtype=torch.int32, device=logits.device)
    return gpu_sample_into(logits, probs_scratch, top1_value_scratch, row_states_scratch, valid, out,
                           temperature, top_k, top_p, random_val)


# ------------------------------------------------------------------ Qwen3.5 gated delta rule, chunk-wise prefill
class GdrChunkwiseScratch35:
    """Pre-allocated scratch of the 7-stage operator (pegainfer-qwen35-4b/src/prefill_buffers.rs:8-95)."""
    CHUNK_SIZE = 64

    def __init__(self, num_value_heads, key_dim, value_dim, seq_len, device="cuda"):
        f32 = dict(dtype=torch.float32, device=device)
        bf = dict(dtype=torch.bfloat16, device=device)
        H, T, C = num_value_heads, seq_len, self.CHUNK_SIZE
        self.num_chunks = (T + C - 1) // C
        self.g_cumsum = torch.zeros(T * H, **f32)
        self.beta = torch.zeros(T * H, **f32)
        self.q_expanded = torch.zeros((T, H * key_dim), **bf)
        self.k_expanded = torch.zeros((T, H * key_dim), **bf)
        self.v_raw = torch.zeros((T, H * value_dim), **bf)
        self.a_tril = torch.zeros(T * H * C, **f32)
        self.a_inv = torch.zeros(T * H * C, **bf)
        self.w = torch.zeros((T, H * key_dim), **bf)
        self.u = torch.zeros((T, H * value_dim), **bf)
        self.v_new = torch.zeros((T, H * value_dim), **bf)
        self.chunk_state = torch.zeros(self.num_chunks * H * value_dim * key_dim, **f32)


def gated_delta_rule_prefill_chunkwise_into(qkv, b_proj, a_proj, dt_bias, a_log, state, scratch, output,
                                            num_key_heads, num_value_heads, key_dim, val_dim):
    """recurrent.rs:368-470: prepare -> cumsum (in place) -> a -> solve -> recompute -> state -> o."""
    L, s, st = ffi.lib(), scratch, _stream()
    T = qkv.shape[0]
    assert key_dim == 128 and val_dim == 128 and output.shape == (T, num_value_heads * val_dim)
    assert s.q_expanded.shape == (T, num_value_heads * key_dim) and state.dtype == torch.float32
    _chk(L.gated_delta_rule_prefill_chunk_prepare_cuda(
        _p(qkv), _p(b_proj), _p(a_proj), _p(dt_bias), _p(a_log), _p(s.q_expanded), _p(s.k_expanded), _p(s.v_raw),
        _p(s.g_cumsum), _p(s.beta), num_key_heads, num_value_heads, qkv.shape[1], T, st), "gdr prepare")
    _chk(L.gated_delta_rule_prefill_chunk_cumsum_cuda(_p(s.g_cumsum), _p(s.g_cumsum), T, num_value_heads, st),
         "gdr cumsum")
    _chk(L.gated_delta_rule_prefill_chunk_a_cuda(_p(s.k_expanded), _p(s.g_cumsum), _p(s.beta), _p(s.a_tril), T,
                                                 num_value_heads, st), "gdr a")
    _chk(L.gated_delta_rule_prefill_chunk_solve_cuda(_p(s.a_tril), _p(s.a_inv), T, num_value_heads, st), "gdr solve")
    _chk(L.gated_delta_rule_prefill_chunk_recompute_cuda(_p(s.k_expanded), _p(s.v_raw), _p(s.beta), _p(s.w), _p(s.u),
                                                         _p(s.a_inv), _p(s.g_cumsum), T, num_value_heads, st),
         "gdr recompute")
    _chk(L.gated_delta_rule_prefill_chunk_state_cuda(_p(s.k_expanded), _p(s.w), _p(s.u), _p(s.g_cumsum), _p(state),
                                                     _p(s.chunk_state), _p(s.v_new), _p(state), T, num_value_heads,
                                                     st), "gdr state")
    _chk(L.gated_delta_rule_prefill_chunk_o_cuda(_p(s.q_expanded), _p(s.k_expanded), _p(s.v_new), _p(s.chunk_state),
                                                 _p(s.g_cumsum), _p(output), T, num_value_heads,
                                                 1.0 / float(key_dim) ** 0.5, st), "gdr o")
