"""numpy restatement of the pegainfer-kernels ops (TEST INFRASTRUCTURE ONLY).

Every function cites the reference source it follows (paths relative to the
reference checkout).  Arrays are float32 holding bf16-representable values
unless noted; "HiddenStates [d, T]" is stored as ``x[T, d]`` (token-major,
pegainfer-kernels/src/tensor.rs:210-217).

Where the arithmetic lives in an un-vendored dependency (FlashInfer — git
submodule pegainfer-kernels/third_party/flashinfer, pinned SHA absent from the
checkout; cuBLAS) the published algorithm is restated and the function says so.
"""
import numpy as np

from .bf16 import bf16_round

F32 = np.float32


# --------------------------------------------------------------------------
# elementwise / embedding
# --------------------------------------------------------------------------
def embedding_batched(embed, token_ids):
    """out[i,:] = embed[token_ids[i],:]   (csrc/elementwise.cu:67-81)."""
    return np.ascontiguousarray(embed[np.asarray(token_ids, dtype=np.int64)])


def embedding_batched_vocab_shard(embed_shard, token_ids, vocab_start, part_vocab_size):
    """TP vocab shard lookup, zeros outside the shard (csrc/elementwise.cu:92-112)."""
    ids = np.asarray(token_ids, dtype=np.int64)
    out = np.zeros((ids.shape[0], embed_shard.shape[1]), dtype=F32)
    inside = (ids >= vocab_start) & (ids < vocab_start + part_vocab_size)
    out[inside] = embed_shard[ids[inside] - vocab_start]
    return out


def add(a, b):
    """bf16(f32(a)+f32(b))   (csrc/elementwise.cu:8-20)."""
    return bf16_round(a.astype(F32) + b.astype(F32))


def _silu(g):
    g = g.astype(F32)
    return (g / (F32(1.0) + np.exp(-g, dtype=F32))).astype(F32)


def silu_mul_fused(gate_up, intermediate_size):
    """out[j,i] = bf16(silu(gu[j,i]) * gu[j,I+i]), single rounding
    (csrc/fused_proj.cu:44-63).  gate_up is [T, 2I]."""
    g = gate_up[:, :intermediate_size]
    u = gate_up[:, intermediate_size:2 * intermediate_size]
    return bf16_round(_silu(g) * u.astype(F32))


def silu_mul(gate, up):
    """Qwen3.5 variant: silu rounded to bf16 before the multiply
    (csrc/elementwise.cu:28-42)."""
    return bf16_round(bf16_round(_silu(gate)) * up.astype(F32))


# --------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------
def rms_norm(x, weight, eps, offset=False):
    """Per-row RMSNorm, fp32 sum, ONE rounding at the store.

    Call sites csrc/flashinfer_norm.cu:49-65 (RMSNorm) and :110-133
    (GemmaRMSNorm, ``offset=True`` -> (1+w)).  FlashInfer norm.cuh (un-vendored)
    computes ``out = bf16(f32(x) * rsqrt(mean(x^2)+eps) * (bias + f32(w)))``.
    """
    x = np.atleast_2d(x).astype(F32)
    ss = (x.astype(np.float64) ** 2).sum(axis=-1, keepdims=True)
    inv = (1.0 / np.sqrt(ss / x.shape[-1] + np.float64(eps))).astype(F32)
    w = weight.astype(F32) + (F32(1.0) if offset else F32(0.0))
    return bf16_round(x * inv * w)


def rms_norm_kat_reference(x, weight, eps, offset=False):
    """The reference's OWN test oracle (pegainfer-server/src/ops/tests.rs:12-34):
    rounds x*inv_rms to bf16 before the weight multiply; used with tol 0.01-0.02."""
    x = np.asarray(x, dtype=F32)
    ss = F32(0.0)
    for v in x:
        ss = F32(ss + v * v)
    inv = F32(1.0) / np.sqrt(F32(ss / F32(len(x))) + F32(eps), dtype=F32)
    normed = bf16_round(x * inv)
    scale = weight.astype(F32) + (F32(1.0) if offset else F32(0.0))
    return bf16_round(normed * scale)


def fused_add_rms_norm(hidden, residual, weight, eps, offset=False):
    """hidden += residual (stored bf16); out = norm(UNROUNDED fp32 sum) * w.

    Wrapper csrc/flashinfer_norm.cu:71-105 (memcpy residual->out, then
    FusedAddRMSNorm(input=out, residual=hidden)).  FlashInfer keeps the fp32 sum
    in shared memory for the normalisation and stores its bf16 rounding to
    ``hidden`` (restated; header not in tree).  Returns (new_hidden, out).
    """
    s = hidden.astype(F32) + residual.astype(F32)
    new_hidden = bf16_round(s)
    ss = (s.astype(np.float64) ** 2).sum(axis=-1, keepdims=True)
    inv = (1.0 / np.sqrt(ss / s.shape[-1] + np.float64(eps))).astype(F32)
    w = weight.astype(F32) + (F32(1.0) if offset else F32(0.0))
    return new_hidden, bf16_round(s * inv * w)


def rms_norm_gated(x, weight_f32, gate, head_dim, eps):
    """Per-head RMSNorm (fp32 weight) * silu(gate), one rounding (csrc/norm.cu:17-61).
    x, gate: [..., heads*head_dim]."""
    shp = x.shape
    xh = x.reshape(-1, head_dim).astype(F32)
    gh = gate.reshape(-1, head_dim).astype(F32)
    ss = (xh.astype(np.float64) ** 2).sum(axis=-1, keepdims=True)
    inv = (1.0 / np.sqrt(ss / head_dim + np.float64(eps))).astype(F32)
    normed = xh * inv * weight_f32.astype(F32)
    return bf16_round(normed * _silu(gh)).reshape(shp)


# --------------------------------------------------------------------------
# RoPE table + per-head QK norm + RoPE
# --------------------------------------------------------------------------
def precompute_rope(head_dim, max_seq_len, theta):
    """cos/sin tables, fp32 math -> bf16, half-split duplicated layout
    (pegainfer-core/src/weight_loader.rs:210-244).  Returns [max_seq_len, head_dim]."""
    half = head_dim // 2
    i = np.arange(half, dtype=F32)
    inv_freq = (F32(1.0) / np.power(F32(theta), i * F32(2.0) / F32(head_dim), dtype=F32)).astype(F32)
    pos = np.arange(max_seq_len, dtype=F32)[:, None]
    freq = (pos * inv_freq[None, :]).astype(F32)
    c = bf16_round(np.cos(freq, dtype=F32))
    s = bf16_round(np.sin(freq, dtype=F32))
    return np.concatenate([c, c], axis=1), np.concatenate([s, s], axis=1)


def qk_norm_rope(q, k, q_w, k_w, cos, sin, positions, num_q_heads, num_kv_heads, head_dim, eps):
    """Per-(head, token) RMSNorm + NeoX half-split RoPE, in the reference's
    rounding order (csrc/prefill_attention.cu:12-88):
        n = bf16(x * inv_rms); m = bf16(f32(n) * f32(w));
        d<half : out = bf16(m[d]*c - m[d+half]*s)
        d>=half: out = bf16(m[d-half]*s + m[d]*c)     c,s = table[pos, d mod half]
    q: [T, Hq*D], k: [T, Hkv*D]; positions: int[T].  Returns (q_out, k_out)."""
    positions = np.asarray(positions, dtype=np.int64)
    half = head_dim // 2

    def one(x, w, heads):
        T = x.shape[0]
        xh = x.reshape(T, heads, head_dim).astype(F32)
        ss = (xh.astype(np.float64) ** 2).sum(axis=-1, keepdims=True)
        inv = (1.0 / np.sqrt(ss / head_dim + np.float64(eps))).astype(F32)
        n = bf16_round(xh * inv)
        m = bf16_round(n * w.astype(F32)[None, None, :])
        c = cos[positions][:, None, :half].astype(F32)
        s = sin[positions][:, None, :half].astype(F32)
        lo, hi = m[..., :half], m[..., half:]
        out = np.empty_like(m)
        out[..., :half] = bf16_round(lo * c - hi * s)
        out[..., half:] = bf16_round(lo * s + hi * c)
        return out.reshape(T, heads * head_dim)

    return one(q, q_w, num_q_heads), one(k, k_w, num_kv_heads)


# --------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------
GEMM_ACCUM = np.float64  # bench.py's cpu_baseline switches to float32 (plain sgemm) for timing only


def gemm(W, X):
    """Y[T,M] = X[T,K] @ W[M,K]^T, bf16 in, fp32 accumulate, bf16 store
    (csrc/linear.cu:45-75; cublasGemmEx COMPUTE_32F — reduction order is cuBLAS
    internal, so this oracle accumulates in float64 and parity is by tolerance)."""
    if GEMM_ACCUM is np.float32:
        return bf16_round(np.asarray(X, dtype=F32) @ W.T)
    y = X.astype(np.float64) @ W.astype(np.float64).T
    return bf16_round(y.astype(F32))


# --------------------------------------------------------------------------
# paged KV geometry + scatter
# --------------------------------------------------------------------------
class PagedKvLayout:
    """Page-first geometry (pegainfer-kernels/src/paged_kv.rs:5-34 ==
    pegainfer-core/src/kv_pool.rs:14-54): cache = [page][layer][K,V][slot][Hkv][D]."""

    def __init__(self, num_layers, num_kv_heads, head_dim, page_size):
        self.page_size = page_size
        self.num_layers = num_layers
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.kv_block_len = page_size * num_kv_heads * head_dim
        self.layer_stride = 2 * self.kv_block_len
        self.page_stride = num_layers * self.layer_stride

    def k_offset(self, layer):
        return layer * self.layer_stride

    def v_offset(self, layer):
        return layer * self.layer_stride + self.kv_block_len


def paged_kv_scatter(kv_data, layout, layer, page_indices, page_indptr, src_k, src_v,
                     batch_indices, positions):
    """Append K,V rows into the page-first cache (wrapper csrc/paged_attention.cu:274-311
    -> FlashInfer AppendPagedKVCache: page = indices[indptr[b] + pos/page_size],
    slot = pos % page_size).  kv_data is the flat pool (modified in place);
    src_k/src_v: [nnz, Hkv*D]."""
    ps, H, D = layout.page_size, layout.num_kv_heads, layout.head_dim
    ko, vo = layout.k_offset(layer), layout.v_offset(layer)
    for i in range(len(batch_indices)):
        b, pos = int(batch_indices[i]), int(positions[i])
        page = int(page_indices[int(page_indptr[b]) + pos // ps])
        base = page * layout.page_stride + (pos % ps) * H * D
        kv_data[base + ko: base + ko + H * D] = src_k[i]
        kv_data[base + vo: base + vo + H * D] = src_v[i]


def _gather_kv(kv_data, layout, layer, pages, kv_len):
    """Return K,V as [kv_len, Hkv, D] float32 for one request."""
    ps, H, D = layout.page_size, layout.num_kv_heads, layout.head_dim
    ko, vo = layout.k_offset(layer), layout.v_offset(layer)
    K = np.empty((kv_len, H, D), dtype=F32)
    V = np.empty((kv_len, H, D), dtype=F32)
    for t0 in range(0, kv_len, ps):
        page = int(pages[t0 // ps])
        n = min(ps, kv_len - t0)
        base = page * layout.page_stride
        K[t0:t0 + n] = kv_data[base + ko: base + ko + n * H * D].reshape(n, H, D)
        V[t0:t0 + n] = kv_data[base + vo: base + vo + n * H * D].reshape(n, H, D)
    return K, V


def paged_kv_len(page_indptr, last_page_len, b, page_size):
    """FlashInfer paged_kv_t::get_length: (num_pages-1)*page_size + last_page_len."""
    n = int(page_indptr[b + 1]) - int(page_indptr[b])
    return 0 if n == 0 else (n - 1) * page_size + int(last_page_len[b])


# --------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------
def _attend(q, K, V, sm_scale, lo=0, hi=None):
    """q [Hq, D]; K,V [L, Hkv, D] -> (o [Hq, D] float64, lse_log2 [Hq])."""
    hi = K.shape[0] if hi is None else hi
    Hq, D = q.shape
    Hkv = K.shape[1]
    g = Hq // Hkv
    o = np.zeros((Hq, D), dtype=np.float64)
    lse = np.full((Hq,), -np.inf, dtype=np.float64)
    if hi <= lo:
        return o, lse
    Kc = K[lo:hi].astype(np.float64)
    Vc = V[lo:hi].astype(np.float64)
    for h in range(Hq):
        kh = h // g
        s = (Kc[:, kh, :] @ q[h].astype(np.float64)) * np.float64(sm_scale)
        m = s.max()
        p = np.exp(s - m)
        d = p.sum()
        o[h] = (p @ Vc[:, kh, :]) / d
        lse[h] = (m + np.log(d)) / np.log(2.0)
    return o, lse


def paged_attention_decode(q, kv_data, layout, layer, page_indices, page_indptr,
                           last_page_len, num_q_heads, sm_scale):
    """GQA decode attention over the paged cache, no KV partition
    (wrapper csrc/paged_attention.cu:77-145 -> FlashInfer
    BatchDecodeWithPagedKVCacheDispatched<128, kNone, DefaultAttention>):
    kv_len comes from the page table, fp32 online softmax, bf16 output.
    q: [bs, Hq*D] -> o: [bs, Hq*D]."""
    bs = q.shape[0]
    D = layout.head_dim
    out = np.zeros_like(q, dtype=F32)
    for b in range(bs):
        kv_len = paged_kv_len(page_indptr, last_page_len, b, layout.page_size)
        pages = page_indices[int(page_indptr[b]): int(page_indptr[b + 1])]
        K, V = _gather_kv(kv_data, layout, layer, pages, kv_len)
        o, _ = _attend(q[b].reshape(num_q_heads, D), K, V, sm_scale)
        out[b] = bf16_round(o.astype(F32)).reshape(-1)
    return out


def paged_attention_decode_split_kv(q, kv_data, layout, layer, page_indices, page_indptr,
                                    last_page_len, split_request_indices, split_kv_tile_indices,
                                    kv_chunk_size, o_indptr, block_valid_mask,
                                    num_q_heads, sm_scale):
    """Partition-KV decode (wrapper csrc/paged_attention.cu:158-230): slot ``s`` handles
    tokens [tile*chunk, min((tile+1)*chunk, kv_len)) of request ``request_indices[s]``,
    writes a bf16 partial o and fp32 log2-sum-exp; VariableLengthMergeStates then merges
    slots o_indptr[b]..o_indptr[b+1] (FlashInfer cascade.cuh, restated).  Returns o [bs, Hq*D]."""
    bs = q.shape[0]
    D = layout.head_dim
    n_slots = len(split_request_indices)
    part_o = np.zeros((n_slots, num_q_heads, D), dtype=F32)
    part_s = np.full((n_slots, num_q_heads), -np.inf, dtype=np.float64)
    for s in range(n_slots):
        if block_valid_mask is not None and not block_valid_mask[s]:
            continue
        b = int(split_request_indices[s])
        t = int(split_kv_tile_indices[s])
        kv_len = paged_kv_len(page_indptr, last_page_len, b, layout.page_size)
        lo = t * kv_chunk_size
        hi = min((t + 1) * kv_chunk_size, kv_len)
        pages = page_indices[int(page_indptr[b]): int(page_indptr[b + 1])]
        K, V = _gather_kv(kv_data, layout, layer, pages, kv_len)
        o, lse = _attend(q[b].reshape(num_q_heads, D), K, V, sm_scale, lo, hi)
        part_o[s] = bf16_round(o.astype(F32))
        part_s[s] = lse
    out = np.zeros_like(q, dtype=F32)
    for b in range(bs):
        lo, hi = int(o_indptr[b]), int(o_indptr[b + 1])
        if hi <= lo:
            continue
        s = part_s[lo:hi]                       # [n, Hq]
        m = s.max(axis=0)
        w = np.exp2(s - m[None, :])             # [n, Hq]
        o = (part_o[lo:hi].astype(np.float64) * w[:, :, None]).sum(axis=0) / w.sum(axis=0)[:, None]
        out[b] = bf16_round(o.astype(F32)).reshape(-1)
    return out


def batch_prefill_paged(q, kv_data, layout, layer, page_indices, page_indptr, last_page_len,
                        q_indptr, num_q_heads, sm_scale, row_block=1024, causal_shift=0):
    """Causal varlen GQA prefill over the paged cache (wrapper
    csrc/paged_attention.cu:399-535 -> FlashInfer BatchPrefillWithPagedKVCacheDispatched,
    MaskMode::kCausal): query row i of a request with qo_len rows and kv_len cached
    tokens attends kv positions <= i + kv_len - qo_len.  q: [T, Hq*D] -> o: [T, Hq*D].

    Same float64 arithmetic as ``batch_prefill_paged_rowwise`` (scores, max-subtracted exp,
    normalised PV, one bf16 rounding), evaluated one (request, kv head, row block) at a time as two
    matrix products so that 2-4k-token cases finish in seconds; tests/test_oracle_kats.py pins the
    two forms against each other."""
    D = layout.head_dim
    out = np.zeros_like(q, dtype=F32)
    nreq = len(q_indptr) - 1
    for b in range(nreq):
        q0, q1 = int(q_indptr[b]), int(q_indptr[b + 1])
        qo_len = q1 - q0
        if qo_len <= 0:
            continue
        kv_len = paged_kv_len(page_indptr, last_page_len, b, layout.page_size)
        pages = page_indices[int(page_indptr[b]): int(page_indptr[b + 1])]
        K, V = _gather_kv(kv_data, layout, layer, pages, kv_len)
        Hkv = K.shape[1]
        g = num_q_heads // Hkv
        Qb = q[q0:q1].reshape(qo_len, num_q_heads, D)
        ob = out[q0:q1].reshape(qo_len, num_q_heads, D)
        for kh in range(Hkv):
            Kh = K[:, kh, :].astype(np.float64)
            Vh = V[:, kh, :].astype(np.float64)
            for r0 in range(0, qo_len, row_block):
                r1 = min(qo_len, r0 + row_block)
                hi = np.arange(r0, r1) + (kv_len - qo_len) + 1            # kv positions < hi are visible
                if causal_shift:   # fault injection of tests/test_depth_harness.py only: an off-by-one mask
                    hi = np.clip(hi + causal_shift, 1, kv_len)
                top = int(hi[-1])
                Qg = Qb[r0:r1, kh * g:(kh + 1) * g, :].astype(np.float64)  # [rows, g, D]
                S = (Qg.reshape(-1, D) @ Kh[:top].T) * np.float64(sm_scale)
                S = S.reshape(r1 - r0, g, top)
                masked = np.arange(top)[None, :] >= hi[:, None]          # [rows, top], shared by the g heads
                S = np.where(masked[:, None, :], -np.inf, S)
                m = S.max(axis=-1, keepdims=True)
                P = np.exp(S - m)
                d = P.sum(axis=-1, keepdims=True)
                O = (P.reshape(-1, top) @ Vh[:top]).reshape(r1 - r0, g, D) / d
                ob[r0:r1, kh * g:(kh + 1) * g, :] = bf16_round(O.astype(F32))
    return out


def batch_prefill_paged_rowwise(q, kv_data, layout, layer, page_indices, page_indptr, last_page_len,
                                q_indptr, num_q_heads, sm_scale):
    """Row-at-a-time form of batch_prefill_paged (the literal restatement: one _attend per query row)."""
    D = layout.head_dim
    out = np.zeros_like(q, dtype=F32)
    nreq = len(q_indptr) - 1
    for b in range(nreq):
        q0, q1 = int(q_indptr[b]), int(q_indptr[b + 1])
        qo_len = q1 - q0
        kv_len = paged_kv_len(page_indptr, last_page_len, b, layout.page_size)
        pages = page_indices[int(page_indptr[b]): int(page_indptr[b + 1])]
        K, V = _gather_kv(kv_data, layout, layer, pages, kv_len)
        for i in range(qo_len):
            hi = i + kv_len - qo_len + 1
            o, _ = _attend(q[q0 + i].reshape(num_q_heads, D), K, V, sm_scale, 0, hi)
            out[q0 + i] = bf16_round(o.astype(F32)).reshape(-1)
    return out


# --------------------------------------------------------------------------
# host-side plan helpers (pure integer; bit-exact parity required)
# --------------------------------------------------------------------------
def fa2_determine_cta_tile_q(avg_packed_qo_len, head_dim):
    """FlashInfer FA2DetermineCtaTileQ (utils.cuh, un-vendored; called from
    csrc/paged_attention.cu:332): >64 && D<256 -> 128; else (sm>=80) >16 -> 64 else 16."""
    if avg_packed_qo_len > 64 and head_dim < 256:
        return 128
    return 64 if avg_packed_qo_len > 16 else 16


def resolve_prefill_cta_tile_q(packed_qo_len, head_dim, override):
    """csrc/paged_attention.cu:312-324: 0 -> heuristic; {16,64,128} accepted; else 0 (invalid)."""
    if override == 0:
        return fa2_determine_cta_tile_q(packed_qo_len, head_dim)
    return override if override in (16, 64, 128) else 0


def batch_prefill_paged_num_tiles(seq_len, num_qo_heads, num_kv_heads, head_dim, override=0):
    """csrc/paged_attention.cu:326-358; -1 for an invalid override."""
    packed = seq_len * (num_qo_heads // num_kv_heads)
    t = resolve_prefill_cta_tile_q(packed, head_dim, override)
    return -1 if t == 0 else (packed + t - 1) // t


def batch_prefill_cta_tile_q(total_seq_len, num_qo_heads, num_kv_heads, head_dim, override=0):
    """csrc/paged_attention.cu:363-397."""
    return resolve_prefill_cta_tile_q(total_seq_len * (num_qo_heads // num_kv_heads), head_dim, override)


def prefill_paged_plan(page_indices, last_page_lens, start_positions, seq_lens,
                       num_q_heads, num_kv_heads, head_dim, cta_tile_q_override=0):
    """PrefillPagedPlan::new_batch_with_cta_tile_q (pegainfer-kernels/src/ops/attention.rs:208-302).
    Returns a dict of the 11 arrays + scalars."""
    group = num_q_heads // num_kv_heads
    total = int(sum(seq_lens))
    all_pages, indptr, kv_chunk = [], [0], []
    for i, pages in enumerate(page_indices):
        all_pages.extend(int(p) for p in pages)
        indptr.append(len(all_pages))
        kv_chunk.append(int(start_positions[i] + seq_lens[i]))
    batch_indices, positions = [], []
    for i, n in enumerate(seq_lens):
        batch_indices.extend([i] * n)
        positions.extend(range(start_positions[i], start_positions[i] + n))
    q_indptr = [0]
    for n in seq_lens:
        q_indptr.append(q_indptr[-1] + n)
    cta = batch_prefill_cta_tile_q(total, num_q_heads, num_kv_heads, head_dim, cta_tile_q_override)
    if cta <= 0:
        raise ValueError(f"invalid prefill CTA tile override {cta_tile_q_override}")
    req, qo_tile, kv_tile = [], [], []
    for r, n in enumerate(seq_lens):
        for t in range((n * group + cta - 1) // cta):
            req.append(r)
            qo_tile.append(t)
            kv_tile.append(0)
    i32 = lambda a: np.asarray(a, dtype=np.int32)
    return dict(page_indices=i32(all_pages), page_indptr=i32(indptr),
                last_page_len=i32(last_page_lens), batch_indices=i32(batch_indices),
                positions=i32(positions), q_indptr=i32(q_indptr), request_indices=i32(req),
                qo_tile_indices=i32(qo_tile), kv_tile_indices=i32(kv_tile),
                kv_chunk_size=i32(kv_chunk), total_num_rows=np.asarray([total], dtype=np.uint32),
                num_tiles=len(req), batch_size=len(seq_lens), total_tokens=total, cta_tile_q=cta)


SPLIT_KV_CHUNK_TOKENS = 256
SPLIT_KV_MAX_CHUNKS_PER_REQUEST = 64
SPLIT_KV_MAX_BATCH_SIZE = 2
SPLIT_KV_MIN_SEQ_LEN = 1024
BATCH_BUCKETS = (1, 2, 4, 8, 16, 32, 64)


def bucket_for(bs):
    """pegainfer-qwen3-4b/src/batch_decode_buffers.rs:35-46."""
    for b in BATCH_BUCKETS:
        if b >= bs:
            return b
    raise ValueError(f"batch size {bs} exceeds largest bucket {BATCH_BUCKETS[-1]}")


def split_kv_plan(seq_lens, padded_bs):
    """BatchDecodeBuffers::sync_split_kv_meta (batch_decode_buffers.rs:229-279)."""
    max_seq = max(seq_lens) if len(seq_lens) else 0
    chunk = max(SPLIT_KV_CHUNK_TOKENS, -(-max_seq // SPLIT_KV_MAX_CHUNKS_PER_REQUEST))
    slots = padded_bs * SPLIT_KV_MAX_CHUNKS_PER_REQUEST
    req, tile, mask, o_indptr = [], [], [], [0]
    for r, n in enumerate(seq_lens):
        chunks = max(1, -(-n // chunk))
        for c in range(chunks):
            req.append(r)
            tile.append(c)
            mask.append(1)
        o_indptr.append(len(req))
    for _ in range(len(seq_lens), padded_bs):
        o_indptr.append(len(req))
    while len(req) < slots:
        req.append(0)
        tile.append(0)
        mask.append(0)
    return dict(request_indices=np.asarray(req, np.int32), kv_tile_indices=np.asarray(tile, np.int32),
                kv_chunk_size=chunk, o_indptr=np.asarray(o_indptr, np.int32),
                block_valid_mask=np.asarray(mask, np.uint8), padded_slots=slots)


def attention_path_is_split(padded_bs, max_seq_len):
    """batch_decode_buffers.rs:281-287."""
    return padded_bs <= SPLIT_KV_MAX_BATCH_SIZE and max_seq_len >= SPLIT_KV_MIN_SEQ_LEN


# --------------------------------------------------------------------------
# sampling
# --------------------------------------------------------------------------
def argmax(x):
    """bf16 argmax, lowest index wins ties (csrc/argmax.cu:18,29-31)."""
    x = np.asarray(x, dtype=F32)
    return int(np.flatnonzero(x == x.max())[0])


def compute_logprobs(logits_f32, sampled_token, top_k):
    """compute_logprobs_from_cpu (pegainfer-qwen3-4b/src/executor.rs:400-434), statement for statement: f32 max fold, f32
    SEQUENTIAL sum of exp(x - max) (Rust's Iterator::sum is a left fold - np.cumsum keeps that order, np.sum would
    not), log_sum_exp = max + ln(sum).  The top list is the reference's ordered insertion, literally: a value enters
    when the list is short or it beats the last entry STRICTLY, at partition_point(v > val) - i.e. in FRONT of entries
    equal to it - so the list is value-descending and, among equal values that made it in, the LATER index comes first;
    a value equal to the last entry of a full list does not enter.
    Returns (logprob, [(token id, logprob), ...]) or None for an empty row."""
    x = np.asarray(logits_f32, dtype=F32)
    if x.size == 0:
        return None
    max_val = F32(x.max())
    e = np.exp(x - max_val, dtype=F32)
    sum_exp = F32(np.cumsum(e, dtype=F32)[-1])
    lse = F32(max_val + np.log(sum_exp, dtype=F32))
    k = min(int(top_k), x.size)
    best = []                                   # (index, value), value descending
    if k > 0:
        for idx, val in enumerate(x.tolist()):
            if len(best) < k or val > best[-1][1]:
                pos = 0
                while pos < len(best) and best[pos][1] > val:      # partition_point(|v| v > val)
                    pos += 1
                best.insert(pos, (idx, val))
                if len(best) > k:
                    best.pop()
    top = [(int(i), float(F32(F32(v) - lse))) for i, v in best]
    return float(F32(x[int(sampled_token)] - lse)), top


def logits_to_probs(logits, inv_temperature):
    """fp32 softmax of logits*inv_T (csrc/flashinfer_sampling.cu:13-70)."""
    v = logits.astype(F32) * F32(inv_temperature)
    e = np.exp(v - v.max(), dtype=F32)
    return (e * (F32(1.0) / e.sum(dtype=F32))).astype(F32)


def top_k_top_p_support(probs, top_k, top_p):
    """Token set that top-k then top-p (FlashInfer joint filter, restated) can emit:
    the top_k most probable tokens intersected with the smallest prefix of the
    descending-sorted distribution whose mass reaches top_p.  Used for the
    distributional check only (RNG stream is 'parity unpinned')."""
    order = np.argsort(-probs, kind="stable")
    keep = np.ones(len(probs), dtype=bool)
    if top_k > 0:
        kth = probs[order[min(top_k, len(probs)) - 1]]
        keep &= probs >= kth
    if top_p < 1.0:
        c = np.cumsum(probs[order].astype(np.float64))
        n = int(np.searchsorted(c, top_p - 1e-6) + 1)
        pth = probs[order[min(n, len(probs)) - 1]]
        keep &= probs >= pth
    return keep


# --------------------------------------------------------------------------
# Qwen3.5 hybrid extras
# --------------------------------------------------------------------------
def conv1d_prefill(x_seq, conv_weight, conv_state):
    """Causal depthwise conv + bf16 round + SiLU, and the state update (csrc/conv1d.cu:18-98).
    x_seq [T, C]; conv_weight [C, K]; conv_state [C, K-1] (modified copy returned).
    Returns (out [T, C], new_state)."""
    T, C = x_seq.shape
    K = conv_weight.shape[1]
    sw = K - 1
    hist = np.concatenate([conv_state.T.astype(F32), x_seq.astype(F32)], axis=0)       # [sw + T, C]
    out = np.empty((T, C), dtype=F32)
    for t in range(T):
        s = np.zeros(C, dtype=F32)
        for k in range(K):
            s = (s + hist[t + k] * conv_weight[:, k].astype(F32)).astype(F32)          # taps in order k = 0..K-1
        r = bf16_round(s)
        out[t] = bf16_round(_silu(r))
    new_state = hist[T:T + sw].T.copy() if sw > 0 else conv_state.copy()
    return out, bf16_round(new_state)


def gated_delta_rule_decode(qkv, b_proj, a_proj, dt_bias, A_log, state, num_key_heads, num_value_heads,
                            key_dim, val_dim):
    """One recurrent decode step per value head, all fp32 (csrc/gated_delta_rule.cu:27-190).
    qkv: [q(kh*key_dim) | k(kh*key_dim) | v(vh*val_dim)]; state [vh, key_dim, val_dim] float32 (updated copy
    returned).  Returns (out [vh*val_dim] bf16-valued, new_state)."""
    q_total = key_dim * num_key_heads
    st = state.astype(np.float64).copy()
    out = np.empty(num_value_heads * val_dim, dtype=F32)
    for h in range(num_value_heads):
        kh = h * num_key_heads // num_value_heads
        q = qkv[kh * key_dim:(kh + 1) * key_dim].astype(np.float64)
        k = qkv[q_total + kh * key_dim: q_total + (kh + 1) * key_dim].astype(np.float64)
        v = qkv[2 * q_total + h * val_dim: 2 * q_total + (h + 1) * val_dim].astype(np.float64)
        q = q / np.sqrt((q * q).sum() + 1e-12) / np.sqrt(key_dim)
        k = k / np.sqrt((k * k).sum() + 1e-12)
        x = float(a_proj[h]) + float(dt_bias[h])
        softplus = x if x > 20.0 else np.log1p(np.exp(x))
        g = -np.exp(float(A_log[h])) * softplus
        beta = 1.0 / (1.0 + np.exp(-float(b_proj[h])))
        S = st[h] * np.exp(g)
        kv = S.T @ k
        delta = (v - kv) * beta
        S = S + np.outer(k, delta)
        st[h] = S
        out[h * val_dim:(h + 1) * val_dim] = (S.T @ q).astype(F32)
    return bf16_round(out), st.astype(F32)


def hd256_norm_partial_rope(x, w, cos, sin, positions, rotary_dim, eps):
    """(1+w) RMSNorm (one rounding) then partial NeoX RoPE on the first rotary_dim dims; table rows are
    rotary_dim wide (csrc/prefill_attention_hd256.cu:7-133).  x: [T, heads, 256]."""
    xs = x.astype(F32)
    ss = (xs.astype(np.float64) ** 2).sum(-1, keepdims=True)
    inv = (1.0 / np.sqrt(ss / 256.0 + np.float64(eps))).astype(F32)
    n = bf16_round(xs * inv * (F32(1.0) + w.astype(F32)))
    half = rotary_dim // 2
    pos = np.asarray(positions, dtype=np.int64)
    c = cos[pos][:, None, :half].astype(F32)
    s = sin[pos][:, None, :half].astype(F32)
    out = n.copy()
    lo, hi = n[..., :half], n[..., half:rotary_dim]
    out[..., :half] = bf16_round(lo * c - hi * s)
    out[..., half:rotary_dim] = bf16_round(lo * s + hi * c)
    return out


def attention_gate_hd256(q_full, attn_out, num_q_heads):
    """attn_out *= sigmoid(gate); gate = second half of each interleaved q_full head
    (csrc/prefill_attention_hd256.cu:135-157).  q_full [T, Hq*512], attn_out [T, Hq*256]."""
    T = attn_out.shape[0]
    gate = q_full.reshape(T, num_q_heads, 2, 256)[:, :, 1, :].reshape(T, -1).astype(F32)
    sig = (F32(1.0) / (F32(1.0) + np.exp(-gate, dtype=F32))).astype(F32)
    return bf16_round(attn_out.astype(F32) * sig)


# --------------------------------------------------------------------------
# Qwen3.5 gated delta rule, chunk-wise prefill (7 stages; chunk 64)
# Follows pegainfer-kernels/tools/triton/gated_delta_rule_chunkwise_kernels.py (the Triton-AOT kernels behind
# ffi.rs gated_delta_rule_prefill_chunk_*_cuda) stage by stage, with its bf16 rounding points, and the
# operator order of pegainfer-qwen35-4b/src/recurrent.rs:368-470.  Matrix products accumulate in float64
# here (the GPU accumulates in fp32: parity by tolerance).
# --------------------------------------------------------------------------
GDR_CHUNK = 64


def gdr_chunk_prepare(qkv, b_proj, a_proj, dt_bias, A_log, num_key_heads, num_value_heads, key_dim, val_dim):
    """(kernels.py:29-84)  qkv [T, 2*kh*K + vh*V]; returns q,k [T, vh, K] bf16 (L2-normalised, head-expanded),
    v [T, vh, V] (raw), g [T, vh] f32 (= -exp(A_log)*softplus(a+dt_bias)), beta [T, vh] f32 (= sigmoid(b))."""
    T = qkv.shape[0]
    qk_total = num_key_heads * key_dim
    kh_of = (np.arange(num_value_heads) * num_key_heads) // num_value_heads
    q_src = qkv[:, :qk_total].reshape(T, num_key_heads, key_dim).astype(F32)[:, kh_of]
    k_src = qkv[:, qk_total:2 * qk_total].reshape(T, num_key_heads, key_dim).astype(F32)[:, kh_of]
    v = qkv[:, 2 * qk_total:].reshape(T, num_value_heads, val_dim).astype(F32).copy()

    def l2(x):
        ss = (x.astype(np.float64) ** 2).sum(-1, keepdims=True)
        return bf16_round((x * (1.0 / np.sqrt(ss + 1e-12))).astype(F32))

    x = a_proj.astype(F32) + dt_bias.astype(F32)[None, :]
    with np.errstate(over="ignore"):
        softplus = np.where(x > 20.0, x, np.log1p(np.exp(x.astype(np.float64)))).astype(F32)
    g = (-np.exp(A_log.astype(F32))[None, :] * softplus).astype(F32)
    beta = (1.0 / (1.0 + np.exp(-b_proj.astype(np.float64)))).astype(F32)
    return l2(q_src), l2(k_src), v, g, beta


def gdr_chunk_cumsum(g):
    """Chunk-local inclusive prefix sum over tokens (kernels.py:143-159).  g [T, vh] f32."""
    out = np.empty_like(g, dtype=F32)
    for c0 in range(0, g.shape[0], GDR_CHUNK):
        out[c0:c0 + GDR_CHUNK] = np.cumsum(g[c0:c0 + GDR_CHUNK].astype(np.float64), axis=0).astype(F32)
    return out


def gdr_chunk_a(k, g_cumsum, beta):
    """A[t, h, j] = beta_t * exp(g_t - g_j) * <k_t, k_j> for j < t inside the chunk, else 0 (kernels.py:162-215).
    k [T, vh, K] -> a_tril [T, vh, 64] f32."""
    T, H, _ = k.shape
    A = np.zeros((T, H, GDR_CHUNK), dtype=F32)
    for c0 in range(0, T, GDR_CHUNK):
        n = min(GDR_CHUNK, T - c0)
        for h in range(H):
            kc = k[c0:c0 + n, h].astype(np.float64)
            gc = g_cumsum[c0:c0 + n, h].astype(np.float64)
            a = (kc @ kc.T) * beta[c0:c0 + n, h].astype(np.float64)[:, None] * np.exp(gc[:, None] - gc[None, :])
            A[c0:c0 + n, h, :n] = np.tril(a, -1).astype(F32)
    return A


def gdr_chunk_solve(a_tril):
    """A_inv = (I + A)^-1 per (chunk, head), rounded to bf16 (kernels.py:218-330).  Rows of a partial last chunk
    beyond seq_len do not exist; columns beyond the chunk length are zero."""
    T, H, _ = a_tril.shape
    out = np.zeros((T, H, GDR_CHUNK), dtype=F32)
    for c0 in range(0, T, GDR_CHUNK):
        n = min(GDR_CHUNK, T - c0)
        for h in range(H):
            M = np.eye(n) + a_tril[c0:c0 + n, h, :n].astype(np.float64)
            out[c0:c0 + n, h, :n] = bf16_round(np.linalg.inv(M).astype(F32))
    return out


def gdr_chunk_recompute(k, v, beta, a_inv, g_cumsum):
    """u = A_inv @ bf16(v * bf16(beta)), w = A_inv @ bf16(bf16(k * bf16(beta)) * bf16(exp(g)))  (kernels.py:333-430).
    Returns (w [T, vh, K], u [T, vh, V]) bf16."""
    T, H, _ = k.shape
    w = np.zeros_like(k, dtype=F32)
    u = np.zeros_like(v, dtype=F32)
    for c0 in range(0, T, GDR_CHUNK):
        n = min(GDR_CHUNK, T - c0)
        for h in range(H):
            ai = a_inv[c0:c0 + n, h, :n].astype(np.float64)
            bb = bf16_round(beta[c0:c0 + n, h].astype(F32))[:, None]
            eg = bf16_round(np.exp(g_cumsum[c0:c0 + n, h].astype(F32)))[:, None]
            vb = bf16_round(v[c0:c0 + n, h].astype(F32) * bb)
            kb = bf16_round(bf16_round(k[c0:c0 + n, h].astype(F32) * bb) * eg)
            u[c0:c0 + n, h] = bf16_round((ai @ vb.astype(np.float64)).astype(F32))
            w[c0:c0 + n, h] = bf16_round((ai @ kb.astype(np.float64)).astype(F32))
    return w, u


def gdr_chunk_state(k, w, u, g_cumsum, state):
    """Sequential over chunks (kernels.py:433-600): snapshot h; v_new = u - w @ bf16(h); store bf16(v_new);
    h = h*exp(g_last) + k^T @ bf16(v_new * exp(g_last - g)).  state [vh, K, V] f32.
    Returns (chunk_state [nchunks, vh, K, V] f32, v_new [T, vh, V] bf16, final_state)."""
    T, H, K = k.shape
    V = u.shape[2]
    nchunks = (T + GDR_CHUNK - 1) // GDR_CHUNK
    chunk_state = np.zeros((nchunks, H, K, V), dtype=F32)
    v_new = np.zeros((T, H, V), dtype=F32)
    h_all = state.astype(F32).copy()
    for ci in range(nchunks):
        c0 = ci * GDR_CHUNK
        n = min(GDR_CHUNK, T - c0)
        chunk_state[ci] = h_all
        for h in range(H):
            hh = h_all[h]
            vn = u[c0:c0 + n, h].astype(np.float64) - w[c0:c0 + n, h].astype(np.float64) @ bf16_round(hh).astype(np.float64)
            vn = vn.astype(F32)
            v_new[c0:c0 + n, h] = bf16_round(vn)
            gc = g_cumsum[c0:c0 + n, h].astype(F32)
            g_last = gc[n - 1]
            gate = np.exp(g_last - gc).astype(F32)
            vg = bf16_round(vn * gate[:, None])
            hh = (hh * np.exp(g_last).astype(F32)).astype(F32)
            hh = (hh.astype(np.float64) + k[c0:c0 + n, h].astype(np.float64).T @ vg.astype(np.float64)).astype(F32)
            h_all[h] = hh
    return chunk_state, v_new, h_all


def gdr_chunk_o(q, k, v_new, chunk_state, g_cumsum, scale):
    """out = (exp(g) * (q @ bf16(h_chunk)) + bf16(tril(q k^T * exp(g_i - g_j))) @ v_new) * scale  (kernels.py:603-709).
    Returns [T, vh*V] bf16."""
    T, H, K = q.shape
    V = v_new.shape[2]
    out = np.zeros((T, H, V), dtype=F32)
    for c0 in range(0, T, GDR_CHUNK):
        n = min(GDR_CHUNK, T - c0)
        ci = c0 // GDR_CHUNK
        for h in range(H):
            qc = q[c0:c0 + n, h].astype(np.float64)
            kc = k[c0:c0 + n, h].astype(np.float64)
            gc = g_cumsum[c0:c0 + n, h].astype(np.float64)
            acc_o = (qc @ bf16_round(chunk_state[ci, h]).astype(np.float64)) * np.exp(gc)[:, None]
            acc_a = np.tril((qc @ kc.T) * np.exp(gc[:, None] - gc[None, :]))
            a_bf = bf16_round(acc_a.astype(F32)).astype(np.float64)
            out[c0:c0 + n, h] = bf16_round(((acc_o + a_bf @ v_new[c0:c0 + n, h].astype(np.float64)) * scale).astype(F32))
    return out.reshape(T, H * V)


def gated_delta_rule_prefill_chunkwise(qkv, b_proj, a_proj, dt_bias, A_log, state, num_key_heads, num_value_heads,
                                       key_dim, val_dim):
    """The 7-stage operator (recurrent.rs:368-470).  Returns (out [T, vh*V] bf16, new_state [vh, K, V] f32)."""
    q, k, v, g, beta = gdr_chunk_prepare(qkv, b_proj, a_proj, dt_bias, A_log, num_key_heads, num_value_heads,
                                         key_dim, val_dim)
    gc = gdr_chunk_cumsum(g)
    a_tril = gdr_chunk_a(k, gc, beta)
    a_inv = gdr_chunk_solve(a_tril)
    w, u = gdr_chunk_recompute(k, v, beta, a_inv, gc)
    chunk_state, v_new, final_state = gdr_chunk_state(k, w, u, gc, state)
    out = gdr_chunk_o(q, k, v_new, chunk_state, gc, 1.0 / np.sqrt(float(key_dim)))
    return out, final_state
