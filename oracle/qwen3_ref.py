"""Whole-model CPU restatement of the reference's Qwen3 DAG (TEST INFRASTRUCTURE ONLY).

Follows the op order of
  * prefill : pegainfer-qwen3-4b/src/prefill.rs:73-285  (forward_layer_batch_paged, batch_prefill)
  * decode  : pegainfer-qwen3-4b/src/batch_decode.rs:82-295 (batch_decode_kernels, batch_decode_layer)
with the fused-weight layout of weights.rs (qkv_proj = vstack(q,k,v); gate_up_proj =
vstack(gate,up)) and the paged KV cache of pegainfer-core/src/kv_pool.rs.

Used (a) as the checker for the HIP model path in tests/ and smoke(), (b) as the
``cpu_baseline`` ("port") in bench.py.  Never a product path.
"""
import numpy as np

from . import ops
from .bf16 import bf16_round

F32 = np.float32


class Qwen3Config:
    def __init__(self, hidden_size, num_hidden_layers, num_attention_heads, num_key_value_heads,
                 head_dim, intermediate_size, vocab_size, rms_norm_eps=1e-6, rope_theta=1e6,
                 tie_word_embeddings=True, max_position_embeddings=40960):
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.head_dim = head_dim
        self.intermediate_size = intermediate_size
        self.vocab_size = vocab_size
        self.rms_norm_eps = rms_norm_eps
        self.rope_theta = rope_theta
        self.tie_word_embeddings = tie_word_embeddings
        self.max_position_embeddings = max_position_embeddings

    @property
    def q_dim(self):
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self):
        return self.num_key_value_heads * self.head_dim

    @staticmethod
    def qwen3_4b():
        """SURVEY.md §2a dims (KERNELS.md:9, docs/models/qwen3/tp-design.md:84-110)."""
        return Qwen3Config(2560, 36, 32, 8, 128, 9728, 151936)

    @staticmethod
    def qwen3_8b():
        return Qwen3Config(4096, 36, 32, 8, 128, 12288, 151936, tie_word_embeddings=False)


class KvState:
    """Per-request page list + seq_len (pegainfer-core/src/kv_pool.rs:147-260)."""

    def __init__(self):
        self.pages = []
        self.seq_len = 0


class Qwen3Oracle:
    """weights: dict of float32 arrays (bf16-valued) with HF names
    (pegainfer-qwen3-4b/src/weights.rs:102-296)."""

    PAGE_SIZE = 16  # weights.rs:309

    def __init__(self, cfg, weights, num_pages=64, rope_positions=4096, all_reduce=None):
        """all_reduce: optional callable(float32 array) -> summed array; applied to the bf16 O-proj and
        down-proj outputs exactly where the reference's TP all-reduces (weights.rs:396-405)."""
        self.all_reduce = all_reduce
        self.taps = None   # a list: every layer appends its output rows (prefill: each request's last row; decode: all rows) -
        #                    the per-layer comparison of docs/playbooks/accuracy-parity-playbook.md:15-24
        self.cfg = cfg
        self.w = weights
        c = cfg
        self.layers = []
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            self.layers.append(dict(
                qkv=np.concatenate([weights[p + "self_attn.q_proj.weight"],
                                    weights[p + "self_attn.k_proj.weight"],
                                    weights[p + "self_attn.v_proj.weight"]], axis=0),
                o=weights[p + "self_attn.o_proj.weight"],
                q_norm=weights[p + "self_attn.q_norm.weight"],
                k_norm=weights[p + "self_attn.k_norm.weight"],
                gate_up=np.concatenate([weights[p + "mlp.gate_proj.weight"],
                                        weights[p + "mlp.up_proj.weight"]], axis=0),
                down=weights[p + "mlp.down_proj.weight"],
                ln1=weights[p + "input_layernorm.weight"],
                ln2=weights[p + "post_attention_layernorm.weight"]))
        self.embed = weights["model.embed_tokens.weight"]
        self.norm = weights["model.norm.weight"]
        self.lm_head = self.embed if c.tie_word_embeddings else weights["lm_head.weight"]
        self.cos, self.sin = ops.precompute_rope(c.head_dim, rope_positions, c.rope_theta)
        self.layout = ops.PagedKvLayout(c.num_hidden_layers, c.num_key_value_heads, c.head_dim,
                                        self.PAGE_SIZE)
        self.kv = np.zeros(num_pages * self.layout.page_stride, dtype=F32)
        self.free_pages = list(range(1, num_pages))  # page 0 = padding page (kv_pool.rs:100-103)
        self.sm_scale = F32(1.0) / np.sqrt(F32(c.head_dim))

    # -- page bookkeeping (kv_pool.rs ensure_capacity/advance) --------------
    def _ensure(self, st, tokens):
        need = -(-tokens // self.PAGE_SIZE)
        while len(st.pages) < need:
            st.pages.append(self.free_pages.pop(0))

    def _meta(self, states):
        pages, indptr, last = [], [0], []
        for st in states:
            pages.extend(st.pages)
            indptr.append(len(pages))
            rem = st.seq_len % self.PAGE_SIZE
            last.append(0 if st.seq_len == 0 else (self.PAGE_SIZE if rem == 0 else rem))
        return (np.asarray(pages, np.int32), np.asarray(indptr, np.int32), np.asarray(last, np.int32))

    # -- one transformer layer body shared by both phases --------------------
    def _attn_block(self, li, L, normed, positions, batch_indices, meta, decode, q_indptr=None):
        c = self.cfg
        q = ops.gemm(L["qkv"][:c.q_dim], normed)
        k = ops.gemm(L["qkv"][c.q_dim:c.q_dim + c.kv_dim], normed)
        v = ops.gemm(L["qkv"][c.q_dim + c.kv_dim:], normed)
        q, k = ops.qk_norm_rope(q, k, L["q_norm"], L["k_norm"], self.cos, self.sin, positions,
                                c.num_attention_heads, c.num_key_value_heads, c.head_dim,
                                c.rms_norm_eps)
        pages, indptr, last = meta
        ops.paged_kv_scatter(self.kv, self.layout, li, pages, indptr, k, v, batch_indices, positions)
        if decode:
            return ops.paged_attention_decode(q, self.kv, self.layout, li, pages, indptr, last,
                                              c.num_attention_heads, self.sm_scale)
        return ops.batch_prefill_paged(q, self.kv, self.layout, li, pages, indptr, last, q_indptr,
                                       c.num_attention_heads, self.sm_scale)

    def _reduce(self, x):
        return x if self.all_reduce is None else bf16_round(self.all_reduce(x).astype(F32))

    def _mlp(self, L, normed):
        gu = ops.gemm(L["gate_up"], normed)
        act = ops.silu_mul_fused(gu, self.cfg.intermediate_size)
        return self._reduce(ops.gemm(L["down"], act))

    # -- prefill (prefill.rs:220-285) ----------------------------------------
    def batch_prefill(self, prompts, states, echo=False):
        """echo=True also returns compute_all_position_logits (prefill.rs:196-212): [total_tokens, vocab]."""
        c = self.cfg
        seq_lens = [len(p) for p in prompts]
        starts = [st.seq_len for st in states]
        tokens = np.concatenate([np.asarray(p, np.int64) for p in prompts])
        hidden = ops.embedding_batched(self.embed, tokens)
        for st, n, s0 in zip(states, seq_lens, starts):
            self._ensure(st, s0 + n)
            st.seq_len += n
        meta = self._meta(states)
        positions = np.concatenate([np.arange(s0, s0 + n) for s0, n in zip(starts, seq_lens)])
        batch_indices = np.concatenate([np.full(n, i) for i, n in enumerate(seq_lens)])
        q_indptr = np.concatenate([[0], np.cumsum(seq_lens)])
        for li, L in enumerate(self.layers):
            normed = ops.rms_norm(hidden, L["ln1"], c.rms_norm_eps)
            attn = self._attn_block(li, L, normed, positions, batch_indices, meta, False, q_indptr)
            o = self._reduce(ops.gemm(L["o"], attn))
            hidden, normed = ops.fused_add_rms_norm(hidden, o, L["ln2"], c.rms_norm_eps)
            mlp = self._mlp(L, normed)
            hidden = ops.add(hidden, mlp)                        # prefill.rs:183 plain add
            if self.taps is not None:
                self.taps.append(hidden[q_indptr[1:] - 1].copy())
        logits = []
        for b in range(len(prompts)):
            last = hidden[int(q_indptr[b + 1]) - 1][None, :]
            normed = ops.rms_norm(last, self.norm, c.rms_norm_eps)
            logits.append(ops.gemm(self.lm_head, normed)[0])
        if echo:
            return logits, ops.gemm(self.lm_head, ops.rms_norm(hidden, self.norm, c.rms_norm_eps))
        return logits

    # -- decode (batch_decode.rs:17-145) --------------------------------------
    def batch_decode(self, token_ids, states):
        c = self.cfg
        positions = []
        for st in states:
            positions.append(st.seq_len)
            self._ensure(st, st.seq_len + 1)
            st.seq_len += 1
        positions = np.asarray(positions)
        meta = self._meta(states)
        batch_indices = np.arange(len(states))
        hidden = ops.embedding_batched(self.embed, token_ids)
        normed = ops.rms_norm(hidden, self.layers[0]["ln1"], c.rms_norm_eps)
        for li, L in enumerate(self.layers):
            attn = self._attn_block(li, L, normed, positions, batch_indices, meta, True)
            o = self._reduce(ops.gemm(L["o"], attn))
            hidden, normed = ops.fused_add_rms_norm(hidden, o, L["ln2"], c.rms_norm_eps)
            mlp = self._mlp(L, normed)
            nxt = self.layers[li + 1]["ln1"] if li + 1 < len(self.layers) else self.norm
            hidden, normed = ops.fused_add_rms_norm(hidden, mlp, nxt, c.rms_norm_eps)
            if self.taps is not None:
                self.taps.append(hidden.copy())
        return ops.gemm(self.lm_head, normed)                   # [bs, vocab]

    # -- greedy generate: the reference's e2e loop (tests/e2e.rs:108-221) ----
    def generate_greedy(self, prompt, max_new_tokens):
        st = KvState()
        logits = self.batch_prefill([prompt], [st])[0]
        out = [ops.argmax(logits)]
        all_logits = [logits]
        for _ in range(max_new_tokens - 1):
            lg = self.batch_decode([out[-1]], [st])[0]
            all_logits.append(lg)
            out.append(ops.argmax(lg))
        return out, all_logits


def synthetic_weights(cfg, seed=42, std=0.02, dtype_round=True, with_bits=False):
    """Seeded N(0, std) bf16 checkpoint of the given shape (used when no
    real weights are on disk; throughput is data-independent).  Every tensor is drawn in fixed
    4 Mi-element chunks, chunk j of tensor i from default_rng([seed, i, j]), so the values do not
    depend on how many host threads fill them (a Qwen3-4B-width embedding table is 389 M draws).
    with_bits=True returns (weights, bits): the uint16 bf16 images are produced in the same threaded pass
    (a separate conversion of 4 G parameters costs minutes)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    CH = 1 << 22
    pool = ThreadPoolExecutor(max_workers=max(1, min(32, os.cpu_count() or 1)))
    counter = [0]
    bits = {}

    def t(name, *shape, scale=std, mean=0.0):
        idx = counter[0]
        counter[0] += 1
        n = int(np.prod(shape))
        out = np.empty(n, dtype=F32)
        ob = np.empty(n, dtype=np.uint16) if with_bits else None

        def fill(j):
            lo, hi = j * CH, min(n, (j + 1) * CH)
            a = np.random.default_rng([seed, idx, j]).standard_normal(hi - lo, dtype=F32)
            a = a * F32(scale) + F32(mean)
            a = bf16_round(a) if dtype_round else a
            out[lo:hi] = a
            if ob is not None:
                ob[lo:hi] = bf16_round(a).view(np.uint32) >> np.uint32(16)
        list(pool.map(fill, range(-(-n // CH))))
        if ob is not None:
            bits[name] = ob.reshape(shape)
        return out.reshape(shape)

    c = cfg
    names = ["model.embed_tokens.weight", "model.norm.weight"]
    w = {"model.embed_tokens.weight": t(names[0], c.vocab_size, c.hidden_size),
         "model.norm.weight": t(names[1], c.hidden_size, scale=0.1, mean=1.0)}
    if not c.tie_word_embeddings:
        w["lm_head.weight"] = t("lm_head.weight", c.vocab_size, c.hidden_size)
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        for suffix, shape, kw in (
                ("self_attn.q_proj.weight", (c.q_dim, c.hidden_size), {}),
                ("self_attn.k_proj.weight", (c.kv_dim, c.hidden_size), {}),
                ("self_attn.v_proj.weight", (c.kv_dim, c.hidden_size), {}),
                ("self_attn.o_proj.weight", (c.hidden_size, c.q_dim), {}),
                ("self_attn.q_norm.weight", (c.head_dim,), dict(scale=0.1, mean=1.0)),
                ("self_attn.k_norm.weight", (c.head_dim,), dict(scale=0.1, mean=1.0)),
                ("mlp.gate_proj.weight", (c.intermediate_size, c.hidden_size), {}),
                ("mlp.up_proj.weight", (c.intermediate_size, c.hidden_size), {}),
                ("mlp.down_proj.weight", (c.hidden_size, c.intermediate_size), {}),
                ("input_layernorm.weight", (c.hidden_size,), dict(scale=0.1, mean=1.0)),
                ("post_attention_layernorm.weight", (c.hidden_size,), dict(scale=0.1, mean=1.0))):
            w[p + suffix] = t(p + suffix, *shape, **kw)
    pool.shutdown()
    return (w, bits) if with_bits else w
