"""Whole-model CPU restatement of the reference's Qwen3.5 hybrid DAG (TEST INFRASTRUCTURE ONLY).

Follows the op order of
  * prefill : pegainfer-qwen35-4b/src/prefill.rs:21-449   (prefill_forward, prefill_layer,
              prefill_full_attention, prefill_linear_attention) - one request per call, like the reference
  * decode  : pegainfer-qwen35-4b/src/batch_decode.rs:43-365 (batch_decode_kernels_graph,
              batch_decode_full_attention, batch_decode_linear_attention_slots)
with the weight names of weights.rs:102-296 (prefix ``model.language_model``), the paged KV pool over the
full-attention layers only (weights.rs:318-345) and the per-request recurrent state of recurrent_state.rs
(conv_state bf16 [C, K-1], state f32 [vh, K, V] per linear layer).
"""
import numpy as np

from . import ops
from .bf16 import bf16_round

F32 = np.float32
WP = "model.language_model"


class Qwen35Config:
    def __init__(self, hidden_size, intermediate_size, num_hidden_layers, vocab_size, num_attention_heads,
                 num_key_value_heads, head_dim, linear_num_key_heads, linear_num_value_heads,
                 linear_key_head_dim=128, linear_value_head_dim=128, linear_conv_kernel_dim=4, rms_norm_eps=1e-6,
                 rope_theta=1e7, partial_rotary_factor=0.25, layer_types=None, full_attention_interval=4):
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.vocab_size = vocab_size
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.head_dim = head_dim
        self.linear_num_key_heads = linear_num_key_heads
        self.linear_num_value_heads = linear_num_value_heads
        self.linear_key_head_dim = linear_key_head_dim
        self.linear_value_head_dim = linear_value_head_dim
        self.linear_conv_kernel_dim = linear_conv_kernel_dim
        self.rms_norm_eps = rms_norm_eps
        self.rope_theta = rope_theta
        self.rotary_dim = int(head_dim * partial_rotary_factor)      # config.rs:101
        if layer_types is None:                                      # full attention at 3, 7, 11, ...
            layer_types = ["full_attention" if (i + 1) % full_attention_interval == 0 else "linear_attention"
                           for i in range(num_hidden_layers)]
        self.layer_types = list(layer_types)

    @property
    def q_dim(self):
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self):
        return self.num_key_value_heads * self.head_dim

    @property
    def lin_qkv_dim(self):                                           # config.rs:141-147
        return 2 * self.linear_num_key_heads * self.linear_key_head_dim + self.lin_z_dim

    @property
    def lin_z_dim(self):
        return self.linear_num_value_heads * self.linear_value_head_dim

    @property
    def num_full_layers(self):
        return sum(t == "full_attention" for t in self.layer_types)

    @staticmethod
    def qwen35_4b():
        """docs/models/qwen35/optimization.md:56-76."""
        return Qwen35Config(2560, 9216, 32, 248320, 16, 4, 256, 16, 32)


class RequestState:
    """KvState (pages + seq_len) and RecurrentState (recurrent_state.rs:1-55) of one request."""

    def __init__(self, cfg):
        self.pages = []
        self.seq_len = 0
        C, K = cfg.lin_qkv_dim, cfg.linear_conv_kernel_dim
        n_lin = cfg.num_hidden_layers - cfg.num_full_layers
        self.conv = [np.zeros((C, K - 1), F32) for _ in range(n_lin)]
        self.state = [np.zeros((cfg.linear_num_value_heads, cfg.linear_key_head_dim, cfg.linear_value_head_dim), F32)
                      for _ in range(n_lin)]


class Qwen35Oracle:
    PAGE_SIZE = 16

    def __init__(self, cfg, weights, num_pages=64, rope_positions=4096):
        self.cfg = c = cfg
        self.taps = None   # a list: every layer appends its output rows (prefill: the last row; decode: all rows)
        self.w = weights
        g = lambda name: weights[f"{WP}.{name}"]
        self.embed = g("embed_tokens.weight")
        self.norm = g("norm.weight")
        self.layers = []
        for i, kind in enumerate(c.layer_types):
            p = f"layers.{i}."
            L = dict(kind=kind, ln1=g(p + "input_layernorm.weight"), ln2=g(p + "post_attention_layernorm.weight"),
                     gate=g(p + "mlp.gate_proj.weight"), up=g(p + "mlp.up_proj.weight"),
                     down=g(p + "mlp.down_proj.weight"))
            if kind == "full_attention":
                a = p + "self_attn."
                L.update(q=g(a + "q_proj.weight"), k=g(a + "k_proj.weight"), v=g(a + "v_proj.weight"),
                         o=g(a + "o_proj.weight"), q_norm=g(a + "q_norm.weight"), k_norm=g(a + "k_norm.weight"))
            else:
                a = p + "linear_attn."
                L.update(qkv=g(a + "in_proj_qkv.weight"), z=g(a + "in_proj_z.weight"), b=g(a + "in_proj_b.weight"),
                         a=g(a + "in_proj_a.weight"),
                         conv=g(a + "conv1d.weight").reshape(c.lin_qkv_dim, c.linear_conv_kernel_dim),
                         dt_bias=g(a + "dt_bias"), A_log=g(a + "A_log").astype(F32),
                         norm_w=g(a + "norm.weight").astype(F32), out=g(a + "out_proj.weight"))
            self.layers.append(L)
        self.cos, self.sin = ops.precompute_rope(c.rotary_dim, rope_positions, c.rope_theta)   # weights.rs:296-297
        self.layout = ops.PagedKvLayout(c.num_full_layers, c.num_key_value_heads, c.head_dim, self.PAGE_SIZE)
        self.kv = np.zeros(num_pages * self.layout.page_stride, dtype=F32)
        self.free_pages = list(range(1, num_pages))
        self.sm_scale = F32(1.0) / np.sqrt(F32(c.head_dim))

    def new_request(self):
        return RequestState(self.cfg)

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

    # ---- shared pieces ----
    def _full_attention(self, L, full_idx, normed, positions, batch_indices, meta, decode, q_indptr=None):
        c = self.cfg
        T = normed.shape[0]
        q_full = ops.gemm(L["q"], normed)                                       # [T, Hq*512], per head [q | gate]
        k = ops.gemm(L["k"], normed)
        v = ops.gemm(L["v"], normed)
        q_raw = q_full.reshape(T, c.num_attention_heads, 2, c.head_dim)[:, :, 0]
        q = ops.hd256_norm_partial_rope(q_raw, L["q_norm"], self.cos, self.sin, positions, c.rotary_dim,
                                        c.rms_norm_eps).reshape(T, -1)
        kk = ops.hd256_norm_partial_rope(k.reshape(T, c.num_key_value_heads, c.head_dim), L["k_norm"], self.cos,
                                         self.sin, positions, c.rotary_dim, c.rms_norm_eps).reshape(T, -1)
        pages, indptr, last = meta
        ops.paged_kv_scatter(self.kv, self.layout, full_idx, pages, indptr, kk, v, batch_indices, positions)
        if decode:
            o = ops.paged_attention_decode(q, self.kv, self.layout, full_idx, pages, indptr, last,
                                           c.num_attention_heads, self.sm_scale)
        else:
            o = ops.batch_prefill_paged(q, self.kv, self.layout, full_idx, pages, indptr, last, q_indptr,
                                        c.num_attention_heads, self.sm_scale)
        o = ops.attention_gate_hd256(q_full, o, c.num_attention_heads)
        return ops.gemm(L["o"], o)

    def _mlp(self, L, normed):
        act = ops.silu_mul(ops.gemm(L["gate"], normed), ops.gemm(L["up"], normed))
        return ops.gemm(L["down"], act)

    # ---- prefill: one request (prefill.rs:21-120) ----
    def prefill(self, tokens, st):
        c = self.cfg
        T = len(tokens)
        base = st.seq_len
        hidden = ops.embedding_batched(self.embed, tokens)
        self._ensure(st, base + T)
        st.seq_len += T
        meta = self._meta([st])
        positions = np.arange(base, base + T)
        batch_indices = np.zeros(T, np.int64)
        q_indptr = np.asarray([0, T])
        lin = full = 0
        for L in self.layers:
            normed = ops.rms_norm(hidden, L["ln1"], c.rms_norm_eps, offset=True)
            if L["kind"] == "full_attention":
                attn = self._full_attention(L, full, normed, positions, batch_indices, meta, False, q_indptr)
                full += 1
            else:
                qkv = ops.gemm(L["qkv"], normed)
                z = ops.gemm(L["z"], normed)
                b = ops.gemm(L["b"], normed)
                a = ops.gemm(L["a"], normed)
                qkv_conv, st.conv[lin] = ops.conv1d_prefill(qkv, L["conv"], st.conv[lin])
                gdr, st.state[lin] = ops.gated_delta_rule_prefill_chunkwise(
                    qkv_conv, b, a, L["dt_bias"], L["A_log"], st.state[lin], c.linear_num_key_heads,
                    c.linear_num_value_heads, c.linear_key_head_dim, c.linear_value_head_dim)
                gated = ops.rms_norm_gated(gdr, L["norm_w"], z, c.linear_value_head_dim, c.rms_norm_eps)
                attn = ops.gemm(L["out"], gated)
                lin += 1
            mid = ops.add(hidden, attn)
            normed = ops.rms_norm(mid, L["ln2"], c.rms_norm_eps, offset=True)
            hidden = ops.add(mid, self._mlp(L, normed))
            if self.taps is not None:
                self.taps.append(hidden[T - 1:T].copy())
        last = ops.rms_norm(hidden[T - 1][None, :], self.norm, c.rms_norm_eps, offset=True)
        return ops.gemm(self.embed, last)[0]

    # ---- batched decode (batch_decode.rs:198-365) ----
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
        lin = full = 0
        for L in self.layers:
            normed = ops.rms_norm(hidden, L["ln1"], c.rms_norm_eps, offset=True)
            if L["kind"] == "full_attention":
                attn = self._full_attention(L, full, normed, positions, batch_indices, meta, True)
                full += 1
            else:
                qkv = ops.gemm(L["qkv"], normed)
                z = ops.gemm(L["z"], normed)
                b = ops.gemm(L["b"], normed)
                a = ops.gemm(L["a"], normed)
                rows = []
                for i, st in enumerate(states):                                  # per slot, batch_decode.rs:315-345
                    conv, st.conv[lin] = ops.conv1d_prefill(qkv[i][None, :], L["conv"], st.conv[lin])
                    o, st.state[lin] = ops.gated_delta_rule_decode(
                        conv[0], b[i], a[i], L["dt_bias"], L["A_log"], st.state[lin], c.linear_num_key_heads,
                        c.linear_num_value_heads, c.linear_key_head_dim, c.linear_value_head_dim)
                    rows.append(o)
                gated = ops.rms_norm_gated(np.stack(rows), L["norm_w"], z, c.linear_value_head_dim, c.rms_norm_eps)
                attn = ops.gemm(L["out"], gated)
                lin += 1
            mid = ops.add(hidden, attn)
            normed = ops.rms_norm(mid, L["ln2"], c.rms_norm_eps, offset=True)
            hidden = ops.add(mid, self._mlp(L, normed))
            if self.taps is not None:
                self.taps.append(hidden.copy())
        normed = ops.rms_norm(hidden, self.norm, c.rms_norm_eps, offset=True)
        return ops.gemm(self.embed, normed)

    def generate_greedy(self, prompt, max_new_tokens):
        st = self.new_request()
        logits = self.prefill(prompt, st)
        out, all_logits = [ops.argmax(logits)], [logits]
        for _ in range(max_new_tokens - 1):
            lg = self.batch_decode([out[-1]], [st])[0]
            all_logits.append(lg)
            out.append(ops.argmax(lg))
        return out, all_logits


def synthetic_weights(cfg, seed=42, std=0.02):
    """Seeded bf16 checkpoint with the reference's tensor names and dtypes (A_log and the gated-norm weight are
    f32, weights.rs:226-241)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(seed)
    c = cfg
    CH = 1 << 22
    pool = ThreadPoolExecutor(max_workers=max(1, min(32, os.cpu_count() or 1)))
    counter = [0]

    def t(*shape, scale=std, mean=0.0):
        """chunk j of tensor i from default_rng([seed, i, j]): values independent of the host thread count (the
        248 320 x 2560 embedding is 636 M draws - a minute on one thread)"""
        idx = counter[0]
        counter[0] += 1
        n = int(np.prod(shape))
        out = np.empty(n, dtype=F32)

        def fill(j):
            lo, hi = j * CH, min(n, (j + 1) * CH)
            a = np.random.default_rng([seed, idx, j]).standard_normal(hi - lo, dtype=F32)
            out[lo:hi] = bf16_round(a * F32(scale) + F32(mean))
        list(pool.map(fill, range(-(-n // CH))))
        return out.reshape(shape)

    w = {f"{WP}.embed_tokens.weight": t(c.vocab_size, c.hidden_size), f"{WP}.norm.weight": t(c.hidden_size, scale=0.1)}
    for i, kind in enumerate(c.layer_types):
        p = f"{WP}.layers.{i}."
        w[p + "input_layernorm.weight"] = t(c.hidden_size, scale=0.1)
        w[p + "post_attention_layernorm.weight"] = t(c.hidden_size, scale=0.1)
        w[p + "mlp.gate_proj.weight"] = t(c.intermediate_size, c.hidden_size)
        w[p + "mlp.up_proj.weight"] = t(c.intermediate_size, c.hidden_size)
        w[p + "mlp.down_proj.weight"] = t(c.hidden_size, c.intermediate_size)
        if kind == "full_attention":
            a = p + "self_attn."
            w[a + "q_proj.weight"] = t(2 * c.q_dim, c.hidden_size)
            w[a + "k_proj.weight"] = t(c.kv_dim, c.hidden_size)
            w[a + "v_proj.weight"] = t(c.kv_dim, c.hidden_size)
            w[a + "o_proj.weight"] = t(c.hidden_size, c.q_dim)
            w[a + "q_norm.weight"] = t(c.head_dim, scale=0.1)
            w[a + "k_norm.weight"] = t(c.head_dim, scale=0.1)
        else:
            a = p + "linear_attn."
            w[a + "in_proj_qkv.weight"] = t(c.lin_qkv_dim, c.hidden_size)
            w[a + "in_proj_z.weight"] = t(c.lin_z_dim, c.hidden_size)
            w[a + "in_proj_b.weight"] = t(c.linear_num_value_heads, c.hidden_size)
            w[a + "in_proj_a.weight"] = t(c.linear_num_value_heads, c.hidden_size)
            w[a + "conv1d.weight"] = t(c.lin_qkv_dim, 1, c.linear_conv_kernel_dim, scale=0.3)
            w[a + "dt_bias"] = t(c.linear_num_value_heads, scale=0.5)
            w[a + "A_log"] = (rng.standard_normal(c.linear_num_value_heads) * 0.5).astype(F32)
            w[a + "norm.weight"] = (1.0 + 0.1 * rng.standard_normal(c.linear_value_head_dim)).astype(F32)
            w[a + "out_proj.weight"] = t(c.hidden_size, c.lin_z_dim)
    return w
