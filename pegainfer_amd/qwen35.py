"""Python driver of the Qwen3.5 hybrid host runtime (libpegainfer_qwen3.so, include/pegainfer_qwen35.h).

Surface of the reference's ``pegainfer_qwen35_4b`` model crate for this path: load -> prefill (one request per
call, prefill.rs:21) -> batched decode (batch_decode.rs:113) -> drop.  Weights: a safetensors checkpoint with the
reference's tensor names (``model.language_model.*``; A_log and linear_attn.norm.weight f32) or a seeded synthetic
checkpoint generated on the device.  No CPU fallback.
"""
import json
import struct

import numpy as np

from . import ffi

QWEN35_4B = dict(hidden_size=2560, intermediate_size=9216, num_hidden_layers=32, vocab_size=248320,
                 num_attention_heads=16, num_key_value_heads=4, head_dim=256, linear_num_key_heads=16,
                 linear_num_value_heads=32, linear_key_head_dim=128, linear_value_head_dim=128,
                 linear_conv_kernel_dim=4, rms_norm_eps=1e-6, rope_theta=1e7, partial_rotary_factor=0.25,
                 layer_types=["full_attention" if (i + 1) % 4 == 0 else "linear_attention" for i in range(32)])


class Qwen35Engine:
    def __init__(self, config, num_kv_pages=1024, max_batch_size=8, enable_graph=True, device=0, max_positions=4096,
                 split_policy=1):
        self.lib = ffi.host_lib()
        self.cfg = c = dict(config)
        if c.get("linear_key_head_dim", 128) != 128 or c.get("linear_value_head_dim", 128) != 128:
            raise ValueError("linear attention head dims are fixed at 128 (chunk-wise kernels)")
        is_full = np.ascontiguousarray([1 if t == "full_attention" else 0 for t in c["layer_types"]], dtype=np.int32)
        assert len(is_full) == c["num_hidden_layers"]
        rotary = int(c["head_dim"] * c.get("partial_rotary_factor", 0.25))
        self.h = self.lib.pegainfer_qwen35_create(
            device, c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"], c["vocab_size"],
            c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"], c["linear_num_key_heads"],
            c["linear_num_value_heads"], c.get("linear_conv_kernel_dim", 4), float(c.get("rms_norm_eps", 1e-6)),
            float(c.get("rope_theta", 1e7)), rotary, is_full.ctypes.data, int(max_positions), int(num_kv_pages),
            int(max_batch_size), int(bool(enable_graph)), int(split_policy))
        if not self.h:
            raise RuntimeError("pegainfer_qwen35_create failed")
        self.vocab = c["vocab_size"]

    # ---- per-layer hidden tap (accuracy-parity-playbook.md:15-24) ----
    def debug_hidden_enable(self, on=True):
        self._chk(self.lib.pegainfer_qwen35_debug_hidden_enable(self.h, int(bool(on))), "debug_hidden_enable")

    def debug_hidden(self, max_rows=8):
        """bf16 bits [layers, rows, hidden] of the last step (decode: every column; prefill: the last prompt position)"""
        H, L = self.cfg["hidden_size"], self.cfg["num_hidden_layers"]
        out = []
        for li in range(L):
            buf = np.zeros((max_rows, H), dtype=np.uint16)
            n = self.lib.pegainfer_qwen35_debug_hidden(self.h, li, buf.ctypes.data, max_rows)
            if n < 0:
                self._chk(n, "debug_hidden")
            out.append(buf[:n])
        return np.stack(out)

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.pegainfer_qwen35_last_error(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def load_safetensors(self, path):
        with open(path, "rb") as f:
            n = struct.unpack("<Q", f.read(8))[0]
            header = json.loads(f.read(n))
            base = 8 + n
            for name, info in header.items():
                if name == "__metadata__":
                    continue
                lo, hi = info["data_offsets"]
                f.seek(base + lo)
                raw = f.read(hi - lo)
                if info["dtype"] == "BF16":
                    a = np.frombuffer(raw, dtype=np.uint16)
                    is_f32 = 0
                elif info["dtype"] == "F32":
                    a = np.frombuffer(raw, dtype=np.float32)
                    is_f32 = 1
                else:
                    raise ValueError(f"{name}: unsupported dtype {info['dtype']}")
                a = np.ascontiguousarray(a)
                self._chk(self.lib.pegainfer_qwen35_load_tensor(self.h, name.encode(), a.ctypes.data, a.size, is_f32),
                          name)
        self._chk(self.lib.pegainfer_qwen35_finalize(self.h), "finalize")
        return self

    def load_state(self, tensors):
        """tensors: {reference tensor name: np.uint16 array of bf16 bits | np.float32 array (A_log, linear_attn.norm.weight)}"""
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr)
            if a.dtype not in (np.uint16, np.float32):
                raise ValueError(f"{name}: expected uint16 bf16 bits or float32, got {a.dtype}")
            self._chk(self.lib.pegainfer_qwen35_load_tensor(self.h, name.encode(), a.ctypes.data, a.size,
                                                            1 if a.dtype == np.float32 else 0), name)
        self._chk(self.lib.pegainfer_qwen35_finalize(self.h), "finalize")
        return self

    def load_safetensors_native(self, path):
        import os
        self._chk(self.lib.pegainfer_qwen35_load_safetensors(self.h, os.fsencode(path)), "load_safetensors")
        return self

    def export_state(self):
        """{reference tensor name: uint16 bf16 bits | float32 (A_log, linear_attn.norm.weight)} of the checkpoint the
        engine computes with, copied back from the device (pegainfer_qwen35_export_tensor)."""
        c = self.cfg
        H, I, V, D = c["hidden_size"], c["intermediate_size"], c["vocab_size"], c["head_dim"]
        qd, kvd = c["num_attention_heads"] * D, c["num_key_value_heads"] * D
        kh, vh, K = c["linear_num_key_heads"], c["linear_num_value_heads"], c.get("linear_conv_kernel_dim", 4)
        C, Z = 2 * kh * 128 + vh * 128, vh * 128
        shapes = {"embed_tokens.weight": (V, H), "norm.weight": (H,)}
        for i, kind in enumerate(c["layer_types"]):
            p = f"layers.{i}."
            shapes.update({p + "input_layernorm.weight": (H,), p + "post_attention_layernorm.weight": (H,),
                           p + "mlp.gate_proj.weight": (I, H), p + "mlp.up_proj.weight": (I, H),
                           p + "mlp.down_proj.weight": (H, I)})
            if kind == "full_attention":
                shapes.update({p + "self_attn.q_proj.weight": (2 * qd, H), p + "self_attn.k_proj.weight": (kvd, H),
                               p + "self_attn.v_proj.weight": (kvd, H), p + "self_attn.o_proj.weight": (H, qd),
                               p + "self_attn.q_norm.weight": (D,), p + "self_attn.k_norm.weight": (D,)})
            else:
                shapes.update({p + "linear_attn.in_proj_qkv.weight": (C, H), p + "linear_attn.in_proj_z.weight": (Z, H),
                               p + "linear_attn.in_proj_b.weight": (vh, H), p + "linear_attn.in_proj_a.weight": (vh, H),
                               p + "linear_attn.conv1d.weight": (C, 1, K), p + "linear_attn.dt_bias": (vh,),
                               p + "linear_attn.A_log": (vh,), p + "linear_attn.norm.weight": (128,),
                               p + "linear_attn.out_proj.weight": (H, Z)})
        out = {}
        for name, shp in shapes.items():
            f32 = name.endswith("A_log") or name.endswith("linear_attn.norm.weight")
            a = np.empty(shp, dtype=np.float32 if f32 else np.uint16)
            full = "model.language_model." + name
            self._chk(self.lib.pegainfer_qwen35_export_tensor(self.h, full.encode(), a.ctypes.data, a.size, 1 if f32 else 0),
                      "export " + name)
            out[full] = a
        return out

    def fill_synthetic(self, seed=42, std=0.02):
        self._chk(self.lib.pegainfer_qwen35_fill_synthetic(self.h, seed, std), "fill_synthetic")
        self._chk(self.lib.pegainfer_qwen35_finalize(self.h), "finalize")
        return self

    def new_request(self):
        r = self.lib.pegainfer_qwen35_new_request(self.h)
        if r < 0:
            self._chk(r, "new_request")
        return r

    def drop_request(self, rid):
        self._chk(self.lib.pegainfer_qwen35_drop_request(self.h, rid), "drop_request")

    def seq_len(self, rid):
        return self.lib.pegainfer_qwen35_request_seq_len(self.h, rid)

    def prefill(self, rid, tokens, want_logits=False):
        toks = np.ascontiguousarray(tokens, dtype=np.uint32)
        out = np.zeros(1, np.int32)
        lg = np.zeros(self.vocab, np.uint16) if want_logits else None
        self._chk(self.lib.pegainfer_qwen35_prefill(self.h, rid, toks.size, toks.ctypes.data, out.ctypes.data,
                                                    lg.ctypes.data if want_logits else None), "prefill")
        return (int(out[0]), _bf16_to_f32(lg)) if want_logits else int(out[0])

    def decode(self, rids, tokens, want_logits=False):
        ids = np.ascontiguousarray(rids, dtype=np.int32)
        toks = np.ascontiguousarray(tokens, dtype=np.uint32)
        out = np.zeros(ids.size, np.int32)
        lg = np.zeros((ids.size, self.vocab), np.uint16) if want_logits else None
        self._chk(self.lib.pegainfer_qwen35_decode(self.h, ids.size, ids.ctypes.data, toks.ctypes.data, out.ctypes.data,
                                                   lg.ctypes.data if want_logits else None), "decode")
        return (out, _bf16_to_f32(lg)) if want_logits else out

    def decode_greedy_chain(self, rids, first_tokens, n_steps):
        """n_steps greedy decode steps enqueued back to back, one host synchronisation -> int32 [n_steps, n_requests]"""
        ids = np.ascontiguousarray(rids, dtype=np.int32)
        toks = np.ascontiguousarray(first_tokens, dtype=np.uint32)
        out = np.zeros((int(n_steps), ids.size), np.int32)
        self._chk(self.lib.pegainfer_qwen35_decode_greedy_chain(self.h, ids.size, ids.ctypes.data, toks.ctypes.data, int(n_steps),
                                                                out.ctypes.data), "decode_greedy_chain")
        return out

    def last_step_ms(self):
        return float(self.lib.pegainfer_qwen35_last_step_ms(self.h))

    def bench_gemv(self, which=0, iters=320):
        return float(self.lib.pegainfer_qwen35_bench_gemv(self.h, which, iters))

    def weight_bytes(self):
        return int(self.lib.pegainfer_qwen35_weight_bytes(self.h))

    def close(self):
        if self.h:
            self.lib.pegainfer_qwen35_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _bf16_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)
