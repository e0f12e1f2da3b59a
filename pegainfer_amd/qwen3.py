"""Python driver of libpegainfer_qwen3.so - the thin test/bench harness above the C++ host
runtime (which is where the hot loop lives: metadata packing, hipGraph replay, sampling).

Mirrors the surface a user of the reference's ``pegainfer_qwen3_4b::runtime`` sees
(executor.rs:541-640): ``from_config`` / ``load`` -> ``prefill`` -> ``decode`` -> ``drop_request``.
Weights: a safetensors checkpoint (HF names, bf16) or a seeded synthetic checkpoint generated on
the device.  No CPU fallback: missing library -> ImportError, device errors -> RuntimeError.
"""
import ctypes
import json
import os
import struct

import numpy as np

from . import ffi

QWEN3_4B = dict(hidden_size=2560, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8,
                head_dim=128, intermediate_size=9728, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6,
                tie_word_embeddings=True, max_position_embeddings=40960)
QWEN3_8B = dict(hidden_size=4096, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8,
                head_dim=128, intermediate_size=12288, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6,
                tie_word_embeddings=False, max_position_embeddings=40960)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Qwen3Engine:
    def __init__(self, config, num_kv_pages=1024, max_batch_size=8, enable_graph=True, decode_mode=0,
                 split_policy=1, device=0, max_positions=None):
        self.lib = ffi.host_lib()
        self.cfg = dict(config)
        c = self.cfg
        max_pos = int(max_positions or c.get("max_position_embeddings", 4096))
        self.h = self.lib.pegainfer_qwen3_create(
            device, c["hidden_size"], c["num_hidden_layers"], c["num_attention_heads"], c["num_key_value_heads"],
            c["head_dim"], c["intermediate_size"], c["vocab_size"], float(c.get("rms_norm_eps", 1e-6)),
            float(c.get("rope_theta", 1e6)), int(bool(c.get("tie_word_embeddings", True))), max_pos,
            int(num_kv_pages), int(max_batch_size), int(bool(enable_graph)), int(decode_mode), int(split_policy))
        if not self.h:
            raise RuntimeError("pegainfer_qwen3_create failed: %s" % self.lib.pegainfer_qwen3_last_error(None))
        self.vocab = c["vocab_size"]

    # ---- error plumbing ----
    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.pegainfer_qwen3_last_error(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # ---- weights ----
    def load_state(self, tensors_u16):
        """tensors_u16: {HF name: np.uint16 array of bf16 bits}."""
        for name, arr in tensors_u16.items():
            a = np.ascontiguousarray(arr, dtype=np.uint16)
            self._chk(self.lib.pegainfer_qwen3_load_tensor(self.h, name.encode(), a.ctypes.data, a.size), name)
        self._chk(self.lib.pegainfer_qwen3_finalize(self.h), "finalize")
        return self

    def load_safetensors(self, path):
        files = [path] if path.endswith(".safetensors") else sorted(
            os.path.join(path, f) for f in os.listdir(path) if f.endswith(".safetensors"))
        for fp in files:
            with open(fp, "rb") as f:
                n = struct.unpack("<Q", f.read(8))[0]
                header = json.loads(f.read(n))
                base = 8 + n
                mm = np.memmap(fp, dtype=np.uint8, mode="r")
                for name, meta in header.items():
                    if name == "__metadata__":
                        continue
                    if meta["dtype"] != "BF16":
                        raise ValueError(f"{name}: expected BF16, got {meta['dtype']}")
                    lo, hi = meta["data_offsets"]
                    a = np.ascontiguousarray(mm[base + lo: base + hi])
                    self._chk(self.lib.pegainfer_qwen3_load_tensor(self.h, name.encode(), a.ctypes.data, a.size // 2), name)
        self._chk(self.lib.pegainfer_qwen3_finalize(self.h), "finalize")
        return self

    def load_safetensors_native(self, path, tp_rank=0, tp_world=1):
        """C++ mmap loader with on-the-fly TP slicing (pegainfer_qwen3_load_safetensors)."""
        self._chk(self.lib.pegainfer_qwen3_load_safetensors(self.h, os.fsencode(path), tp_rank, tp_world),
                  "load_safetensors")
        return self

    def fill_synthetic(self, seed=42, std=0.02):
        self._chk(self.lib.pegainfer_qwen3_fill_synthetic(self.h, seed, std), "fill_synthetic")
        self._chk(self.lib.pegainfer_qwen3_finalize(self.h), "finalize")
        return self

    def export_state(self):
        """{HF tensor name: uint16 bf16 bits} of the checkpoint the engine computes with (device-generated synthetic, or
        loaded weights), copied back from the device (pegainfer_qwen3_export_tensor): a checker can run on EXACTLY these
        weights."""
        c = self.cfg
        hd = c["head_dim"]
        qd, kvd, H, I, V = (c["num_attention_heads"] * hd, c["num_key_value_heads"] * hd, c["hidden_size"],
                            c["intermediate_size"], c["vocab_size"])
        shapes = {"model.embed_tokens.weight": (V, H), "model.norm.weight": (H,)}
        if not c.get("tie_word_embeddings", True):
            shapes["lm_head.weight"] = (V, H)
        for i in range(c["num_hidden_layers"]):
            p = f"model.layers.{i}."
            shapes.update({p + "self_attn.q_proj.weight": (qd, H), p + "self_attn.k_proj.weight": (kvd, H),
                           p + "self_attn.v_proj.weight": (kvd, H), p + "self_attn.o_proj.weight": (H, qd),
                           p + "self_attn.q_norm.weight": (hd,), p + "self_attn.k_norm.weight": (hd,),
                           p + "mlp.gate_proj.weight": (I, H), p + "mlp.up_proj.weight": (I, H),
                           p + "mlp.down_proj.weight": (H, I), p + "input_layernorm.weight": (H,),
                           p + "post_attention_layernorm.weight": (H,)})
        bits = {}
        for name, shp in shapes.items():
            a = np.empty(shp, dtype=np.uint16)
            self._chk(self.lib.pegainfer_qwen3_export_tensor(self.h, name.encode(), a.ctypes.data, a.size), "export " + name)
            bits[name] = a
        return bits

    # ---- requests ----
    def new_request(self):
        r = self.lib.pegainfer_qwen3_new_request(self.h)
        if r < 0:
            self._chk(r, "new_request")
        return r

    def drop_request(self, rid):
        self._chk(self.lib.pegainfer_qwen3_drop_request(self.h, rid), "drop_request")

    def seq_len(self, rid):
        return self.lib.pegainfer_qwen3_request_seq_len(self.h, rid)

    def available_pages(self):
        return self.lib.pegainfer_qwen3_available_pages(self.h)

    # ---- forward ----
    def prefill(self, request_ids, prompts, return_logits=False, echo=False):
        """batch_prefill (prefill.rs:220-285).  echo=True (prefill.rs:196-212) also returns the logits of every prompt
        position, bf16 bits [total_tokens, vocab]: -> (tokens, last_logits, all_logits)."""
        ids = _i32(request_ids)
        lens = _i32([len(p) for p in prompts])
        toks = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint32) for p in prompts]))
        out = np.zeros(len(ids), dtype=np.int32)
        lg = np.zeros((len(ids), self.vocab), dtype=np.uint16) if (return_logits or echo) else None
        if echo:
            allg = np.zeros((int(lens.sum()), self.vocab), dtype=np.uint16)
            self._chk(self.lib.pegainfer_qwen3_prefill_echo(self.h, len(ids), ids.ctypes.data, lens.ctypes.data,
                                                            toks.ctypes.data, out.ctypes.data, lg.ctypes.data,
                                                            allg.ctypes.data), "prefill_echo")
            return out, lg, allg
        self._chk(self.lib.pegainfer_qwen3_prefill(self.h, len(ids), ids.ctypes.data, lens.ctypes.data, toks.ctypes.data,
                                                   out.ctypes.data, lg.ctypes.data if return_logits else None), "prefill")
        return (out, lg) if return_logits else out

    def unified_step(self, prefill_ids, prompts, decode_ids, decode_tokens, return_logits=False):
        """Qwen3Model::unified_step (unified_forward.rs:78-198): prompts + active decodes in one forward."""
        ids = _i32(list(prefill_ids) + list(decode_ids))
        lens = _i32([len(p) for p in prompts] + [1] * len(decode_ids))
        toks = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint32) for p in prompts] +
                                                   [np.asarray(decode_tokens, dtype=np.uint32)]))
        out = np.zeros(len(ids), dtype=np.int32)
        lg = np.zeros((len(ids), self.vocab), dtype=np.uint16) if return_logits else None
        self._chk(self.lib.pegainfer_qwen3_unified_step(self.h, len(prefill_ids), len(decode_ids), ids.ctypes.data,
                                                        lens.ctypes.data, toks.ctypes.data, out.ctypes.data,
                                                        lg.ctypes.data if return_logits else None), "unified_step")
        np_ = len(prefill_ids)
        return ((out[:np_], out[np_:]), (lg[:np_], lg[np_:])) if return_logits else (out[:np_], out[np_:])

    def decode(self, request_ids, token_ids, return_logits=False):
        ids = _i32(request_ids)
        toks = np.ascontiguousarray(token_ids, dtype=np.uint32)
        out = np.zeros(len(ids), dtype=np.int32)
        lg = np.zeros((len(ids), self.vocab), dtype=np.uint16) if return_logits else None
        self._chk(self.lib.pegainfer_qwen3_decode(self.h, len(ids), ids.ctypes.data, toks.ctypes.data, out.ctypes.data,
                                                  lg.ctypes.data if return_logits else None), "decode")
        return (out, lg) if return_logits else out

    def decode_greedy_chain(self, request_ids, first_tokens, n_steps):
        """n_steps greedy decode steps enqueued back to back, one host synchronisation (pegainfer_qwen3_decode_greedy_chain).
        -> int32 [n_steps, n_requests]"""
        ids = _i32(request_ids)
        toks = np.ascontiguousarray(first_tokens, dtype=np.uint32)
        out = np.zeros((int(n_steps), len(ids)), dtype=np.int32)
        self._chk(self.lib.pegainfer_qwen3_decode_greedy_chain(self.h, len(ids), ids.ctypes.data, toks.ctypes.data,
                                                               int(n_steps), out.ctypes.data), "decode_greedy_chain")
        return out

    def sample(self, column, temperature, top_k, top_p, random_val):
        out = ctypes.c_int32(0)
        self._chk(self.lib.pegainfer_qwen3_sample(self.h, column, temperature, top_k, top_p, random_val,
                                                  ctypes.addressof(out)), "sample")
        return out.value

    def last_step_ms(self):
        return self.lib.pegainfer_qwen3_last_step_ms(self.h)

    # ---- per-layer hidden tap (accuracy-parity-playbook.md:15-24) ----
    def debug_hidden_enable(self, on=True):
        self._chk(self.lib.pegainfer_qwen3_debug_hidden_enable(self.h, int(bool(on))), "debug_hidden_enable")

    def debug_hidden(self, max_rows=64):
        """bf16 bits [layers, rows, hidden] of the last step: the residual stream leaving every layer (decode: every
        column; prefill: each request's last prompt position)."""
        H, L = self.cfg["hidden_size"], self.cfg["num_hidden_layers"]
        out = []
        for li in range(L):
            buf = np.zeros((max_rows, H), dtype=np.uint16)
            n = self.lib.pegainfer_qwen3_debug_hidden(self.h, li, buf.ctypes.data, max_rows)
            if n < 0:
                self._chk(n, "debug_hidden")
            out.append(buf[:n])
        return np.stack(out)

    def bench_gemv(self, which, iters=200, bs=1):
        return self.lib.pegainfer_qwen3_bench_gemv(self.h, which, iters, bs)

    def last_attention_path(self):
        return self.lib.pegainfer_qwen3_last_attention_path(self.h)

    def weight_bytes(self):
        return self.lib.pegainfer_qwen3_weight_bytes(self.h)

    def generate_greedy(self, prompt, max_new_tokens):
        """The reference's e2e loop (tests/e2e.rs:108-221): prefill then decode, greedy."""
        rid = self.new_request()
        try:
            out = [int(self.prefill([rid], [prompt])[0])]
            for _ in range(max_new_tokens - 1):
                out.append(int(self.decode([rid], [out[-1]])[0]))
        finally:
            self.drop_request(rid)
        return out

    def close(self):
        if self.h:
            self.lib.pegainfer_qwen3_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
