#!/usr/bin/env python3
"""Generate the tiny-Qwen3.5 (hybrid linear/full attention) golden fixture with HF Transformers on CPU.

Same truth engine as the reference (``scripts/generate_test_data.py``: HF Transformers bf16 greedy,
``do_sample=False``) applied to a tiny seeded checkpoint with the Qwen3.5-4B topology shrunk: layers
[linear, full, linear], full attention 2 q / 1 kv heads x 256 with partial RoPE 64 and an output gate, linear
attention 2 key heads / 4 value heads x 128 with conv k=4, tied embeddings.  HF runs its torch fallbacks
(torch_chunk_gated_delta_rule for prefill, torch_recurrent_gated_delta_rule for decode).  Tensor names are saved
with the reference's prefix ``model.language_model.`` (pegainfer-qwen35-4b/src/weights.rs:118) and its dtypes
(A_log and linear_attn.norm.weight as f32, weights.rs:226-241).

  tests/golden/qwen35_tiny.safetensors / qwen35_tiny_golden.json / qwen35_tiny_logits.npz

Run (container with transformers; NOT on the GPU box):
    python tests/golden/make_qwen35_tiny_golden.py
"""
import json
import os
import sys

import numpy as np
import torch
from safetensors.torch import save_file
from transformers import Qwen3_5ForCausalLM, Qwen3_5TextConfig

HERE = os.path.dirname(os.path.abspath(__file__))

CFG = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=3, vocab_size=1024, num_attention_heads=2,
           num_key_value_heads=1, head_dim=256, linear_num_key_heads=2, linear_num_value_heads=4,
           linear_key_head_dim=128, linear_value_head_dim=128, linear_conv_kernel_dim=4, rms_norm_eps=1e-6,
           rope_theta=10000000.0, partial_rotary_factor=0.25,
           layer_types=["linear_attention", "full_attention", "linear_attention"])

# prompt lengths straddle the 16-token page and the 64-token GDR chunk
CASES = [("short5", 5, 16), ("page16", 16, 16), ("chunk70", 70, 16), ("long150", 150, 12)]


def main():
    torch.manual_seed(20260926)
    torch.set_num_threads(8)
    hf = {k: v for k, v in CFG.items() if k not in ("rope_theta", "partial_rotary_factor")}
    cfg = Qwen3_5TextConfig(**hf, tie_word_embeddings=True, max_position_embeddings=4096, attention_bias=False,
                            rope_parameters=dict(rope_type="default", rope_theta=CFG["rope_theta"],
                                                 partial_rotary_factor=CFG["partial_rotary_factor"],
                                                 mrope_section=[11, 11, 10]),
                            attn_implementation="eager")
    model = Qwen3_5ForCausalLM(cfg)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("linear_attn.norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))          # plain-weight gated norm
            elif name.endswith("norm.weight"):
                p.copy_(0.1 * torch.randn_like(p))                # (1 + w) norms
            elif name.endswith("A_log"):
                p.copy_(torch.log(torch.empty_like(p).uniform_(0.5, 4.0)))
            elif name.endswith("dt_bias"):
                p.copy_(0.5 * torch.randn_like(p))
            elif "conv1d" in name:
                p.copy_(0.4 * torch.randn_like(p))
            elif "embed_tokens" in name:
                p.copy_(0.25 * torch.randn_like(p))
            else:
                p.copy_(0.12 * torch.randn_like(p))
    model = model.to(torch.bfloat16).eval()

    state = {}
    for k, v in model.state_dict().items():
        if k == "lm_head.weight":
            continue
        assert k.startswith("model.")
        name = "model.language_model." + k[len("model."):]
        if k.endswith("A_log") or k.endswith("linear_attn.norm.weight"):
            v = v.float()
        state[name] = v.contiguous()
    save_file(state, os.path.join(HERE, "qwen35_tiny.safetensors"))

    g = torch.Generator().manual_seed(11)
    cases, logits_out = [], {}
    for name, plen, max_new in CASES:
        prompt = torch.randint(0, CFG["vocab_size"], (1, plen), generator=g)
        with torch.no_grad():
            out = model.generate(prompt, max_new_tokens=max_new, do_sample=False, temperature=None, top_p=None,
                                 top_k=None, output_logits=True, return_dict_in_generate=True, pad_token_id=0)
        gen = out.sequences[0, plen:].tolist()
        lg = torch.stack([x[0].float() for x in out.logits]).numpy()
        srt = np.sort(lg, axis=-1)
        margins = (srt[:, -1] - srt[:, -2]).tolist()
        cases.append(dict(name=name, prompt_tokens=prompt[0].tolist(), max_new_tokens=max_new, output_tokens=gen,
                          top1_margin=margins))
        logits_out[name] = lg.astype(np.float32)
        print(name, gen[:12], "min margin %.4f" % min(margins), "logit scale %.2f" % np.abs(lg).max())

    meta = dict(engine="transformers", transformers_version=__import__("transformers").__version__,
                torch_version=torch.__version__, device="cpu", dtype="bfloat16",
                generator="tests/golden/make_qwen35_tiny_golden.py", config=CFG, cases=cases)
    with open(os.path.join(HERE, "qwen35_tiny_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "qwen35_tiny_logits.npz"), **logits_out)
    print("wrote golden fixtures to", HERE)


if __name__ == "__main__":
    sys.exit(main())
