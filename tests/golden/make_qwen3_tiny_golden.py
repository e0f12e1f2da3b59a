#!/usr/bin/env python3
"""Generate the tiny-Qwen3 golden fixture with HF Transformers on CPU.

The reference's own ground truth for this path is HF Transformers bf16 greedy
decode (``/root/reference/scripts/generate_test_data.py``:
``AutoModelForCausalLM ... torch_dtype=bfloat16``, ``do_sample=False``,
``add_special_tokens=False``).  Real Qwen3-4B weights are not on disk and there
is no network, so this script applies the same engine and the same generate()
arguments to a tiny random Qwen3 checkpoint (head_dim 128, GQA 4:1, tied
embeddings - the Qwen3-4B topology, shrunk) and commits:

  tests/golden/qwen3_tiny.safetensors   the checkpoint (bf16)
  tests/golden/qwen3_tiny_golden.json   config, prompts (token ids), HF greedy token ids
  tests/golden/qwen3_tiny_logits.npz    HF per-step logits (float32) for tolerance checks

Run (container with transformers; NOT on the GPU box):
    python tests/golden/make_qwen3_tiny_golden.py
"""
import json
import os
import sys

import numpy as np
import torch
from safetensors.torch import save_file
from transformers import Qwen3Config, Qwen3ForCausalLM

HERE = os.path.dirname(os.path.abspath(__file__))

CFG = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=1,
           head_dim=128, intermediate_size=512, vocab_size=1024, rms_norm_eps=1e-6,
           rope_theta=1000000.0, tie_word_embeddings=True, max_position_embeddings=4096)

# (name, prompt length, max_new_tokens) - prompt lengths straddle the 16-token page size.
CASES = [("short5", 5, 24), ("page16", 16, 24), ("ragged37", 37, 24), ("long70", 70, 16)]


def main():
    torch.manual_seed(20260925)
    torch.set_num_threads(8)
    cfg = Qwen3Config(**CFG, attention_bias=False, use_sliding_window=False,
                      attn_implementation="eager")
    model = Qwen3ForCausalLM(cfg)
    # Widen the init (default std 0.02 gives near-flat logits); exact bf16 ties still occur,
    # so the tests teacher-force the golden tokens and gate argmax checks on top1_margin.
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif "embed_tokens" in name:
                p.copy_(0.25 * torch.randn_like(p))
            else:
                p.copy_(0.12 * torch.randn_like(p))
    model = model.to(torch.bfloat16).eval()

    state = {k: v.contiguous() for k, v in model.state_dict().items() if k != "lm_head.weight"}
    save_file(state, os.path.join(HERE, "qwen3_tiny.safetensors"))

    g = torch.Generator().manual_seed(7)
    cases, logits_out = [], {}
    for name, plen, max_new in CASES:
        prompt = torch.randint(0, CFG["vocab_size"], (1, plen), generator=g)
        with torch.no_grad():
            out = model.generate(prompt, max_new_tokens=max_new, do_sample=False,
                                 temperature=None, top_p=None, top_k=None,
                                 output_logits=True, return_dict_in_generate=True,
                                 pad_token_id=0)
        gen = out.sequences[0, plen:].tolist()
        lg = torch.stack([x[0].float() for x in out.logits]).numpy()     # [steps, vocab]
        srt = np.sort(lg, axis=-1)
        margins = (srt[:, -1] - srt[:, -2]).tolist()
        cases.append(dict(name=name, prompt_tokens=prompt[0].tolist(), max_new_tokens=max_new,
                          output_tokens=gen, top1_margin=margins))
        logits_out[name] = lg.astype(np.float32)
        print(name, gen[:12], "min margin %.4f" % min(margins))

    meta = dict(engine="transformers", transformers_version=__import__("transformers").__version__,
                torch_version=torch.__version__, device="cpu", dtype="bfloat16",
                generator="tests/golden/make_qwen3_tiny_golden.py", config=CFG, cases=cases)
    with open(os.path.join(HERE, "qwen3_tiny_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "qwen3_tiny_logits.npz"), **logits_out)
    print("wrote golden fixtures to", HERE)


if __name__ == "__main__":
    sys.exit(main())
