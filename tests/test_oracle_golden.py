"""Pin the whole-model oracle against HF Transformers (the engine behind the reference's
scripts/generate_test_data.py) on the committed tiny-Qwen3 fixture.  CPU only.

Teacher-forced: the golden tokens are fed back so one near-tie cannot derail the rest.
Tolerances (stated per SURVEY.md §8c "model level"): per-step logits cosine > 0.999,
max |diff| < 0.75 (logit scale ~16, bf16 ulp 0.125), argmax equal wherever the golden
top-1 margin exceeds 0.5.
"""
import json
import os

import numpy as np
import pytest

from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle
from oracle.safetensors_io import load_safetensors

G = os.path.join(os.path.dirname(__file__), "golden")


def load_golden():
    meta = json.load(open(os.path.join(G, "qwen3_tiny_golden.json")))
    logits = np.load(os.path.join(G, "qwen3_tiny_logits.npz"))
    weights = load_safetensors(os.path.join(G, "qwen3_tiny.safetensors"))
    keys = ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
            "intermediate_size", "vocab_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings"]
    cfg = Qwen3Config(**{k: meta["config"][k] for k in keys})
    return meta, logits, weights, cfg


def check_against_golden(case, got_logits, hf_logits):
    L, H = np.stack(got_logits), hf_logits
    cos = (L * H).sum(-1) / np.linalg.norm(L, axis=-1) / np.linalg.norm(H, axis=-1)
    assert cos.min() > 0.999, (case["name"], cos.min())
    assert np.abs(L - H).max() < 0.75, (case["name"], np.abs(L - H).max())
    am, gold, margin = L.argmax(-1), np.array(case["output_tokens"]), np.array(case["top1_margin"])
    strong = margin > 0.5
    assert np.array_equal(am[strong], gold[strong]), case["name"]


@pytest.mark.parametrize("idx", range(4))
def test_oracle_matches_hf_golden(idx):
    meta, logits, weights, cfg = load_golden()
    case = meta["cases"][idx]
    m = Qwen3Oracle(cfg, weights, num_pages=32)
    st = KvState()
    got = [m.batch_prefill([case["prompt_tokens"]], [st])[0]]
    for tok in case["output_tokens"][:-1]:
        got.append(m.batch_decode([tok], [st])[0])
    check_against_golden(case, got, logits[case["name"]])


def test_oracle_batch_equals_sequential():
    """batch prefill == sequential prefill, batch decode == bs=1 decode
    (reference batch_decode.rs:505-606 `batch_matches_sequential`); exact for the oracle."""
    meta, _, weights, cfg = load_golden()
    prompts = [c["prompt_tokens"] for c in meta["cases"][:3]]
    mb = Qwen3Oracle(cfg, weights, num_pages=64)
    sb = [KvState() for _ in prompts]
    lb = mb.batch_prefill(prompts, sb)
    ms = Qwen3Oracle(cfg, weights, num_pages=64)
    ss = [KvState() for _ in prompts]
    ls = [ms.batch_prefill([p], [s])[0] for p, s in zip(prompts, ss)]
    for a, b in zip(lb, ls):
        assert np.array_equal(a, b)
    toks = [int(x.argmax()) for x in lb]
    db = mb.batch_decode(toks, sb)
    for i in range(len(prompts)):
        assert np.array_equal(db[i], ms.batch_decode([toks[i]], [ss[i]])[0])
