"""Pin the Qwen3.5 hybrid oracle (oracle/qwen35_ref.py: chunk-wise gated-delta-rule prefill, recurrent decode,
HD256 gated attention with partial RoPE) against HF Transformers on the committed tiny fixture.  CPU only.

Teacher-forced like the Qwen3 test.  Tolerances (SURVEY.md §8c "model level"): per-step logits cosine > 0.999,
max |diff| < 0.75 on a logit scale of ~18 (bf16 ulp 0.125), argmax equal wherever the golden top-1 margin > 0.5.
HF rounds differently inside the linear-attention block (fp32 chunk math, l2norm eps 1e-6) - the tolerance is the
same one the Qwen3 oracle is held to.
"""
import json
import os

import numpy as np
import pytest

from oracle.qwen35_ref import Qwen35Config, Qwen35Oracle
from oracle.safetensors_io import load_safetensors

G = os.path.join(os.path.dirname(__file__), "golden")


def load_golden35():
    meta = json.load(open(os.path.join(G, "qwen35_tiny_golden.json")))
    logits = np.load(os.path.join(G, "qwen35_tiny_logits.npz"))
    weights = load_safetensors(os.path.join(G, "qwen35_tiny.safetensors"))
    return meta, logits, weights, Qwen35Config(**meta["config"])


def check35(case, got_logits, hf_logits):
    L, H = np.stack(got_logits), hf_logits
    cos = (L * H).sum(-1) / np.linalg.norm(L, axis=-1) / np.linalg.norm(H, axis=-1)
    assert cos.min() > 0.999, (case["name"], cos.min())
    assert np.abs(L - H).max() < 0.75, (case["name"], np.abs(L - H).max())
    am, gold, margin = L.argmax(-1), np.array(case["output_tokens"]), np.array(case["top1_margin"])
    strong = margin > 0.5
    assert np.array_equal(am[strong], gold[strong]), case["name"]


@pytest.mark.parametrize("idx", range(4))
def test_qwen35_oracle_matches_hf_golden(idx):
    meta, logits, weights, cfg = load_golden35()
    case = meta["cases"][idx]
    m = Qwen35Oracle(cfg, weights, num_pages=32)
    st = m.new_request()
    got = [m.prefill(case["prompt_tokens"], st)]
    for tok in case["output_tokens"][:-1]:
        got.append(m.batch_decode([tok], [st])[0])
    check35(case, got, logits[case["name"]])


def test_qwen35_oracle_batch_decode_equals_single_and_prefill_handoff():
    """Batched decode == per-request decode (exact for the oracle), and a prompt prefilled in two pieces (the
    recurrent + conv state handed from one prefill call to the next, prefill.rs:52-54,84-87) stays within the
    model tolerance of the one-shot prefill."""
    meta, _, weights, cfg = load_golden35()
    prompts = [c["prompt_tokens"] for c in meta["cases"][:3]]
    mb = Qwen35Oracle(cfg, weights, num_pages=64)
    sb = [mb.new_request() for _ in prompts]
    toks = [int(mb.prefill(p, s).argmax()) for p, s in zip(prompts, sb)]
    ms = Qwen35Oracle(cfg, weights, num_pages=64)
    ss = [ms.new_request() for _ in prompts]
    for p, s in zip(prompts, ss):
        ms.prefill(p, s)
    db = mb.batch_decode(toks, sb)
    for i in range(len(prompts)):
        assert np.array_equal(db[i], ms.batch_decode([toks[i]], [ss[i]])[0])
    p = meta["cases"][2]["prompt_tokens"]
    m1, m2 = Qwen35Oracle(cfg, weights, num_pages=32), Qwen35Oracle(cfg, weights, num_pages=32)
    s1, s2 = m1.new_request(), m2.new_request()
    one = m1.prefill(p, s1)
    m2.prefill(p[:41], s2)
    two = m2.prefill(p[41:], s2)
    cos = float((one * two).sum() / np.linalg.norm(one) / np.linalg.norm(two))
    assert cos > 0.999 and np.abs(one - two).max() < 0.75
