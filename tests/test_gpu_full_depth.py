"""GPU parity at FULL DEPTH on the benchmarked configuration.

Every number bench.py reports is a 36-layer Qwen3-4B-shaped model (BASELINE.json configs[1]); tests/test_gpu_real_dims.py
stops at two layers.  Here the whole model is checked, on the reference's own ``decode_heavy`` prompt (1024 tokens,
``100 + i % 1000``, bench_serving.rs:37-43) followed by decode steps, graph on, ``decode_mode`` 0 and 1:

  * vs the ORACLE (oracle/qwen3_ref.py) on the same host-generated seeded checkpoint: cosine and max |dlogit| of every
    step's logits row, greedy token wherever the oracle's top-1 margin exceeds the logit bar;
  * vs HF TRANSFORMERS - the engine behind the reference's golden texts (scripts/generate_test_data.py) - through the
    committed fixture tests/golden/qwen3_4b_depth36_hf.{json,npz} (made by tests/golden/make_qwen3_4b_depth_golden.py
    from the same seed): greedy tokens of the engine's OWN free-running generation against HF's, first-difference step
    reported the way docs/accuracy-parity-playbook.md:15-24 asks, logits on a fixed 4096-index subset;
  * `batch_matches_sequential`'s idea (batch_decode.rs:505-606) at depth: decode_mode 1 == decode_mode 0 in every bit.

The real north-star gate - token ids vs test_data/Qwen3-4B.json - needs the real checkpoint (tests/test_e2e_golden.py,
skipped without PEGAINFER_TEST_MODEL_PATH); this is its reachable proxy.  Same for BASELINE.json configs[2]: an 8-layer
Qwen3-8B-shaped model with top-k / top-p sampling on top of oracle-checked logits.

The oracle GEMM accumulates in fp32 here (plain sgemm, what cuBLAS COMPUTE_32F does); both sides round every
activation to bf16 at the same points and differ in summation order only.  Measured numbers are also written to
gpurun_out/full_depth_parity.json when that directory exists.
"""
import json
import os

import numpy as np
import pytest

from oracle import ops as O
from oracle.bf16 import bf16_from_bits
from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle, synthetic_weights

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "qwen3_4b_depth36_hf")
N_STEPS = 8                      # decode steps checked (the fixture holds 16)

# Bars.  Two-layer bar of tests/test_gpu_real_dims.py: cosine > 0.9998, max |dlogit| <= 2 % of the largest |logit|.
# 36 layers accumulate 18 x as many independent rounding flips (random-walk growth ~ sqrt(18) = 4.2 on the relative
# error, ~18 on 1 - cos); the bars below are the measured values of the first GPU run with ~2 x margin and are what a
# regression has to stay inside (the oracle itself sits at cosine 0.9977 / 7 % from HF on this checkpoint).  HF rounds at different points than the reference (norm output rounded before the
# weight product, separate RoPE roundings), so its bar is the oracle-vs-HF distance recorded in the fixture, x 1.5.
COS_MIN_ORACLE, REL_MAX_ORACLE = 0.998, 0.06


def _report(name, payload):
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(HERE)
    d = os.path.join(root, "gpurun_out")
    if not os.path.isdir(d):
        return
    fp = os.path.join(d, "full_depth_parity.json")
    cur = json.load(open(fp)) if os.path.exists(fp) else {}
    cur[name] = payload
    json.dump(cur, open(fp, "w"), indent=1)


def _cos_rows(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)


@pytest.fixture(scope="module")
def depth36():
    """The seeded 36-layer checkpoint (host generated, the same bits go to the engine and to the oracle), the HF fixture
    and the oracle's logits for prompt + N_STEPS decode steps, teacher-forced on HF's greedy tokens."""
    meta = json.load(open(GOLD + ".json"))
    npz = np.load(GOLD + ".npz")
    cfgd = dict(meta["config"])
    cfg = Qwen3Config(**cfgd)
    w, bits = synthetic_weights(cfg, seed=meta["seed"], std=meta["std"], with_bits=True)
    prompt = [100 + (i % 1000) for i in range(meta["prompt_tokens"])]
    feed = meta["hf_tokens"][:N_STEPS]
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        orc = Qwen3Oracle(cfg, w, num_pages=len(prompt) // 16 + 8, rope_positions=4096)
        st = KvState()
        rows = [orc.batch_prefill([prompt], [st])[0]]
        for tk in feed:
            rows.append(orc.batch_decode([tk], [st])[0])
    finally:
        O.GEMM_ACCUM = old
    del orc, w
    return dict(cfg=cfgd, bits=bits, prompt=prompt, feed=feed, oracle=np.stack(rows), meta=meta, npz=npz)


def _engine(d, **kw):
    from pegainfer_amd.qwen3 import Qwen3Engine
    kw.setdefault("num_kv_pages", len(d["prompt"]) // 16 + 16)
    kw.setdefault("max_batch_size", 2)
    kw.setdefault("max_positions", 4096)
    return Qwen3Engine(d["cfg"], **kw).load_state(d["bits"])


def _forced(eng, d):
    rid = eng.new_request()
    _, lg = eng.prefill([rid], [d["prompt"]], return_logits=True)
    rows = [lg[0].copy()]
    for tk in d["feed"]:
        _, lg = eng.decode([rid], [tk], return_logits=True)
        rows.append(lg[0].copy())
    eng.drop_request(rid)
    return np.stack(rows)          # bf16 bits [1 + N_STEPS, V]


@pytest.fixture(scope="module")
def engine_bits(built_libs, depth36):
    """logits bits of decode_mode 0 and 1 (graph on) on the teacher-forced stream, and decode_mode 1's own greedy run"""
    out = {}
    for mode in (0, 1):
        eng = _engine(depth36, decode_mode=mode, split_policy=1, enable_graph=True)
        out[mode] = _forced(eng, depth36)
        if mode == 1:
            out["greedy"] = eng.generate_greedy(depth36["prompt"], len(depth36["meta"]["hf_tokens"]))
            name = "model.layers.17.mlp.up_proj.weight"     # export_tensor hands back exactly what was loaded
            got = np.zeros(depth36["bits"][name].size, np.uint16)
            eng._chk(eng.lib.pegainfer_qwen3_export_tensor(eng.h, name.encode(), got.ctypes.data, got.size), "export")
            out["export_equal"] = bool(np.array_equal(got, depth36["bits"][name].ravel()))
        eng.close()
    return out


@pytest.mark.parametrize("mode", [0, 1])
def test_full_depth_logits_match_the_oracle(depth36, engine_bits, mode):
    got, ref = bf16_from_bits(engine_bits[mode]), depth36["oracle"]
    cos = _cos_rows(got, ref)
    scale = float(np.abs(ref).max())
    dmax = np.abs(got - ref).max(-1)
    srt = np.sort(ref, axis=-1)
    margin = srt[:, -1] - srt[:, -2]
    agree = got.argmax(-1) == ref.argmax(-1)
    # accuracy-parity-playbook.md:15-24: a greedy token may differ only at a near-tie - the oracle's top-1 margin at that
    # step must be inside twice that step's own max |dlogit| (two logits moving towards each other)
    explained = agree | (margin <= 2 * dmax)
    _report(f"oracle_mode{mode}", dict(steps=int(len(cos)), cos=[float(x) for x in cos], cos_min=float(cos.min()),
                                       max_dlogit=float(dmax.max()), logit_scale=scale, rel=float(dmax.max() / scale),
                                       tokens_equal=[int(agree.sum()), int(len(agree))],
                                       oracle_margin=[float(x) for x in margin], dlogit_per_step=[float(x) for x in dmax]))
    assert cos.min() > COS_MIN_ORACLE, ("cosine per step", cos)
    assert dmax.max() <= REL_MAX_ORACLE * scale, ("max |dlogit| per step", dmax, scale)
    assert explained.all(), ("greedy token differs away from a near-tie", agree, margin, dmax)


def test_full_depth_fused_decode_equals_reference_sequence_bitwise(engine_bits):
    """decode_mode 1 (4-5 launches per layer, attention + o_proj in one) == decode_mode 0 (the reference's 14) in every
    logit bit of every step, at 36 layers, ctx 1024 -> 1032 (prefill is the same code in both modes)."""
    assert np.array_equal(engine_bits[0], engine_bits[1]), int((engine_bits[0] != engine_bits[1]).sum())
    assert engine_bits["export_equal"]


def test_full_depth_against_hf_transformers_fixture(depth36, engine_bits):
    """HF Transformers bf16 CPU (the reference's truth engine) on the same seeded checkpoint, from the committed
    fixture: (a) teacher-forced logits on the fixture's 4096-index subset and its top-64 set, (b) the engine's own
    free-running greedy tokens against HF's, compared up to the first step whose HF top-1 margin is inside the bar."""
    meta, npz = depth36["meta"], depth36["npz"]
    got = bf16_from_bits(engine_bits[1])
    n = got.shape[0]
    idx, ref_sub = npz["idx"], npz["idx_vals"][:n]
    cos = _cos_rows(got[:, idx], ref_sub)
    scale = float(meta["logit_absmax"])
    dmax = float(np.abs(got[:, idx] - ref_sub).max())
    top_ids, top_vals = npz["top_ids"][:n], npz["top_vals"][:n]
    dtop = float(np.abs(np.take_along_axis(got, top_ids, axis=-1) - top_vals).max())
    ovh = meta.get("oracle_vs_hf", {})
    bar_cos = 1.0 - 1.5 * (1.0 - float(ovh.get("cos_min", 0.995)))
    bar_d = 1.5 * float(ovh.get("max_dlogit", 0.05 * scale))
    margins = np.asarray(meta["top1_margin"])
    hf_tokens = meta["hf_tokens"]
    mine = [int(x) for x in engine_bits["greedy"]]
    first_diff = next((i for i, (a, b) in enumerate(zip(mine, hf_tokens)) if a != b), None)
    dtop_step = np.abs(np.take_along_axis(got, top_ids, axis=-1) - top_vals).max(-1)
    forced_agree = got.argmax(-1) == top_ids[:, 0]
    _report("hf", dict(cos_min_subset=float(cos.min()), max_dlogit_subset=dmax, max_dlogit_top64=dtop, logit_scale=scale,
                       bars=dict(cos=bar_cos, dlogit=bar_d), hf_tokens=hf_tokens, engine_tokens=mine,
                       first_diff_step=first_diff, hf_margin=[float(x) for x in margins],
                       teacher_forced_argmax_equal=[int(forced_agree.sum()), int(n)],
                       oracle_vs_hf=ovh))
    # as close to the reference's truth engine as the oracle is (x 1.5): HF rounds at other points than the reference
    assert cos.min() > bar_cos, (cos, bar_cos)
    assert max(dmax, dtop) <= bar_d, (dmax, dtop, bar_d)
    # teacher-forced on HF's stream: a different greedy token only at a near-tie of the golden logits (random weights give
    # margins of 0.06 ... 0.7 on a logit scale of 4.9, so this bites at almost every step)
    assert (forced_agree | (margins[:n] <= 2 * dtop_step)).all(), (forced_agree, margins[:n], dtop_step)
    assert forced_agree.sum() >= n - 2, (forced_agree, margins[:n])
    # free-running (the e2e loop, tests/e2e.rs:108-221): token ids identical to HF's up to a first difference that must
    # itself sit on a near-tie (e2e-gibberish.md:80 - "sensitive to equal-logit top1 choices")
    if first_diff is not None:
        assert first_diff >= 1 or margins[0] <= 2 * bar_d
        assert margins[first_diff] <= 2 * bar_d, (first_diff, margins[first_diff], mine, hf_tokens)


# ------------------------------------------------------------------ configs[2]: Qwen3-8B shape, 8 layers, top-k / top-p
CFG8 = dict(hidden_size=4096, num_hidden_layers=8, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
            intermediate_size=12288, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False,
            max_position_embeddings=4096)


def test_qwen3_8b_shape_8_layers_logits_and_topk_topp_sampling(built_libs):
    """BASELINE.json configs[2] at depth 8: 1024-token prefill + 4 decode steps against the oracle (decode_mode 1, graph
    on), then the reference's sampling shapes (ops_embedding_sampling_bench.rs:49-90: T 0.8 / top_k 50 / top_p 0.95 and
    T 0.8 / top_k -1 / top_p 0.9) on the engine's logits: every sampled token lies in the oracle's top-k / top-p support
    of that row (FlashInfer's Philox stream itself is parity-unpinned)."""
    from pegainfer_amd.qwen3 import Qwen3Engine
    cfg = Qwen3Config(**CFG8)
    w, bits = synthetic_weights(cfg, seed=808, std=0.02, with_bits=True)
    prompt = [100 + (i % 1000) for i in range(1024)]
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        orc = Qwen3Oracle(cfg, w, num_pages=80, rope_positions=4096)
        st = KvState()
        rows = [orc.batch_prefill([prompt], [st])[0]]
        feed = []
        for _ in range(4):
            feed.append(int(rows[-1].argmax()))
            rows.append(orc.batch_decode([feed[-1]], [st])[0])
    finally:
        O.GEMM_ACCUM = old
    ref = np.stack(rows)
    del orc, w
    eng = Qwen3Engine(CFG8, num_kv_pages=96, max_batch_size=2, decode_mode=1, max_positions=4096).load_state(bits)
    rid = eng.new_request()
    _, lg = eng.prefill([rid], [prompt], return_logits=True)
    got = [bf16_from_bits(lg[0])]
    rng = np.random.default_rng(3)
    sampled = []
    for tk in feed:
        _, lg = eng.decode([rid], [tk], return_logits=True)
        row = bf16_from_bits(lg[0])
        got.append(row)
        for (T, k, p) in ((0.8, 50, 0.95), (0.8, -1, 0.9)):
            keep = O.top_k_top_p_support(O.logits_to_probs(row, 1.0 / T), k, p)
            for _ in range(8):
                t = eng.sample(0, T, k, p, float(rng.random()))
                sampled.append(bool(keep[t]))
    eng.close()
    got = np.stack(got)
    cos = _cos_rows(got, ref)
    scale = float(np.abs(ref).max())
    rel = float(np.abs(got - ref).max() / scale)
    _report("qwen3_8b_8_layers", dict(cos_min=float(cos.min()), rel=rel, logit_scale=scale, sampled_in_support=[
        int(sum(sampled)), len(sampled)]))
    assert cos.min() > 0.9995 and rel <= 0.03, (cos, rel)
    assert all(sampled)
