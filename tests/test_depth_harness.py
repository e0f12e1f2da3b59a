"""CPU tests of the derived-bar harness the GPU full-depth tests assert with (tests/depth_common.py).

The GPU tests claim: "err(engine vs fp32 truth) <= 1.25 x err(bf16 oracle vs fp32 truth) separates summation-order noise
from a wrong rounding point or kernel route".  That claim is checked HERE, where no GPU is needed, with stand-in engines
built from the oracle itself on a small Qwen3-shaped model:

  * another summation order (fp64 instead of fp32 accumulation in every GEMM: the roundings flip at different places
    and the flips cascade through the layers) must sit INSIDE the band - and this is also the measurement of how far two
    honest realisations of the same rounding noise differ per step (the reason STEP_MAX is wider than AGG_MAX);
  * gross errors - a GEMM that drops the last 64 elements of K, a GEMM whose running sum lives in bf16 - must fall
    OUTSIDE it (ratios of 1.7 and 100);
  * and, stated honestly, what the band can NOT see: ONE extra rounding point (split-K partials rounded to bf16 before the
    slice sum, an attention output rounded twice) adds 9-16 % to a noise floor made of ~10 rounding points per layer and
    stays inside 1.25.  That class is what the BITWISE tests are for (fused == reference op sequence, graph == eager,
    batch == single, tests/test_gpu_fused.py / test_gpu_model.py / test_gpu_full_depth.py) together with the <= 1 ulp op
    tests; the derived bar is the referee for "is the remaining distance to the oracle summation order or something
    bigger", not a replacement for them;
  * FAULT INJECTION at ctx 1024: an off-by-one causal mask, a RoPE position shifted by one at the
    decode steps, a KV page of another request in the page table, a decode append that lands one slot late - each run
    through the same band, on the flat-logit N(0, 0.02) model the GPU fixtures use AND on a "sharp attention" variant
    (q / k norm weights ~ N(4, 0.4): score std ~ 16).  Result, pinned below: the flat model SEES the first three (ratios
    6 ... 28 at 8 layers, 3 ... 36 at 24) and barely sees the one missing token of 1024 (1.3 on the 1024-token request);
    the sharp model is CHAOTIC - two honest realisations of the same arithmetic are cosine 0.65 apart, every ratio
    collapses towards 1 - so a sharp-attention fixture makes the referee blind, not sharp-eyed, and none was added;
  * the exact-activation context really is rounding-free and leaves checkpoints alone;
  * the per-layer taps of both oracles line up with the logits they lead to.
"""
import numpy as np
import pytest

import depth_common as dc
from oracle import ops as O
from oracle.bf16 import bf16_round, exact_activations
from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle, synthetic_weights

CFG = dict(hidden_size=512, num_hidden_layers=8, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
           intermediate_size=1536, vocab_size=4096)
STEPS = 5


@pytest.fixture(scope="module")
def small():
    cfg = Qwen3Config(**CFG)
    w = synthetic_weights(cfg, seed=5, std=0.02)
    rng = np.random.default_rng(0)
    prompts = [rng.integers(0, CFG["vocab_size"], n).tolist() for n in (1, 8, 17, 48, 100)]
    feeds = [rng.integers(0, CFG["vocab_size"], STEPS).tolist() for _ in prompts]
    orc, otaps = dc.qwen3_pass(cfg, w, prompts, feeds, exact=False, taps=True)
    tru, ttaps = dc.qwen3_pass(cfg, w, prompts, feeds, exact=True, taps=True)
    return dict(cfg=cfg, w=w, prompts=prompts, feeds=feeds, oracle=orc, truth=tru, otaps=otaps, ttaps=ttaps)


def _stand_in(s, gemm=None, attend_round_twice=False, taps=False):
    """an 'engine' made of the oracle: the reference's rounding points, GEMMs replaced by `gemm` (default: fp64 accumulation)"""
    old_gemm, old_acc, old_dec = O.gemm, O.GEMM_ACCUM, O.paged_attention_decode
    O.GEMM_ACCUM = np.float64
    if gemm is not None:
        O.gemm = gemm
    if attend_round_twice:
        O.paged_attention_decode = lambda *a, **k: bf16_round(old_dec(*a, **k) * np.float32(1.0 + 2.0 ** -7))
    try:
        o = Qwen3Oracle(s["cfg"], s["w"], num_pages=64, rope_positions=4096)
        sts = [KvState() for _ in s["prompts"]]
        tap_steps = []
        if taps:
            o.taps = []
        rows = [np.stack(o.batch_prefill(s["prompts"], sts))]
        if taps:
            tap_steps.append(np.stack(o.taps))
        for st in range(STEPS):
            if taps:
                o.taps = []
            rows.append(o.batch_decode([f[st] for f in s["feeds"]], sts))
            if taps:
                tap_steps.append(np.stack(o.taps))
    finally:
        O.gemm, O.GEMM_ACCUM, O.paged_attention_decode = old_gemm, old_acc, old_dec
    out = np.stack(rows, axis=1)
    return (out, tap_steps) if taps else out


def test_another_summation_order_sits_inside_the_band(small):
    d = dc.derived(_stand_in(small), small["oracle"], small["truth"])
    dc.assert_derived(d, "fp64-accumulating stand-in")
    # two honest realisations of the same rounding noise: pooled within a few percent of 1, single rows within ~ +-15 %
    assert 0.9 <= d["ratio_pooled"] <= 1.1, d["ratio_pooled"]
    assert min(d["ratio"]) > 0.75 and d["ratio_max"] < 1.3, (min(d["ratio"]), d["ratio_max"])
    # and they are about as far from each other as each is from the truth: the flips cascade, they do not cancel
    assert d["cos_engine_vs_oracle_min"] < 1.0 - 0.3 * (1.0 - d["cos_oracle_vs_truth_min"])


def _gemm_bf16_partials(W, X, slices=4):
    K = W.shape[1]
    step = -(-K // slices)
    acc = np.zeros((X.shape[0], W.shape[0]), np.float32)
    for k0 in range(0, K, step):
        acc = acc + bf16_round((X[:, k0:k0 + step].astype(np.float64) @ W[:, k0:k0 + step].astype(np.float64).T).astype(np.float32))
    return bf16_round(acc)


def _gemm_bf16_accumulator(W, X, slices=16):
    K = W.shape[1]
    step = -(-K // slices)
    acc = np.zeros((X.shape[0], W.shape[0]), np.float32)
    for k0 in range(0, K, step):
        acc = bf16_round(acc + (X[:, k0:k0 + step].astype(np.float64) @ W[:, k0:k0 + step].astype(np.float64).T).astype(np.float32))
    return acc


def _gemm_dropped_tail(W, X):
    K = W.shape[1] - 64
    return bf16_round((X[:, :K].astype(np.float64) @ W[:, :K].astype(np.float64).T).astype(np.float32))


@pytest.mark.parametrize("name,kw", [("GEMM whose running sum is kept in bf16", dict(gemm=_gemm_bf16_accumulator)),
                                     ("GEMM that drops the last 64 of K", dict(gemm=_gemm_dropped_tail))])
def test_a_wrong_route_falls_outside_the_band(small, name, kw):
    d = dc.derived(_stand_in(small, **kw), small["oracle"], small["truth"])
    assert d["ratio_pooled"] > 1.3 * dc.AGG_MAX, (name, d["ratio_pooled"])
    with pytest.raises(AssertionError):
        dc.assert_derived(d, name)


@pytest.mark.parametrize("name,kw", [("split-K with bf16 partials", dict(gemm=_gemm_bf16_partials)),
                                     ("decode attention scaled + rounded twice", dict(attend_round_twice=True))])
def test_one_extra_rounding_point_is_visible_but_inside_the_band(small, name, kw):
    """the limit of the derived bar, pinned so nobody reads more into a green run than it says (module docstring)"""
    d = dc.derived(_stand_in(small, **kw), small["oracle"], small["truth"])
    assert 1.03 < d["ratio_pooled"] <= dc.AGG_MAX, (name, d["ratio_pooled"])


def test_exact_context_is_rounding_free_and_scoped():
    x = np.float32([1.0 + 2.0 ** -10, 3.14159, -2.5e-3])
    assert not np.array_equal(bf16_round(x), x)
    with exact_activations():
        assert np.array_equal(bf16_round(x), x)
        with exact_activations():
            pass
        assert np.array_equal(bf16_round(x), x)          # nesting restores the outer state, not "off"
    assert not np.array_equal(bf16_round(x), x)
    # checkpoints generated outside the context stay bf16-valued whatever runs later
    w = synthetic_weights(Qwen3Config(**dict(CFG, num_hidden_layers=1)), seed=1)["model.norm.weight"]
    assert np.array_equal(bf16_round(w), w)


def test_truth_is_closer_to_fp64_than_the_bf16_oracle_is(small):
    """the truth pass accumulates in fp32 sgemm; against an fp64-accumulating exact pass it must be orders of magnitude
    closer than the bf16 oracle is to it - otherwise it could not referee"""
    old = O.GEMM_ACCUM
    s = small
    try:
        O.GEMM_ACCUM = np.float64
        with exact_activations():
            o = Qwen3Oracle(s["cfg"], s["w"], num_pages=64, rope_positions=4096)
            sts = [KvState() for _ in s["prompts"]]
            ref = np.stack(o.batch_prefill(s["prompts"], sts))
    finally:
        O.GEMM_ACCUM = old
    e_truth = dc.rms_rows(s["truth"][:, 0], ref).max()
    e_orc = dc.rms_rows(s["oracle"][:, 0], ref).min()
    assert e_truth < 1e-2 * e_orc, (e_truth, e_orc)


def test_layer_taps_line_up(small):
    s = small
    L, n = CFG["num_hidden_layers"], len(s["prompts"])
    assert len(s["otaps"]) == 1 + STEPS and s["otaps"][0].shape == (L, n, CFG["hidden_size"])
    # the last tap of a PREFILL is the residual stream the final norm + lm_head consume (a decode step normalises the
    # unrounded fp32 sum inside fused_add_rms_norm, so there only the stored, rounded stream is tapped)
    w = s["w"]
    last = s["otaps"][0][-1]
    normed = O.rms_norm(last, w["model.norm.weight"], 1e-6)
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        lg = O.gemm(w["model.embed_tokens.weight"], normed)
    finally:
        O.GEMM_ACCUM = old
    assert np.array_equal(lg, s["oracle"][:, 0])
    eng, etaps = _stand_in(s, taps=True)
    for st in (0, 1, STEPS):
        c = dc.layer_curve(etaps[st], s["otaps"][st], s["ttaps"][st])
        assert len(c["ratio"]) == L and max(c["ratio"]) < 1.6 and min(c["cos_engine_vs_oracle"]) > 0.999
        # relative error grows with depth (rounding events accumulate) in both realisations
        assert c["rel_err_oracle"][-1] > c["rel_err_oracle"][0]
    bad, btaps = _stand_in(s, gemm=_gemm_bf16_accumulator, taps=True)
    c = dc.layer_curve(btaps[0], s["otaps"][0], s["ttaps"][0])
    assert c["ratio"][0] > dc.AGG_MAX          # the tap localises the wrong route to the FIRST layer


# ===================================================================== fault injection at ctx 1024
FAULT_LENS = (1024, 300)


def _fault_model(sharp):
    cfg = Qwen3Config(**CFG)
    w = synthetic_weights(cfg, seed=5, std=0.02)
    if sharp:   # q / k norm weights ~ N(4, 0.4): scores 16 x larger (std ~ 16 instead of ~ 1)
        rng = np.random.default_rng(77)
        for i in range(cfg.num_hidden_layers):
            for n in ("q_norm", "k_norm"):
                w[f"model.layers.{i}.self_attn.{n}.weight"] = bf16_round((rng.standard_normal(128) * 0.4 + 4.0).astype(np.float32))
    rng = np.random.default_rng(0)
    prompts = [rng.integers(0, CFG["vocab_size"], n).tolist() for n in FAULT_LENS]
    feeds = [rng.integers(0, CFG["vocab_size"], STEPS).tolist() for _ in prompts]
    return dict(cfg=cfg, w=w, prompts=prompts, feeds=feeds, oracle=dc.qwen3_pass(cfg, w, prompts, feeds, exact=False),
                truth=dc.qwen3_pass(cfg, w, prompts, feeds, exact=True))


@pytest.fixture(scope="module")
def flat1024():
    return _fault_model(False)


@pytest.fixture(scope="module")
def sharp1024():
    return _fault_model(True)


def _causal(shift):
    return lambda f: (lambda *a, **k: f(*a, causal_shift=shift, **k))


def _rope_plus_one_at_decode(f):
    def g(q, k, qw, kw, cos, sin, positions, *a):
        p = np.asarray(positions)
        return f(q, k, qw, kw, cos, sin, p + 1 if len(p) == len(FAULT_LENS) else p, *a)   # decode steps only: a shift of
    return g                                                                             # EVERY position is no fault (RoPE is relative)


def _page_of_another_request(f):
    def g(q, kv, layout, layer, pages, indptr, last, *a):
        pg = np.array(pages).copy()
        pg[int(indptr[0]) + 5] = pg[int(indptr[1]) + 2]    # 16 of request 0's 1024 tokens read from request 1's cache
        return f(q, kv, layout, layer, pg, indptr, last, *a)
    return g


def _append_one_slot_late(f):
    def g(kv, layout, layer, pages, indptr, k, v, bi, positions):
        p = np.asarray(positions)
        if len(p) == len(FAULT_LENS):   # the decode append (where pos + 1 stays inside the page): the newest token's K / V are
            p = np.where((p + 1) % 16 != 0, p + 1, p)   # missing from what the step attends to
        return f(kv, layout, layer, pages, indptr, k, v, bi, p)
    return g


FAULTS = {"causal mask -1 (own token hidden)": {"batch_prefill_paged": _causal(-1)},
          "causal mask +1 (one future token visible)": {"batch_prefill_paged": _causal(+1)},
          "RoPE position + 1 at the decode steps": {"qk_norm_rope": _rope_plus_one_at_decode},
          "KV page of another request in the table": {"paged_attention_decode": _page_of_another_request},
          "decode append one slot late": {"paged_kv_scatter": _append_one_slot_late}}


def _faulty(s, patches):
    saved = {k: getattr(O, k) for k in patches}
    old_acc = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float64
    for k, f in patches.items():
        setattr(O, k, f(saved[k]))
    try:
        o = Qwen3Oracle(s["cfg"], s["w"], num_pages=200, rope_positions=4096)
        sts = [KvState() for _ in s["prompts"]]
        rows = [np.stack(o.batch_prefill(s["prompts"], sts))]
        for st in range(STEPS):
            rows.append(o.batch_decode([f[st] for f in s["feeds"]], sts))
    finally:
        O.GEMM_ACCUM = old_acc
        for k, f in saved.items():
            setattr(O, k, f)
    return np.stack(rows, axis=1)


@pytest.mark.parametrize("name,floor", [("causal mask -1 (own token hidden)", 8.0), ("causal mask +1 (one future token visible)", 8.0),
                                        ("RoPE position + 1 at the decode steps", 3.0),
                                        ("KV page of another request in the table", 3.0)])
def test_the_band_sees_an_attention_fault_at_ctx_1024(flat1024, name, floor):
    """Measured (8 layers; 24 layers in docs/lab-notes/round-6.md): mask -1 / +1: 21 / 29 (24 / 36), RoPE + 1: 6.9 (2.9), wrong
    page: 6.6 (4.1) - pooled, both requests; the 1024-token request alone: 15 / 23 / 6.8 / 9.0."""
    s = flat1024
    got = _faulty(s, FAULTS[name])
    d = dc.derived(got, s["oracle"], s["truth"])
    d0 = dc.derived(got[:1], s["oracle"][:1], s["truth"][:1])            # the 1024-token request alone
    assert d["ratio_pooled"] > floor and d0["ratio_pooled"] > floor, (name, d["ratio_pooled"], d0["ratio_pooled"])
    with pytest.raises(AssertionError):
        dc.assert_derived(d0, name)
    if "causal" not in name:   # a decode-side fault leaves the prefill row alone: the per-step ratios localise it
        assert d0["ratio"][0] < 1.1 and min(d0["ratio"][1:]) > 2.0, d0["ratio"]


def test_one_missing_token_of_1024_is_at_the_edge_of_the_band(flat1024):
    """The limit at long context, stated: the newest token's K / V missing from ONE decode step's attention is 1 of 1024
    (1 of 300) tokens - pooled 2.0 over both requests, 1.35 on the 1024-token request alone, with single steps inside
    1.5.  That class (an append off by a slot) is pinned by the exact-bits scatter test and batch == single instead."""
    s = flat1024
    got = _faulty(s, FAULTS["decode append one slot late"])
    d = dc.derived(got, s["oracle"], s["truth"])
    d0 = dc.derived(got[:1], s["oracle"][:1], s["truth"][:1])
    assert d["ratio_pooled"] > 1.5 and 1.1 < d0["ratio_pooled"] < 1.8, (d["ratio_pooled"], d0["ratio_pooled"])


def test_a_sharp_attention_fixture_is_chaotic_not_sharp_eyed(sharp1024):
    """VERDICT r5 asked for a sharp-attention fixture (score std ~ 16) should the flat one be blind.  It is not blind (above)
    - and the sharp one would be: with softmax weights of O(1) on single keys the rounding noise is amplified to the size
    of the logits themselves (two honest realisations: cosine < 0.9, measured 0.65), so err(oracle vs truth) is the whole
    signal and every fault disappears in it (ratios 1.0 ... 1.3).  No such fixture was added to the 36-layer generator."""
    s = sharp1024
    honest = dc.derived(_faulty(s, {}), s["oracle"], s["truth"])
    assert honest["cos_engine_vs_oracle_min"] < 0.9 and 0.9 < honest["ratio_pooled"] < 1.1, honest
    for name in ("causal mask +1 (one future token visible)", "KV page of another request in the table"):
        d = dc.derived(_faulty(s, FAULTS[name]), s["oracle"], s["truth"])
        assert d["ratio_pooled"] < 1.5, (name, d["ratio_pooled"])


def test_concurrent_pass_pair_equals_two_passes_in_a_row(small):
    """oracle/parity.py runs the bf16 pass and the fp32-truth pass of a model side by side on two threads (the exact-activation
    switch is per thread): same rows and taps, bit for bit, as the two calls one after the other."""
    s = small
    (orc, otaps), (tru, ttaps) = dc.qwen3_pass_pair(s["cfg"], s["w"], s["prompts"], s["feeds"], taps=True)
    assert np.array_equal(orc, s["oracle"]) and np.array_equal(tru, s["truth"])
    assert all(np.array_equal(a, b) for a, b in zip(otaps, s["otaps"])) and all(np.array_equal(a, b) for a, b in zip(ttaps, s["ttaps"]))
    assert not np.array_equal(orc, tru)
    x = np.float32([1.0 + 2.0 ** -10])
    assert not np.array_equal(bf16_round(x), x)          # the caller's thread is back in bf16 mode
