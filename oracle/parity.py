"""Parity arithmetic shared by the tests and by bench.py's parity legs (TEST INFRASTRUCTURE ONLY - never a product path).

THE DERIVED BAR.  A full-depth model on N(0, 0.02) weights has flat logits, so "cosine > 0.998"
cannot tell summation-order noise from a wrong rounding point, and a bar copied from the first GPU run is a regression
fence, not a tolerance.  Instead three evaluations of the SAME DAG on the SAME weights and the SAME token stream are
compared:

    truth   oracle with no activation rounding at all (oracle.bf16.exact_activations: fp32 storage, fp32 / fp64
            accumulation, unrounded RoPE tables) - what both others approximate
    oracle  the bf16 restatement of the reference (every rounding point of the reference, sgemm accumulation)
    engine  the HIP path

and the assertion is  err(engine vs truth) <= AGG_MAX * err(oracle vs truth)  on the RMS logit error pooled over all
steps of a case (STEP_MAX for any single step).  An engine that rounds where the reference rounds and only sums in
another order sits at a ratio of ~1.0 (two independent realisations of the same rounding noise); a route that keeps a
running sum in bf16 or drops a K tail adds its own error on top and leaves the band.  1.25 is the verdict's figure; a
single step pools 1 / n of the rounding events, so two independent realisations of the same noise legitimately differ by
more there - hence STEP_MAX = 1.5.  tests/test_depth_harness.py pins on the CPU what the band sees (gross routes: ratios
1.7 ... 100) and what it cannot (ONE extra rounding point: +9 ... 16 %, the territory of the bitwise fused == reference
tests).  The same comparison per LAYER (hidden-state taps) is the reference's own debugging method
(docs/playbooks/accuracy-parity-playbook.md:15-24).
"""
import contextlib

import numpy as np

from . import ops as O
from .bf16 import exact_activations

AGG_MAX, STEP_MAX = 1.25, 1.5


def cos_rows(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)


def rms_rows(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return np.sqrt((d * d).mean(-1))


def derived(engine, oracle, truth):
    """rows [..., n] of the three evaluations -> dict with the per-row and pooled error ratios"""
    e_eng, e_orc = rms_rows(engine, truth).ravel(), rms_rows(oracle, truth).ravel()
    ratio = e_eng / np.maximum(e_orc, 1e-30)
    agg = float(np.sqrt((e_eng ** 2).sum() / max((e_orc ** 2).sum(), 1e-60)))
    return dict(ratio=[float(x) for x in ratio], ratio_max=float(ratio.max()), ratio_pooled=agg,
                rms_engine_vs_truth=[float(x) for x in e_eng], rms_oracle_vs_truth=[float(x) for x in e_orc],
                cos_engine_vs_truth_min=float(cos_rows(engine, truth).min()),
                cos_oracle_vs_truth_min=float(cos_rows(oracle, truth).min()),
                cos_engine_vs_oracle_min=float(cos_rows(engine, oracle).min()),
                scale=float(np.abs(truth).max()))


def assert_derived(d, what, agg_max=AGG_MAX, step_max=STEP_MAX):
    assert d["ratio_pooled"] <= agg_max, (what, "pooled err(engine vs truth) / err(oracle vs truth)", d["ratio_pooled"])
    assert d["ratio_max"] <= step_max, (what, "worst single row", d["ratio_max"], d["ratio"])


def near_tie_ok(engine, ref, margin_of):
    """accuracy-parity-playbook.md:15-24: a greedy token may differ from the reference's only at a near-tie - the
    reference's top-1 margin at that row inside twice that row's own max |dlogit|"""
    dmax = np.abs(engine - ref).max(-1)
    srt = np.sort(margin_of, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    agree = engine.argmax(-1) == ref.argmax(-1)
    return agree | (margin <= 2 * dmax), agree, margin, dmax


@contextlib.contextmanager
def oracle_mode(exact):
    """sgemm accumulation (what cuBLAS COMPUTE_32F does; fp64 copies of 4-8 G parameters would dominate the run) and,
    for the truth pass, no activation rounding"""
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        with (exact_activations() if exact else contextlib.nullcontext()):
            yield
    finally:
        O.GEMM_ACCUM = old


@contextlib.contextmanager
def _thread_mode(exact):
    """inside a concurrent pair: only the per-thread part of oracle_mode (the shared GEMM_ACCUM is set by the pair's caller)"""
    with (exact_activations() if exact else contextlib.nullcontext()):
        yield


def qwen3_pass(cfg, w, prompts, feeds, exact, taps=False, max_pos=4096, _set_mode=True):
    """Qwen3 oracle / truth pass: prompts (list of token lists) prefilled as ONE batch, then len(feeds[0]) decode steps
    teacher-forced on feeds[r][s].  -> rows [n_req, 1 + steps, V] (and, with taps, a list per step of [L, n_req, H])"""
    from .qwen3_ref import KvState, Qwen3Oracle
    n, steps = len(prompts), len(feeds[0]) if feeds else 0
    pages = sum(-(-(len(p) + steps) // 16) for p in prompts) + 8
    with (oracle_mode(exact) if _set_mode else _thread_mode(exact)):
        orc = Qwen3Oracle(cfg, w, num_pages=pages, rope_positions=max_pos)
        sts = [KvState() for _ in prompts]
        tap_steps = []
        if taps:
            orc.taps = []
        rows = [np.stack(orc.batch_prefill(prompts, sts))]
        if taps:
            tap_steps.append(np.stack(orc.taps))
        for s in range(steps):
            if taps:
                orc.taps = []
            rows.append(orc.batch_decode([feeds[r][s] for r in range(n)], sts))
            if taps:
                tap_steps.append(np.stack(orc.taps))
    out = np.stack(rows, axis=1)
    return (out, tap_steps) if taps else out


def _pair(fn):
    """bf16 pass and fp32-truth pass of the same model SIDE BY SIDE on two threads (round 6): `exact_activations` is per thread,
    GEMM_ACCUM is the same for both (set once around the pair), numpy releases the GIL inside BLAS and its large element-wise
    kernels - on the pool's 256-core hosts the two ~75 s passes of a 36-layer model then cost the wall time of one.  Same
    functions, same arguments: the same bits as two calls in a row (tests/test_depth_harness.py compares)."""
    from concurrent.futures import ThreadPoolExecutor
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        with ThreadPoolExecutor(max_workers=2) as ex:
            fo, ft = ex.submit(fn, False), ex.submit(fn, True)
            return fo.result(), ft.result()
    finally:
        O.GEMM_ACCUM = old


def qwen3_pass_pair(cfg, w, prompts, feeds, taps=False, max_pos=4096):
    """(oracle result, truth result) of qwen3_pass, computed concurrently"""
    return _pair(lambda exact: qwen3_pass(cfg, w, prompts, feeds, exact, taps=taps, max_pos=max_pos, _set_mode=False))


def qwen35_pass_pair(cfg, w, prompt, feed, taps=False, max_pos=2048):
    """(oracle result, truth result) of qwen35_pass, computed concurrently"""
    return _pair(lambda exact: qwen35_pass(cfg, w, prompt, feed, exact, taps=taps, max_pos=max_pos, _set_mode=False))


def qwen35_pass(cfg, w, prompt, feed, exact, taps=False, max_pos=2048, _set_mode=True):
    """Qwen3.5 oracle / truth pass on one request -> rows [1 + steps, V] (and taps: per step [L, 1, H])"""
    from .qwen35_ref import Qwen35Oracle
    with (oracle_mode(exact) if _set_mode else _thread_mode(exact)):
        orc = Qwen35Oracle(cfg, w, num_pages=(len(prompt) + len(feed)) // 16 + 8, rope_positions=max_pos)
        st = orc.new_request()
        tap_steps = []
        if taps:
            orc.taps = []
        rows = [orc.prefill(prompt, st)]
        if taps:
            tap_steps.append(np.stack(orc.taps))
        for tk in feed:
            if taps:
                orc.taps = []
            rows.append(orc.batch_decode([tk], [st])[0])
            if taps:
                tap_steps.append(np.stack(orc.taps))
    out = np.stack(rows)
    return (out, tap_steps) if taps else out


def layer_curve(engine_taps, oracle_taps, truth_taps):
    """per-layer comparison of hidden-state taps [L, rows, H]: cosine engine vs oracle, and the derived error ratio"""
    L = engine_taps.shape[0]
    cos_eo = [float(cos_rows(engine_taps[l], oracle_taps[l]).min()) for l in range(L)]
    e_eng = np.array([np.sqrt((rms_rows(engine_taps[l], truth_taps[l]) ** 2).mean()) for l in range(L)])
    e_orc = np.array([np.sqrt((rms_rows(oracle_taps[l], truth_taps[l]) ** 2).mean()) for l in range(L)])
    rel = np.array([np.sqrt((truth_taps[l].astype(np.float64) ** 2).mean()) for l in range(L)])
    return dict(cos_engine_vs_oracle=cos_eo, rel_err_engine=[float(x) for x in e_eng / rel],
                rel_err_oracle=[float(x) for x in e_orc / rel], ratio=[float(x) for x in e_eng / np.maximum(e_orc, 1e-30)])
