"""GPU parity of BATCHED decode at FULL DEPTH (VERDICT r5 item 1a / Missing 3).

bench.py publishes `batch_sweep` (bs 2 / 4 / 8 / 16) on the 36-layer Qwen3-4B shape; until round 6 the 36-layer tests ran one
request and batches were checked at 2 layers only.  Here, on the same seeded 36-layer checkpoint as tests/test_gpu_full_depth.py:

  * a bs-16 stream with RAGGED contexts 1 ... 1024 tokens and a bs-4 stream (1024 / 257 / 17 / 1 tokens), 8 decode steps each,
    decode_mode 0 (the reference's 14-launch layer over the 3..16-column skinny MFMA GEMMs) and decode_mode 1 (fused forms,
    lazy-ticket flush), hipGraph on - the shape of the reference's `batch_matches_sequential` (batch_decode.rs:505-606);
  * every column of every step against the ORACLE and its fp32 TRUTH pass under the derived bar (oracle/parity.py), greedy
    tokens by the near-tie rule; ONE oracle pass and ONE truth pass over the 16 streams serve both batch sizes and the
    per-request runs (the feeds are seeded tokens, independent of the engine);
  * decode_mode 0 == decode_mode 1 in every logit bit of every column and step;
  * batch == batch: the bs-4 columns against the same requests' columns of the bs-16 steps - bitwise where the two plans
    coincide (a request whose whole context is one KV chunk under both plans: the 3..16-column GEMM family gives a column the
    same bits at every batch size, and an un-partitioned scan does not depend on the neighbours), within the bf16 noise of
    the partitioned-KV merge where they do not (bs 4 cuts 1032 tokens into 80-token chunks, bs 16 into 272-token ones);
  * batch == per-request: each of the four requests decoded ALONE (dot2 GEMV family, fused attention + o_proj launch) - other
    kernels, other summation orders: inside the derived bar against the same truth rows, greedy tokens equal away from
    near-ties (the reference's check is token-level too).

Every request is prefilled alone (the same launches in every run), so the KV cache a decode step reads is bit-identical across
the runs and the comparisons above isolate the batched DECODE path.  Measured numbers: gpurun_out/full_depth_parity.json.
"""
import json
import os
import time

import numpy as np
import pytest

import depth_common as dc
from oracle.bf16 import bf16_from_bits
from oracle.qwen3_ref import Qwen3Config, synthetic_weights

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD_DIR = os.environ.get("PEGAINFER_DEPTH_GOLD_DIR") or os.path.join(HERE, "golden")
GOLD = os.path.join(GOLD_DIR, "qwen3_4b_depth36_hf")
LENS = (1, 2, 5, 9, 16, 17, 31, 33, 48, 64, 65, 100, 128, 200, 257, 1024)   # ragged contexts 1 ... 1024, 2000 tokens in all
BS4 = (1024, 257, 17, 1)                                                     # the bs-4 stream: a subset of the same requests
STEPS = 8
SEED = 20261001
SANITY_COS = 0.98


def _timed(label, t0):
    dc.report_kv("durations", label, round(time.time() - t0, 1))


@pytest.fixture(scope="module")
def ckpt():
    t0 = time.time()
    meta = json.load(open(GOLD + ".json"))
    cfgd = dict(meta["config"])
    cfg = Qwen3Config(**cfgd)
    w, bits = synthetic_weights(cfg, seed=meta["seed"], std=meta["std"], with_bits=True)
    rng = np.random.default_rng(SEED)
    prompts = {n: [int(x) for x in rng.integers(0, cfgd["vocab_size"], n)] for n in LENS}
    feeds = {n: [int(x) for x in rng.integers(0, cfgd["vocab_size"], STEPS)] for n in LENS}
    _timed("batch_ckpt36", t0)
    return dict(cfgd=cfgd, cfg=cfg, w=w, bits=bits, prompts=prompts, feeds=feeds)


def _engine(d, **kw):
    from pegainfer_amd.qwen3 import Qwen3Engine
    kw.setdefault("num_kv_pages", 512)
    kw.setdefault("max_batch_size", 16)
    kw.setdefault("max_positions", 4096)
    return Qwen3Engine(d["cfgd"], **kw).load_state(d["bits"])


def _batched(eng, d, lens):
    """every request prefilled ALONE, then STEPS decode steps of the whole batch teacher-forced on the seeded feeds
    -> bf16 bits [len(lens), 1 + STEPS, V]"""
    rids, rows = [], []
    for n in lens:
        rid = eng.new_request()
        _, lg = eng.prefill([rid], [d["prompts"][n]], return_logits=True)
        rids.append(rid)
        rows.append([lg[0].copy()])
    for s in range(STEPS):
        _, lg = eng.decode(rids, [d["feeds"][n][s] for n in lens], return_logits=True)
        for i in range(len(lens)):
            rows[i].append(lg[i].copy())
    for r in rids:
        eng.drop_request(r)
    return np.stack([np.stack(r) for r in rows])


@pytest.fixture(scope="module")
def engine_rows(built_libs, ckpt):
    t0 = time.time()
    d, out = ckpt, {}
    for mode in (1, 0):
        eng = _engine(d, decode_mode=mode, split_policy=1, enable_graph=True)
        out[(16, mode)] = _batched(eng, d, LENS)
        out[(4, mode)] = _batched(eng, d, BS4)
        if mode == 1:
            out["alone"] = {n: _batched(eng, d, (n,))[0] for n in BS4}
        eng.close()
    _timed("batch_engine36", t0)
    return out


@pytest.fixture(scope="module")
def oracle_rows(ckpt):
    t0 = time.time()
    d = ckpt
    prompts, feeds = [d["prompts"][n] for n in LENS], [d["feeds"][n] for n in LENS]
    orc, tru = dc.qwen3_pass_pair(d["cfg"], d["w"], prompts, feeds)     # both passes side by side on two threads
    _timed("batch_oracle36_pair", t0)
    return dict(oracle=orc, truth=tru)                 # [16, 1 + STEPS, V] each


def _cols(lens):
    return [LENS.index(n) for n in lens]


@pytest.mark.parametrize("bs,mode", [(16, 1), (16, 0), (4, 1), (4, 0)])
def test_batched_decode_at_full_depth_matches_the_oracle(engine_rows, oracle_rows, bs, mode):
    """bs 4 / bs 16, ragged contexts, 36 layers: every column and step under the derived bar against the truth pass (pooled
    per request and over the whole batch), greedy tokens by the near-tie rule."""
    lens = LENS if bs == 16 else BS4
    c = _cols(lens)
    got = bf16_from_bits(engine_rows[(bs, mode)])
    ref, tru = oracle_rows["oracle"][c], oracle_rows["truth"][c]
    per_req = {}
    for i, n in enumerate(lens):
        dv = dc.derived(got[i], ref[i], tru[i])
        per_req[str(n)] = dict(ratio_pooled=dv["ratio_pooled"], ratio_max=dv["ratio_max"], cos_engine_vs_oracle_min=dv["cos_engine_vs_oracle_min"])
        assert dv["cos_engine_vs_oracle_min"] > SANITY_COS, (n, dv["cos_engine_vs_oracle_min"])
        dc.assert_derived(dv, f"Qwen3-4B x 36, bs {bs}, decode_mode {mode}, request of {n} tokens")
    dec = dc.derived(got[:, 1:], ref[:, 1:], tru[:, 1:])          # the batched decode steps alone (row 0 is each prefill)
    ok, agree, margin, dmax = dc.near_tie_ok(got, ref, ref)
    dc.report(f"batch_bs{bs}_mode{mode}", dict(decode_ratio_pooled=dec["ratio_pooled"], decode_ratio_max=dec["ratio_max"],
                                               cos_engine_vs_oracle_min=dec["cos_engine_vs_oracle_min"], per_request=per_req,
                                               tokens_equal=[int(agree.sum()), int(agree.size)], max_dlogit=float(dmax.max()),
                                               scale=dec["scale"], contexts=list(lens), steps=STEPS))
    assert dec["ratio_pooled"] <= dc.AGG_MAX, dec["ratio_pooled"]
    assert ok.all(), ("greedy token differs away from a near-tie", np.argwhere(~ok), margin[~ok], dmax[~ok])


def test_batched_fused_decode_equals_the_reference_sequence_bitwise(engine_rows):
    """decode_mode 1 (fused add + norm prologues, SwiGLU epilogues, lazy-ticket flush, fused attention forms) == decode_mode 0
    (the reference's 14-launch layer) in every logit bit of every column and step, bs 4 and bs 16, 36 layers."""
    for bs in (4, 16):
        a, b = engine_rows[(bs, 0)], engine_rows[(bs, 1)]
        assert np.array_equal(a, b), (bs, int((a != b).sum()))


def _chunks(lens_now, padded):
    """chunks per request of a decode step over contexts `lens_now` (the host plan through the C ABI: pegainfer_split_kv_plan)"""
    from pegainfer_amd import ffi
    lib = ffi.host_lib()
    slots = padded * 64
    ri, kt = np.zeros(slots, np.int32), np.zeros(slots, np.int32)
    oi, va = np.zeros(padded + 1, np.int32), np.zeros(slots, np.uint8)
    chunk, use = np.zeros(1, np.int32), np.zeros(1, np.int32)
    L = np.asarray(lens_now, np.int32)
    lib.pegainfer_split_kv_plan(1, len(lens_now), L.ctypes.data, padded, 8, ri.ctypes.data, kt.ctypes.data, oi.ctypes.data,
                                va.ctypes.data, chunk.ctypes.data, use.ctypes.data)
    return [int(oi[r + 1] - oi[r]) if use[0] else 1 for r in range(len(lens_now))], int(chunk[0])


def test_batch_size_does_not_change_a_column_where_the_plans_coincide(engine_rows):
    """The same four requests at bs 4 and inside the bs-16 steps.  The host plan (kv_pool.h make_split_plan, policy 1) cuts a
    bs-4 step's contexts into ~80-token chunks and a bs-16 step's into ~272-token ones.  A request that is ONE chunk under both
    plans at every step must come out bit-identical (3..16-column GEMM family: a column's bits do not depend on the batch
    size; an un-partitioned scan does not depend on the neighbours).  A request whose partials differ between the plans
    carries the bf16 noise of the split-KV merge: bounded against its own distance to the oracle by the derived-bar tests
    above, here by a cosine fence and reported."""
    a, b = engine_rows[(4, 1)], engine_rows[(16, 1)][_cols(BS4)]
    one_chunk = {n: True for n in BS4}
    plans = {}
    for s in range(STEPS):
        c4, k4 = _chunks([n + s + 1 for n in BS4], 4)
        c16, k16 = _chunks([n + s + 1 for n in LENS], 16)
        plans[s] = dict(chunk_bs4=k4, chunk_bs16=k16)
        for i, n in enumerate(BS4):
            one_chunk[n] = one_chunk[n] and c4[i] == 1 and c16[LENS.index(n)] == 1
    rep = {"plans": plans}
    for i, n in enumerate(BS4):
        same = bool(np.array_equal(a[i], b[i]))
        fa, fb = bf16_from_bits(a[i]), bf16_from_bits(b[i])
        rep[str(n)] = dict(one_chunk_under_both_plans=one_chunk[n], bitwise_equal=same, max_dlogit=float(np.abs(fa - fb).max()),
                           cos_min=float(dc.cos_rows(fa, fb).min()))
        dc.report("batch_bs4_vs_bs16_columns", rep)
        assert np.array_equal(a[i][0], b[i][0]), n                     # the lone prefill of the request: same launches, same bits
        if one_chunk[n]:
            assert same, (n, rep[str(n)])
        else:
            assert rep[str(n)]["cos_min"] > 0.998, (n, rep[str(n)])
    assert one_chunk[1] and one_chunk[17] and not one_chunk[1024]      # the workload really has both kinds


def test_batched_columns_against_each_request_alone(engine_rows, oracle_rows):
    """batch == sequential (batch_decode.rs:505-606) across kernel families: the lone request runs the dot2 GEMVs and the fused
    attention + o_proj launch, the batch the skinny MFMA GEMMs - not bitwise; both inside the derived bar against the SAME truth
    rows, greedy tokens equal wherever the oracle's margin exceeds twice the larger of the two distances to the oracle."""
    c = _cols(BS4)
    ref, tru = oracle_rows["oracle"][c], oracle_rows["truth"][c]
    rep = {}
    for i, n in enumerate(BS4):
        alone = bf16_from_bits(engine_rows["alone"][n])
        batch = bf16_from_bits(engine_rows[(16, 1)][c[i]])
        dv = dc.derived(alone, ref[i], tru[i])
        dc.assert_derived(dv, f"request of {n} tokens decoded alone")
        assert np.array_equal(engine_rows["alone"][n][0], engine_rows[(16, 1)][c[i]][0]), n        # same lone prefill
        srt = np.sort(ref[i], axis=-1)
        margin = srt[:, -1] - srt[:, -2]
        d = np.maximum(np.abs(alone - ref[i]).max(-1), np.abs(batch - ref[i]).max(-1))
        agree = alone.argmax(-1) == batch.argmax(-1)
        rep[str(n)] = dict(ratio_pooled_alone=dv["ratio_pooled"], tokens_equal=[int(agree.sum()), int(agree.size)],
                           max_dlogit_alone_vs_batch=float(np.abs(alone - batch).max()),
                           cos_alone_vs_batch_min=float(dc.cos_rows(alone, batch).min()))
        assert (agree | (margin <= 2 * d)).all(), (n, agree, margin, d)
    dc.report("batch_vs_alone", rep)
