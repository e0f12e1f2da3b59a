"""GPU parity of the whole Qwen3.5 hybrid path (C++ host runtime + HIP kernels through the C ABI: chunk-wise
gated-delta-rule prefill, recurrent decode, conv1d, HD256 gated attention) against
  (1) the committed HF-Transformers golden (tests/golden/qwen35_tiny_*, make_qwen35_tiny_golden.py),
  (2) the CPU oracle (oracle/qwen35_ref.py) on the same checkpoint,
and the invariants the reference relies on: graph == eager (bitwise), batched decode == per-request decode (bitwise),
prefill in two pieces (state hand-off, prefill.rs:52-54) == one piece within the model tolerance, rerun determinism.
Tolerances as in tests/test_oracle_qwen35_golden.py.
"""
import json
import os

import numpy as np
import pytest

from oracle.qwen35_ref import Qwen35Config, Qwen35Oracle
from oracle.safetensors_io import load_safetensors

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CKPT = os.path.join(G, "qwen35_tiny.safetensors")


@pytest.fixture(scope="module")
def golden35():
    meta = json.load(open(os.path.join(G, "qwen35_tiny_golden.json")))
    return meta, np.load(os.path.join(G, "qwen35_tiny_logits.npz"))


def make_engine(meta, **kw):
    from pegainfer_amd.qwen35 import Qwen35Engine
    kw.setdefault("num_kv_pages", 128)
    kw.setdefault("max_batch_size", 4)
    return Qwen35Engine(meta["config"], **kw).load_safetensors(CKPT)


def teacher_forced(eng, case):
    rid = eng.new_request()
    _, lg = eng.prefill(rid, case["prompt_tokens"], want_logits=True)
    rows = [lg]
    for tok in case["output_tokens"][:-1]:
        _, lg = eng.decode([rid], [tok], want_logits=True)
        rows.append(lg[0])
    eng.drop_request(rid)
    return np.stack(rows)


@pytest.mark.parametrize("graph,split", [(True, 1), (False, 1), (True, 0)])
def test_qwen35_matches_hf_golden(built_libs, golden35, graph, split):
    meta, hf = golden35
    eng = make_engine(meta, enable_graph=graph, split_policy=split)
    for case in meta["cases"]:
        L, H = teacher_forced(eng, case), hf[case["name"]]
        cos = (L * H).sum(-1) / np.linalg.norm(L, axis=-1) / np.linalg.norm(H, axis=-1)
        assert cos.min() > 0.999, (case["name"], cos.min())
        assert np.abs(L - H).max() < 0.75, (case["name"], np.abs(L - H).max())
        strong = np.array(case["top1_margin"]) > 0.5
        assert np.array_equal(L.argmax(-1)[strong], np.array(case["output_tokens"])[strong]), case["name"]
    eng.close()


def test_qwen35_matches_cpu_oracle(built_libs, golden35):
    meta, _ = golden35
    oracle = Qwen35Oracle(Qwen35Config(**meta["config"]), load_safetensors(CKPT), num_pages=64)
    eng = make_engine(meta)
    for case in meta["cases"][1:4]:
        L = teacher_forced(eng, case)
        st = oracle.new_request()
        ref = [oracle.prefill(case["prompt_tokens"], st)]
        for tok in case["output_tokens"][:-1]:
            ref.append(oracle.batch_decode([tok], [st])[0])
        R = np.stack(ref)
        cos = (L * R).sum(-1) / np.linalg.norm(L, axis=-1) / np.linalg.norm(R, axis=-1)
        assert cos.min() > 0.9995 and np.abs(L - R).max() <= 0.4, (case["name"], cos.min(), np.abs(L - R).max())
    eng.close()


def test_qwen35_graph_eager_batch_and_handoff(built_libs, golden35):
    meta, _ = golden35
    prompts = [c["prompt_tokens"] for c in meta["cases"][:3]]
    runs = {}
    for graph in (True, False):
        # split_policy 0: the partition plan depends on the batch size, so bitwise batch invariance is a property of
        # the reference's non-partition call; graph == eager holds for both (checked with policy 1 below)
        eng = make_engine(meta, enable_graph=graph, split_policy=0)
        rids = [eng.new_request() for _ in prompts]
        toks = np.array([eng.prefill(r, p) for r, p in zip(rids, prompts)], np.int32)
        rows = []
        for _ in range(5):
            toks, lg = eng.decode(rids, toks, want_logits=True)
            rows.append(lg.copy())
        runs[graph] = np.stack(rows)
        if graph:
            # per-request decode on fresh requests vs the batched columns: the batch of 3 runs in bucket 4, whose
            # GEMMs are the skinny MFMA family (3..16 columns) while a single request runs the dot2 GEMV - same
            # tokens, logits within bf16 accumulation noise (the reference's cuBLAS switches kernels with N too)
            for i, p in enumerate(prompts):
                r = eng.new_request()
                t = eng.prefill(r, p)
                for step in range(5):
                    out, lg = eng.decode([r], [t], want_logits=True)
                    if step == 0:
                        assert int(out[0]) == int(np.argmax(runs[True][0, i]))
                        assert np.abs(lg[0] - runs[True][0, i]).max() <= 0.25, (i, step)
                    t = int(out[0])
                eng.drop_request(r)
            # inside one kernel family (1..2 columns, dot2 GEMV) batched == per-request bit for bit
            pair = [eng.new_request() for _ in prompts[:2]]
            tk = np.array([eng.prefill(r, p) for r, p in zip(pair, prompts[:2])], np.int32)
            solo_r = eng.new_request()
            solo_t = eng.prefill(solo_r, prompts[1])
            for step in range(4):
                tk, lg2 = eng.decode(pair, tk, want_logits=True)
                so, lg1 = eng.decode([solo_r], [solo_t], want_logits=True)
                assert np.array_equal(lg2[1].view(np.uint32), lg1[0].view(np.uint32)), step
                solo_t = int(so[0])
        eng.close()
    assert np.array_equal(runs[True].view(np.uint32), runs[False].view(np.uint32))      # graph == eager
    long_prompt = (meta["cases"][3]["prompt_tokens"] * 3)[:300]                          # >= 2 KV chunks -> split path
    outs = []
    for graph in (True, False):
        eng = make_engine(meta, enable_graph=graph, split_policy=1)
        r = eng.new_request()
        t = eng.prefill(r, long_prompt)
        rows = []
        for _ in range(4):
            o, lg = eng.decode([r], [t], want_logits=True)
            rows.append(lg[0].copy())
            t = int(o[0])
        outs.append(np.stack(rows))
        eng.close()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    # prefill hand-off: 70 tokens as 41 + 29 vs in one call
    eng = make_engine(meta)
    p = meta["cases"][2]["prompt_tokens"]
    r1, r2 = eng.new_request(), eng.new_request()
    _, one = eng.prefill(r1, p, want_logits=True)
    eng.prefill(r2, p[:41])
    _, two = eng.prefill(r2, p[41:], want_logits=True)
    assert eng.seq_len(r1) == eng.seq_len(r2) == len(p)
    cos = float((one * two).sum() / np.linalg.norm(one) / np.linalg.norm(two))
    assert cos > 0.999 and np.abs(one - two).max() < 0.75
    _, again = eng.prefill(eng.new_request(), p, want_logits=True)                      # rerun determinism
    assert np.array_equal(one.view(np.uint32), again.view(np.uint32))
    eng.close()


@pytest.mark.parametrize("split", [0, 1])
def test_qwen35_fused_decode_bitwise_equals_reference_sequence(built_libs, golden35, split, monkeypatch):
    """bs = 1 fused step (stacked-projection GEMVs with the norm / residual-add prologues and the SwiGLU epilogue in
    their Qwen3.5 rounding forms, single-launch conv step) == the reference-order op sequence, logits bit for bit,
    across prompts that end inside / across KV chunks and GDR chunks."""
    meta, _ = golden35
    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("PEGAINFER_Q35_DECODE_MODE", mode)
        eng = make_engine(meta, split_policy=split)
        rows = []
        for case in meta["cases"]:
            r = eng.new_request()
            t = eng.prefill(r, case["prompt_tokens"])
            for _ in range(6):
                o, lg = eng.decode([r], [t], want_logits=True)
                rows.append(lg[0].copy())
                t = int(o[0])
            eng.drop_request(r)
        runs.append(np.stack(rows))
        eng.close()
    assert np.array_equal(runs[0].view(np.uint32), runs[1].view(np.uint32))


def test_qwen35_scheduler_streams_equal_per_request_generation(built_libs, golden35):
    """The continuous-batching scheduler over the hybrid runtime (one prefill per admitted prompt, batched decode with
    per-slot recurrent state): overlapping requests produce exactly the tokens each prompt gets when generated alone
    (non-partition attention: batch invariance is bitwise there)."""
    from pegainfer_amd.scheduler import FINISHED, TOKEN, Scheduler
    meta, _ = golden35
    cases = meta["cases"]
    eng = make_engine(meta, split_policy=0, max_batch_size=4)
    s = Scheduler.over_engine(eng)
    arrivals = {0: [(cases[0]["prompt_tokens"], 5)], 1: [(cases[2]["prompt_tokens"], 7), (cases[1]["prompt_tokens"], 1)],
                3: [(cases[3]["prompt_tokens"], 4)]}
    streams, fin, it, rid_prompt = {}, 0, 0, {}
    while fin < 4:
        for prompt, mx in arrivals.get(it, []):
            rid_prompt[s.submit(prompt, mx, (0.0, -1, 1.0, True))] = (prompt, mx)
        assert s.step() >= 0, s.last_message()
        for rid, kind, tok, *_ in s.poll():
            if kind == TOKEN:
                streams.setdefault(rid, []).append(tok)
            elif kind == FINISHED:
                fin += 1
        it += 1
    s.close()
    for rid, (prompt, mx) in rid_prompt.items():
        r = eng.new_request()
        t = eng.prefill(r, prompt)
        alone = [t]
        for _ in range(mx - 1):
            t = int(eng.decode([r], [t])[0])
            alone.append(t)
        eng.drop_request(r)
        assert streams[rid] == alone, rid
    eng.close()


def test_qwen35_scheduler_sampling_and_logprobs(built_libs, golden35):
    """Round 4: the Qwen3.5 executor is no longer greedy-only.  A sampled request (T 0.8, top_k 5, top_p 0.9) draws every
    token from the top-k / top-p support of that step's logits (FlashInfer's Philox stream is parity-unpinned: the support is
    what can be checked), and a greedy request with logprobs = 3 carries TokenLogprobs equal to compute_logprobs_from_cpu
    over the logits the engine returns for the same prompt - prompt row and decode rows."""
    from oracle import ops as O
    from pegainfer_amd.scheduler import FINISHED, TOKEN, Scheduler
    meta, _ = golden35
    prompt = meta["cases"][1]["prompt_tokens"]
    eng = make_engine(meta, split_policy=0, max_batch_size=4)
    # expected logprobs from the engine's own logits
    r = eng.new_request()
    t, lg = eng.prefill(r, prompt, want_logits=True)
    want = [O.compute_logprobs(lg, t, 3)]
    toks = [t]
    for _ in range(3):
        o, lgd = eng.decode([r], [toks[-1]], want_logits=True)
        toks.append(int(o[0]))
        want.append(O.compute_logprobs(lgd[0], toks[-1], 3))
    eng.drop_request(r)
    s = Scheduler.over_engine(eng)
    a = s.submit(prompt, 4, (0.0, -1, 1.0, True), logprobs=3)
    b = s.submit(prompt, 4, (0.8, 5, 0.9, True))
    ev = []
    while s.step() != 0:
        ev += s.poll()
    ev += s.poll()
    s.close()
    got = [e for e in ev if e[0] == a and e[1] == TOKEN]
    assert [e[2] for e in got] == toks
    for e, w in zip(got, want):
        assert e[7] is not None and abs(e[7][0] - w[0]) < 2e-5 and [i for i, _ in e[7][1]] == [i for i, _ in w[1]]
    sampled = [e[2] for e in ev if e[0] == b and e[1] == TOKEN]
    assert len(sampled) == 4 and all(e[7] is None for e in ev if e[0] == b)
    # every sampled token lies in the support of the logits the same engine produces on that token stream
    r = eng.new_request()
    _, lg = eng.prefill(r, prompt, want_logits=True)
    for i, tk in enumerate(sampled):
        keep = O.top_k_top_p_support(O.logits_to_probs(lg if i == 0 else lg[0], 1.0 / 0.8), 5, 0.9)
        assert keep[tk], (i, tk)
        _, lg = eng.decode([r], [tk], want_logits=True)
    eng.drop_request(r)
    eng.close()


def test_qwen35_native_loader_equals_python_loader(built_libs, golden35):
    from pegainfer_amd.qwen35 import Qwen35Engine
    meta, _ = golden35
    prompt = meta["cases"][2]["prompt_tokens"]
    a = make_engine(meta)
    b = Qwen35Engine(meta["config"], num_kv_pages=128, max_batch_size=4).load_safetensors_native(CKPT)
    _, la = a.prefill(a.new_request(), prompt, want_logits=True)
    _, lb = b.prefill(b.new_request(), prompt, want_logits=True)
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
    a.close()
    b.close()


def test_qwen35_split_kv_after_batch_shrink_equals_unsplit(built_libs, golden35):
    """the partition-KV launch covered bs * 64 slots while only the first plan.slots entries of the
    slot arrays are refreshed per step, so after a batch shrink stale `valid` slots of the larger batch (naming
    requests that are still live) ran again, bumped merge counters and could fire the in-launch merge early.  The launch
    now covers exactly plan.slots.  Scenario sized for this checkpoint (1 kv head): 32 long requests -> 16 chunks per
    request, entries 14 b + c; then the first 20 of them -> 260 refreshed slots, old launch 1280 of which 260..447
    were stale and named requests 18..31.  Every logit of the shrunk batch must match the un-split engine within the
    split-vs-non-partition bar (partials are bf16-rounded before the merge)."""
    meta, _ = golden35
    rng = np.random.default_rng(35)
    V = meta["config"]["vocab_size"]
    prompts = [rng.integers(0, V, 1090 + 3 * i).tolist() for i in range(32)]
    outs = {}
    for split in (1, 0):
        eng = make_engine(meta, split_policy=split, num_kv_pages=32 * 72 + 8, max_batch_size=32, max_positions=2048)
        rids = [eng.new_request() for _ in prompts]
        toks = [eng.prefill(r, p) for r, p in zip(rids, prompts)]
        rows = []
        for step in range(3):                                   # 32 requests, long context
            t, lg = eng.decode(rids, toks, want_logits=True)
            toks = [int(x) for x in lg.argmax(-1)] if split == 1 else outs[1]["toks"][step]
            rows.append(lg)
            outs.setdefault(split, {}).setdefault("toks", []).append(toks)
        keep, ktoks = rids[:20], toks[:20]
        for step in range(4):                                   # shrink to 20: stale slots would name requests 18, 19
            t, lg = eng.decode(keep, ktoks, want_logits=True)
            ktoks = [int(x) for x in lg.argmax(-1)] if split == 1 else outs[1]["ktoks"][step]
            rows.append(lg)
            outs[split].setdefault("ktoks", []).append(ktoks)
        outs[split]["rows"] = rows
        eng.close()
    for a, b in zip(outs[1]["rows"], outs[0]["rows"]):
        assert np.abs(a - b).max() <= 0.25, np.abs(a - b).max()


def test_qwen35_decode_greedy_chain_equals_step_by_step(built_libs, golden35, monkeypatch):
    """pegainfer_qwen35_decode_greedy_chain (n greedy steps back to back, the token handed over on the device, one host
    synchronisation) == n calls of pegainfer_qwen35_decode: tokens, request state and the next step's logits, bit for bit -
    fused bs = 1 decode and the reference-order sequence, one request and a batch of two (recurrent state + KV advance on
    the device across the chain)."""
    meta = golden35[0]
    cases = meta["cases"]
    for q35_mode in (1, 0):
        monkeypatch.setenv("PEGAINFER_Q35_DECODE_MODE", str(q35_mode))
        for prompts in ([cases[3]["prompt_tokens"]], [cases[2]["prompt_tokens"], cases[0]["prompt_tokens"]]):
            outs = []
            for chained in (False, True):
                eng = make_engine(meta, max_batch_size=2)
                rids = [eng.new_request() for _ in prompts]
                toks = np.array([eng.prefill(r, p) for r, p in zip(rids, prompts)], np.int32)
                if chained:
                    seq = eng.decode_greedy_chain(rids, toks, 5)
                    seq = np.concatenate([seq, eng.decode_greedy_chain(rids, seq[-1], 19)])
                else:
                    seq, t = [], toks
                    for _ in range(24):
                        t = eng.decode(rids, t)
                        seq.append(t.copy())
                    seq = np.stack(seq)
                t1, l1 = eng.decode(rids, seq[-1], want_logits=True)
                outs.append((seq, [eng.seq_len(r) for r in rids], t1, l1))
                eng.close()
            a, b = outs
            assert np.array_equal(a[0], b[0]) and a[1] == b[1], (q35_mode, len(prompts))
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
