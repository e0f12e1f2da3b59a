"""GPU parity at the dimensions the bench quotes ("What's weak: parity").

  (a) whole-model HIP-vs-oracle at Qwen3-4B REAL dims (hidden 2560, 32/8 heads, I 9728, V 151 936,
      tied lm_head) with 2 layers of seeded synthetic weights: 1024-token prefill + 8 decode steps,
      decode_mode 0 and 1, split policy 0 and 1, graph on/off, bs 1 / 8 / 32 - every kernel variant the
      benchmarked path takes (K = 2560 / 4096 / 9728 GEMV families, persistent grids, KSPLIT 4,
      split-slot attention grid at 8 kv heads, skinny MFMA and the 7-launch mid-batch layer);
  (b) op-level GEMM / GEMV at the Qwen3-8B and Qwen3.5 shapes (K 4096 / 12288, M 24576, lm_head
      M = 151 936 x 4096 and 248 320 x 2560) at T in {1, 8, 32, 1024};
  (c) batch_prefill_paged at 2048 and 4096 tokens, Hq 32 / Hkv 8, against the oracle.

The oracle GEMM accumulates in fp32 here (ops.GEMM_ACCUM, plain sgemm - what cuBLAS COMPUTE_32F does);
float64 copies of 1.2 G parameters per call would dominate the run time.  Tolerances are stated per test.
"""
import os

import numpy as np
import pytest

from conftest import from_dev, to_dev
from oracle import ops as O
from oracle.bf16 import bf16_bits, bf16_from_bits, bf16_round
from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle, synthetic_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(built_libs):
    import pegainfer_amd.ops as P
    return P


def rnd(rng, *shape, scale=1.0):
    return bf16_round((rng.standard_normal(shape, dtype=np.float32) * np.float32(scale)))


# ------------------------------------------------------------------ (b) GEMM shapes of configs[2] / configs[3]
def gemm_check_rows(P, M, T, K, nrows=None, seed=0):
    """HIP gemm (all M x T outputs computed) vs a float64 reference; for the big T = 1024 shapes the reference is
    evaluated on `nrows` seeded output rows (every token column of those rows).  Bar = the envelope of
    tests/test_gpu_ops.py::gemm_check: |diff| <= 0.05 * 2^-7 * sum|w||x| + 2^-8 |y| (fp32 accumulation order +
    one bf16 store)."""
    rng = np.random.default_rng(seed + M + T + K)
    W, X = rnd(rng, M, K, scale=0.05), rnd(rng, T, K)
    got = from_dev(P.gemm(to_dev(W), to_dev(X)))
    rows = np.arange(M) if nrows is None or nrows >= M else np.sort(rng.choice(M, size=nrows, replace=False))
    Wr = W[rows].astype(np.float64)
    X64 = X.astype(np.float64)
    ref = X64 @ Wr.T
    env = np.abs(X64) @ np.abs(Wr).T
    tol = env * 2.0 ** -7 * 0.05 + np.abs(ref) * 2.0 ** -8 + 1e-6
    err = np.abs(got[:, rows] - ref)
    assert np.all(err <= tol), (M, T, K, float((err / tol).max()))


QWEN3_8B_SITES = [(6144, 4096), (4096, 4096), (24576, 4096), (4096, 12288)]   # qkv, o, gate|up, down  (M, K)
QWEN35_SITES = [(18432, 2560), (2560, 9216), (8192, 2560), (2560, 4096)]      # gate|up, down, qkv(full-attn q|gate), o


@pytest.mark.parametrize("T", [1, 8, 32, 1024])
@pytest.mark.parametrize("M,K", QWEN3_8B_SITES + QWEN35_SITES)
def test_gemm_qwen3_8b_and_qwen35_layer_shapes(P, M, K, T):
    gemm_check_rows(P, M, T, K, nrows=1536 if T == 1024 else None)


@pytest.mark.parametrize("T", [1, 8, 32])
@pytest.mark.parametrize("M,K", [(151936, 4096), (248320, 2560)])
def test_gemm_lm_head_qwen3_8b_and_qwen35(P, M, K, T):
    """Untied Qwen3-8B lm_head (K = 4096) and the Qwen3.5 vocabulary (M = 248 320): 4096 seeded rows + the last
    64 rows (tail workgroups) checked for every token column."""
    rng = np.random.default_rng(M + T)
    W, X = rnd(rng, M, K, scale=0.05), rnd(rng, T, K)
    got = from_dev(P.gemm(to_dev(W), to_dev(X)))
    rows = np.unique(np.concatenate([rng.choice(M, size=4096, replace=False), np.arange(M - 64, M), np.arange(64)]))
    Wr, X64 = W[rows].astype(np.float64), X.astype(np.float64)
    ref, env = X64 @ Wr.T, np.abs(X64) @ np.abs(Wr).T
    tol = env * 2.0 ** -7 * 0.05 + np.abs(ref) * 2.0 ** -8 + 1e-6
    err = np.abs(got[:, rows] - ref)
    assert np.all(err <= tol), float((err / tol).max())
    assert np.isfinite(got).all()


def test_gemv_fused_qwen3_8b_shapes_equal_unfused_sequence(P):
    """The fused decode GEMV forms (add+RMSNorm prologue, SwiGLU epilogue) at Qwen3-8B shapes == the reference-named
    op sequence over the same GEMV, bit for bit (the fused path of bench.py --model qwen3-8b)."""
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(8)
    H, I = 4096, 12288
    for T in (1, 2, 8):
        hid, res = rnd(rng, T, H, scale=2), rnd(rng, T, H, scale=0.5)
        g = bf16_round(1 + rnd(rng, H, scale=0.1))
        Wgu, Wd = rnd(rng, 2 * I, H, scale=0.03), rnd(rng, H, I, scale=0.03)
        s = torch.cuda.current_stream().cuda_stream
        # reference-named sequence
        h_ref = to_dev(hid)
        n_ref = torch.empty_like(h_ref)
        P.fused_add_rms_norm_batch_into(h_ref, to_dev(res), to_dev(g), 1e-6, n_ref)
        act_ref = torch.empty((T, I), dtype=torch.bfloat16, device="cuda")
        P.silu_mul_fused_batch_into(P.gemm(to_dev(Wgu), n_ref), act_ref)
        y_ref = P.gemm(to_dev(Wd), act_ref)
        # fused forms
        h_in, h_out = to_dev(hid), torch.empty((T, H), dtype=torch.bfloat16, device="cuda")
        act = torch.empty((T, I), dtype=torch.bfloat16, device="cuda")
        keep = [to_dev(Wgu), to_dev(res), to_dev(g), to_dev(Wd)]
        assert ffi.lib().pegainfer_gemv_fused(keep[0].data_ptr(), h_in.data_ptr(), act.data_ptr(), 2 * I, T, H,
                                              keep[1].data_ptr(), keep[2].data_ptr(), h_out.data_ptr(), 1e-6, I, s) == 0
        y = torch.empty((T, H), dtype=torch.bfloat16, device="cuda")
        assert ffi.lib().pegainfer_gemv_fused(keep[3].data_ptr(), act.data_ptr(), y.data_ptr(), H, T, I, None, None,
                                              None, 0.0, 0, s) == 0
        assert np.array_equal(bf16_bits(from_dev(h_out)), bf16_bits(from_dev(h_ref))), T
        assert np.array_equal(bf16_bits(from_dev(act)), bf16_bits(from_dev(act_ref))), T
        assert np.array_equal(bf16_bits(from_dev(y)), bf16_bits(from_dev(y_ref))), T
        # and against the oracle
        eh, en = O.fused_add_rms_norm(hid, res, g, 1e-6)
        ea = O.silu_mul_fused(O.gemm(Wgu, en), I)
        assert np.abs(from_dev(act) - ea).max() <= 2.0 ** -6 * max(1.0, np.abs(ea).max())


# ------------------------------------------------------------------ (c) long prefill attention
def make_paged(rng, lens, Hkv=8, D=128, layers=2, ps=16, extra_pages=3):
    lay = O.PagedKvLayout(layers, Hkv, D, ps)
    need = [-(-n // ps) for n in lens]
    total = sum(need) + extra_pages
    perm = rng.permutation(total)
    pages, indptr, last, c = [], [0], [], 0
    for n, k in zip(lens, need):
        pages.extend(perm[c:c + k].tolist()); c += k
        indptr.append(len(pages))
        last.append(0 if n == 0 else ((n - 1) % ps) + 1)
    kv = bf16_round(rng.standard_normal(total * lay.page_stride, dtype=np.float32))
    return lay, kv, np.int32(pages), np.int32(indptr), np.int32(last)


@pytest.mark.parametrize("seq_lens,starts", [([2048], [0]), ([4096], [0]), ([1500, 548], [0, 300]), ([1024], [3072])])
def test_batch_prefill_paged_long(P, seq_lens, starts):
    """2048 / 4096-token causal prefill at Hq 32 / Hkv 8 / D 128 (the regime TTFT is quoted in), a 2-request batch
    with a chunked continuation, and a 1024-token chunk on top of 3072 cached tokens; every output element against
    the float64 oracle.  Bar: 2 * (2^-7 max|o| + 1e-3) - P is rounded to bf16 before the PV MFMA."""
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(sum(seq_lens) + sum(starts))
    lens = [s + n for s, n in zip(starts, seq_lens)]
    lay, kv, pages, indptr, last = make_paged(rng, lens)
    T = sum(seq_lens)
    q = rnd(rng, T, 32 * 128)
    page_lists = [pages[indptr[i]:indptr[i + 1]].tolist() for i in range(len(lens))]
    plan = P.PrefillPagedPlan(page_lists, last.tolist(), starts, seq_lens, 32, 8, 128, 0)
    oplan = O.prefill_paged_plan(page_lists, last.tolist(), starts, seq_lens, 32, 8, 128, 0)
    for key in ("batch_indices", "positions", "q_indptr", "request_indices", "qo_tile_indices", "kv_chunk_size"):
        assert np.array_equal(getattr(plan, key + "_d").cpu().numpy(), oplan[key]), key
    L = P.PagedKvLayout(2, 8, 128, 16)
    out = torch.zeros((T, 32 * 128), dtype=torch.bfloat16, device="cuda")
    qd, kvd = to_dev(q), to_dev(kv)
    sm = 1.0 / np.sqrt(128.0)
    rc = ffi.lib().batch_prefill_paged_cuda_with_cta_tile_q(
        qd.data_ptr(), out.data_ptr(), kvd.data_ptr(), L.layer_stride, L.layer_stride + L.kv_block_len,
        plan.page_indices_d.data_ptr(), plan.page_indptr_d.data_ptr(), plan.last_page_len_d.data_ptr(),
        plan.q_indptr_d.data_ptr(), plan.request_indices_d.data_ptr(), plan.qo_tile_indices_d.data_ptr(),
        plan.kv_tile_indices_d.data_ptr(), plan.kv_chunk_size_d.data_ptr(), plan.total_num_rows_d.data_ptr(),
        32, 8, 128, 16, T, len(lens), plan.num_tiles, L.page_stride, sm, plan.cta_tile_q,
        torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = O.batch_prefill_paged(q, kv, lay, 1, pages, indptr, last, oplan["q_indptr"], 32, sm)
    err = np.abs(from_dev(out) - ref).max()
    assert err <= 2 * (2.0 ** -7 * np.abs(ref).max() + 1e-3), err


@pytest.mark.parametrize("ps,seq_lens,starts", [(16, [300, 77], [0, 41]),      # 16-token pages: the full-tile fast path
                                                 (32, [300, 77], [0, 41]),      # other power of two: shift / mask path
                                                 (12, [300, 77], [0, 41]),      # not a power of two: division path
                                                 (2, [4700], [0]),              # 2350 pages: the 2048-entry LDS window reloads
                                                 (64, [300, 77], [0, 41]),      # page == KV tile: LDS-DMA pieces, per-tile page id
                                                 (128, [300, 77], [0, 41]),     # page > KV tile: the DMA form's general addressing
                                                 (128, [4700], [0]),            # ... on paired 128-row tiles
                                                 (4, [4700], [0])])             # a DMA piece (4 rows) = one page, paired tiles
def test_batch_prefill_paged_page_sizes(P, ps, seq_lens, starts):
    """The prefill kernel's three page-addressing forms and the page-id window in LDS (a request with more than 2048
    pages - 32 k tokens at the usual page size, here 4700 tokens on 2-token pages) against the oracle."""
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(ps * 1000 + sum(seq_lens))
    lens = [s + n for s, n in zip(starts, seq_lens)]
    lay, kv, pages, indptr, last = make_paged(rng, lens, ps=ps)
    T = sum(seq_lens)
    q = rnd(rng, T, 32 * 128)
    page_lists = [pages[indptr[i]:indptr[i + 1]].tolist() for i in range(len(lens))]
    oplan = O.prefill_paged_plan(page_lists, last.tolist(), starts, seq_lens, 32, 8, 128, 64)   # page-size agnostic
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device="cuda")
    L = P.PagedKvLayout(2, 8, 128, ps)
    out = torch.zeros((T, 32 * 128), dtype=torch.bfloat16, device="cuda")
    qd, kvd = to_dev(q), to_dev(kv)
    keep = [i32(pages), i32(indptr), i32(last), i32(oplan["q_indptr"]), i32(oplan["request_indices"]),
            i32(oplan["qo_tile_indices"]), i32(oplan["kv_tile_indices"]), i32(oplan["kv_chunk_size"]), i32([T])]
    sm = 1.0 / np.sqrt(128.0)
    rc = ffi.lib().batch_prefill_paged_cuda_with_cta_tile_q(
        qd.data_ptr(), out.data_ptr(), kvd.data_ptr(), L.layer_stride, L.layer_stride + L.kv_block_len,
        keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
        keep[5].data_ptr(), keep[6].data_ptr(), keep[7].data_ptr(), keep[8].data_ptr(), 32, 8, 128, ps, T, len(lens),
        int(oplan["num_tiles"]), L.page_stride, sm, 64, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = O.batch_prefill_paged(q, kv, lay, 1, pages, indptr, last, oplan["q_indptr"], 32, sm)
    err = np.abs(from_dev(out) - ref).max()
    assert err <= 2 * (2.0 ** -7 * np.abs(ref).max() + 1e-3), (ps, err)


# ------------------------------------------------------------------ (a) Qwen3-4B real dims, 2 layers
CFG2 = dict(hidden_size=2560, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
            intermediate_size=9728, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True,
            max_position_embeddings=4096)
PROMPT_1024 = [100 + (i % 1000) for i in range(1024)]      # reference decode_heavy prompt (bench_serving.rs:37-43)
N_DECODE = 8


def _prompt(rng, n):
    return rng.integers(0, CFG2["vocab_size"], n).tolist()


@pytest.fixture(scope="module")
def real2():
    """Seeded 2-layer Qwen3-4B-shaped checkpoint + the oracle's logits for three workloads (teacher-forced on the
    oracle's own greedy tokens so every engine configuration is compared on the same token stream)."""
    cfg = Qwen3Config(**CFG2)
    w = synthetic_weights(cfg, seed=1234, std=0.02)
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        rng = np.random.default_rng(77)
        work = {
            "bs1": [PROMPT_1024],
            # round 5: two requests in the fused attention + o_proj launch.  "bs2": <= 9 x 64 tokens each, where the capped
            # plan of the fused form (10 of the launch's 20 chunks per request) and the un-capped plan of decode_mode 0
            # (16) cut the KV into the SAME 64-token chunks, so the two modes are comparable bit for bit; "bs2long":
            # 128-token chunks, vs the oracle
            "bs2": [_prompt(rng, 540), _prompt(rng, 300)],
            "bs2long": [PROMPT_1024, _prompt(rng, 777)],
            "bs8": [_prompt(rng, n) for n in (700, 333, 129, 64, 17, 16, 5, 1)],
            "bs32": [_prompt(rng, n) for n in [600] + rng.integers(1, 97, 31).tolist()],
        }
        steps = {"bs1": N_DECODE, "bs2": 6, "bs2long": 4, "bs8": 4, "bs32": 2}
        ref = {}
        for name, prompts in work.items():
            orc = Qwen3Oracle(cfg, w, num_pages=512, rope_positions=4096)
            sts = [KvState() for _ in prompts]
            pf = np.stack(orc.batch_prefill(prompts, sts))                    # [bs, V]
            toks = [pf.argmax(-1)]
            dec = []
            for _ in range(steps[name]):
                lg = orc.batch_decode(toks[-1].tolist(), sts)
                dec.append(lg)
                toks.append(lg.argmax(-1))
            ref[name] = dict(prompts=prompts, prefill=pf, decode=np.stack(dec), tokens=np.stack(toks))
    finally:
        O.GEMM_ACCUM = old
    state = {k: bf16_bits(v) for k, v in w.items()}
    return cfg, state, ref


def _engine(state, **kw):
    from pegainfer_amd.qwen3 import Qwen3Engine
    kw.setdefault("num_kv_pages", 512)
    kw.setdefault("max_batch_size", 32)
    return Qwen3Engine(CFG2, **kw).load_state(state)


def _close(a, b):
    """logits rows vs oracle rows: (min cosine, max |diff| / max |ref|)."""
    a, b = a.reshape(-1, a.shape[-1]).astype(np.float64), b.reshape(-1, b.shape[-1]).astype(np.float64)
    cos = (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)
    return float(cos.min()), float(np.abs(a - b).max() / np.abs(b).max())


def _run(eng, case):
    rids = [eng.new_request() for _ in case["prompts"]]
    _, lg = eng.prefill(rids, case["prompts"], return_logits=True)
    pf = bf16_from_bits(lg)
    dec, paths = [], []
    for step in range(case["decode"].shape[0]):
        _, lg = eng.decode(rids, case["tokens"][step], return_logits=True)
        dec.append(bf16_from_bits(lg))
        paths.append(eng.last_attention_path())
    for r in rids:
        eng.drop_request(r)
    return pf, np.stack(dec), paths


# Bars.  Both sides accumulate in fp32 and round every activation to bf16 at the same points; they differ in
# summation order only.  Two layers + lm_head at real widths: cosine > 0.9998 and max |dlogit| <= 2 % of the
# largest |logit| (3 bf16 ulp of the top logit), the bar of tests/test_gpu_model.py::test_matches_cpu_oracle
# (0.4 on a scale of 16 = 2.5 %) restated relative to the logit scale of this checkpoint.
COS_MIN, REL_MAX = 0.9998, 0.02


@pytest.mark.parametrize("mode,policy,graph", [(0, 0, True), (0, 1, True), (1, 0, True), (1, 1, True), (1, 1, False)])
def test_real_dims_bs1_prefill_1024_then_decode(built_libs, real2, mode, policy, graph):
    _, state, ref = real2
    eng = _engine(state, decode_mode=mode, split_policy=policy, enable_graph=graph, max_batch_size=8)
    pf, dec, paths = _run(eng, ref["bs1"])
    eng.close()
    c, r = _close(pf, ref["bs1"]["prefill"])
    assert c > COS_MIN and r <= REL_MAX, ("prefill", c, r)
    c, r = _close(dec, ref["bs1"]["decode"])
    assert c > COS_MIN and r <= REL_MAX, ("decode", c, r)
    assert all(p == 1 for p in paths)     # ctx >= 1024 at bs 1: both policies partition the KV (batch_decode_buffers.rs:229)
    # greedy token of every step == the oracle's wherever the oracle's top-1 margin exceeds the logit bar
    for got, want in ((pf, ref["bs1"]["prefill"]), (dec.reshape(-1, dec.shape[-1]), ref["bs1"]["decode"].reshape(-1, dec.shape[-1]))):
        srt = np.sort(want, axis=-1)
        strong = (srt[:, -1] - srt[:, -2]) > 2 * REL_MAX * np.abs(want).max()
        assert np.array_equal(got.argmax(-1)[strong], want.argmax(-1)[strong])


@pytest.mark.parametrize("name,mode,policy", [("bs2", 1, 1), ("bs2long", 1, 1), ("bs2long", 1, 0), ("bs8", 0, 1), ("bs8", 1, 1), ("bs8", 1, 0), ("bs32", 0, 1), ("bs32", 1, 1)])
def test_real_dims_batched_prefill_and_decode(built_libs, real2, name, mode, policy):
    """bs 8 = skinny-MFMA fused decode family, bs 32 = the 7-launch mid-batch layer (decode_mode 1) or the reference
    op sequence over tiled GEMMs (decode_mode 0); ragged prompt lengths 1..700."""
    _, state, ref = real2
    eng = _engine(state, decode_mode=mode, split_policy=policy)
    pf, dec, _ = _run(eng, ref[name])
    eng.close()
    c, r = _close(pf, ref[name]["prefill"])
    assert c > COS_MIN and r <= REL_MAX, ("prefill", c, r)
    c, r = _close(dec, ref[name]["decode"])
    assert c > COS_MIN and r <= REL_MAX, ("decode", c, r)


def test_real_dims_fused_path_bit_identical_to_reference_sequence(built_libs, real2):
    """decode_mode 1 (5 launches per layer) == decode_mode 0 (the reference's 14) in every logit bit at real dims,
    bs 1 and bs 8, both split policies - the invariant tests/test_gpu_fused.py holds on the tiny checkpoint."""
    _, state, ref = real2
    for name in ("bs1", "bs2", "bs8"):
        outs = []
        for mode in (0, 1):
            eng = _engine(state, decode_mode=mode, split_policy=1)
            _, dec, _ = _run(eng, ref[name])
            eng.close()
            outs.append(bf16_bits(dec))
        assert np.array_equal(outs[0], outs[1]), name


def test_real_dims_two_requests_fused_attention_oproj(built_libs, real2, monkeypatch):
    """bs 2 at the real widths (round 5): attention + o_proj of BOTH requests in one launch (nine KV
    chunks each, the padding-slot workgroups hold the o_proj rows, per-(request, head group) arrival counters) must give
    the bits of the reference op sequence on the same plan (decode_mode 0) and of the eager launches; each request decoded
    alone agrees within the bf16 bound (batch_decode.rs:505-606 is a token-level check in the reference too)."""
    _, state, ref = real2
    case = ref["bs2"]
    runs = {}
    for tag, kw in (("fused", dict(decode_mode=1)), ("eager", dict(decode_mode=1, enable_graph=False)), ("reference", dict(decode_mode=0))):
        eng = _engine(state, split_policy=1, max_batch_size=2, **kw)
        _, dec, paths = _run(eng, case)
        runs[tag] = bf16_bits(dec)
        assert all(p == 1 for p in paths), tag          # the KV is partitioned: the fused launch's plan
        eng.close()
    assert np.array_equal(runs["fused"], runs["reference"]) and np.array_equal(runs["fused"], runs["eager"])
    # each request alone, teacher-forced on the same tokens: its column of the batch within the bf16 bound (NOT bitwise: a
    # lone request is cut into 13-16 chunks of 64 tokens, in the pair into 7-8 of 128 - other partials, other roundings;
    # bit-exact batch == single holds where the KV is not partitioned, tests/test_gpu_model.py)
    scale = float(np.abs(case["decode"]).max())
    for col in range(2):
        eng = _engine(state, decode_mode=1, split_policy=1, max_batch_size=2)
        rid = eng.new_request()
        eng.prefill([rid], [case["prompts"][col]])
        for step in range(case["decode"].shape[0]):
            _, lg = eng.decode([rid], case["tokens"][step][col:col + 1], return_logits=True)
            d = np.abs(bf16_from_bits(lg[0]) - bf16_from_bits(runs["fused"][step][col])).max()
            assert d <= REL_MAX * scale, (col, step, float(d), scale)
        eng.close()


def test_real_dims_short_prompt_prefill_path(built_libs, real2, monkeypatch):
    """Prompts of <= 16 tokens take the 6-launch prefill layer (stacked q|k|v GEMV with the layer-input norm in its
    prologue, "add, then norm" between layers, fused gate_up) - at the REAL widths: K = 2560 register-staged prologue of
    the skinny kernel at 5 / 16 columns, the dot2 kernel at 1 / 3 columns, 6144-row stacked output.  Bit-identical to the
    reference op sequence 1:1 (PEGAINFER_PREFILL_SHORT=0) and inside the oracle bar."""
    cfg, state, _ = real2
    rng = np.random.default_rng(99)
    groups = [[_prompt(rng, n)] for n in (1, 3, 5, 16)] + [[_prompt(rng, 7), _prompt(rng, 2), _prompt(rng, 6)]]
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        want = []
        for g in groups:
            orc = Qwen3Oracle(cfg, {k: bf16_from_bits(v) for k, v in state.items()}, num_pages=64, rope_positions=4096)
            want.append(np.stack(orc.batch_prefill(g, [KvState() for _ in g])))
    finally:
        O.GEMM_ACCUM = old
    runs = []
    for short in ("0", "1"):
        monkeypatch.setenv("PEGAINFER_PREFILL_SHORT", short)
        eng = _engine(state, decode_mode=1, max_batch_size=4)
        rows = []
        for g in groups:
            rids = [eng.new_request() for _ in g]
            tok, lg = eng.prefill(rids, g, return_logits=True)
            rows.append(lg.copy())
            _, lg = eng.decode(rids, tok, return_logits=True)     # the KV the short path appended
            rows.append(lg.copy())
            for r in rids:
                eng.drop_request(r)
        eng.close()
        runs.append(rows)
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    for g, ref, got in zip(groups, want, runs[1][0::2]):
        c, r = _close(bf16_from_bits(got), ref)
        assert c > COS_MIN and r <= REL_MAX, ([len(p) for p in g], c, r)


def test_real_dims_fused_attention_oproj_expired_wait_reruns_the_step(built_libs, real2):
    """The o_proj workgroups of the fused attention + o_proj launch wait for the attention rows with a BOUND; when it expires
    the launch's output is void, the status word says so, and the host re-runs the step on the two stand-alone launches in the
    same call (disabling the form).  PEGAINFER_OPROJ_WAIT_TICKS=1 makes the bound expire at once - in a subprocess, the
    knob is read once per process - for a single decode call, for two requests, and inside a greedy chain: the tokens and
    logits must be those of an engine that never had the form (the re-run restores the token inputs the failed attempt's
    top-1 overwrote)."""
    import subprocess
    import sys
    import tempfile
    _, state, ref = real2
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from pegainfer_amd.qwen3 import Qwen3Engine
cfg = %r
st = np.load(sys.argv[1])
state = {k: st[k] for k in st.files}
out = {}
for nb in (1, 2):
    eng = Qwen3Engine(cfg, num_kv_pages=256, max_batch_size=2, decode_mode=1, split_policy=1).load_state(state)
    prompts = [[100 + (i %% 1000) for i in range(520)], [7 + (i %% 900) for i in range(300)]][:nb]
    rids = [eng.new_request() for _ in prompts]
    toks = eng.prefill(rids, prompts)
    if nb == 1:      # the bound expires inside a single decode call ...
        toks, lg = eng.decode(rids, toks, return_logits=True)
    else:            # ... or inside a chain (its re-run restarts the whole chain on the two launches)
        first = eng.decode_greedy_chain(rids, toks, 3)
        toks, lg = first[-1], first
    msg = eng.lib.pegainfer_qwen3_last_error(eng.h).decode()
    chain = eng.decode_greedy_chain(rids, toks, 5)
    t2, lg2 = eng.decode(rids, chain[-1], return_logits=True)
    out["tok%%d" %% nb], out["lg%%d" %% nb], out["chain%%d" %% nb], out["lg2_%%d" %% nb] = toks, lg, chain, lg2
    out["msg%%d" %% nb] = np.array(msg)
    eng.close()
np.savez(sys.argv[2], **out)
''' % (CFG2,)
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, "state.npz"), **state)
        res = {}
        for tag, env in (("forced", {"PEGAINFER_OPROJ_WAIT_TICKS": "1"}), ("never", {"PEGAINFER_ATTN_OPROJ": "0"})):
            out = os.path.join(d, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, os.path.join(d, "state.npz"), out], env=dict(os.environ, **env),
                               cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res[tag] = np.load(out)
    for nb in (1, 2):
        assert "bounded wait" in str(res["forced"]["msg%d" % nb]), str(res["forced"]["msg%d" % nb])      # the re-run really happened
        # the plans differ (the form's chunk cap vs none) only beyond 9 x 64 tokens: these prompts stay below, so bit for bit
        for k in ("tok%d", "lg%d", "chain%d", "lg2_%d"):
            assert np.array_equal(res["forced"][k % nb], res["never"][k % nb]), (nb, k)


# ------------------------------------------------------------------ (d) stream-K over 256 x 256 tiles (round 6)
@pytest.fixture
def streamk_on():
    """the stream-K route is off by default since it measured slower (gemm256.h); these tests keep the kernel correct"""
    from pegainfer_amd import ffi
    ffi.lib().pegainfer_debug_streamk(1)
    yield
    ffi.lib().pegainfer_debug_streamk(-1)


def _route(M, T, K, silu_I=0):
    from pegainfer_amd import ffi
    out = np.zeros(3, np.int32)
    assert ffi.lib().pegainfer_debug_gemm_route(M, T, K, silu_I, out.ctypes.data) == 0
    return int(out[0])


@pytest.mark.parametrize("M,K,T", [(19456, 2560, 512), (19456, 2560, 1024), (19456, 2560, 1100), (19456, 2560, 2048),
                                   (6144, 2560, 2048), (24576, 4096, 768), (18432, 2560, 1024)])
def test_streamk_gemm_against_float64(P, streamk_on, M, K, T):
    """The persistent stream-K launch (kind 258: teams of workgroups over (row tile, K-tile pair) units, fp32 partials
    published through the split-K workspace, the owner adds them in ascending K): every shape whose 256 x 256 tiling
    leaves >= 8 % of a CU round idle - gate_up of the three configs at 512 ... 2048 tokens (ragged T too), the stacked
    qkv at 2048 - against float64 on 1536 seeded rows, the GEMM envelope of the other routes; twice (same bits: the
    K order is fixed per shape), with another stream-K shape in between (owners clear the flags they consumed)."""
    assert _route(M, T, K) == 258, _route(M, T, K)
    rng = np.random.default_rng(M + T + K)
    W, X = rnd(rng, M, K, scale=0.05), rnd(rng, T, K)
    Wd, Xd = to_dev(W), to_dev(X)
    got = from_dev(P.gemm(Wd, Xd))
    rows = np.sort(rng.choice(M, size=1536, replace=False))
    Wr, X64 = W[rows].astype(np.float64), X.astype(np.float64)
    ref, env = X64 @ Wr.T, np.abs(X64) @ np.abs(Wr).T
    tol = env * 2.0 ** -7 * 0.05 + np.abs(ref) * 2.0 ** -8 + 1e-6
    err = np.abs(got[:, rows] - ref)
    assert np.all(err <= tol), (M, T, K, float((err / tol).max()))
    assert np.isfinite(got).all()
    other = P.gemm(to_dev(rnd(rng, 19456, 2560, scale=0.05)), to_dev(rnd(rng, 600, 2560)))     # 3 x 76 = 228 tiles: stream-K
    assert np.isfinite(from_dev(other)).all()
    again = from_dev(P.gemm(Wd, Xd))
    assert np.array_equal(bf16_bits(again), bf16_bits(got))


@pytest.mark.parametrize("I,K,T,rounded", [(9728, 2560, 1024, False), (9728, 2560, 520, False), (12288, 4096, 1024, False),
                                           (9216, 2560, 1024, True)])
def test_streamk_swiglu_gemm_against_the_pair(P, streamk_on, I, K, T, rounded):
    """gate_up GEMM with SwiGLU in the epilogue on the stream-K route against the oracle's pair gemm -> silu_mul on float64
    sums: gate and up rounded to bf16 before the activation (fused_proj.cu:57-62; the Qwen3.5 form rounds silu too).  A gate
    / up value may differ from the float64-rounded one by one bf16 step (fp32 accumulation order), so the bar is the
    activation evaluated over those one-step neighbourhoods plus one output rounding."""
    import torch
    from pegainfer_amd import ffi
    assert _route(0, T, K, I) == 258
    rng = np.random.default_rng(I + T)
    W, X = rnd(rng, 2 * I, K, scale=0.03), rnd(rng, T, K)
    Wd, Xd = to_dev(W), to_dev(X)
    out = torch.empty((T, I), dtype=torch.bfloat16, device="cuda")
    fn = ffi.lib().pegainfer_gemm_silu_rounded if rounded else ffi.lib().pegainfer_gemm_silu
    s = torch.cuda.current_stream().cuda_stream
    assert fn(Wd.data_ptr(), Xd.data_ptr(), out.data_ptr(), None, I, T, K, s) == 0
    got = from_dev(out)
    cols = np.sort(rng.choice(I, size=1024, replace=False))
    X64 = X.astype(np.float64)
    gate, up = X64 @ W[cols].astype(np.float64).T, X64 @ W[I + cols].astype(np.float64).T
    env_g, env_u = np.abs(X64) @ np.abs(W[cols]).astype(np.float64).T, np.abs(X64) @ np.abs(W[I + cols]).astype(np.float64).T
    silu = lambda g: g / (1.0 + np.exp(-g))
    ref = silu(gate) * up
    # first-order error propagation of the two GEMM envelopes through silu(g) * u, plus bf16 roundings of g, u, (silu,) out
    dg = env_g * 2.0 ** -7 * 0.05 + np.abs(gate) * 2.0 ** -8
    du = env_u * 2.0 ** -7 * 0.05 + np.abs(up) * 2.0 ** -8
    dsilu = np.abs(silu(gate + 1e-3) - silu(gate - 1e-3)) / 2e-3          # |silu'|
    tol = dsilu * dg * np.abs(up) + np.abs(silu(gate)) * (du + (2.0 ** -8 * np.abs(up) if rounded else 0)) + np.abs(ref) * 2.0 ** -7 + 1e-5
    err = np.abs(got[:, cols] - ref)
    assert np.all(err <= 1.5 * tol), (I, T, float((err / tol).max()))
    out2 = torch.empty_like(out)
    assert fn(Wd.data_ptr(), Xd.data_ptr(), out2.data_ptr(), None, I, T, K, s) == 0
    assert torch.equal(out, out2)


def test_streamk_split3_outputs(P, streamk_on):
    """the stacked q | k | v projection at 2048 tokens (24 x 8 = 192 tiles: stream-K) writes its three buffers"""
    import torch
    rng = np.random.default_rng(77)
    T, K = 2048, 2560
    W, X = rnd(rng, 6144, K, scale=0.05), rnd(rng, T, K)
    q = torch.empty((T, 4096), dtype=torch.bfloat16, device="cuda")
    k = torch.empty((T, 1024), dtype=torch.bfloat16, device="cuda")
    v = torch.empty((T, 1024), dtype=torch.bfloat16, device="cuda")
    Wd, Xd = to_dev(W), to_dev(X)
    P.gemm_split3_into(Wd, Xd, q, k, v)
    full = from_dev(P.gemm(Wd, Xd))
    assert np.array_equal(bf16_bits(np.concatenate([from_dev(q), from_dev(k), from_dev(v)], axis=1)), bf16_bits(full))
