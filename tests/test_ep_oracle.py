"""CPU: the expert-parallel dispatch / combine oracle (oracle/ep_ref.py) against its defining properties."""
import numpy as np

from oracle import ep_ref


def _case(seed, world=4, E=16, topk=3, H=8, Ts=(5, 0, 7, 3)):
    rng = np.random.default_rng(seed)
    xs = [rng.standard_normal((t, H)).astype(np.float32) for t in Ts]
    idx = [np.stack([rng.permutation(E)[:topk] for _ in range(t)]).astype(np.int32) if t else np.zeros((0, topk), np.int32)
           for t in Ts]
    w = [rng.random((t, topk)).astype(np.float32) for t in Ts]
    return xs, idx, w


def test_dispatch_counts_and_grouping():
    xs, idx, _ = _case(1)
    out = ep_ref.dispatch(xs, idx, 16)
    total = sum(len(o[0]) for o in out)
    assert total == sum(len(x) for x in xs) * 3                      # every (token, k) pair arrives exactly once
    for r, (rows, tpe, origin) in enumerate(out):
        assert tpe.sum() == len(rows)
        hist = np.zeros(4, np.int64)
        for (src, t, k) in origin:
            e = idx[src][t, k]
            assert e // 4 == r                                       # on the rank that owns the expert
            hist[e % 4] += 1
        assert np.array_equal(hist, tpe)
        les = [idx[s][t, k] % 4 for (s, t, k) in origin]
        assert les == sorted(les)                                    # expert-major
        for (src, t, k), row in zip(origin, rows):
            assert np.array_equal(row, xs[src][t])


def test_identity_experts_combine_to_weighted_sum():
    xs, idx, w = _case(2)
    out = ep_ref.dispatch(xs, idx, 16)
    res = ep_ref.combine([o[0] for o in out], [o[2] for o in out], w, [len(x) for x in xs], 8)
    for r in range(4):
        want = xs[r].astype(np.float64) * w[r].astype(np.float64).sum(axis=1, keepdims=True)
        assert np.allclose(res[r], want, atol=1e-12)


def test_combine_accumulate_is_linear():
    xs, idx, w = _case(3)
    out = ep_ref.dispatch(xs, idx, 16)
    prev = [np.ones((len(x), 8)) for x in xs]
    a = ep_ref.combine([o[0] for o in out], [o[2] for o in out], w, [len(x) for x in xs], 8)
    b = ep_ref.combine([o[0] for o in out], [o[2] for o in out], w, [len(x) for x in xs], 8, prev=prev)
    for r in range(4):
        assert np.allclose(b[r] - a[r], 1.0)


def test_pairs_with_invalid_expert_ids_travel_nowhere_and_combine_as_zero():
    xs, idx, w = _case(4)
    idx[0][1, 2] = -1          # not expert ids
    idx[2][0, 0] = 16
    out = ep_ref.dispatch(xs, idx, 16)
    assert sum(len(o[0]) for o in out) == sum(len(x) for x in xs) * 3 - 2
    res = ep_ref.combine([o[0] for o in out], [o[2] for o in out], w, [len(x) for x in xs], 8)
    for r in range(4):
        wr = w[r].astype(np.float64).copy()
        if r == 0:
            wr[1, 2] = 0.0
        if r == 2:
            wr[0, 0] = 0.0
        assert np.allclose(res[r], xs[r].astype(np.float64) * wr.sum(axis=1, keepdims=True), atol=1e-12)


# ---------------------------------------------------------------- pinned to the reference's Python truth
import pytest  # noqa: E402

from oracle.bf16 import bf16_round  # noqa: E402
import ep_golden  # noqa: E402


@pytest.mark.parametrize("cid", ep_golden.ids())
def test_oracle_reproduces_the_reference_a2a_test(cid):
    """The reference's own all-to-all test (test_p2p_all_to_all.py:95-232) with its own inputs
    (RankTestData.create, fixture made by tests/golden/make_ep_golden.py), run against the oracle: per-expert counts,
    padded expert groups, token membership, and combine == ref_out_tokens within torch's assert_close tolerance."""
    case = ep_golden.load(cid)
    W, E = case["world"], case["E"]
    xs = [ep_golden.as_f32(d["dp_x"]) for d in case["ranks"]]
    idx = [d["indices"] for d in case["ranks"]]
    ws = [d["weights"] for d in case["ranks"]]
    sc = [d["dp_x_scale"] for d in case["ranks"]] if case["Hs"] else None
    for r, d in enumerate(case["ranks"]):                      # data.py:52-56: the fixture's own bincount
        assert np.array_equal(np.bincount(idx[r].ravel(), minlength=E), d["expected_num_tokens"])
    out = ep_ref.dispatch(xs, idx, E, expert_padding=case["pad"], scales=sc)
    ys = []
    for r in range(W):
        rows, tpe, origin = out[r][:3]
        extent = ep_golden.check_dispatch(case, r, tpe, rows)
        assert extent == len(rows)                             # the oracle's padded extent == the reference's walk
        y = ep_golden.act(rows, out[r][3] if sc else None)     # expert_y = _act(out_expert_x, scale).to(out_dtype)
        ys.append(bf16_round(y) if case["out_el"] == 2 else y)
    res = ep_ref.combine_f32(ys, [o[2] for o in out], ws, [case["T"]] * W, case["H"])
    for r in range(W):
        got = bf16_round(res[r]) if case["out_el"] == 2 else res[r]
        ep_golden.check_combine(case, r, got)
        exact = ep_ref.combine(ys, [o[2] for o in out], ws, [case["T"]] * W, case["H"])[r]
        assert np.abs(res[r] - exact).max() <= 2.0 ** -21 * max(1.0, np.abs(exact).max())   # the f32 chain vs the exact sum


def test_bound_m_limits_the_tokens_that_take_part():
    """`bound_m_ptr ? *bound_m_ptr : num_tokens` (a2a_dispatch_send.cu:172, a2a_combine_recv.cu:52)"""
    xs, idx, w = _case(5)
    bound = [2, 0, 7, 1]
    out = ep_ref.dispatch(xs, idx, 16, bound_m=bound)
    assert sum(int(o[1].sum()) for o in out) == sum(bound) * 3
    cut = ep_ref.dispatch([x[:b] for x, b in zip(xs, bound)], [i[:b] for i, b in zip(idx, bound)], 16)
    for a, b in zip(out, cut):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    prev = [np.full((len(x), 8), 3.0) for x in xs]
    res = ep_ref.combine([o[0] for o in out], [o[2] for o in out], w, [len(x) for x in xs], 8, prev=prev, bound_m=bound)
    for r in range(4):
        assert np.all(res[r][bound[r]:] == 3.0)                # rows at or beyond the bound are not touched


def test_expert_padding_aligns_every_group():
    xs, idx, _ = _case(6)
    for pad in (1, 4, 16):
        out = ep_ref.dispatch(xs, idx, 16, expert_padding=pad)
        for rows, tpe, origin in out:
            base = 0
            for n in tpe.tolist():
                assert base % pad == 0
                assert np.all(origin[base:base + n, 0] >= 0)
                gap = -(-n // pad) * pad - n
                assert np.all(origin[base + n:base + n + gap, 0] == -1)
                base += n + gap
            assert base == len(rows)
