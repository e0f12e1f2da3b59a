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
