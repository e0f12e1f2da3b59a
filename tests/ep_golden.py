"""Loader for tests/golden/ep_a2a_golden.npz (made by tests/golden/make_ep_golden.py from the reference's
RankTestData.create) and the reference test's own checks (pegainfer-comm/tests/p2p_all_to_all/
test_p2p_all_to_all.py:95-232) restated over plain arrays, shared by the oracle test and the GPU test."""
import os

import numpy as np

from oracle.bf16 import bf16_from_bits

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ep_a2a_golden.npz")


def ids():
    return [str(i) for i in np.load(PATH)["ids"]]


def load(cid):
    z = np.load(PATH)
    world, T, E, H, Hs, topk, pad, in_el, out_el = (int(v) for v in z[f"{cid}/meta"])
    case = dict(id=cid, world=world, T=T, E=E, H=H, Hs=Hs, topk=topk, pad=pad, in_el=in_el, out_el=out_el,
                expected_sum=z[f"{cid}/expected_num_tokens_sum"], ranks=[])
    for r in range(world):
        d = {k: z[f"{cid}/r{r}/{k}"] for k in ("indices", "weights", "dp_x", "expected_num_tokens", "ref_out_tokens")}
        d["dp_x_scale"] = z[f"{cid}/r{r}/dp_x_scale"] if Hs else None
        case["ranks"].append(d)
    return case


def as_f32(a):
    """fixture payloads: f32 arrays or bf16 bit images (uint16)"""
    return bf16_from_bits(a) if a.dtype == np.uint16 else np.asarray(a, np.float32)


def act(x, x_scale):
    """the reference test's "expert" (test_p2p_all_to_all.py:44-50): y = 2 x, or 2 x * tiled scales in f32"""
    x = np.asarray(x, np.float32)
    if x_scale is None:
        return x * np.float32(2)
    return x * np.tile(np.asarray(x_scale, np.float32), (1, x.shape[1] // x_scale.shape[1])) * np.float32(2)


def hash_token(row):
    return ",".join(f"{v:.2f}" for v in np.asarray(row, np.float64).tolist())   # test_p2p_all_to_all.py:193-194


def check_dispatch(case, rank, expert_num_tokens, out_expert_x):
    """test_p2p_all_to_all.py:188-229 for one rank: the per-expert counts, then every token of this rank's dp_x that is
    routed to a local expert must be found among the rows of the padded expert groups."""
    epr = case["E"] // case["world"]
    first, last = rank * epr, (rank + 1) * epr
    expected_local = case["expected_sum"][first:last]
    assert np.array_equal(np.asarray(expert_num_tokens, np.int64), expected_local.astype(np.int64))
    on_rank, index = set(), 0
    for n in expected_local.tolist():
        for row in out_expert_x[index:index + n]:
            on_rank.add(hash_token(row))
        index = -(-(index + n) // case["pad"]) * case["pad"]            # round_up(index + n, expert_padding)
    d = case["ranks"][rank]
    x = as_f32(d["dp_x"])
    missing = [i for i, (tok, routes) in enumerate(zip(x, d["indices"].tolist()))
               if any(first <= e < last for e in routes) and hash_token(tok) not in on_rank]
    assert not missing, f"missing {len(missing)} tokens on rank {rank}"
    return index


def check_combine(case, rank, out_tokens):
    """torch.testing.assert_close(out_tokens, ref_out_tokens) (test_p2p_all_to_all.py:232) with torch's default
    tolerances for the output dtype: bf16 rtol 1.6e-2 / atol 1e-5, f32 rtol 1.3e-6 / atol 1e-5."""
    ref = as_f32(case["ranks"][rank]["ref_out_tokens"])
    got = np.asarray(out_tokens, np.float32)
    rtol, atol = (1.6e-2, 1e-5) if case["out_el"] == 2 else (1.3e-6, 1e-5)
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert not bad.any(), (case["id"], rank, int(bad.sum()), float(np.abs(got - ref).max()))
