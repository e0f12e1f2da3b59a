#!/usr/bin/env python3
"""Golden vectors for the expert-parallel dispatch / combine, generated FROM THE REFERENCE'S OWN
PYTHON TRUTH: this script imports /root/reference/pegainfer-comm/tests/p2p_all_to_all/data.py (RankTestData.create,
data.py:35-107 - the exact generator the reference's all-to-all test feeds its kernels with) and records, per
configuration of the reference's parametrisation (test_p2p_all_to_all.py:235-470), every quantity that test checks
(test_p2p_all_to_all.py:95-232):

  * per rank: indices, weights, dp_x (+ dp_x_scale) exactly as RankTestData.create drew them with the test's own
    per-rank generator seed (`_generator(device, rank)`, :53-56; device = CPU here: no GPU in this container)
  * expected_num_tokens per rank (data.py:52-56 bincount) and their sum over ranks - the reference asserts
    `expected_num_tokens[first_expert:last_expert] == expert_num_tokens` on every rank (:189-190)
  * ref_out_tokens = _act(dp_x, dp_x_scale).to(out_dtype) (:44-50, :98): with the "expert" y = _act(out_expert_x) the
    combine must return it (`torch.testing.assert_close(out_tokens, ref_out_tokens)`, :232)
  * expert_padding: groups start at `index = round_up(index + n, expert_padding)` (:196-204)

Run in the build container (needs /root/reference and torch):  python tests/golden/make_ep_golden.py
Output: tests/golden/ep_a2a_golden.npz (committed).  The GPU box never needs /root/reference.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/pegainfer-comm/tests/p2p_all_to_all/data.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ep_a2a_golden.npz")

# (id, world_size, max_num_tokens, num_experts, hidden_dim, hidden_dim_scale, topk, in_dtype, out_dtype, scale_dtype,
#  expert_padding) - the reference's parameter sets that fit one node (test_p2p_all_to_all.py:235-470); the two
# hidden_dim 7168 sets are carried at hidden_dim 512 (same routing, smaller rows) to keep the fixture small
CONFIGS = [
    ("TP2-NIC1-FP32", 2, 128, 16, 128, None, 2, torch.float32, torch.float32, None, 1),
    ("TP2-NIC1-BF16", 2, 128, 16, 128, None, 2, torch.bfloat16, torch.bfloat16, None, 1),
    ("TP2-NIC1-BF16-PADDED", 2, 8, 16, 16, None, 2, torch.bfloat16, torch.bfloat16, None, 16),
    ("TP2-NIC1-FP8", 2, 2, 16, 128, 16, 2, torch.bfloat16, torch.bfloat16, torch.float32, 1),
    ("TP4-NIC1-FP32", 4, 128, 128, 128, None, 8, torch.float32, torch.float32, None, 1),
    ("TP4-NIC2-BF16-H512", 4, 128, 256, 512, None, 8, torch.bfloat16, torch.bfloat16, None, 1),
    ("EP8-BF16-PADDED8-H512", 8, 32, 256, 512, None, 6, torch.bfloat16, torch.bfloat16, None, 8),
]


def _act(x, x_scale):
    """test_p2p_all_to_all.py:44-50"""
    if x_scale is None:
        return x * 2
    _, hidden_dim = x.shape
    _, hidden_dim_scale = x_scale.shape
    return x.to(torch.float32) * x_scale.repeat(1, hidden_dim // hidden_dim_scale) * 2


def _generator(device, rank):
    """test_p2p_all_to_all.py:53-56"""
    g = torch.Generator(device=device)
    g.manual_seed(rank)
    return g


def _np(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)      # bf16 bits
    if t.dtype == torch.uint32:
        return t.to(torch.int64).numpy().astype(np.int32)
    return t.numpy()


def main():
    if not os.path.exists(REF):
        sys.exit("needs the reference checkout at /root/reference (build container only)")
    spec = importlib.util.spec_from_file_location("ref_a2a_data", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    dev = torch.device("cpu")
    out = {"ids": np.array([c[0] for c in CONFIGS])}
    for (cid, world, T, E, H, Hs, topk, in_dt, out_dt, sc_dt, pad) in CONFIGS:
        ranks = [ref.RankTestData.create(num_experts=E, num_experts_per_token=topk, max_num_tokens=T, hidden_dim=H,
                                         hidden_dim_scale=Hs, in_dtype=in_dt, scale_dtype=sc_dt,
                                         generator=_generator(dev, r), device=dev) for r in range(world)]
        total = torch.sum(torch.stack([d.expected_num_tokens for d in ranks], dim=0), dim=0, dtype=torch.int32)
        out[f"{cid}/meta"] = np.array([world, T, E, H, Hs or 0, topk, pad, 4 if in_dt == torch.float32 else 2,
                                       4 if out_dt == torch.float32 else 2], dtype=np.int64)
        out[f"{cid}/expected_num_tokens_sum"] = _np(total)
        for r, d in enumerate(ranks):
            out[f"{cid}/r{r}/indices"] = _np(d.indices)
            out[f"{cid}/r{r}/weights"] = _np(d.weights)
            out[f"{cid}/r{r}/dp_x"] = _np(d.dp_x)
            if d.dp_x_scale is not None:
                out[f"{cid}/r{r}/dp_x_scale"] = _np(d.dp_x_scale)
            out[f"{cid}/r{r}/expected_num_tokens"] = _np(d.expected_num_tokens)
            out[f"{cid}/r{r}/ref_out_tokens"] = _np(_act(d.dp_x, d.dp_x_scale).to(out_dt))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
