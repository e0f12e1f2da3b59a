"""N > 1 paths on CPU: world_size-2 gloo process groups (127.0.0.1 rendezvous).

* replica sharding + max-over-ranks timing (what bench.py --gpus N does),
* the collective verbs of pegainfer_amd.parallel.Comm (reference cudarc Comm + deepseek-v4
  runtime/collectives.rs, moe.rs AG/RS) against dense single-process results,
* Qwen3 tensor-parallel sharding (weights.rs:121-291): two ranks, each running the oracle DAG on its weight
  shard with the all-reduce at the reference's two call sites, must reproduce the unsharded oracle.
"""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle
from oracle.safetensors_io import load_safetensors
from pegainfer_amd import parallel as P

G = os.path.join(os.path.dirname(__file__), "golden")
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, fn, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        out[rank] = fn(rank)
    finally:
        dist.destroy_process_group()


def run2(fn):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), fn, out), nprocs=WORLD, join=True)
    return [out[r] for r in range(WORLD)]


# ------------------------------------------------------------------ pure helpers
def test_shard_helpers():
    assert P.shard_range(4096, 1, 4) == (1024, 1024)
    with pytest.raises(ValueError):
        P.shard_range(10, 0, 4)
    assert P.shard_requests(5, 1, 2) == [1, 3] and P.shard_requests(5, 0, 2) == [0, 2, 4]
    cfg = dict(num_attention_heads=32, num_key_value_heads=8, intermediate_size=9728, head_dim=128, hidden_size=2560)
    loc = P.tp_local_config(cfg, 8)
    assert (loc["num_attention_heads"], loc["num_key_value_heads"], loc["intermediate_size"]) == (4, 1, 1216)
    with pytest.raises(ValueError):
        P.tp_local_config(cfg, 16)   # 8 kv heads: world in {1,2,4,8}
    assert P.max_over_ranks(1.5) == 1.5   # no process group -> identity


def _replica_job(rank):
    mine = P.shard_requests(7, rank, WORLD)
    elapsed = 1.0 + rank            # rank 1 is slower
    return mine, P.max_over_ranks(elapsed)


def test_replicas_and_max_time():
    (r0, t0), (r1, t1) = run2(_replica_job)
    assert sorted(r0 + r1) == list(range(7)) and not set(r0) & set(r1)
    assert t0 == t1 == 2.0        # every rank reports the slowest rank's time


# ------------------------------------------------------------------ collective verbs
def _verbs(rank):
    c = P.Comm()
    g = torch.Generator().manual_seed(5)
    full = torch.randn(WORLD, 6, 16, generator=g)                       # same on both ranks
    mine = full[rank].clone()
    res = {}
    res["ar"] = c.all_reduce_in_place(mine.clone()).numpy()
    hb = mine.to(torch.bfloat16)
    res["ar_f32"] = c.all_reduce_hidden_fp32_in_place(hb.clone()).float().numpy()
    res["ag"] = c.all_gather(mine).numpy()
    res["rs"] = c.reduce_scatter(full.reshape(WORLD * 6, 16) * (rank + 1)).numpy()
    # decode MoE AG/RS: each rank owns "experts" = a slice of a weight; dense result = tokens @ W
    W = torch.randn(16, 16, generator=g)
    half = 16 // WORLD
    expert = lambda x: (x.float()[:, rank * half:(rank + 1) * half] @ W[rank * half:(rank + 1) * half]).float()
    shared = lambda x: x * 0.5
    res["moe"] = c.moe_all_gather_reduce_scatter(hb, expert, shared).numpy()
    res["full"], res["W"] = full.numpy(), W.numpy()
    return res


def test_comm_verbs_match_dense():
    r = run2(_verbs)
    full, W = r[0]["full"], r[0]["W"]
    for rank in range(WORLD):
        assert np.allclose(r[rank]["ar"], full.sum(0))
        bf = torch.from_numpy(full).to(torch.bfloat16).float()
        assert np.array_equal(r[rank]["ar_f32"], bf.sum(0).to(torch.bfloat16).float().numpy())  # f32 sum, one rounding
        assert np.array_equal(r[rank]["ag"], full.reshape(WORLD * 6, 16))                      # rank-major order
        expect_rs = (full.reshape(WORLD * 6, 16) * 1 + full.reshape(WORLD * 6, 16) * 2)[rank * 6:(rank + 1) * 6]
        assert np.allclose(r[rank]["rs"], expect_rs)
        tok = bf[rank].numpy()
        assert np.allclose(r[rank]["moe"], tok @ W + 0.5 * tok, atol=1e-4)


# ------------------------------------------------------------------ Qwen3 tensor parallel
def _tp_oracle(rank):
    meta = json.load(open(os.path.join(G, "qwen3_tiny_golden.json")))
    # the tiny fixture has 1 kv head: widen to a 4q/2kv variant by duplicating the kv head so TP=2 is legal
    full_w = load_safetensors(os.path.join(G, "qwen3_tiny.safetensors"))
    cfg = dict(meta["config"], num_key_value_heads=2)
    for i in range(cfg["num_hidden_layers"]):
        for n in ("k_proj", "v_proj"):
            k = f"model.layers.{i}.self_attn.{n}.weight"
            full_w[k] = np.concatenate([full_w[k], full_w[k][::-1].copy()], axis=0)
    keys = ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
            "intermediate_size", "vocab_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings"]
    loc = P.tp_local_config(cfg, WORLD)
    w = P.shard_qwen3_state(full_w, cfg, rank, WORLD)

    def all_reduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.all_reduce(t)
        return t.numpy()
    m = Qwen3Oracle(Qwen3Config(**{k: loc[k] for k in keys}), w, num_pages=32, all_reduce=all_reduce)
    st = KvState()
    prompt = meta["cases"][2]["prompt_tokens"]
    lg = [m.batch_prefill([prompt], [st])[0]]
    for tok in meta["cases"][2]["output_tokens"][:4]:
        lg.append(m.batch_decode([tok], [st])[0])
    out = {"logits": np.stack(lg)}
    if rank == 0:
        ref = Qwen3Oracle(Qwen3Config(**{k: cfg[k] for k in keys}), full_w, num_pages=32)
        st = KvState()
        rl = [ref.batch_prefill([prompt], [st])[0]]
        for tok in meta["cases"][2]["output_tokens"][:4]:
            rl.append(ref.batch_decode([tok], [st])[0])
        out["ref"] = np.stack(rl)
    return out


def test_tp2_sharded_oracle_matches_unsharded():
    r = run2(_tp_oracle)
    a, b, ref = r[0]["logits"], r[1]["logits"], r[0]["ref"]
    assert np.array_equal(a, b)                       # replicated lm_head: every rank holds the same logits
    cos = (a * ref).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(ref, axis=-1)
    # partial sums are rounded to bf16 before the reduce (bf16 all-reduce): a few ulp on the logits
    assert cos.min() > 0.9995 and np.abs(a - ref).max() <= 0.5, (cos.min(), np.abs(a - ref).max())


# ------------------------------------------------------------------ MP8 collective microbench (bench.py, N > 1)
def _mp8_bench(rank):
    out = P.bench_mp8_collectives(P.Comm(), hidden=64, token_counts=(1, 8), iters=2, device="cpu")
    return out


def test_mp8_collective_bench_runs_on_gloo():
    """The side measurement bench.py adds at N > 1 (BASELINE.json configs[4]): every verb x token count yields a
    positive time on both ranks."""
    for r in run2(_mp8_bench):
        assert set(r) == {"all_reduce_f32", "all_gather_bf16", "reduce_scatter_f32", "all_to_all_bf16"}
        for verb in r.values():
            assert set(verb) == {"1", "8"} and all(v > 0 for v in verb.values())


def _a2a(rank):
    c = P.Comm()
    send = torch.arange(WORLD * 3 * 4, dtype=torch.float32).reshape(WORLD * 3, 4) + 100 * rank     # 3 rows per peer
    got = c.all_to_all(send)
    counts = [1 + rank, 2]                                                                      # ragged: rows per peer
    rows = torch.arange(sum(counts) * 2, dtype=torch.float32).reshape(sum(counts), 2) + 1000 * rank
    gv, rc = c.all_to_allv(rows, counts)
    return {"got": got.numpy(), "gv": gv.numpy(), "rc": rc}


def test_expert_all_to_all_matches_dense_routing():
    """all_to_all: slab r of every rank ends up on rank r, rank-major; all_to_allv: the ragged variant."""
    r = run2(_a2a)
    for me in range(WORLD):
        exp = np.concatenate([(np.arange(WORLD * 3 * 4, dtype=np.float32).reshape(WORLD * 3, 4) + 100 * src)[me * 3:(me + 1) * 3]
                              for src in range(WORLD)])
        assert np.array_equal(r[me]["got"], exp)
        parts = []
        for src in range(WORLD):
            counts = [1 + src, 2]
            rows = np.arange(sum(counts) * 2, dtype=np.float32).reshape(sum(counts), 2) + 1000 * src
            off = sum(counts[:me])
            parts.append(rows[off:off + counts[me]])
        assert r[me]["rc"] == [[1 + src, 2][me] for src in range(WORLD)]
        assert np.array_equal(r[me]["gv"], np.concatenate(parts))


# ------------------------------------------------------------------ the sharded leg of every N > 1 bench line
class _StandInTpEngine:
    """host-only stand-in for an attached tensor-parallel engine: every decode step joins an all-reduce like the real one"""

    def __init__(self):
        self.steps = 0

    def new_request(self):
        return 0

    def prefill(self, rids, prompts):
        t = torch.ones(4)
        dist.all_reduce(t)
        return [int(t[0])]

    def decode(self, rids, toks):
        t = torch.ones(4)
        dist.all_reduce(t)
        self.steps += 1
        return [int(t[0])]

    def last_step_ms(self):
        return 0.5

    def close(self):
        pass


def _tp_leg(rank):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from pegainfer_amd.qwen3 import QWEN3_4B
    eng = _StandInTpEngine()
    r = bench.run_tp_leg(lambda: eng, bench.synthetic_prompt(64), 6, 2, WORLD, dict(QWEN3_4B), dist.barrier, P.max_over_ranks)
    r["engine_steps"] = eng.steps
    return r


def test_bench_tp_leg_line_shape_at_world_2():
    """bench.py --gpus N (N > 1) always adds the sharded config next to the replicas value: `tp` = {tok_s, ms_per_step,
    ttft_ms, scaling "strong", parallelism "tpN", ...}, timed by warm-up + K steps between fences with the MAX over ranks -
    so every rank reports the SAME numbers."""
    a, b = run2(_tp_leg)
    for key in ("tok_s", "ms_per_step", "ttft_ms", "scaling", "parallelism", "all_reduces_per_step", "frac_of_aggregate_8TBps"):
        assert key in a, key
    assert a["scaling"] == "strong" and a["parallelism"] == "tp2" and a["all_reduces_per_step"] == 72
    assert a["engine_steps"] == 8 and a["steps"] == 6 and a["warmup"] == 2
    assert a["tok_s"] == b["tok_s"] and a["ms_per_step"] == b["ms_per_step"] and a["tok_s"] > 0
    assert abs(a["tok_s"] * a["ms_per_step"] / 1000.0 - 1.0) < 0.02


def test_traffic_probe_kernel_is_found_by_call_site():
    """bench.py's PMC pass identifies the dominant GEMV by the probe's launch count, whatever instantiation the shape takes
    (VERDICT r5 Missing 6: the 8B / Qwen3.5 lines had traffic null because a template-argument prefix missed their kernels)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = bench.PROBE_LAUNCHES
    per = {"fill_synthetic_kernel(...)": (5.0, 30), "gemv_fused_kernel<1, 2, 4, 1, 5>(GemvFusedArgs)": (48000.0 * n, n),
           "gemv_fused_kernel<1, 1, 1, 0, 4>(GemvFusedArgs)": (100.0 * 7, 7), "rms_norm_kernel": (1.0 * n, n)}
    assert bench.pick_call_site_kernel(per) == "gemv_fused_kernel<1, 2, 4, 1, 5>(GemvFusedArgs)"
    assert bench.pick_call_site_kernel({"skinny_resident_kernel<1, 1>": (9.0 * n, n), "x": (1.0, 2)}) == "skinny_resident_kernel<1, 1>"
    assert bench.pick_call_site_kernel({"gemv_fused_kernel<1>": (1.0, n - 1)}) is None
