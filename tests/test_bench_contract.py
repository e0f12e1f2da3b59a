"""bench.py's host-side contract, checked without a GPU: the algorithmic-bytes and FLOP models the roofline objects
are priced with, the launcher re-exec for `--gpus N`, and the refusal to run without a device
(the product path has no CPU fallback)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def qwen3_4b():
    from pegainfer_amd.qwen3 import QWEN3_4B
    return dict(QWEN3_4B)


def test_algorithmic_bytes_per_token_qwen3_4b():
    c = qwen3_4b()
    H, L, I, V, D = c["hidden_size"], c["num_hidden_layers"], c["intermediate_size"], c["vocab_size"], c["head_dim"]
    qd, kvd = c["num_attention_heads"] * D, c["num_key_value_heads"] * D
    # weights read once: per layer qkv + o + gate_up + down, tied lm_head, norm vectors (bf16)
    weights = 2 * (L * ((qd + 2 * kvd) * H + H * qd + 3 * I * H) + V * H + L * (2 * H + 2 * D) + H)
    kv_per_token = L * 2 * kvd * 2
    ctx = 1024
    want = weights + kv_per_token * ctx + kv_per_token + 2 * H + 2 * V
    assert bench.algorithmic_bytes_per_token(c, ctx, 1) == want
    assert 8.0e9 < want < 8.3e9              # the 8.2 GB per token the 8 TB/s step roofline is priced on
    # a second request adds its own KV scan, KV append, embedding row and logits - the weights are shared
    assert bench.algorithmic_bytes_per_token(c, ctx, 2) - want == kv_per_token * (ctx + 1) + 2 * H + 2 * V


def test_prefill_roofline_flop_model():
    c = qwen3_4b()
    r = bench.prefill_roofline(c, 1024, 10.0)
    H, L, I, V, D = c["hidden_size"], c["num_hidden_layers"], c["intermediate_size"], c["vocab_size"], c["head_dim"]
    qd, kvd = c["num_attention_heads"] * D, c["num_key_value_heads"] * D
    flops = 2.0 * ((qd + 2 * kvd) * H + H * qd + 3 * I * H) * L * 1024 + 2.0 * V * H + 2.0 * qd * 1024 * 1024 * L
    assert r["flops"] == flops and r["bound"] == "mfma" and r["peak"] == 2500.0   # dense bf16, never the sparse figure
    assert r["frac"] == round(flops / 10e-3 / 1e12 / 2500.0, 4)
    assert bench.prefill_roofline(c, 1024, 10.0, tp_world=2)["peak"] == 5000.0


def test_synthetic_prompt_is_the_reference_profile():
    p = bench.synthetic_prompt(2048)
    assert p[:3] == [100, 101, 102] and p[999] == 1099 and p[1000] == 100 and len(p) == 2048  # bench_serving.rs


def test_gpus_flag_becomes_the_launcher(monkeypatch):
    """`python bench.py --gpus 4` without WORLD_SIZE re-executes itself under torch.distributed.run on 127.0.0.1 with
    one process per GPU and the same arguments (the flag was parsed and ignored)."""
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert bench.main() == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_world_size_mismatch_and_missing_gpu_are_loud(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "needs a GPU" in str(e.value)       # no silent CPU path behind the bench line


def test_committed_bench_line_carries_the_contract_fields():
    """The bench line kept under profiles/ for the current round (the same command the driver runs) has every field of
    the driver's contract plus the roofline / cpu_baseline objects, and its numbers are self-consistent."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_default_run.json")))
    assert paths, "no committed bench line under profiles/"
    d = json.load(open(paths[-1]))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 0.02          # tokens/s x s/step == 1 at batch 1
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_us"] * 1e-3) / r["achieved"] < 0.01   # GB/s
    assert r["traffic"] is None or 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.10
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    if "parity" in d:   # since round 4: the benchmarked 36-layer model against the oracle (and HF) on its own checkpoint
        p = d["parity"]
        for key in ("steps", "cos_min", "max_dlogit", "tokens_equal"):
            assert key in p, key
        assert p["cos_min"] > 0.99 and p["tokens_differing_away_from_a_near_tie"] == 0


def test_cpu_legs_run_on_a_tiny_checkpoint():
    """bench.py's CPU side end to end on a tiny seeded checkpoint (no GPU): the HF reference engine and the oracle run on
    the same exported-style bits, the parity block compares them with 'GPU' rows (here: the oracle's own, so the oracle
    distance is exactly zero and HF sits at the oracle-vs-HF distance), and the baseline carries the contract keys."""
    pytest.importorskip("transformers")
    import numpy as np
    from oracle import ops as O
    from oracle.bf16 import bf16_bits
    from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle, synthetic_weights
    cfgd = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                intermediate_size=512, vocab_size=1024, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True,
                max_position_embeddings=4096)
    cfg = Qwen3Config(**cfgd)
    w, bits = synthetic_weights(cfg, seed=3, std=0.05, with_bits=True)
    prompt = [100 + (i % 900) for i in range(40)]
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        m = Qwen3Oracle(cfg, w, num_pages=16)
        st = KvState()
        rows = [m.batch_prefill([prompt], [st])[0]]
        toks = [int(rows[0].argmax())]
        for _ in range(4):
            rows.append(m.batch_decode([toks[-1]], [st])[0])
            toks.append(int(rows[-1].argmax()))
        gb = bf16_bits(np.stack(rows))
        base, par = bench.cpu_legs(cfgd, bits, prompt, toks, gb, 4, 4, hf_repeats=1, hf_new_tokens=6,
                                   batch_rows={"2": np.stack([gb, gb])})
    finally:
        O.GEMM_ACCUM = old
    # the batch sweep's parity entries: every column of a bs-N step against the same oracle / truth stream (here the columns ARE
    # the oracle's rows: ratio exactly 1, every token equal)
    assert par["batch"]["2"]["ratio_pooled"] == 1.0 and par["batch"]["2"]["within_bar"] and par["batch"]["2"]["tokens_equal"] == [10, 10]
    assert base["kind"] == "reference" and base["value"] > 0 and base["port"]["kind"] == "port"
    assert par["cos_min"] == 1.0 and par["max_dlogit"] == 0.0 and par["tokens_equal"] == 5
    assert par["hf"]["cos_min"] > 0.999 and par["hf"]["steps_compared"] >= 1
    # HF teacher-forced on the 'GPU' tokens: all 5 rows compared whatever the free-running streams did
    tf = par["hf"]["teacher_forced_on_engine_tokens"]
    assert tf["steps_compared"] == 5 and tf["cos_min"] > 0.999 and tf["argmax_differing_away_from_a_near_tie"] == 0


def test_cpu_leg_qwen35_runs_on_the_tiny_golden_checkpoint():
    """configs[3]'s CPU side end to end (no GPU): HF Qwen3_5ForCausalLM built by oracle/hf_engine.py from the committed tiny
    hybrid checkpoint reproduces the committed HF golden's tokens on a case without near-ties (the fixture came from another
    CPU: bf16 GEMMs differ by ~1 % of the logit scale between hosts, so logits are compared by tolerance), and the baseline /
    parity blocks carry their keys."""
    pytest.importorskip("transformers")
    import json
    import struct

    import numpy as np
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(g, "qwen35_tiny_golden.json")))
    lgz = np.load(os.path.join(g, "qwen35_tiny_logits.npz"))
    t = {}
    with open(os.path.join(g, "qwen35_tiny.safetensors"), "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        hdr = json.loads(f.read(n))
        for name, info in hdr.items():
            if name == "__metadata__":
                continue
            lo, hi = info["data_offsets"]
            f.seek(8 + n + lo)
            t[name] = np.frombuffer(f.read(hi - lo), dtype=np.uint16 if info["dtype"] == "BF16" else np.float32).copy()
    c = meta["cases"][1]
    base, par = bench.cpu_leg_qwen35(meta["config"], t, c["prompt_tokens"], c["output_tokens"][:9], lgz[c["name"]][:9], 4, 4,
                                     hf_repeats=1, hf_new_tokens=9)
    assert base["kind"] == "reference" and base["value"] > 0
    assert par["hf"]["tokens_equal_prefix"] == 9 and par["hf"]["cos_min"] > 0.9995
