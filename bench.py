#!/usr/bin/env python3
"""bench.py - decode tokens/s (+ TTFT) of the Qwen3-4B bf16 greedy forward-pass hot path on MI355X.

Workload (BASELINE.json configs[1], reference method bench_serving.rs:37-43,761-763,856-890):
  Qwen3-4B shape (36 layers, hidden 2560, 32/8 heads x 128, MLP 9728, vocab 151936, tied lm_head),
  bf16, greedy, hipGraph on, ONE request: synthetic prompt token_id = 100 + (i % 1000) of --ctx tokens
  (default 1024 = the reference's decode_heavy profile), then decode steps.  A "step" = one decode step
  (one new token through all 36 layers + lm_head + on-device greedy sampling + 4-byte D2H).
  Weights are a seeded synthetic checkpoint generated on the device (no weights on disk / no network;
  decode throughput is data-independent) unless PEGAINFER_TEST_MODEL_PATH points at a safetensors dir.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from rank 0.
  value = whole-job decode tokens/s over the K timed steps (max over ranks of the wall time).
  N > 1 (torchrun, one rank per GPU): independent requests shard across ranks with no data-path
  collective ("replicas": weak scaling, every rank decodes its own request); see DESIGN.md §multi-GPU.
Extra objects: roofline (dominant kernel, HBM bound), cpu_baseline (the reference's CPU path - HF Transformers bf16 - on the
host cores, on the engine's own checkpoint), parity (the benchmarked 36-layer model vs the oracle and vs HF, outside the
timed region).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def synthetic_prompt(n):
    return [100 + (i % 1000) for i in range(n)]  # bench_serving.rs synthetic_prompt_tokens


def algorithmic_bytes_per_token(cfg, ctx, batch=1):
    """weights once + KV read per request + KV write + logits."""
    H, L, I, V = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"], cfg["vocab_size"]
    q_dim = cfg["num_attention_heads"] * cfg["head_dim"]
    kv_dim = cfg["num_key_value_heads"] * cfg["head_dim"]
    per_layer = (q_dim + 2 * kv_dim) * H + H * q_dim + 2 * I * H + H * I
    norms = L * (2 * H + 2 * cfg["head_dim"]) + H
    lm = V * H * (1 if cfg["tie_word_embeddings"] else 2)
    params = L * per_layer + lm + norms - (0 if cfg["tie_word_embeddings"] else V * H)  # embed row read, not table
    weights = 2 * params
    kv_tok = L * 2 * kv_dim * 2
    return weights + batch * (kv_tok * ctx + kv_tok + 2 * H + 2 * V)


def export_checkpoint(eng, cfg):
    """The checkpoint the engine computes with (device-generated synthetic, or real weights), as HF name -> bf16 bits on
    the host: the CPU legs below run on EXACTLY these weights, so their logits can be compared with the engine's."""
    return eng.export_state()


def gpu_parity_run(eng, prompt, steps):
    """Outside the timed region: the engine's own greedy run on the bench prompt (prefill + `steps` decode steps, the
    same graph-captured decode path the timed steps used), logits rows kept for the CPU checkers."""
    rid = eng.new_request()
    tok, lg = eng.prefill([rid], [prompt], return_logits=True)
    toks, rows = [int(tok[0])], [lg[0].copy()]
    for _ in range(steps):
        tok, lg = eng.decode([rid], [toks[-1]], return_logits=True)
        toks.append(int(tok[0]))
        rows.append(lg[0].copy())
    eng.drop_request(rid)
    return toks, np.stack(rows)


def cpu_legs(cfg, bits, prompt, gpu_tokens, gpu_rows_bits, steps, threads, hf_repeats=3, hf_new_tokens=33, hf_threads=None,
             batch_rows=None):
    """bench.py's CPU side, on the host cores of the GPU box, on the SAME checkpoint and the SAME prompt as the GPU leg
    (rank 0, N = 1 only; a bounded sample).  Two engines:

      * the reference's CPU path: HF Transformers bf16 greedy `generate` (scripts/generate_test_data.py --device cpu;
        oracle/hf_engine.py) - timed: `cpu_baseline` with kind "reference" (the reference's own engine for this path, imported - it is Python) (TTFT, steady decode tok/s as
        bench_serving.rs:1005-1008 over `hf_new_tokens` - 1 inter-token gaps, median and spread of `hf_repeats`);
      * the oracle (oracle/qwen3_ref.py, numpy restatement with the reference's rounding points), teacher-forced on the
        GPU's own greedy tokens: the parity checker (cosine / max |dlogit| / greedy agreement per step) and the "port"
        timing of the previous rounds.

    Both are checkers / baselines only: nothing here is on the product path."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import ops as oracle_ops
    from oracle.bf16 import bf16_from_bits
    from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle
    keys = ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
            "intermediate_size", "vocab_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings"]
    c = Qwen3Config(**{k: cfg[k] for k in keys})
    gpu_rows = bf16_from_bits(gpu_rows_bits)
    V = gpu_rows.shape[-1]

    def cos_rows(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)

    parity = {"checkpoint": "the engine's own weights exported to the host (pegainfer_qwen3_export_tensor)",
              "prompt_tokens": len(prompt), "steps": len(gpu_tokens), "gpu_tokens": gpu_tokens}
    baseline = None
    # ---- the reference's CPU path: HF Transformers on the same weights ----
    try:
        from oracle import hf_engine
        t0 = time.perf_counter()
        hf_threads = int(hf_threads or threads)
        model = hf_engine.build_qwen3(cfg, bits, threads=hf_threads)
        build_s = time.perf_counter() - t0
        rates, ttfts, hf_tokens = [], [], None
        for rep in range(hf_repeats):
            t0 = time.perf_counter()
            toks, stamps, lg = hf_engine.generate_greedy(model, prompt, hf_new_tokens, return_logits=(rep == 0))
            if rep == 0:
                hf_tokens, hf_logits = toks, lg
            ttfts.append(stamps[1] - t0)
            rates.append((len(stamps) - 2) / (stamps[-1] - stamps[1]))
        # ... and HF teacher-forced on the ENGINE's own tokens (the free-running comparison ends at the first near-tie -
        # two steps on the driver's box): every step of the parity run compared, whatever the greedy choices did
        try:
            tf = hf_engine.teacher_forced_logits(model, prompt, gpu_tokens[:-1])
            m_tf = min(len(tf), len(gpu_rows))
            d_tf = np.abs(gpu_rows[:m_tf] - tf[:m_tf]).max(-1)
            srt_tf = np.sort(tf[:m_tf], axis=-1)
            agree_tf = gpu_rows[:m_tf].argmax(-1) == tf[:m_tf].argmax(-1)
            hf_forced = {"steps_compared": int(m_tf), "cos_min": round(float(cos_rows(gpu_rows[:m_tf], tf[:m_tf]).min()), 6),
                         "max_dlogit": round(float(d_tf.max()), 4), "argmax_equal": int(agree_tf.sum()),
                         "argmax_differing_away_from_a_near_tie": int((~agree_tf & ((srt_tf[:, -1] - srt_tf[:, -2]) > 2 * d_tf)).sum())}
        except Exception as e:  # noqa: BLE001
            hf_forced = {"error": f"{type(e).__name__}: {e}"[:200]}
        del model
        n = min(len(gpu_tokens), len(hf_tokens))
        first_diff = next((i for i in range(n) if gpu_tokens[i] != hf_tokens[i]), None)
        # logits are comparable while both engines are on the same token stream, i.e. up to the first difference
        m = n if first_diff is None else first_diff + 1
        cs = cos_rows(gpu_rows[:m], hf_logits[:m])
        srt = np.sort(hf_logits[:n], axis=-1)
        parity["hf"] = {"engine": "transformers bf16 generate(do_sample=False), the reference's truth engine",
                        "tokens": hf_tokens[:n], "tokens_equal_prefix": n if first_diff is None else first_diff,
                        "first_diff_step": first_diff, "steps_compared": m, "cos_min": round(float(cs.min()), 6),
                        "max_dlogit": round(float(np.abs(gpu_rows[:m] - hf_logits[:m]).max()), 4),
                        "hf_top1_margin": [round(float(x), 4) for x in (srt[:, -1] - srt[:, -2])[:m]],
                        "teacher_forced_on_engine_tokens": hf_forced}
        baseline = {"value": round(float(np.median(rates)), 3), "unit": "tokens/s", "cores": int(hf_threads),
                    "kind": "reference", "engine": "hf-transformers " + __import__("transformers").__version__,
                    "spread": {"min": round(float(min(rates)), 3), "max": round(float(max(rates)), 3), "repeats": hf_repeats},
                    "ttft_s": round(float(np.median(ttfts)), 3),
                    "sample": f"HF Transformers Qwen3ForCausalLM bf16 on the host cores ({hf_threads} torch threads of {threads} cores), the "
                              f"reference's CPU path (scripts/generate_test_data.py --device cpu) on the engine's own "
                              f"checkpoint: {hf_repeats} x generate({len(prompt)}-token bench prompt -> {hf_new_tokens} tokens, "
                              f"greedy); value = steady decode tokens/s over the {hf_new_tokens - 1} inter-token gaps "
                              f"(median of the repeats), model build {build_s:.0f} s untimed"}
    except Exception as e:  # noqa: BLE001 - transformers absent / out of memory: the port below still reports
        parity["hf"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # ---- the oracle on the same weights, teacher-forced on the GPU's tokens ----
    oracle_ops.GEMM_ACCUM = np.float32   # fp32 sgemm: what a CPU engine (and cuBLAS COMPUTE_32F) accumulates in
    t_setup = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, min(32, threads))) as pool:
        names = list(bits)
        w = dict(zip(names, pool.map(lambda k: bf16_from_bits(bits[k]), names)))
    m = Qwen3Oracle(c, w, num_pages=-(-(len(prompt) + steps + 1) // 16) + 2, rope_positions=len(prompt) + steps + 16)
    st = KvState()
    rows = [m.batch_prefill([prompt], [st])[0]]
    setup_s = time.perf_counter() - t_setup
    t0 = time.perf_counter()
    for tk in gpu_tokens[:-1]:
        rows.append(m.batch_decode([tk], [st])[0])
    dt = time.perf_counter() - t0
    R = np.stack(rows)
    cs = cos_rows(gpu_rows, R)
    dl = np.abs(gpu_rows - R).max(-1)
    agree = gpu_rows.argmax(-1) == R.argmax(-1)
    srt = np.sort(R, axis=-1)
    margin = srt[:, -1] - srt[:, -2]
    # the fp32 truth pass of the same oracle (no activation rounding) and the DERIVED bar of oracle/parity.py:
    # err(engine vs truth) / err(oracle vs truth), ~1.0 when the engine differs from the reference by summation order only
    try:
        from oracle import parity as par
        from oracle.bf16 import exact_activations
        with exact_activations():
            mt = Qwen3Oracle(c, w, num_pages=-(-(len(prompt) + steps + 1) // 16) + 2, rope_positions=len(prompt) + steps + 16)
            stt = KvState()
            trows = [mt.batch_prefill([prompt], [stt])[0]]
            for tk in gpu_tokens[:-1]:
                trows.append(mt.batch_decode([tk], [stt])[0])
        dv = par.derived(gpu_rows, R, np.stack(trows))
        parity["derived"] = {"ratio_pooled": round(dv["ratio_pooled"], 4), "ratio_max": round(dv["ratio_max"], 4),
                             "bar": {"pooled": par.AGG_MAX, "single_step": par.STEP_MAX},
                             "within_bar": bool(dv["ratio_pooled"] <= par.AGG_MAX and dv["ratio_max"] <= par.STEP_MAX),
                             "cos_engine_vs_truth_min": round(dv["cos_engine_vs_truth_min"], 6),
                             "cos_oracle_vs_truth_min": round(dv["cos_oracle_vs_truth_min"], 6),
                             "what": "RMS logit error of the engine against the fp32-activation pass of the oracle, over that of "
                                     "the bf16 oracle against the same pass, per step and pooled (oracle/parity.py)"}
        del mt
        # the batch sweep's columns (every request of a bs-N step carries the bench prompt and is teacher-forced on the same
        # tokens, so ONE oracle / truth stream referees all of them): the derived ratio per batch size
        if batch_rows:
            T_ = np.stack(trows)
            parity["batch"] = {}
            for label, rb in batch_rows.items():
                got = bf16_from_bits(rb)                                  # [bs, 1 + steps, V]
                m_ = min(got.shape[1], len(R))
                bd = par.derived(got[:, :m_], np.broadcast_to(R[:m_], got[:, :m_].shape), np.broadcast_to(T_[:m_], got[:, :m_].shape))
                b_ok, b_agree, _, _ = par.near_tie_ok(got[:, :m_], R[:m_][None], R[:m_][None])
                parity["batch"][label] = {"ratio_pooled": round(bd["ratio_pooled"], 4), "ratio_max": round(bd["ratio_max"], 4),
                                          "within_bar": bool(bd["ratio_pooled"] <= par.AGG_MAX and bd["ratio_max"] <= par.STEP_MAX),
                                          "cos_engine_vs_oracle_min": round(bd["cos_engine_vs_oracle_min"], 6),
                                          "tokens_equal": [int(b_agree.sum()), int(b_agree.size)],
                                          "tokens_differing_away_from_a_near_tie": int((~b_ok).sum())}
    except Exception as e:  # noqa: BLE001
        parity["derived"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    parity.update({"cos_min": round(float(cs.min()), 6), "max_dlogit": round(float(dl.max()), 4),
                   "logit_scale": round(float(np.abs(R).max()), 3),
                   "tokens_equal": int(agree.sum()), "tokens_differing_away_from_a_near_tie": int((~agree & (margin > 2 * dl)).sum()),
                   "oracle": "oracle/qwen3_ref.py teacher-forced on the GPU's greedy tokens (prefill + every decode step)"})
    port = {"value": round((len(gpu_tokens) - 1) / dt, 3), "unit": "tokens/s", "cores": int(threads), "kind": "port",
            "sample": f"{len(gpu_tokens) - 1} decode steps of oracle/qwen3_ref.py (numpy fp32 sgemm) at ctx {len(prompt)}, same "
                      f"checkpoint; {dt:.1f} s timed, {setup_s:.0f} s untimed (weights to f32 + the {len(prompt)}-token prefill)"}
    if baseline is None:
        baseline = port
    else:
        baseline["port"] = port
    return baseline, parity


def traffic_probe_main(model, bs):
    """`bench.py --traffic-probe` (run UNDER rocprofv3 --pmc by measure_traffic): a 2-layer engine of the benchmarked
    shape launches the dominant kernel - the gate_up GEMV with its add + RMSNorm prologue and SwiGLU epilogue - 40
    times over cold weights.  No torch, no timing: the counters are the product."""
    if model == "qwen3.5-4b":   # configs[3]: the gate|up GEMV of the hybrid model ((1 + w) norm prologue, SwiGLU epilogue)
        from pegainfer_amd.qwen35 import QWEN35_4B, Qwen35Engine
        cfg = dict(QWEN35_4B, num_hidden_layers=4, layer_types=QWEN35_4B["layer_types"][:4])
        eng = Qwen35Engine(cfg, num_kv_pages=8, max_batch_size=1, max_positions=4096)
        eng.fill_synthetic(seed=42, std=0.02)
        eng.bench_gemv(0, PROBE_LAUNCHES - 3)
        eng.close()
        return 0
    from pegainfer_amd.qwen3 import QWEN3_4B, QWEN3_8B, Qwen3Engine
    cfg = dict(QWEN3_4B if model == "qwen3-4b" else QWEN3_8B, num_hidden_layers=2)
    eng = Qwen3Engine(cfg, num_kv_pages=8, max_batch_size=max(bs, 1), decode_mode=1, max_positions=4096)
    eng.fill_synthetic(seed=42, std=0.02)
    eng.bench_gemv(5, PROBE_LAUNCHES - 3, bs)
    eng.close()
    return 0


PROBE_LAUNCHES = 43   # bench_gemv(5, 40, bs): 3 warm-up + 40 timed launches of the gate_up call site


def pick_call_site_kernel(per_kernel, launches=PROBE_LAUNCHES):
    """per_kernel: {kernel name: (sum, count)} of one counter pass over the traffic probe.  The dominant GEMV is whatever
    kernel the gate_up call site dispatched - a dot2 instantiation for Qwen3-4B, another one for hidden 4096, the skinny MFMA
    kernel at bs >= 3 - so it is identified by the probe's launch count (the only kernel launched exactly `launches` times
    whose name says GEMV / GEMM; the largest mean value wins a tie), not by a template-argument prefix."""
    cand = [(tot / n, k) for k, (tot, n) in per_kernel.items()
            if n == launches and any(t in k for t in ("gemv", "skinny", "gemm"))]
    return max(cand)[1] if cand else None


def measure_traffic(model, bs, kernel_prefix=None):
    """HBM bytes per launch of the dominant kernel from the PMC counters, measured NOW: FETCH_SIZE and WRITE_SIZE in their
    own rocprofv3 passes (they do not fit one pass: MI355X_MICROARCH.md 'rocprofv3 PMC slots'), --kernel-trace only, KiB
    per dispatch, FETCH_SIZE x 2 (the guide's gfx950 correction: a wide coalesced stream is reported at half).  Returns
    (bytes, description) or (None, reason)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not on this box"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "run", "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-probe", "--model", model, "--batch", str(bs)]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                                   stderr=subprocess.STDOUT, timeout=180)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} timed out"
            dbs = [os.path.join(r_, f) for r_, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            db = sqlite3.connect(dbs[0])
            cols = [c[1] for c in db.execute("pragma table_info(counters_collection)")]
            kname = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "kernel" in c][0]
            cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
            vname = "value" if "value" in cols else [c for c in cols if "value" in c][0]
            per = {}
            for k, c, v in db.execute(f"select {kname}, {cname}, {vname} from counters_collection"):
                if c == counter:
                    k = k.replace("void ", "").replace("pk::", "")
                    t_, n_ = per.get(k, (0.0, 0))
                    per[k] = (t_ + float(v), n_ + 1)
            db.close()
            if kernel_prefix is None:
                kernel_prefix = pick_call_site_kernel(per)
                if kernel_prefix is None:
                    return None, f"no kernel with {PROBE_LAUNCHES} dispatches in the {counter} pass: " + str(sorted((n_, k[:40]) for k, (_, n_) in per.items())[-4:])
            hit = [(t_, n_) for k, (t_, n_) in per.items() if kernel_prefix in k]
            if not hit:
                return None, f"no {kernel_prefix} dispatch in the {counter} pass"
            vals[counter] = (sum(t_ for t_, _ in hit) / sum(n_ for _, n_ in hit), sum(n_ for _, n_ in hit))
            vals["kernel"] = kernel_prefix
    traffic = int(2 * vals["FETCH_SIZE"][0] * 1024 + vals["WRITE_SIZE"][0] * 1024)
    return traffic, (f"kernel {vals['kernel'][:60]} (found by the probe's launch count); "
                     "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) "
                     f"over {vals['FETCH_SIZE'][1]} dispatches of the kernel on cold weights; KiB per dispatch, FETCH_SIZE x 2 "
                     "(gfx950 correction, MI355X_MICROARCH.md)")


def run_tp_leg(make_engine, prompt, steps, warmup, world, full_cfg, fence, max_over_ranks):
    """The sharded leg of an N > 1 bench line: ONE request through the tensor-parallel engine of every rank, timed by the
    contract's protocol (warm-up, `steps` decode steps between two fences = barrier + device sync, MAX over ranks).
    make_engine() -> an attached TP engine; host-only logic, so tests/test_parallel_gloo.py runs it at world 2 over gloo with
    a stand-in engine."""
    e = make_engine()
    r_ = e.new_request()
    fence()
    t0 = time.perf_counter()
    tk = e.prefill([r_], [prompt])
    ttft = max_over_ranks(time.perf_counter() - t0) * 1e3
    for _ in range(warmup):
        tk = e.decode([r_], tk)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        tk = e.decode([r_], tk)
    fence()
    el = max_over_ranks(time.perf_counter() - t0)
    dev = e.last_step_ms()
    e.close()
    ctx = len(prompt) + warmup + steps / 2
    b = algorithmic_bytes_per_token(full_cfg, ctx, 1)
    return {"tok_s": round(steps / el, 2), "ms_per_step": round(el / steps * 1e3, 4), "device_ms_last_step": round(float(dev), 4),
            "ttft_ms": round(ttft, 3), "scaling": "strong", "parallelism": "tp%d" % world, "steps": steps, "warmup": warmup,
            "all_reduces_per_step": 2 * full_cfg["num_hidden_layers"],
            "frac_of_aggregate_8TBps": round(b / (el / steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "what": "ONE request sharded over the ranks (q/k/v/gate/up by rows, o/down by columns, weights.rs:121-291), sum "
                    "all-reduce of [hidden] bf16 after o_proj and down_proj inside the captured graph"}


MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md); never the 2:1-sparsity figure


def prefill_roofline(c, T, ttft_ms, tp_world=1):
    """TTFT against the MFMA roofline: layer GEMMs 2*params*T, lm_head on the last token only,
    causal attention 2*Hq*D*T^2 per layer (QK^T + PV over the causal half)."""
    H, I, L, V = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"], c["vocab_size"]
    qd, kvd = c["num_attention_heads"] * c["head_dim"], c["num_key_value_heads"] * c["head_dim"]
    layer_params = (qd + 2 * kvd) * H + H * qd + 3 * I * H
    flops = 2.0 * layer_params * L * T + 2.0 * V * H + 2.0 * qd * T * T * L
    ach = flops / (ttft_ms * 1e-3) / 1e12
    return {"bound": "mfma", "flops": flops, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS * tp_world,
            "unit": "TFLOP/s", "frac": round(ach / (MFMA_PEAK_TFLOPS * tp_world), 4), "prompt_tokens": T}

def qwen35_bytes_per_token(c, ctx, batch):
    """Algorithmic HBM bytes of one Qwen3.5 decode step: every weight once (bf16; A_log / gated-norm f32), the KV of
    the full-attention layers (2 * Hkv * 256 * 2 B per token per layer), and per request the fp32 delta-rule state
    read + written (2 * vh*128*128*4 B per linear layer) plus the conv window."""
    H, I, V = c["hidden_size"], c["intermediate_size"], c["vocab_size"]
    qd, kvd = c["num_attention_heads"] * c["head_dim"], c["num_key_value_heads"] * c["head_dim"]
    kh, vh = c["linear_num_key_heads"], c["linear_num_value_heads"]
    C, Z = 2 * kh * 128 + vh * 128, vh * 128
    n_full = sum(t == "full_attention" for t in c["layer_types"])
    n_lin = len(c["layer_types"]) - n_full
    mlp = 3 * I * H
    full = 2 * qd * H + 2 * kvd * H + H * qd
    lin = C * H + Z * H + 2 * vh * H + H * Z + C * c.get("linear_conv_kernel_dim", 4)
    params = V * H + n_full * (full + mlp) + n_lin * (lin + mlp)
    kv = n_full * 2 * kvd * 2 * ctx
    state = n_lin * (2 * vh * 128 * 128 * 4 + 2 * C * 3 * 2)
    return 2.0 * params + batch * (kv + state)


def run_qwen35(args, rank, world, local, dist, torch):
    """configs[3]: Qwen3.5-4B hybrid (24 linear + 8 full-attention layers) - same decode_heavy profile, one
    independent request stream per GPU (the reference supports exactly one device for this model: replicas)."""
    from pegainfer_amd import parallel
    from pegainfer_amd.qwen35 import QWEN35_4B, Qwen35Engine
    cfg = dict(QWEN35_4B)
    q35_mode = int(os.environ.get("PEGAINFER_Q35_DECODE_MODE", "1"))
    total_ctx = args.ctx + args.warmup + args.steps + 8 + 66
    pages = (args.batch + 1) * (-(-total_ctx // 16) + 1) + 8
    eng = Qwen35Engine(cfg, num_kv_pages=pages, max_batch_size=max(args.batch, 1) + 1, enable_graph=not args.no_graph,
                       device=local, max_positions=max(4096, total_ctx + 16))
    eng.fill_synthetic(seed=42 + rank, std=0.02)
    prompt = synthetic_prompt(args.ctx)
    ttfts = []
    for _ in range(1 + args.ttft_iters):
        r = eng.new_request()
        t0 = time.perf_counter()
        eng.prefill(r, prompt)
        ttfts.append((time.perf_counter() - t0) * 1e3)
        eng.drop_request(r)
    ttfts = sorted(ttfts[1:]) if len(ttfts) > 1 else ttfts
    rids = [eng.new_request() for _ in range(args.batch)]
    toks = np.array([eng.prefill(r, prompt) for r in rids], np.int32)
    for _ in range(args.warmup):
        toks = eng.decode(rids, toks)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_ms, dev_ms = [], []
    chain = args.chain if args.batch <= 4 else 0     # greedy steps in chains (pegainfer_qwen35_decode_greedy_chain), as in main()
    if chain:
        toks = eng.decode_greedy_chain(rids, toks, 2)[-1]      # allocates the pinned ring outside the timed region
    barrier()
    t_start = time.perf_counter()
    done = 0
    while done < args.steps:
        t0 = time.perf_counter()
        m = min(chain, args.steps - done) if chain else 1
        toks = eng.decode_greedy_chain(rids, toks, m)[-1] if chain else eng.decode(rids, toks)
        dt = (time.perf_counter() - t0) * 1e3 / m
        step_ms.extend([dt] * m)
        dev_ms.extend([eng.last_step_ms()] * m)
        done += m
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t_start, device="cuda")
    sync_ms = []
    if chain:   # the reference's loop shape (one synchronisation per step) right after, for the per-step percentiles
        for _ in range(min(args.steps, 64)):
            t0 = time.perf_counter()
            toks = eng.decode(rids, toks)
            sync_ms.append((time.perf_counter() - t0) * 1e3)
    value = args.steps * args.batch * world / elapsed
    step_bytes = qwen35_bytes_per_token(cfg, args.ctx + args.warmup + args.steps / 2, args.batch)
    out = {
        "metric": "decode tokens/sec + TTFT, Qwen3.5-4B bf16 greedy, 1xMI355X",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"qwen3.5-4b hybrid (24 linear + 8 full-attention layers) greedy decode, hipGraph "
                               f"{'off' if args.no_graph else 'on'}, bs={args.batch}/GPU, ctx {args.ctx}->"
                               f"{args.ctx + args.warmup + args.steps}, "
                               + ("fused bs=1 decode kernels" if q35_mode == 1 and args.batch == 1 else "reference op sequence"),
                   "batch_per_gpu": args.batch, "ctx": args.ctx, "decode_mode": q35_mode,
                   "parallelism": "replicas%d" % world if world > 1 else "single"},
        "ttft_ms": {"prompt_tokens": args.ctx, "p50": round(float(np.median(ttfts)), 3),
                    "min": round(float(min(ttfts)), 3), "iters": len(ttfts)},
        "tpot_ms": ({"p50": round(float(np.median(sync_ms)), 4), "p95": round(float(np.percentile(sync_ms, 95)), 4),
                     "percentiles_from": f"{len(sync_ms)} steps of the sync-per-step loop after the timed region",
                     "device_p50": round(float(np.median(dev_ms)), 4), "mean": round(elapsed / args.steps * 1e3, 4),
                     "mean_is": f"timed region: chains of {chain} greedy steps, one host synchronisation per chain"}
                    if chain and sync_ms else
                    {"p50": round(float(np.median(step_ms)), 4), "p95": round(float(np.percentile(step_ms, 95)), 4),
                     "percentiles_from": "the timed region (one host synchronisation per step)",
                     "device_p50": round(float(np.median(dev_ms)), 4)}),
        "host_loop": ({"form": "chained", "chain_steps": chain,
                       "sync_per_step": {"tpot_ms_p50": round(float(np.median(sync_ms)), 4),
                                         "tok_s": round(args.batch * world * 1e3 / float(np.mean(sync_ms)), 2), "steps": len(sync_ms)}}
                      if chain and sync_ms else {"form": "sync_per_step"}),
        "roofline": None,
        "step_roofline": {"algorithmic_bytes_per_step": int(step_bytes),
                          "achieved_GBps": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
                          "frac_of_8TBps": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                          "timed_by": "ms_per_step (wall, the contract's timed region)",
                          "device_frac_of_8TBps": round(step_bytes / (np.median(dev_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "cpu_baseline": None,
    }
    if rank == 0:
        # the dominant kernel of the step: the gate|up GEMV (residual add + (1 + w) RMSNorm prologue, SwiGLU epilogue), 32
        # launches per step, 2 * I * H bf16 weights each = 42 % of the step's bytes; timed live over the layers' weights
        H, I = cfg["hidden_size"], cfg["intermediate_size"]
        kbytes = 2 * I * H * 2 + (2 * H * 2 + H * 2 + I * 2) + H * 2
        ms = eng.bench_gemv(0, 320) if (q35_mode == 1 and args.batch == 1) else -1.0
        if ms > 0:
            out["roofline"] = {"bound": "hbm", "kernel": "gemv_fused_kernel (gate|up, M=%d K=%d N=1, add + (1+w) RMSNorm prologue, SwiGLU epilogue)" % (2 * I, H),
                               "achieved": round(kbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(kbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "step_frac": out["step_roofline"]["frac_of_8TBps"], "traffic": None,
                               "bytes_per_launch": kbytes, "avg_launch_us": round(ms * 1e3, 2)}
            if world == 1 and os.environ.get("PEGAINFER_BENCH_TRAFFIC", "1") != "0":
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measure_traffic("qwen3.5-4b", 1)
        else:
            out["roofline"] = {"bound": "hbm", "kernel": "whole decode step (graph)", "achieved": out["step_roofline"]["achieved_GBps"],
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": out["step_roofline"]["frac_of_8TBps"],
                               "traffic": None, "bytes_per_launch": int(step_bytes)}
    # ---- CPU side (rank 0, N = 1): the reference's CPU path (HF Transformers bf16, the hybrid model with its torch
    #      fallbacks of the gated delta rule) on the engine's own exported checkpoint: cpu_baseline + a parity block ----
    cpu_in = None
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        try:
            for r in rids:
                eng.drop_request(r)
            r = eng.new_request()
            tok, lg = eng.prefill(r, prompt, want_logits=True)
            gtoks, grows = [int(tok)], [lg]
            for _ in range(args.cpu_steps):
                t, lg = eng.decode([r], [gtoks[-1]], want_logits=True)
                gtoks.append(int(t[0]))
                grows.append(lg[0])
            cpu_in = (eng.export_state(), gtoks, np.stack(grows))
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    eng.close()
    if cpu_in is not None:
        try:
            out["cpu_baseline"], out["parity"] = cpu_leg_qwen35(cfg, cpu_in[0], prompt, cpu_in[1], cpu_in[2],
                                                                min(32, os.cpu_count() or 1), os.cpu_count() or 1)
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return 0


def cpu_leg_qwen35(cfg, tensors, prompt, gpu_tokens, gpu_rows, hf_threads, cores, hf_repeats=2, hf_new_tokens=17):
    """configs[3]'s CPU side: HF Transformers Qwen3_5ForCausalLM bf16 greedy generate (the reference's truth engine for this
    model too, scripts/generate_test_data.py) on the engine's exported weights - timed as the cpu_baseline, and compared
    with the GPU's greedy tokens / logits up to the first difference.  Checker / baseline only."""
    from oracle import hf_engine
    t0 = time.perf_counter()
    model = hf_engine.build_qwen35(dict(cfg, max_position_embeddings=max(4096, len(prompt) + 64)), tensors, threads=hf_threads)
    build_s = time.perf_counter() - t0
    rates, ttfts, hf_tokens, hf_logits = [], [], None, None
    for rep in range(hf_repeats):
        t0 = time.perf_counter()
        toks, stamps, lg = hf_engine.generate_greedy(model, prompt, hf_new_tokens, return_logits=(rep == 0))
        if rep == 0:
            hf_tokens, hf_logits = toks, lg
        ttfts.append(stamps[1] - t0)
        rates.append((len(stamps) - 2) / (stamps[-1] - stamps[1]))
    n = min(len(gpu_tokens), len(hf_tokens))
    first_diff = next((i for i in range(n) if gpu_tokens[i] != hf_tokens[i]), None)
    m = n if first_diff is None else first_diff + 1
    a, b = gpu_rows[:m].astype(np.float64), hf_logits[:m].astype(np.float64)
    cs = (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)
    srt = np.sort(hf_logits[:m], axis=-1)
    parity = {"checkpoint": "the engine's own weights exported to the host (pegainfer_qwen35_export_tensor)",
              "prompt_tokens": len(prompt), "steps": n, "gpu_tokens": gpu_tokens[:n],
              "hf": {"engine": "transformers Qwen3_5ForCausalLM bf16 generate(do_sample=False)", "tokens": hf_tokens[:n],
                     "tokens_equal_prefix": n if first_diff is None else first_diff, "first_diff_step": first_diff,
                     "steps_compared": m, "cos_min": round(float(cs.min()), 6),
                     "max_dlogit": round(float(np.abs(a - b).max()), 4), "logit_scale": round(float(np.abs(b).max()), 3),
                     "hf_top1_margin": [round(float(x), 4) for x in (srt[:, -1] - srt[:, -2])]}}
    del model
    # ---- the oracle (oracle/qwen35_ref.py) and its fp32 truth pass, teacher-forced on the GPU's own greedy tokens: where
    #      the engine stands between them (derived bar, oracle/parity.py).  At 32 layers HF, the oracle and the truth pass are
    #      mutually ~0.993 apart on N(0, 0.02) weights (tests/golden/qwen35_4b_depth32_hf.json): the cosine against HF above is
    #      this model's bf16 noise floor, the ratio below says whether the engine adds to it ----
    try:
        from oracle import parity as par
        from oracle.bf16 import bf16_from_bits
        from oracle.qwen35_ref import Qwen35Config
        oc = Qwen35Config(**{k: v for k, v in cfg.items() if k != "max_position_embeddings"})
        w = {k: (v if v.dtype == np.float32 else bf16_from_bits(v)) for k, v in tensors.items()}
        feed = gpu_tokens[:-1]
        t0 = time.perf_counter()
        R, T_ = par.qwen35_pass_pair(oc, w, prompt, feed, max_pos=len(prompt) + len(feed) + 16)   # both passes side by side
        dv = par.derived(gpu_rows[:len(R)], R, T_)
        ok, agree, margin, dmax = par.near_tie_ok(gpu_rows[:len(R)], R, R)
        parity["oracle"] = {"engine": "oracle/qwen35_ref.py teacher-forced on the GPU's greedy tokens (prefill + every decode step)",
                            "cos_min": round(dv["cos_engine_vs_oracle_min"], 6), "max_dlogit": round(float(dmax.max()), 4),
                            "tokens_equal": int(agree.sum()), "tokens_differing_away_from_a_near_tie": int((~ok).sum()),
                            "seconds": round(time.perf_counter() - t0, 1)}
        parity["derived"] = {"ratio_pooled": round(dv["ratio_pooled"], 4), "ratio_max": round(dv["ratio_max"], 4),
                             "bar": {"pooled": par.AGG_MAX, "single_step": par.STEP_MAX},
                             "within_bar": bool(dv["ratio_pooled"] <= par.AGG_MAX and dv["ratio_max"] <= par.STEP_MAX),
                             "cos_engine_vs_truth_min": round(dv["cos_engine_vs_truth_min"], 6),
                             "cos_oracle_vs_truth_min": round(dv["cos_oracle_vs_truth_min"], 6)}
    except Exception as e:  # noqa: BLE001
        parity["oracle"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    baseline = {"value": round(float(np.median(rates)), 3), "unit": "tokens/s", "cores": int(hf_threads), "kind": "reference",
                "engine": "hf-transformers " + __import__("transformers").__version__,
                "spread": {"min": round(float(min(rates)), 3), "max": round(float(max(rates)), 3), "repeats": hf_repeats},
                "ttft_s": round(float(np.median(ttfts)), 3),
                "sample": f"HF Transformers Qwen3_5ForCausalLM bf16 ({hf_threads} torch threads of {cores} cores; torch fallbacks "
                          f"of the gated delta rule) on the engine's own checkpoint: {hf_repeats} x generate({len(prompt)}-token "
                          f"bench prompt -> {hf_new_tokens} tokens, greedy); value = steady decode tokens/s over the "
                          f"{hf_new_tokens - 1} inter-token gaps, model build {build_s:.0f} s untimed"}
    return baseline, parity



def run_serving(args, rank, world, local, dist, torch):
    """Serving profile: C requests in flight through the C++ scheduler (admission by KV budget, batched prefill /
    unified / decode steps).  One engine + scheduler per GPU (replicas); value = all output tokens / wall time."""
    from pegainfer_amd import parallel
    from pegainfer_amd.qwen3 import QWEN3_4B, QWEN3_8B, Qwen3Engine
    from pegainfer_amd.scheduler import FINISHED, TOKEN, Scheduler
    cfg = dict(QWEN3_4B if args.model == "qwen3-4b" else QWEN3_8B)
    C, out_len = args.concurrency, args.steps
    per_req_pages = -(-(args.ctx + out_len) // 16) + 1
    eng = Qwen3Engine(cfg, num_kv_pages=C * per_req_pages + 16, max_batch_size=max(C, 1), enable_graph=not args.no_graph,
                      decode_mode=args.decode_mode, split_policy=args.split_policy, device=local,
                      max_positions=max(4096, args.ctx + out_len + 16))
    eng.fill_synthetic(seed=42 + rank, std=0.02)
    prompt = synthetic_prompt(args.ctx)
    warm = Scheduler.over_engine(eng)                       # warm-up: graphs / workspaces for this batch shape
    for _ in range(C):
        warm.submit(prompt, min(args.warmup + 1, out_len), (0.0, -1, 1.0, True))
    while warm.step() != 0:
        warm.poll()
    warm.close()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sched = Scheduler.over_engine(eng)
    barrier()
    t0 = time.perf_counter()
    for _ in range(C):
        sched.submit(prompt, out_len, (0.0, -1, 1.0, True))
    first, done, ntok, plans = {}, 0, 0, {1: 0, 2: 0, 3: 0}
    while done < C:
        plan = sched.step()
        now = time.perf_counter()
        if plan in plans:
            plans[plan] += 1
        for rid, kind, *_ in sched.poll():
            if kind == TOKEN:
                ntok += 1
                first.setdefault(rid, (now - t0) * 1e3)
            elif kind == FINISHED:
                done += 1
            else:
                raise SystemExit("scheduler reported an error: " + sched.last_message())
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device="cuda")
    ttft = sorted(first.values())
    out = {
        "metric": "decode tokens/sec + TTFT, Qwen3-4B bf16 greedy, 1xMI355X" if args.model == "qwen3-4b"
                  else "decode tokens/sec + TTFT, Qwen3-8B bf16 greedy, 1xMI355X",
        "value": round(ntok * world / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / max(sum(plans.values()), 1), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.model} serving through the continuous-batching scheduler: {C} concurrent "
                               f"requests/GPU, prompt {args.ctx}, {out_len} output tokens each, greedy, ignore_eos "
                               "(reference bench_serving decode_heavy with concurrency)",
                   "concurrency": C, "ctx": args.ctx, "decode_mode": args.decode_mode,
                   "parallelism": "replicas%d" % world if world > 1 else "single"},
        "serving": {"wall_s": round(elapsed, 4), "output_tokens": ntok,
                    "ttft_ms": {"p50": round(float(np.median(ttft)), 2), "p95": round(float(np.percentile(ttft, 95)), 2),
                                "max": round(ttft[-1], 2)},
                    "scheduler_steps": {"prefill": plans[1], "decode": plans[2], "unified": plans[3]}},
        "roofline": None, "cpu_baseline": None,
    }
    if rank == 0:
        print(json.dumps(out))
    sched.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ctx", type=int, default=1024, help="prompt length before the timed decode steps")
    ap.add_argument("--batch", type=int, default=1, help="requests decoded together per rank")
    ap.add_argument("--model", default="qwen3-4b", choices=["qwen3-4b", "qwen3-8b", "qwen3.5-4b"])
    ap.add_argument("--decode-mode", type=int, choices=[0, 1], default=int(os.environ.get("PEGAINFER_DECODE_MODE", "1")),
                    help="0 = reference op sequence 1:1, 1 = fused MI355X decode kernels (bit-identical)")
    ap.add_argument("--split-policy", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8,
                    help="decode steps of the parity run (GPU vs oracle vs HF on the same checkpoint) and of the oracle's "
                         "timing; 0 = skip the whole CPU side")
    ap.add_argument("--ttft-iters", type=int, default=20, help="TTFT iterations at --ctx (reference: warmup 5, iters 20)")
    ap.add_argument("--ttft10k-iters", type=int, default=5,
                    help="iterations of the reference's prefill_heavy profile (10 000-token prompt -> 1 token, "
                         "bench_serving.rs:37-43); 0 = skip")
    ap.add_argument("--profile-iters", type=int, default=5,
                    help="iterations of the reference's decode_heavy profile run after the timed steps (prompt --ctx -> "
                         "256 tokens, TTFT + steady TPOT percentiles as bench_serving.rs:972-1032); 0 = skip")
    ap.add_argument("--concurrency", type=int, default=0,
                    help="serving mode (reference bench_serving.rs): this many requests (prompt --ctx, --steps output "
                         "tokens each, ignore_eos) go through the continuous-batching scheduler; reports aggregate "
                         "tok/s and the TTFT distribution")
    ap.add_argument("--sampling", default="greedy", choices=["greedy", "topk_topp", "topp"],
                    help="configs[2]: per-request gpu_sample after every step (ops_embedding_sampling_bench.rs:49-90): "
                         "topk_topp = T 0.8, top_k 50, top_p 0.95; topp = T 0.8, top_k -1, top_p 0.9")
    ap.add_argument("--parallelism", default="replicas", choices=["replicas", "tp"],
                    help="N>1: 'replicas' = one independent request stream per GPU (weak scaling, no data-path "
                         "collective; default); 'tp' = the reference's Qwen3 tensor parallel over RCCL "
                         "(strong scaling, 72 all-reduces per step)")
    ap.add_argument("--chain", type=int, default=32,
                    help="greedy decode steps enqueued per host synchronisation (pegainfer_qwen3_decode_greedy_chain); 0 = the "
                         "reference's loop shape, one synchronisation per step")
    ap.add_argument("--sweep-steps", type=int, default=20,
                    help="decode steps per point of the side batch sweep (bs 2 / 4 / 8 / 16 at --ctx) and of the drop-in-ABI "
                         "(decode_mode 0) side number; 0 = skip both")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.traffic_probe:
        return traffic_probe_main(args.model, args.batch)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher - one process per GPU over RCCL, exactly what the
        # driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` does
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        return subprocess.call(cmd, env=env)

    import torch
    import torch.distributed as dist
    from pegainfer_amd.qwen3 import QWEN3_4B, QWEN3_8B, Qwen3Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if world > 1:
        if torch.cuda.device_count() < world:
            raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) are visible")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != args.gpus or (world > 1 and dist.get_world_size() != args.gpus):
        raise SystemExit(f"--gpus {args.gpus} does not match the launched world size {world}: the line would "
                         "mis-report n_gpus")

    from pegainfer_amd import parallel
    if args.model == "qwen3.5-4b":
        return run_qwen35(args, rank, world, local, dist, torch)
    if args.concurrency > 0:
        return run_serving(args, rank, world, local, dist, torch)
    cfg = dict(QWEN3_4B if args.model == "qwen3-4b" else QWEN3_8B)
    full_cfg = dict(cfg)
    tp = args.parallelism == "tp" and world > 1
    if tp:
        cfg = parallel.tp_local_config(cfg, world)
    total_ctx = args.ctx + args.warmup + args.steps + 8 + 64   # + the sync-per-step side loop behind a chained run
    heavy_out = 256                                   # decode_heavy output length (bench_serving.rs:37-43)
    single_ctx = max(total_ctx, args.ctx + heavy_out + 8 if args.profile_iters > 0 else 0,
                     10000 + 16 if args.ttft10k_iters > 0 else 0)
    pages = max(args.batch * (-(-total_ctx // 16) + 1), -(-single_ctx // 16) + 1) + 8
    total_ctx = max(total_ctx, single_ctx)
    eng = Qwen3Engine(cfg, num_kv_pages=pages, max_batch_size=max(args.batch, 1), enable_graph=not args.no_graph,
                      decode_mode=args.decode_mode, split_policy=args.split_policy, device=local,
                      max_positions=max(4096, total_ctx + 16))
    path = os.environ.get("PEGAINFER_TEST_MODEL_PATH")
    if path and os.path.isdir(path):
        eng.load_safetensors(path)
        data = "synthetic prompt, real weights"
    else:
        eng.fill_synthetic(seed=42 if tp else 42 + rank, std=0.02)   # TP: replicated tensors must agree
        data = "synthetic"
    if tp:
        parallel.attach_tp(eng)

    prompt = synthetic_prompt(args.ctx)
    # ---- TTFT (submit -> first token): prefill of the ctx-token prompt, fresh request each time ----
    ttfts = []
    for _ in range(1 + args.ttft_iters):
        r = eng.new_request()
        t0 = time.perf_counter()
        eng.prefill([r], [prompt])
        ttfts.append((time.perf_counter() - t0) * 1e3)
        eng.drop_request(r)
    ttfts = sorted(ttfts[1:]) if len(ttfts) > 1 else ttfts

    rids = [eng.new_request() for _ in range(args.batch)]
    toks = eng.prefill(rids, [prompt] * args.batch)
    # greedy steps are enqueued in chains (pegainfer_qwen3_decode_greedy_chain: the token of step s reaches step s + 1 on
    # the device, every step's tokens still travel to the host asynchronously, ONE host synchronisation per chain) unless
    # --chain 0; sampling with temperature needs the host between steps, tensor parallel checks its status block per step
    # (measured, profiles/r5_chain_ab*.txt: bs 1 +1.3 ... 1.7 %, bs 4 +0.9 %, bs 16 -1 % - there the per-step copies queue up on
    # the stream behind 3.4 ms graphs and the host was never the limit - so chains are used up to 4 requests)
    chain = args.chain if (args.sampling == "greedy" and not tp and args.batch <= 4) else 0
    n_warm_chain = min(2, args.warmup) if chain else 0      # the pinned metadata ring is allocated by the first chain
    for _ in range(args.warmup - n_warm_chain):
        toks = eng.decode(rids, toks)
    if n_warm_chain:
        toks = eng.decode_greedy_chain(rids, toks, n_warm_chain)[-1]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_ms, dev_ms = [], []
    # everything with a first-call cost stays OUTSIDE the timed region (numpy's Generator import alone is ~12 ms)
    samp = {"greedy": None, "topk_topp": (0.8, 50, 0.95), "topp": (0.8, -1, 0.9)}[args.sampling]
    srng = np.random.default_rng(42 + rank)
    float(srng.random())
    eng.last_step_ms()
    barrier()
    t_start = time.perf_counter()
    done = 0
    while done < args.steps:
        t0 = time.perf_counter()
        if chain:
            m = min(chain, args.steps - done)
            toks = eng.decode_greedy_chain(rids, toks, m)[-1]
        else:
            m = 1
            toks = eng.decode(rids, toks)
            if samp:   # the reference samples request by request after the step (executor.rs:324-328)
                toks = np.array([eng.sample(i, samp[0], samp[1], samp[2], float(srng.random())) for i in range(len(rids))],
                                dtype=np.int32)
        dt = (time.perf_counter() - t0) * 1e3 / m
        step_ms.extend([dt] * m)
        dev_ms.extend([eng.last_step_ms()] * m)
        done += m
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = parallel.max_over_ranks(elapsed, device="cuda")
    # the reference's own loop shape - one host synchronisation per step (executor.rs:541-640) - on the same requests, after
    # the contract's timed region: the per-step TPOT percentiles, and what the chain is worth
    sync_ms = []
    if chain:
        for _ in range(min(args.steps, 64)):
            t0 = time.perf_counter()
            toks = eng.decode(rids, toks)
            sync_ms.append((time.perf_counter() - t0) * 1e3)

    tokens = args.steps * args.batch * (1 if tp else world)
    value = tokens / elapsed
    mean_ms, p50_ms = elapsed / args.steps * 1e3, float(np.median(step_ms))

    def pct(a, q):
        return round(float(np.percentile(a, q)), 4)

    # ---- the reference's two bench_serving profiles, run AFTER the contract's timed steps (not part of `value`) ----
    # decode_heavy (bench_serving.rs:37-43,972-1032): prompt --ctx -> 256 tokens, ignore_eos; TTFT = submit -> first
    # token, steady TPOT excludes the first decode step, decode tok/s = steps / sum of inter-token times
    for r in rids:
        eng.drop_request(r)
    rids = []
    heavy = None
    if args.profile_iters > 0:
        h_ttft, h_tpot, h_rate = [], [], []
        for it in range(args.profile_iters + 1):          # first iteration is warm-up
            r = eng.new_request()
            t0 = time.perf_counter()
            tk = eng.prefill([r], [prompt])
            t1 = time.perf_counter()
            gaps = []
            for _ in range(heavy_out - 1):
                tk = eng.decode([r], tk)
                t2 = time.perf_counter()
                gaps.append((t2 - t1) * 1e3)
                t1 = t2
            eng.drop_request(r)
            if it:
                h_ttft.append((t1 - t0) * 1e3 - sum(gaps))
                h_tpot.append(float(np.median(gaps[1:])))
                h_rate.append(len(gaps[1:]) / (sum(gaps[1:]) * 1e-3))
        heavy = {"profile": f"{args.ctx} -> {heavy_out} tokens, ignore_eos, iters {args.profile_iters} (+1 warm-up)",
                 "ttft_ms": {"p50": pct(h_ttft, 50), "p95": pct(h_ttft, 95)},
                 "steady_tpot_ms": {"p50": pct(h_tpot, 50), "p95": pct(h_tpot, 95)},
                 "decode_tok_s": round(float(np.median(h_rate)), 2)}
    # prefill_heavy (bench_serving.rs:37-43): 10 000-token prompt -> 1 token
    ttft10k = None
    if args.ttft10k_iters > 0:
        long_prompt = synthetic_prompt(10000)
        t10 = []
        for it in range(args.ttft10k_iters + 1):
            r = eng.new_request()
            t0 = time.perf_counter()
            eng.prefill([r], [long_prompt])
            t10.append((time.perf_counter() - t0) * 1e3)
            eng.drop_request(r)
        t10 = t10[1:]
        ttft10k = {"prompt_tokens": 10000, "p50": round(float(np.median(t10)), 3), "min": round(min(t10), 3),
                   "iters": len(t10),
                   "prefill_roofline": prefill_roofline(full_cfg, 10000, float(np.median(t10)), world if tp else 1)}
    # short prompts (the reference's golden prompts and README TTFT are this short, README.md:28-34): TTFT at 32 and 128
    # tokens, same protocol as ttft_ms (one warm-up, then --ttft-iters timed submissions), outside `value`
    ttft_short = None
    if args.ttft_iters > 1 and args.batch == 1:
        ttft_short = {}
        for n in (32, 128):
            sp, ts = synthetic_prompt(n), []
            for it in range(args.ttft_iters + 1):
                r = eng.new_request()
                t0 = time.perf_counter()
                eng.prefill([r], [sp])
                ts.append((time.perf_counter() - t0) * 1e3)
                eng.drop_request(r)
            ts = ts[1:]
            ttft_short[str(n)] = {"p50": round(float(np.median(ts)), 3), "min": round(min(ts), 3), "iters": len(ts)}
    # configs[4] side measurement (not part of `value`): the DeepSeek-V4 MP8 collective verbs over the N ranks
    mp8, mp8_hung = None, False
    if world > 1:
        # never let the side measurement take the headline line down with it: exceptions are reported in the line, and a
        # collective that never completes (a rank missing, a fabric problem) is abandoned after a deadline - every rank
        # runs the same watchdog, prints / exits on its own and skips the process-group teardown that would block too
        import threading
        box = {}

        def side():
            try:
                torch.cuda.set_device(local)
                ncomm = parallel.NativeComm(device=local)   # include/pegainfer_comm.h: RCCL on the caller's stream
                r = parallel.bench_mp8_collectives(ncomm, device=torch.device("cuda", local))
                r["transport"] = "native C ABI over RCCL (pegainfer_comm.h)"
                ncomm.close()
                box["mp8"] = r
            except Exception as e:  # noqa: BLE001
                box["mp8"] = {"error": f"{type(e).__name__}: {e}"[:300]}

        deadline = float(os.environ.get("PEGAINFER_BENCH_MP8_TIMEOUT", "120"))
        th = threading.Thread(target=side, daemon=True)
        th.start()
        th.join(timeout=deadline)
        mp8_hung = th.is_alive()
        mp8 = {"error": f"collective microbench did not finish within {deadline:.0f} s; abandoned"} if mp8_hung else box.get("mp8")
    # the SHARDED config next to the replicas value: north_star asks for tokens/s "at 2/4/8 GPUs for the
    # sharded config", and the driver's scaling run passes no flag - so every N > 1 line also carries the reference's Qwen3
    # tensor parallel (weights.rs:121-291; two sum all-reduces per layer inside the captured graph) as `tp`, timed by the same
    # protocol (warm-up, K steps between barriers, MAX over ranks).  One request, strong scaling.  Same watchdog as above.
    tp_leg, tp_hung = None, False
    if (world > 1 and not tp and not mp8_hung and os.environ.get("PEGAINFER_BENCH_TP_LEG", "1") != "0"
            and full_cfg["num_key_value_heads"] % world == 0):
        import threading
        tbox = {}

        def tp_side():
            try:
                torch.cuda.set_device(local)
                n_ctx = args.ctx + args.warmup + args.steps + 16

                def make_engine():
                    e = Qwen3Engine(parallel.tp_local_config(full_cfg, world), num_kv_pages=-(-n_ctx // 16) + 9, max_batch_size=1,
                                    enable_graph=not args.no_graph, decode_mode=args.decode_mode, split_policy=args.split_policy,
                                    device=local, max_positions=max(4096, n_ctx + 16))
                    e.fill_synthetic(seed=42, std=0.02)            # replicated tensors must agree across the ranks
                    parallel.attach_tp(e)
                    return e

                def fence():
                    dist.barrier()
                    torch.cuda.synchronize()
                tbox["tp"] = run_tp_leg(make_engine, prompt, args.steps, args.warmup, world, full_cfg, fence,
                                        lambda x: parallel.max_over_ranks(x, device="cuda"))
            except Exception as e_:  # noqa: BLE001
                tbox["tp"] = {"error": f"{type(e_).__name__}: {e_}"[:300]}

        th = threading.Thread(target=tp_side, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("PEGAINFER_BENCH_TP_TIMEOUT", "240")))
        tp_hung = th.is_alive()
        tp_leg = {"error": "tensor-parallel leg did not finish within its deadline; abandoned"} if tp_hung else tbox.get("tp")
        if tp_leg and "error" not in tp_leg and isinstance(mp8, dict) and "all_reduce_small" in mp8:
            tp_leg["all_reduce_us"] = mp8["all_reduce_small"].get("bf16_5KB")   # the payload of a decode step's all-reduce
        mp8_hung = mp8_hung or tp_hung
    ctx_mid = args.ctx + args.warmup + args.steps / 2
    step_bytes = algorithmic_bytes_per_token(full_cfg, ctx_mid, args.batch)

    sampling_name = {"greedy": "greedy", "topk_topp": "top-k/top-p sampling (T 0.8, k 50, p 0.95)",
                     "topp": "top-p sampling (T 0.8, p 0.9)"}[args.sampling]
    out = {
        "metric": "decode tokens/sec + TTFT, %s bf16 %s, 1xMI355X" % ("Qwen3-4B" if args.model == "qwen3-4b" else "Qwen3-8B", sampling_name),
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(mean_ms, 4), "higher_is_better": True,
        "scaling": "strong" if tp else "weak",
        "vs_baseline": None, "dtype": "bf16", "data": data,
        "config": {"workload": f"{args.model} {sampling_name} decode, hipGraph {'off' if args.no_graph else 'on'}, "
                               f"bs={args.batch}/GPU, ctx {args.ctx}->{args.ctx + args.warmup + args.steps} "
                               f"(reference decode_heavy: synthetic prompt 100+(i%1000))",
                   "batch_per_gpu": args.batch, "ctx": args.ctx, "decode_mode": args.decode_mode,
                   "sampling": args.sampling, "split_policy": args.split_policy, "parallelism": ("tp%d" % world if tp else "replicas%d" % world) if world > 1 else "single"},
        "ttft_ms": {"prompt_tokens": args.ctx, "p50": round(float(np.median(ttfts)), 3),
                    "min": round(float(min(ttfts)), 3), "iters": len(ttfts)},
        # per-step percentiles come from PER-STEP samples only: with chained steps the timed region yields one
        # average per chain, so p50 / p95 are taken from the sync-per-step loop run right after it on the same requests and
        # the chained figure is labelled a mean
        "tpot_ms": ({"p50": round(float(np.median(sync_ms)), 4), "p95": round(float(np.percentile(sync_ms, 95)), 4),
                     "percentiles_from": f"{len(sync_ms)} steps of the sync-per-step loop after the timed region",
                     "device_p50": round(float(np.median(dev_ms)), 4), "mean": round(mean_ms, 4),
                     "mean_is": f"timed region: chains of {chain} greedy steps, one host synchronisation per chain",
                     "chain_mean_p50": round(p50_ms, 4)} if chain and sync_ms else
                    {"p50": round(p50_ms, 4), "p95": round(float(np.percentile(step_ms, 95)), 4),
                     "percentiles_from": "the timed region (one host synchronisation per step)",
                     "device_p50": round(float(np.median(dev_ms)), 4), "mean": round(mean_ms, 4),
                     "mean_over_p50": round(mean_ms / p50_ms, 4)}),
        # how the timed steps were driven: chains of `chain_steps` greedy steps with the token handed over on the device and
        # one host synchronisation per chain (every step's tokens are still copied to the host, asynchronously), or the
        # reference's loop shape - a host synchronisation per step; `sync_per_step` is the second form measured right after
        "host_loop": ({"form": "chained", "chain_steps": chain,
                       "sync_per_step": {"tpot_ms_p50": round(float(np.median(sync_ms)), 4),
                                         "tpot_ms_p95": round(float(np.percentile(sync_ms, 95)), 4),
                                         "tok_s": round(args.batch * (1 if tp else world) * 1e3 / float(np.mean(sync_ms)), 2),
                                         "steps": len(sync_ms)}} if chain and sync_ms else {"form": "sync_per_step"}),
        "decode_heavy": heavy,
        "ttft_ms_10000": ttft10k,
        "ttft_ms_short": ttft_short,
        "mp8_collectives_us": mp8,
        # N > 1: `value` is the replicas leg (independent requests, weak scaling); `tp` is the sharded leg (one request, strong)
        "tp": tp_leg,
        "scaling_legs": ({"value": "strong (tensor parallel)" if tp else "weak (replicas: one independent request stream per GPU)",
                          "tp": "strong (one request over all ranks)"} if world > 1 else None),
        "prefill_roofline": prefill_roofline(full_cfg, args.ctx, float(np.median(ttfts)), world if tp else 1),
        # the whole step against the HBM roofline, from the DRIVER-TIMED ms_per_step (wall clock of the contract's timed
        # region: metadata upload, graph replay, token D2H, host loop - what `value` is made of); the graph's own device time
        # (hipEvent pair, device_p50) is the side figure
        "step_roofline": {"algorithmic_bytes_per_step": int(step_bytes),
                          "achieved_GBps": round(step_bytes / (mean_ms * 1e-3) / 1e9, 1),
                          "frac_of_8TBps": round(step_bytes / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "timed_by": "ms_per_step (wall, the contract's timed region)",
                          "device_frac_of_8TBps": round(step_bytes / (np.median(dev_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        # which ABI the timed steps ran on: "ext" = the host DAG calls the fused entry points of
        # include/pegainfer_kernels_ext.h (4-5 launches per layer); "reference" = only the symbols of
        # pegainfer-kernels/src/ffi.rs (14 launches per layer) - what an UNMODIFIED Rust host gets;
        # `drop_in_abi` below carries that second number whenever the headline is the first
        "abi_path": "ext" if args.decode_mode >= 1 else "reference",
    }
    if rank == 0:
        # ---- roofline of the dominant kernel: the gate_up weight-streaming GEMV (45 % of the step's bytes).
        # decode_mode 1 launches it with the add+RMSNorm prologue and SwiGLU epilogue fused in; the timing loop
        # launches exactly that kernel over the 36 layers' weights, hipEvents on the model stream. ----
        H, I = cfg["hidden_size"], cfg["intermediate_size"]
        fused = args.decode_mode >= 1 and args.batch <= 16
        # weights once + (hidden, residual, norm weight in; hidden out by one workgroup; act out)
        gate_up_bytes = 2 * I * H * 2 + args.batch * (2 * H * 2 + H * 2 + I * 2) + H * 2
        ms = eng.bench_gemv(5 if fused else 2, 360, args.batch)
        achieved = gate_up_bytes / (ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters.  Measured live (two rocprofv3 --pmc passes over a probe process that
        # launches this kernel, after the timed region) when rocprofv3 is on the box; otherwise REPLAYED from the committed
        # CSVs of the newest round and labelled so.  PEGAINFER_BENCH_TRAFFIC=0 skips the live passes.
        traffic, traffic_src, replayed = None, None, False
        if fused and world == 1 and os.environ.get("PEGAINFER_BENCH_TRAFFIC", "1") != "0":
            traffic, traffic_src = measure_traffic(args.model, args.batch)
        prof = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_fused_pmc_FETCH_SIZE.csv") for r in (5, 4, 3, 2, 1))
                     if os.path.exists(q)), None)
        if traffic is None and prof and fused and args.batch == 1 and args.model == "qwen3-4b":
            import csv
            why = traffic_src
            fetch = {r["kernel"]: float(r["avg_value"]) for r in csv.DictReader(open(prof)) if r["counter"] == "FETCH_SIZE"}
            wprof = prof.replace("FETCH_SIZE", "WRITE_SIZE")
            wr = {r["kernel"]: float(r["avg_value"]) for r in csv.DictReader(open(wprof))} if os.path.exists(wprof) else {}
            # the gate_up site: NT 1, RPW 1, KSPLIT 1, SwiGLU epilogue (+ the in-flight depth U since round 3)
            kname = next((k for k in fetch if k.startswith("gemv_fused_kernel<1, 1, 1, 1")), None)
            if kname:   # KiB per dispatch; gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream -> x2
                traffic = int(2 * fetch[kname] * 1024 + wr.get(kname, 0.0) * 1024)
                replayed = True
                traffic_src = ("replayed from " + os.path.relpath(prof, ROOT) + " (x2 gfx950 correction) + WRITE_SIZE, separate "
                               "rocprofv3 --pmc passes; not measured in this run" + (f" ({why})" if why else ""))
        out["roofline"] = {"bound": "hbm",
                           "kernel": ("gemv_fused_kernel<NT=1,RPW=1,KSPLIT=1,EPI=silu> (gate_up, M=%d K=%d N=%d)" if fused
                                      else "gemv_fused_kernel<NT,RPW=2,KSPLIT=1,EPI=store> (gate_up, M=%d K=%d N=%d)") % (2 * I, H, args.batch),
                           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4),
                           # the dominant kernel is the best-behaved 45 % of the step; the time-weighted figure is the whole
                           # step's (every launch, every boundary, the host loop), from ms_per_step
                           "step_frac": out["step_roofline"]["frac_of_8TBps"],
                           "traffic": traffic, "traffic_replayed": replayed,
                           "traffic_source": traffic_src,
                           "bytes_per_launch": gate_up_bytes, "avg_launch_us": round(ms * 1e3, 2)}
        per_site = {}
        for which, name, M, K in [(0, "qkv", (cfg["num_attention_heads"] + 2 * cfg["num_key_value_heads"]) * cfg["head_dim"], H),
                                  (1, "o", H, cfg["num_attention_heads"] * cfg["head_dim"]), (2, "gate_up", 2 * I, H),
                                  (3, "down", H, I), (4, "lm_head", cfg["vocab_size"], H)]:
            t = eng.bench_gemv(which, 72 if which != 4 else 20, args.batch)
            per_site[name] = {"us": round(t * 1e3, 2), "GBps": round(M * K * 2 / (t * 1e-3) / 1e9, 1)}
        out["gemv_sites"] = per_site
    # ---- CPU side (rank 0, N = 1): parity of the benchmarked model + the reference's CPU path as the baseline ----
    cpu_inputs, batch_rows = None, None
    if rank == 0 and args.cpu_steps > 0 and world == 1 and not mp8_hung:
        try:
            gtoks, grows = gpu_parity_run(eng, prompt, args.cpu_steps)
            cpu_inputs = (export_checkpoint(eng, cfg), gtoks, grows)
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if not mp8_hung:   # freeing device memory synchronises the device - behind a stuck collective it never returns
        eng.close()
    # ---- side numbers on fresh engines (rank 0, N = 1, outside `value`): the reference's batch sweep at --ctx
    #      (bench_serving.rs batch matrix; model-crate.md:151) and the step rate on the STRICT drop-in ABI ----
    if rank == 0 and world == 1 and args.batch == 1 and args.sweep_steps > 0 and not path:
        def side_engine(mode, max_bs):
            e = Qwen3Engine(cfg, num_kv_pages=max_bs * (-(-(args.ctx + args.sweep_steps + 16) // 16) + 1) + 8, max_batch_size=max_bs,
                            enable_graph=not args.no_graph, decode_mode=mode, split_policy=args.split_policy, device=local,
                            max_positions=max(4096, args.ctx + args.sweep_steps + 32))
            return e.fill_synthetic(seed=42 + rank, std=0.02)

        def rate(e, bs):
            ids = [e.new_request() for _ in range(bs)]
            tk = e.prefill(ids, [prompt] * bs)
            for _ in range(5):
                tk = e.decode(ids, tk)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.sweep_steps):
                tk = e.decode(ids, tk)
            dt = (time.perf_counter() - t0) / args.sweep_steps
            for i in ids:
                e.drop_request(i)
            b = algorithmic_bytes_per_token(full_cfg, args.ctx + 5 + args.sweep_steps / 2, bs)
            return {"ms_per_step": round(dt * 1e3, 4), "tok_s": round(bs / dt, 1), "frac_of_8TBps": round(b / dt / 1e9 / HBM_PEAK_GBS, 4)}
        def forced_rows(e, bs, feed):
            """every column = the bench prompt teacher-forced on the bs-1 parity run's tokens -> bf16 bits [bs, 1 + len(feed), V]"""
            ids = [e.new_request() for _ in range(bs)]
            _, lg = e.prefill(ids, [prompt] * bs, return_logits=True)
            rows = [lg.copy()]
            for tk in feed:
                _, lg = e.decode(ids, [tk] * bs, return_logits=True)
                rows.append(lg.copy())
            for i in ids:
                e.drop_request(i)
            return np.stack(rows, axis=1)
        try:
            e2 = side_engine(args.decode_mode, 16)
            out["batch_sweep"] = {"ctx": args.ctx, "steps": args.sweep_steps, **{str(bs): rate(e2, bs) for bs in (2, 4, 8, 16)}}
            if cpu_inputs is not None:   # parity of the batched steps, refereed by the bs-1 oracle / truth streams (cpu_legs)
                batch_rows = {str(bs): forced_rows(e2, bs, cpu_inputs[1][:-1]) for bs in (2, 4, 8, 16)}
            e2.close()
            if args.decode_mode >= 1:
                e3 = side_engine(0, 1)
                out["drop_in_abi"] = dict(rate(e3, 1), decode_mode=0, note="only the symbols of pegainfer-kernels/src/ffi.rs, 14 launches per "
                                          "layer, hipGraph on: what an unmodified Rust host gets; the headline needs the host DAG to call "
                                          "the five fused entry points of include/pegainfer_kernels_ext.h (INTEGRATION.md section 4)")
                e3.close()
        except Exception as e:  # noqa: BLE001 - side numbers never take the line down
            out["batch_sweep"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        out["cpu_baseline"] = None
        if cpu_inputs is not None:
            threads = os.cpu_count() or 1
            # torch threads of the HF leg: measured on the pool's 256-core hosts with tools/hf_threads_probe.py (8 layers of
            # this shape, decode tok/s): 32 threads 15.1, 64 -> 7.4, 128 -> 2.3 (a 2560-wide GEMV per op does not feed 256
            # threads; their barriers dominate) - so 32 unless PEGAINFER_CPU_THREADS says otherwise
            hf_threads = int(os.environ.get("PEGAINFER_CPU_THREADS", "0")) or min(32, threads)
            try:
                out["cpu_baseline"], out["parity"] = cpu_legs(cfg, cpu_inputs[0], prompt, cpu_inputs[1], cpu_inputs[2],
                                                              args.cpu_steps, threads, hf_threads=hf_threads, batch_rows=batch_rows)
                for label, pb in (out["parity"].pop("batch", None) or {}).items():
                    if isinstance(out.get("batch_sweep"), dict) and label in out["batch_sweep"]:
                        out["batch_sweep"][label]["parity"] = pb
            except MemoryError:
                out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": threads, "kind": "port",
                                       "sample": "skipped: host RAM too small for the fp32 oracle weights"}
        print(json.dumps(out), flush=True)
    if mp8_hung:
        os._exit(0)   # a stuck collective would also block the teardown; the line is out
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
