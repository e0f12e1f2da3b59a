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
Extra objects: roofline (dominant kernel, HBM bound), cpu_baseline (oracle port on host cores).
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


def cpu_baseline(cfg, steps, threads, ctx=1024):
    """Oracle ("port") decode on the host cores: the numpy restatement of the same DAG on a synthetic
    checkpoint of the SAME shape, bounded to `steps` decode steps at the bench's own context length.
    Reported baseline only."""
    from oracle import ops as oracle_ops
    from oracle.bf16 import bf16_round
    from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle
    oracle_ops.GEMM_ACCUM = np.float32   # timing leg: fp32 sgemm like a CPU engine would use
    keys = ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
            "intermediate_size", "vocab_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings"]
    c = Qwen3Config(**{k: cfg[k] for k in keys})
    rng = np.random.default_rng(42)
    t_gen = time.perf_counter()

    class Lazy(dict):
        """weights materialised layer by layer from a small seeded pool (shape-faithful, values irrelevant
        for timing; avoids 16 GB of RNG work before the timed region)."""
    pool = bf16_round((rng.standard_normal(1 << 22) * 0.02).astype(np.float32))

    def t(*shape, mean=0.0):
        n = int(np.prod(shape))
        reps = -(-n // pool.size)
        a = np.tile(pool, reps)[:n].reshape(shape)
        return a + np.float32(mean) if mean else a
    w = {"model.embed_tokens.weight": t(c.vocab_size, c.hidden_size), "model.norm.weight": t(c.hidden_size, mean=1.0)}
    if not c.tie_word_embeddings:
        w["lm_head.weight"] = t(c.vocab_size, c.hidden_size)
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        w[p + "self_attn.q_proj.weight"] = t(c.q_dim, c.hidden_size)
        w[p + "self_attn.k_proj.weight"] = t(c.kv_dim, c.hidden_size)
        w[p + "self_attn.v_proj.weight"] = t(c.kv_dim, c.hidden_size)
        w[p + "self_attn.o_proj.weight"] = t(c.hidden_size, c.q_dim)
        w[p + "self_attn.q_norm.weight"] = t(c.head_dim, mean=1.0)
        w[p + "self_attn.k_norm.weight"] = t(c.head_dim, mean=1.0)
        w[p + "mlp.gate_proj.weight"] = t(c.intermediate_size, c.hidden_size)
        w[p + "mlp.up_proj.weight"] = t(c.intermediate_size, c.hidden_size)
        w[p + "mlp.down_proj.weight"] = t(c.hidden_size, c.intermediate_size)
        w[p + "input_layernorm.weight"] = t(c.hidden_size, mean=1.0)
        w[p + "post_attention_layernorm.weight"] = t(c.hidden_size, mean=1.0)
    # the SAME workload as the timed GPU leg: one request at --ctx cached tokens.  The prompt's KV is not computed
    # (a 36-layer numpy prefill of 1024 tokens would be minutes of untimed CPU work): the request's pages are
    # filled with seeded bf16 values directly - decode cost depends on the KV bytes scanned, not on their values
    npages = -(-(ctx + steps + 1) // 16) + 2
    m = Qwen3Oracle(c, w, num_pages=npages, rope_positions=ctx + steps + 16)
    st = KvState()
    m._ensure(st, ctx)
    st.seq_len = ctx
    kvpool = bf16_round((np.random.default_rng(7).standard_normal(1 << 20) * 0.5).astype(np.float32))
    m.kv[m.layout.page_stride:] = np.resize(kvpool, m.kv.size - m.layout.page_stride)
    setup_s = time.perf_counter() - t_gen
    t0 = time.perf_counter()
    tok = 100
    for _ in range(steps):
        tok = int(m.batch_decode([tok], [st])[0].argmax())
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{steps} greedy decode steps, bs=1, ctx {ctx}->{ctx + steps} (the GPU leg's workload; KV pages "
                      f"seeded directly, no CPU prefill), oracle/qwen3_ref.py (numpy fp32 matmul over bf16-valued "
                      f"weights) on a synthetic checkpoint of the same shape; {dt:.1f} s timed, {setup_s:.0f} s "
                      f"untimed setup"}




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
    total_ctx = args.ctx + args.warmup + args.steps + 8
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
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        toks = eng.decode(rids, toks)
        step_ms.append((time.perf_counter() - t0) * 1e3)
        dev_ms.append(eng.last_step_ms())
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t_start, device="cuda")
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
        "tpot_ms": {"p50": round(float(np.median(step_ms)), 4), "p95": round(float(np.percentile(step_ms, 95)), 4),
                    "device_p50": round(float(np.median(dev_ms)), 4)},
        "roofline": {"bound": "hbm", "kernel": "whole decode step (graph)", "achieved": round(
            step_bytes / (np.median(dev_ms) * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(step_bytes / (np.median(dev_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_launch": int(step_bytes)},
        "cpu_baseline": None,
    }
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0



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
    ap.add_argument("--decode-mode", type=int, default=int(os.environ.get("PEGAINFER_DECODE_MODE", "1")),
                    help="0 = reference op sequence 1:1, 1 = fused MI355X decode kernels (bit-identical)")
    ap.add_argument("--split-policy", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=12, help="decode steps for the CPU baseline (0 = skip)")
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
    args = ap.parse_args()

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
    total_ctx = args.ctx + args.warmup + args.steps + 8
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
    for _ in range(args.warmup):
        toks = eng.decode(rids, toks)

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
    for _ in range(args.steps):
        t0 = time.perf_counter()
        toks = eng.decode(rids, toks)
        if samp:   # the reference samples request by request after the step (executor.rs:324-328)
            toks = np.array([eng.sample(i, samp[0], samp[1], samp[2], float(srng.random())) for i in range(len(rids))],
                            dtype=np.int32)
        step_ms.append((time.perf_counter() - t0) * 1e3)
        dev_ms.append(eng.last_step_ms())
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = parallel.max_over_ranks(elapsed, device="cuda")

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
    ctx_mid = args.ctx + args.warmup + args.steps / 2
    step_bytes = algorithmic_bytes_per_token(full_cfg, ctx_mid, args.batch)

    out = {
        "metric": "decode tokens/sec + TTFT, Qwen3-4B bf16 greedy, 1xMI355X" if args.model == "qwen3-4b"
                  else "decode tokens/sec + TTFT, Qwen3-8B bf16 greedy, 1xMI355X",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(mean_ms, 4), "higher_is_better": True,
        "scaling": "strong" if tp else "weak",
        "vs_baseline": None, "dtype": "bf16", "data": data,
        "config": {"workload": f"{args.model} greedy decode, hipGraph {'off' if args.no_graph else 'on'}, "
                               f"bs={args.batch}/GPU, ctx {args.ctx}->{args.ctx + args.warmup + args.steps} "
                               f"(reference decode_heavy: synthetic prompt 100+(i%1000))",
                   "batch_per_gpu": args.batch, "ctx": args.ctx, "decode_mode": args.decode_mode,
                   "sampling": args.sampling, "split_policy": args.split_policy, "parallelism": ("tp%d" % world if tp else "replicas%d" % world) if world > 1 else "single"},
        "ttft_ms": {"prompt_tokens": args.ctx, "p50": round(float(np.median(ttfts)), 3),
                    "min": round(float(min(ttfts)), 3), "iters": len(ttfts)},
        "tpot_ms": {"p50": round(p50_ms, 4), "p95": round(float(np.percentile(step_ms, 95)), 4),
                    "device_p50": round(float(np.median(dev_ms)), 4), "mean": round(mean_ms, 4),
                    "mean_over_p50": round(mean_ms / p50_ms, 4)},
        "decode_heavy": heavy,
        "ttft_ms_10000": ttft10k,
        "mp8_collectives_us": mp8,
        "prefill_roofline": prefill_roofline(full_cfg, args.ctx, float(np.median(ttfts)), world if tp else 1),
        "step_roofline": {"algorithmic_bytes_per_step": int(step_bytes),
                          "achieved_GBps": round(step_bytes / (np.median(dev_ms) * 1e-3) / 1e9, 1),
                          "frac_of_8TBps": round(step_bytes / (np.median(dev_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
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
        # HBM bytes per launch from the PMC counters: collected in their own rocprofv3 --pmc passes (tools/
        # gpu_refresh_profiles.sh) and REPLAYED here from the committed CSVs of the newest round - not measured in this run
        traffic, traffic_src = None, None
        prof = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_fused_pmc_FETCH_SIZE.csv") for r in (5, 4, 3, 2, 1))
                     if os.path.exists(q)), None)
        if prof and fused and args.batch == 1 and args.model == "qwen3-4b":
            import csv
            fetch = {r["kernel"]: float(r["avg_value"]) for r in csv.DictReader(open(prof)) if r["counter"] == "FETCH_SIZE"}
            wprof = prof.replace("FETCH_SIZE", "WRITE_SIZE")
            wr = {r["kernel"]: float(r["avg_value"]) for r in csv.DictReader(open(wprof))} if os.path.exists(wprof) else {}
            # the gate_up site: NT 1, RPW 1, KSPLIT 1, SwiGLU epilogue (+ the in-flight depth U since round 3)
            kname = next((k for k in fetch if k.startswith("gemv_fused_kernel<1, 1, 1, 1")), None)
            if kname:   # KiB per dispatch; gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream -> x2
                traffic = int(2 * fetch[kname] * 1024 + wr.get(kname, 0.0) * 1024)
                traffic_src = ("replayed from " + os.path.relpath(prof, ROOT) + " (x2 gfx950 correction) + WRITE_SIZE, separate "
                               "rocprofv3 --pmc passes; not measured in this run")
        out["roofline"] = {"bound": "hbm",
                           "kernel": ("gemv_fused_kernel<NT=1,RPW=1,KSPLIT=1,EPI=silu> (gate_up, M=%d K=%d N=%d)" if fused
                                      else "gemv_fused_kernel<NT,RPW=2,KSPLIT=1,EPI=store> (gate_up, M=%d K=%d N=%d)") % (2 * I, H, args.batch),
                           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_replayed": traffic is not None,
                           "traffic_source": traffic_src,
                           "bytes_per_launch": gate_up_bytes, "avg_launch_us": round(ms * 1e3, 2)}
        per_site = {}
        for which, name, M, K in [(0, "qkv", (cfg["num_attention_heads"] + 2 * cfg["num_key_value_heads"]) * cfg["head_dim"], H),
                                  (1, "o", H, cfg["num_attention_heads"] * cfg["head_dim"]), (2, "gate_up", 2 * I, H),
                                  (3, "down", H, I), (4, "lm_head", cfg["vocab_size"], H)]:
            t = eng.bench_gemv(which, 72 if which != 4 else 20, args.batch)
            per_site[name] = {"us": round(t * 1e3, 2), "GBps": round(M * K * 2 / (t * 1e-3) / 1e9, 1)}
        out["gemv_sites"] = per_site
    if not mp8_hung:   # freeing device memory synchronises the device - behind a stuck collective it never returns
        eng.close()
    if rank == 0:
        if args.cpu_steps > 0 and world == 1:
            threads = os.cpu_count() or 1
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_steps, threads, args.ctx)
            except MemoryError:
                out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": threads, "kind": "port",
                                       "sample": "skipped: host RAM too small for the fp32 oracle weights"}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if mp8_hung:
        os._exit(0)   # a stuck collective would also block the teardown; the line is out
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
