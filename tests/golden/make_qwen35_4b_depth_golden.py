#!/usr/bin/env python3
"""Full-depth golden fixture for BASELINE.json configs[3]: HF Transformers ``Qwen3_5ForCausalLM`` (the reference's truth
engine, scripts/generate_test_data.py) on a seeded **32-layer Qwen3.5-4B-shaped** checkpoint (24 linear-attention + 8
full-attention layers, V = 248 320) - the configuration ``bench.py --model qwen3.5-4b`` times - plus the oracle and the
"fp32 truth" pass (oracle.bf16.exact_activations) on the same weights.

Real Qwen3.5-4B weights are not on disk (no network), so the reference's own gate (pegainfer-qwen35-4b/tests/e2e.rs,
greedy text vs test_data/Qwen3.5-4B.json) cannot run; this is its reachable proxy, same recipe as
make_qwen3_4b_depth_golden.py: checkpoint = ``oracle.qwen35_ref.synthetic_weights(qwen35_4b, SEED, STD)`` (regenerated
bit for bit from the seed on any box), prompt = the reference's ``decode_heavy`` rule ``100 + i % 1000``
(bench_serving.rs:37-43), ``generate(do_sample=False)``.

Why three passes.  bench.py's HF leg read cosine 0.9938 / 11 % of the logit scale against the engine at 32 layers
(profiles/r4_qwen35_4b_bench.json) and nobody could say whose error that was.  The fp32 truth pass answers it: the
fixture records err(HF vs truth), err(bf16 oracle vs truth) and HF vs oracle, so the GPU test can place the engine on
the same axis (tests/test_gpu_full_depth.py).

Written (the checkpoint itself is 9 GB and is regenerated from the seed by the tests):
  tests/golden/qwen35_4b_depth32_hf.json   seed, std, config, HF greedy tokens, top-1 margins, versions, the three
                                           distances per step
  tests/golden/qwen35_4b_depth32_hf.npz    per step (last prompt position + every decode step, HF on its own greedy
                                           tokens): top-64 ids / logits and the logits at 4096 seeded vocabulary indices,
                                           for HF and for the fp32 truth pass

Run in the build container (needs ~45 GB of RAM and transformers; NOT on the GPU box):
    python tests/golden/make_qwen35_4b_depth_golden.py [--layers 32] [--prompt 1024] [--steps 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SEED, STD = 20260935, 0.02
N_TOP, N_IDX = 64, 4096


def state_for_engines(w):
    """the tensors as the engines take them: bf16 bits, except A_log / gated-norm weight which stay f32 (weights.rs:226-241)"""
    from oracle.bf16 import bf16_bits_exact
    return {k: (v if (k.endswith("A_log") or k.endswith("linear_attn.norm.weight")) else bf16_bits_exact(v))
            for k, v in w.items()}


def cfg_dict(layers):
    return dict(hidden_size=2560, intermediate_size=9216, num_hidden_layers=layers, vocab_size=248320,
                num_attention_heads=16, num_key_value_heads=4, head_dim=256, linear_num_key_heads=16,
                linear_num_value_heads=32, linear_key_head_dim=128, linear_value_head_dim=128, linear_conv_kernel_dim=4,
                rms_norm_eps=1e-6, rope_theta=1e7, partial_rotary_factor=0.25)


def oracle_rows(cfg, w, prompt, feed, exact):
    """logits rows (last prompt position + one per fed token) of the bf16 oracle, or of the fp32 truth pass"""
    import contextlib
    from oracle import ops as O
    from oracle.bf16 import exact_activations
    from oracle.qwen35_ref import Qwen35Oracle
    old = O.GEMM_ACCUM
    O.GEMM_ACCUM = np.float32
    try:
        with (exact_activations() if exact else contextlib.nullcontext()):
            orc = Qwen35Oracle(cfg, w, num_pages=len(prompt) // 16 + 8, rope_positions=2048)
            st = orc.new_request()
            rows = [orc.prefill(prompt, st)]
            for tk in feed:
                rows.append(orc.batch_decode([tk], [st])[0])
    finally:
        O.GEMM_ACCUM = old
    return np.stack(rows)


def dist(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    cos = (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)
    d = np.abs(a - b)
    return dict(cos=[float(x) for x in cos], cos_min=float(cos.min()), max_dlogit=[float(x) for x in d.max(-1)],
                rms_dlogit=[float(x) for x in np.sqrt((d * d).mean(-1))])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--out", default=HERE, help="output directory (dry runs of the tests write reduced-depth fixtures elsewhere)")
    args = ap.parse_args()
    import torch
    import transformers
    from oracle import hf_engine
    from oracle.qwen35_ref import Qwen35Config, synthetic_weights
    cfgd = cfg_dict(args.layers)
    cfg = Qwen35Config(**cfgd)
    cfgd["layer_types"] = cfg.layer_types
    t0 = time.time()
    w = synthetic_weights(cfg, seed=SEED, std=STD)
    print("checkpoint generated in %.0f s" % (time.time() - t0), flush=True)
    prompt = [100 + (i % 1000) for i in range(args.prompt)]
    model = hf_engine.build_qwen35(dict(cfgd, max_position_embeddings=4096), state_for_engines(w), threads=os.cpu_count())
    t0 = time.time()
    toks, _, lg = hf_engine.generate_greedy(model, prompt, args.steps + 1, return_logits=True)
    print("HF generate: %.1f s, tokens %s" % (time.time() - t0, toks), flush=True)
    del model
    srt = np.sort(lg, axis=-1)
    margins = (srt[:, -1] - srt[:, -2]).tolist()
    idx = np.sort(np.random.default_rng(SEED).choice(cfgd["vocab_size"], size=N_IDX, replace=False)).astype(np.int32)
    top_ids = np.argsort(-lg, axis=-1, kind="stable")[:, :N_TOP].astype(np.int32)
    top_vals = np.take_along_axis(lg, top_ids, axis=-1).astype(np.float32)
    meta = dict(engine="transformers", transformers_version=transformers.__version__, torch_version=torch.__version__,
                device="cpu", dtype="bfloat16", generator="tests/golden/make_qwen35_4b_depth_golden.py", seed=SEED,
                std=STD, config=cfgd, prompt_rule="100 + i % 1000", prompt_tokens=args.prompt, hf_tokens=toks,
                top1_margin=margins, logit_absmax=float(np.abs(lg).max()))
    feed = toks[:-1]
    t0 = time.time()
    R = oracle_rows(cfg, w, prompt, feed, exact=False)
    print("bf16 oracle: %.0f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    Tr = oracle_rows(cfg, w, prompt, feed, exact=True)
    print("fp32 truth: %.0f s" % (time.time() - t0), flush=True)
    meta["oracle_vs_hf"] = dict(dist(R, lg), argmax_equal=[bool(x) for x in (R.argmax(-1) == lg.argmax(-1))])
    meta["hf_vs_truth"] = dist(lg, Tr)
    meta["oracle_vs_truth"] = dict(dist(R, Tr), argmax_equal=[bool(x) for x in (R.argmax(-1) == Tr.argmax(-1))])
    meta["truth_tokens"] = [int(x) for x in Tr.argmax(-1)]
    tsrt = np.sort(Tr, axis=-1)
    meta["truth_top1_margin"] = (tsrt[:, -1] - tsrt[:, -2]).tolist()
    for k in ("oracle_vs_hf", "hf_vs_truth", "oracle_vs_truth"):
        print("%-16s cos_min %.6f  max|dlogit| %.4f  rms %.4f (scale %.2f)" % (
            k, meta[k]["cos_min"], max(meta[k]["max_dlogit"]), max(meta[k]["rms_dlogit"]), meta["logit_absmax"]), flush=True)
    tag = "qwen35_4b_depth%d_hf" % args.layers
    with open(os.path.join(args.out, tag + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    t_top = np.argsort(-Tr, axis=-1, kind="stable")[:, :N_TOP].astype(np.int32)
    np.savez_compressed(os.path.join(args.out, tag + ".npz"), top_ids=top_ids, top_vals=top_vals, idx=idx,
                        idx_vals=lg[:, idx].astype(np.float32), truth_idx_vals=Tr[:, idx].astype(np.float32),
                        truth_top_ids=t_top, truth_top_vals=np.take_along_axis(Tr, t_top, axis=-1).astype(np.float32),
                        truth_at_hf_top=np.take_along_axis(Tr, top_ids, axis=-1).astype(np.float32))
    print("wrote", tag)


if __name__ == "__main__":
    sys.exit(main())
