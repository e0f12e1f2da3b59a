#!/usr/bin/env python3
"""Full-depth golden fixture: HF Transformers (the reference's truth engine) on a seeded 36-layer Qwen3-4B-shaped
checkpoint - the exact configuration bench.py times (BASELINE.json configs[1]) - plus the oracle on the same weights.

Real Qwen3-4B weights are not on disk and there is no network, so the north-star gate itself (greedy token ids vs
``test_data/Qwen3-4B.json``, pegainfer-qwen3-4b/tests/e2e.rs:108-221) cannot run.  This is its reachable proxy: the same
engine and the same ``generate`` call as ``scripts/generate_test_data.py --device cpu`` (bf16, greedy,
``add_special_tokens=False`` i.e. raw token ids) on ``oracle.qwen3_ref.synthetic_weights(qwen3_4b, seed, std)`` - a
checkpoint any box regenerates bit for bit from the seed - with the reference's ``decode_heavy`` prompt
(``100 + i % 1000``, 1024 tokens, bench_serving.rs:37-43).

Written (small; the checkpoint itself is 8 GB and is regenerated from the seed by the tests):
  tests/golden/qwen3_4b_depth36_hf.json   seed, std, config, prompt rule, HF greedy tokens, top-1 margins, versions,
                                          oracle-vs-HF distances measured at generation time
  tests/golden/qwen3_4b_depth36_hf.npz    per step (last prompt position + every decode step, HF teacher-forced on
                                          its own greedy tokens): top-64 token ids / logits, and the logits at 4096
                                          seeded vocabulary indices (float32) - enough for argmax, margin, top-k
                                          overlap and a cosine over a fixed random subset without a 20 MB blob

Run in the build container (needs ~40 GB of RAM and transformers; NOT on the GPU box):
    python tests/golden/make_qwen3_4b_depth_golden.py [--layers 36] [--prompt 1024] [--steps 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SEED, STD = 20260926, 0.02
N_TOP, N_IDX = 64, 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=36)
    ap.add_argument("--prompt", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--out", default=HERE, help="output directory (dry runs of the tests write reduced-depth fixtures elsewhere)")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    import torch
    import transformers
    from oracle import hf_engine, ops as O
    from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle, synthetic_weights
    cfgd = dict(hidden_size=2560, num_hidden_layers=args.layers, num_attention_heads=32, num_key_value_heads=8,
                head_dim=128, intermediate_size=9728, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6,
                tie_word_embeddings=True, max_position_embeddings=4096)
    cfg = Qwen3Config(**cfgd)
    t0 = time.time()
    w, bits = synthetic_weights(cfg, seed=SEED, std=STD, with_bits=True)
    print("checkpoint generated in %.0f s" % (time.time() - t0), flush=True)
    prompt = [100 + (i % 1000) for i in range(args.prompt)]
    model = hf_engine.build_qwen3(cfgd, bits, threads=os.cpu_count())
    t0 = time.time()
    toks, stamps, lg = hf_engine.generate_greedy(model, prompt, args.steps + 1, return_logits=True)
    print("HF generate: %.1f s, tokens %s" % (time.time() - t0, toks), flush=True)
    # generate()'s own per-step logits ARE the teacher-forced ones (it feeds its own greedy tokens)
    srt = np.sort(lg, axis=-1)
    margins = (srt[:, -1] - srt[:, -2]).tolist()
    idx = np.sort(np.random.default_rng(SEED).choice(cfgd["vocab_size"], size=N_IDX, replace=False)).astype(np.int32)
    top_ids = np.argsort(-lg, axis=-1, kind="stable")[:, :N_TOP].astype(np.int32)
    top_vals = np.take_along_axis(lg, top_ids, axis=-1).astype(np.float32)
    meta = dict(engine="transformers", transformers_version=transformers.__version__, torch_version=torch.__version__,
                device="cpu", dtype="bfloat16", generator="tests/golden/make_qwen3_4b_depth_golden.py",
                seed=SEED, std=STD, config=cfgd, prompt_rule="100 + i % 1000", prompt_tokens=args.prompt,
                hf_tokens=toks, top1_margin=margins, logit_absmax=float(np.abs(lg).max()))
    del model
    if not args.no_oracle:
        O.GEMM_ACCUM = np.float32
        orc = Qwen3Oracle(cfg, w, num_pages=args.prompt // 16 + 8, rope_positions=4096)
        st = KvState()
        t0 = time.time()
        rows = [orc.batch_prefill([prompt], [st])[0]]
        for tk in toks[:-1]:
            rows.append(orc.batch_decode([tk], [st])[0])
        R = np.stack(rows)
        print("oracle prefill + %d steps: %.1f s" % (len(toks) - 1, time.time() - t0), flush=True)
        cos = (R * lg).sum(-1) / np.linalg.norm(R, axis=-1) / np.linalg.norm(lg, axis=-1)
        meta["oracle_vs_hf"] = dict(cos_min=float(cos.min()), max_dlogit=float(np.abs(R - lg).max()),
                                    argmax_equal=[bool(x) for x in (R.argmax(-1) == lg.argmax(-1))],
                                    oracle_tokens=[int(x) for x in R.argmax(-1)])
        print("oracle vs HF: cos_min %.6f  max|dlogit| %.4f (scale %.2f)  argmax equal %d / %d" % (
            cos.min(), np.abs(R - lg).max(), np.abs(lg).max(), int((R.argmax(-1) == lg.argmax(-1)).sum()), len(cos)))
    tag = "qwen3_4b_depth%d_hf" % args.layers
    with open(os.path.join(args.out, tag + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(args.out, tag + ".npz"), top_ids=top_ids, top_vals=top_vals, idx=idx,
                        idx_vals=lg[:, idx].astype(np.float32))
    print("wrote", tag)


if __name__ == "__main__":
    sys.exit(main())
