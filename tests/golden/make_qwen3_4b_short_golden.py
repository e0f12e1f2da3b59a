#!/usr/bin/env python3
"""HF Transformers fixtures for SHORT prompts at full depth: the 36-layer Qwen3-4B-shaped seeded checkpoint of
make_qwen3_4b_depth_golden.py (same seed, regenerated bit for bit), prompts of 48 and 100 seeded random tokens
(tests/depth_common.py:short_prompts), 30 greedy tokens each - the shape of the reference's own golden file
(test_data/Qwen3-4B.json: 10 prompts below 80 tokens, 30-50 new tokens; pegainfer-qwen3-4b/tests/e2e.rs:108-221), which
needs the real checkpoint.  Every prefill GEMM route round 4 added runs at these lengths; the 1024-token fixture never
touches them.

  tests/golden/qwen3_4b_depth36_short_hf.json   HF greedy tokens + top-1 margins per prompt, versions
  tests/golden/qwen3_4b_depth36_short_hf.npz    per prompt and step: top-64 ids / logits, logits at 4096 seeded indices

Run in the build container (transformers, ~30 GB of RAM; NOT on the GPU box):
    python tests/golden/make_qwen3_4b_short_golden.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

N_NEW, N_TOP, N_IDX = 30, 64, 4096


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=HERE, help="directory of the base fixture and of the output (dry runs use a scratch one)")
    ap.add_argument("--base", default="qwen3_4b_depth36_hf.json")
    args = ap.parse_args()
    import torch
    import transformers
    import depth_common as dc
    from oracle import hf_engine
    from oracle.qwen3_ref import Qwen3Config, synthetic_weights
    base = json.load(open(os.path.join(args.dir, args.base)))
    cfgd, seed, std = base["config"], base["seed"], base["std"]
    t0 = time.time()
    _, bits = synthetic_weights(Qwen3Config(**cfgd), seed=seed, std=std, with_bits=True)
    print("checkpoint generated in %.0f s" % (time.time() - t0), flush=True)
    model = hf_engine.build_qwen3(cfgd, bits, threads=os.cpu_count())
    prompts = dc.short_prompts(cfgd["vocab_size"])
    idx = np.sort(np.random.default_rng(seed).choice(cfgd["vocab_size"], size=N_IDX, replace=False)).astype(np.int32)
    meta = dict(engine="transformers", transformers_version=transformers.__version__, torch_version=torch.__version__,
                device="cpu", dtype="bfloat16", generator="tests/golden/make_qwen3_4b_short_golden.py", seed=seed, std=std,
                config=cfgd, prompt_seed=dc.SHORT_SEED, new_tokens=N_NEW, cases={})
    arrays = dict(idx=idx)
    for n in dc.SHORT_HF:
        t0 = time.time()
        toks, _, lg = hf_engine.generate_greedy(model, prompts[n], N_NEW, return_logits=True)
        srt = np.sort(lg, axis=-1)
        top_ids = np.argsort(-lg, axis=-1, kind="stable")[:, :N_TOP].astype(np.int32)
        meta["cases"][str(n)] = dict(prompt_tokens=n, hf_tokens=toks, top1_margin=(srt[:, -1] - srt[:, -2]).tolist(),
                                     logit_absmax=float(np.abs(lg).max()))
        arrays["top_ids_%d" % n] = top_ids
        arrays["top_vals_%d" % n] = np.take_along_axis(lg, top_ids, axis=-1).astype(np.float32)
        arrays["idx_vals_%d" % n] = lg[:, idx].astype(np.float32)
        print("prompt %d: %.1f s, tokens %s" % (n, time.time() - t0, toks), flush=True)
    with open(os.path.join(args.dir, "qwen3_4b_depth36_short_hf.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(args.dir, "qwen3_4b_depth36_short_hf.npz"), **arrays)
    print("wrote qwen3_4b_depth36_short_hf")


if __name__ == "__main__":
    sys.exit(main())
