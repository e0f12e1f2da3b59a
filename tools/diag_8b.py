#!/usr/bin/env python3
"""Why did the last decode row of the Qwen3-8B sampling loop differ from the forced runs (round 5, GPU call 2)?
Variants on ONE engine, every variant = new request, prefill of the 1024-token prompt, 4 decode steps on fixed tokens:
  base     prefill(return_logits=True), decode(return_logits=True)
  nolog    prefill(return_logits=False)
  samp     base + eng.sample(top-k / top-p) calls after every decode step
  samp1    base + ONE greedy eng.sample (top1 branch) after every decode step
  again    base once more (did anything above leave damage behind?)
Prints, per variant, the number of differing logits per step against `base`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from pegainfer_amd.qwen3 import QWEN3_4B, QWEN3_8B, Qwen3Engine
    for name, cfg in (("qwen3-8b", dict(QWEN3_8B, num_hidden_layers=int(sys.argv[1]) if len(sys.argv) > 1 else 36)),
                      ("qwen3-4b", dict(QWEN3_4B, num_hidden_layers=8))):
        eng = Qwen3Engine(cfg, num_kv_pages=96, max_batch_size=2, decode_mode=1, max_positions=4096).fill_synthetic(seed=808, std=0.02)
        prompt = [100 + (i % 1000) for i in range(1024)]
        feed = [11, 2222, 33333, 44444]

        def run(pf_logits=True, sample=None):
            rid = eng.new_request()
            eng.prefill([rid], [prompt], return_logits=True) if pf_logits else eng.prefill([rid], [prompt])
            rows = []
            for tk in feed:
                _, lg = eng.decode([rid], [tk], return_logits=True)
                rows.append(lg[0].copy())
                if sample == "topk":
                    for (T, k, p) in ((0.8, 50, 0.95), (0.8, -1, 0.9)):
                        eng.sample(0, T, k, p, 0.37)
                elif sample == "top1":
                    eng.sample(0, 0.0, 1, 1.0, 0.37)
            eng.drop_request(rid)
            return np.stack(rows)

        base = run()
        for tag, kw in (("nolog", dict(pf_logits=False)), ("samp", dict(sample="topk")), ("samp1", dict(sample="top1")),
                        ("again", {}), ("nolog+samp", dict(pf_logits=False, sample="topk")), ("again2", {})):
            r = run(**kw)
            print(name, tag, "differing logits per step vs base:", [int((r[i] != base[i]).sum()) for i in range(len(feed))], flush=True)
        eng.close()


if __name__ == "__main__":
    main()
