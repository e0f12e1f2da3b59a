#!/usr/bin/env python3
"""Round 5 diagnosis: on a 36-layer Qwen3-8B-shaped engine, ONE eng.sample() call changes every logit of every later decode
step, for good (GPU call 3: tools/diag_8b.py, first version).  Which state does it damage - weights, graph kernel arguments,
something else?  Per configuration (model shape x depth x graph on / off): decode rows before a sample() call, after it,
a weight checksum (every tensor exported and hashed) before and after, and the same after dropping all graphs (a new engine
is too expensive; eager mode is the graph-free control)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def wsum(eng):
    h = hashlib.sha1()
    st = eng.export_state()
    for k in sorted(st):
        h.update(st[k].tobytes())
    return h.hexdigest()[:12]


def main():
    from pegainfer_amd.qwen3 import QWEN3_4B, QWEN3_8B, Qwen3Engine
    prompt = [100 + (i % 1000) for i in range(1024)]
    feed = [11, 2222, 33333, 44444]
    for name, cfg, graph in (("qwen3-8b x36 graph", dict(QWEN3_8B), True), ("qwen3-8b x36 eager", dict(QWEN3_8B), False),
                             ("qwen3-8b x8 graph", dict(QWEN3_8B, num_hidden_layers=8), True),
                             ("qwen3-4b x36 graph", dict(QWEN3_4B), True)):
        eng = Qwen3Engine(cfg, num_kv_pages=96, max_batch_size=2, decode_mode=1, max_positions=4096,
                          enable_graph=graph).fill_synthetic(seed=808, std=0.02)

        def run(sample=None, when=1):
            rid = eng.new_request()
            _, lg0 = eng.prefill([rid], [prompt], return_logits=True)
            rows = [lg0[0].copy()]
            for i, tk in enumerate(feed):
                _, lg = eng.decode([rid], [tk], return_logits=True)
                rows.append(lg[0].copy())
                if sample and i == when:
                    eng.sample(0, *sample)
            eng.drop_request(rid)
            return np.stack(rows)

        base = run()
        w0 = wsum(eng)
        again = run()
        print(name, "| rerun differing per row:", [int((again[i] != base[i]).sum()) for i in range(len(base))], flush=True)
        s1 = run(sample=(0.8, 50, 0.95, 0.37))
        w1 = wsum(eng)
        print(name, "| with ONE top-k sample after decode step 1, differing per row:",
              [int((s1[i] != base[i]).sum()) for i in range(len(base))], "| weights", "unchanged" if w0 == w1 else "CHANGED", flush=True)
        after = run()
        print(name, "| plain run afterwards:", [int((after[i] != base[i]).sum()) for i in range(len(base))], flush=True)
        s2 = run(sample=(0.0, 1, 1.0, 0.37))
        print(name, "| with ONE greedy sample (top1 kernel):", [int((s2[i] != base[i]).sum()) for i in range(len(base))], flush=True)
        eng.close()


if __name__ == "__main__":
    main()
