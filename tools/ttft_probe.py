#!/usr/bin/env python3
"""TTFT (submit -> first token) of single prompts of the given lengths on a synthetic Qwen3-4B, p50 over 8 runs after 2
warm-ups; environment knobs (PEGAINFER_PREFILL_SHORT, PEGAINFER_PREFILL_SPLIT3_MIN, PEGAINFER_PREFILL_FUSE, ...) select the
variant, so one GPU call can A/B them.    usage: python tools/ttft_probe.py 4 8 16 32 64 128"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pegainfer_amd.qwen3 import QWEN3_4B, Qwen3Engine  # noqa: E402

lens = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32, 64, 128]
tag = " ".join(f"{k[10:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PEGAINFER_PREFILL")) or "default"
eng = Qwen3Engine(dict(QWEN3_4B), num_kv_pages=max(lens) // 16 + 8, max_batch_size=2, decode_mode=1, max_positions=max(4096, max(lens) + 16))
eng.fill_synthetic(seed=1, std=0.02)
for n in lens:
    p, ts = [100 + (i % 1000) for i in range(n)], []
    for _ in range(10):
        r = eng.new_request()
        t0 = time.perf_counter()
        eng.prefill([r], [p])
        ts.append((time.perf_counter() - t0) * 1e3)
        eng.drop_request(r)
    print(f"[{tag}] TTFT({n}) p50 {np.median(ts[2:]):.3f} ms  min {min(ts):.3f}", flush=True)
eng.close()
