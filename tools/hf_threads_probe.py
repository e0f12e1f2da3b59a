#!/usr/bin/env python3
"""How many torch threads should bench.py's HF CPU baseline use on this box?  An 8-layer Qwen3-4B-shaped model (seeded),
128-token prompt -> 12 greedy tokens, steady decode tok/s per thread count.  CPU only.
usage: python tools/hf_threads_probe.py [threads ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hf_engine  # noqa: E402
from oracle.qwen3_ref import Qwen3Config, synthetic_weights  # noqa: E402

cfgd = dict(hidden_size=2560, num_hidden_layers=8, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
            intermediate_size=9728, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True,
            max_position_embeddings=4096)
t0 = time.time()
_, bits = synthetic_weights(Qwen3Config(**cfgd), seed=1, with_bits=True)
print("cpu_count", os.cpu_count(), "checkpoint %.0f s" % (time.time() - t0), flush=True)
prompt = [100 + i for i in range(128)]
for th in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, os.cpu_count()]:
    m = hf_engine.build_qwen3(cfgd, bits, threads=th)
    _, st, _ = hf_engine.generate_greedy(m, prompt, 12)
    print("threads %4d: prefill %.2f s, decode %.2f tok/s (8 layers; x 36/8 slower at full depth)" % (
        th, st[1] - st[0], (len(st) - 2) / (st[-1] - st[1])), flush=True)
    del m
