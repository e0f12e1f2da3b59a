#!/usr/bin/env python3
"""Timeline of one decode GEMV launch on a real MI355X from the in-kernel wall-clock stamps of
pegainfer_debug_gemv_trace (100 MHz): per workgroup entry -> x staged -> first weight block consumed -> K loops done ->
exit.  Sites: 1 = o_proj, 3 = down_proj, 5 = fused gate_up (add + RMSNorm prologue, SwiGLU), 6 = fused qkv, 7 = lm_head,
8 / 9 = 5 / 6 with one (hot) norm weight for every launch.
usage: python tools/gemv_probe.py [--sites 6 1 5 3] [--layers 4]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = {0: "qkv (plain)", 1: "o_proj", 2: "gate_up (plain)", 3: "down_proj", 4: "lm_head (plain)", 5: "gate_up fused",
         6: "qkv fused", 7: "lm_head fused", 8: "gate_up fused, ONE hot norm weight", 9: "qkv fused, ONE hot norm weight"}
BYTES = lambda c: {1: 2 * c["hidden_size"] * c["num_attention_heads"] * c["head_dim"],
                   3: 2 * c["hidden_size"] * c["intermediate_size"],
                   5: 4 * c["hidden_size"] * c["intermediate_size"],
                   6: 2 * c["hidden_size"] * (c["num_attention_heads"] + 2 * c["num_key_value_heads"]) * c["head_dim"],
                   7: 2 * c["hidden_size"] * c["vocab_size"],
                   8: 4 * c["hidden_size"] * c["intermediate_size"],
                   9: 2 * c["hidden_size"] * (c["num_attention_heads"] + 2 * c["num_key_value_heads"]) * c["head_dim"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, nargs="+", default=[6, 1, 5, 3])
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch
    from pegainfer_amd import ffi
    from pegainfer_amd.qwen3 import QWEN3_4B, Qwen3Engine
    cfg = dict(QWEN3_4B, num_hidden_layers=args.layers)
    eng = Qwen3Engine(cfg, num_kv_pages=64, max_batch_size=2, decode_mode=1, split_policy=1, enable_graph=False)
    eng.fill_synthetic(seed=42, std=0.02)
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    nbytes = BYTES(cfg)
    for site in args.sites:
        ms_plain = eng.bench_gemv(site, args.iters, 1)
        ffi.lib().pegainfer_debug_gemv_trace(buf.data_ptr())
        buf.zero_()
        eng.bench_gemv(site, 1, 1)          # 3 warm-up launches + 1: the last launch's stamps remain
        torch.cuda.synchronize()
        ffi.lib().pegainfer_debug_gemv_trace(None)
        t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
        t = t[t[:, 0] > 0]
        us = lambda a: a * 10.0 / 1e3
        t0 = t[:, 0].min()
        print(f"site {site} {NAMES.get(site, '?')}: {ms_plain * 1e3:.2f} us per launch (events), {len(t)} workgroups, "
              f"{nbytes.get(site, 0) / (ms_plain * 1e-3) / 1e12:.2f} TB/s")
        print(f"   entry spread {us(t[:, 0].max() - t0):.2f} us | x staged at {us(t[:, 1].mean() - t0):.2f} (max {us(t[:, 1].max() - t0):.2f})"
              f" | first block consumed at {us(t[:, 2].mean() - t0):.2f} (max {us(t[:, 2].max() - t0):.2f})"
              f" | K loops done at {us(t[:, 3].mean() - t0):.2f} (max {us(t[:, 3].max() - t0):.2f})"
              f" | exit mean {us(t[:, 4].mean() - t0):.2f} min {us(t[:, 4].min() - t0):.2f} max {us(t[:, 4].max() - t0):.2f}")
        if (t[:, 6] > 0).all():   # resident-x prologue: where its time goes
            print(f"   prologue: wave 0's x-side loads landed at {us(t[:, 6].mean() - t0):.2f} (max {us(t[:, 6].max() - t0):.2f}) | "
                  f"all waves' at {us(t[:, 7].mean() - t0):.2f} (max {us(t[:, 7].max() - t0):.2f}) | norm arithmetic + barriers "
                  f"{us((t[:, 1] - t[:, 7]).mean()):.2f} (max {us((t[:, 1] - t[:, 7]).max()):.2f})")
        per_xcc = [us(t[t[:, 5] == x][:, 4].max() - t0) for x in range(8) if (t[:, 5] == x).any()]
        print("   last exit per XCC:", " ".join(f"{v:.2f}" for v in per_xcc))
    eng.close()


if __name__ == "__main__":
    main()
