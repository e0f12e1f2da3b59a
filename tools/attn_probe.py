#!/usr/bin/env python3
"""Phase breakdown of the bs = 1 fused decode-attention launch on a real MI355X, from the in-kernel wall-clock
stamps of pegainfer_debug_attn_trace (100 MHz): per workgroup, entry -> slot record -> q prologue -> KV scan ->
partials published -> ticket -> merge.  Prints the mean / max of every phase over the workgroups of the LAST launch
(last layer of the last step) and the span first entry -> last exit.
usage: python tools/attn_probe.py [--ctx 1024] [--steps 8] [--layers 4] [--batch 1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["record", "q_prologue", "scan", "publish", "ticket", "merge"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    import torch
    from pegainfer_amd import ffi
    from pegainfer_amd.qwen3 import QWEN3_4B, Qwen3Engine
    cfg = dict(QWEN3_4B, num_hidden_layers=args.layers)
    prompt = [100 + (i % 1000) for i in range(args.ctx)]
    pages = args.batch * (-(-(args.ctx + args.steps + 64) // 16)) + 8
    eng = Qwen3Engine(cfg, num_kv_pages=pages, max_batch_size=max(2, args.batch), decode_mode=1, split_policy=1,
                      enable_graph=False, max_positions=max(4096, args.ctx + args.steps + 64))
    eng.fill_synthetic(seed=42, std=0.02)
    rids = [eng.new_request() for _ in range(args.batch)]
    toks = [int(eng.prefill([r], [prompt])[0]) for r in rids]
    nslots, hkv = 64 * max(args.batch, 1), cfg["num_key_value_heads"]
    buf = torch.zeros((nslots * hkv + 256) * 8, dtype=torch.int64, device="cuda")   # + the o_proj phase of the fused launch
    ffi.lib().pegainfer_debug_attn_trace(buf.data_ptr())
    for _ in range(args.steps):
        buf.zero_()
        toks = [int(t) for t in eng.decode(rids, toks)]
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(nslots * hkv + 256, 8).astype(np.float64)
    t, op = raw[:nslots * hkv], raw[64 * hkv:64 * hkv + 256]
    ffi.lib().pegainfer_debug_attn_trace(None)
    live = t[:, 0] > 0
    t = t[live]
    print(f"workgroups stamped: {len(t)}  (ctx {args.ctx}, batch {args.batch})")
    ns = 10.0  # 100 MHz
    for i, name in enumerate(NAMES):
        a, b = t[:, i], t[:, i + 1]
        # merge: only the workgroups that merged in THIS launch (an earlier layer's launch may have left its [6] / [7])
        ok = (b > 0) & (a > 0) if name != "merge" else ((t[:, 7] > 0) & (b > a))
        if ok.any():
            d = (b[ok] - a[ok]) * ns / 1e3
            print(f"  {name:<11} mean {d.mean():6.2f} us  max {d.max():6.2f} us  (n={ok.sum()})")
    last = np.where(t[:, 7] > 0, t[:, 6], np.where(t[:, 5] > 0, t[:, 5], t[:, 3]))
    print(f"  first entry -> last exit: {(last.max() - t[:, 0].min()) * ns / 1e3:.2f} us;"
          f" entry spread {(t[:, 0].max() - t[:, 0].min()) * ns / 1e3:.2f} us")
    if args.batch == 1 and (op[:, 4] > 0).any():   # fused attention + o_proj launch: the o_proj workgroups (padding slots)
        op = op[op[:, 4] > 0]
        t0 = min(op[:, 0].min(), t[:, 0].min())
        us = lambda a: (a - t0) * ns / 1e3
        print(f"  fused launch: {len(t)} attention + {len(op)} o_proj workgroups; attention's last exit at {us(last.max()):.2f} | o_proj rows "
              f"requested at {us(op[:, 1]).mean():.2f} | attention row seen at {us(op[:, 2]).mean():.2f} (min {us(op[:, 2]).min():.2f} max "
              f"{us(op[:, 2]).max():.2f}) | x staged at {us(op[:, 3]).mean():.2f} (max {us(op[:, 3]).max():.2f}) | exit {us(op[:, 4]).mean():.2f} "
              f"(max {us(op[:, 4]).max():.2f}) | dot products done {us(op[:, 5]).mean():.2f}, wave sums done {us(op[:, 6]).mean():.2f}")
    eng.close()


if __name__ == "__main__":
    main()
