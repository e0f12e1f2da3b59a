#!/usr/bin/env python3
"""Probe of the persistent decode engine (decode_mode 2) on a real MI355X:
  * diff : per decode step, how many logit bits differ from decode_mode 1 (same synthetic checkpoint, teacher forced)
  * trace: phase breakdown of one engine step from the in-kernel cycle counters (PEGAINFER_ENGINE_TRACE=1)
usage: python tools/engine_probe.py [--layers 36] [--ctx 1024] [--steps 16] [--policy 1] [--diff] [--trace]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAMES = ["wait_down", "stage_qkv", "gemv_qkv", "store_qkv", "wait_qkv", "attention", "wait_attn", "stage_o", "gemv_o",
         "store_o", "wait_o", "stage_gu", "gemv_gu", "store_gu", "wait_gu", "stage_dn", "gemv_dn", "store_dn",
         "ready_wait(c0)", "-", "loader_free_wait", "loader_total", "consumer_total"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=36)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--policy", type=int, default=1)
    ap.add_argument("--diff", action="store_true")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    if args.trace:
        os.environ["PEGAINFER_ENGINE_TRACE"] = "1"
    from pegainfer_amd.qwen3 import QWEN3_4B, Qwen3Engine
    cfg = dict(QWEN3_4B, num_hidden_layers=args.layers)
    prompt = [100 + (i % 1000) for i in range(args.ctx)]
    pages = -(-(args.ctx + args.steps + 64) // 16) + 8

    def run(mode, feed=None):
        eng = Qwen3Engine(cfg, num_kv_pages=pages, max_batch_size=2, decode_mode=mode, split_policy=args.policy,
                          enable_graph=not args.no_graph, max_positions=max(4096, args.ctx + args.steps + 64))
        eng.fill_synthetic(seed=42, std=0.02)
        rid = eng.new_request()
        tok = int(eng.prefill([rid], [prompt])[0])
        toks, rows, ms = [tok], [], []
        for s in range(args.steps):
            t = toks[-1] if feed is None else feed[s]
            o, lg = eng.decode([rid], [t], return_logits=True)
            rows.append(lg[0].copy())
            toks.append(int(o[0]))
            ms.append(eng.last_step_ms())
        tr = None
        if args.trace and mode == 2:
            buf = (ctypes.c_uint64 * (256 * 32))()
            n = eng.lib.pegainfer_qwen3_engine_trace(eng.h, buf, 256 * 32)
            tr = np.frombuffer(buf, dtype=np.uint64)[:n].reshape(-1, 32).astype(np.float64)
        active = eng.lib.pegainfer_qwen3_engine_active(eng.h)
        eng.close()
        return toks, np.stack(rows), ms, tr, active

    t1, b1, ms1, _, _ = run(1)
    t2, b2, ms2, tr, active = run(2, feed=t1[:-1])
    print(f"engine_active={active} mode1 median {np.median(ms1):.3f} ms/step, mode2 median {np.median(ms2):.3f} ms/step")
    if args.diff:
        for s in range(args.steps):
            nd = int((b1[s] != b2[s]).sum())
            if nd:
                a = (b1[s].astype(np.uint32) << 16).view(np.float32)
                b = (b2[s].astype(np.uint32) << 16).view(np.float32)
                print(f"step {s}: {nd} logits differ, max |d| {np.abs(a - b).max():.4f}, tok {t1[s + 1]} vs {t2[s + 1]}")
        print("first differing step:", next((s for s in range(args.steps) if (b1[s] != b2[s]).any()), None))
    if tr is not None:
        L = args.layers
        clk = 100e6 if tr[:, 22].mean() < 1e6 * 10 and np.median(ms2) > 0 and tr[:, 22].mean() / (np.median(ms2) * 1e-3) < 5e8 else None
        tick_per_us = tr[:, 22].mean() / (np.median(ms2) * 1e3)   # calibrate ticks against the event-timed step
        print(f"s_memtime ticks per us (calibrated on consumer_total vs step time): {tick_per_us:.1f}")
        print(f"{'phase':18s} {'mean us/layer':>14s} {'min':>8s} {'max':>8s}   (over {tr.shape[0]} workgroups)")
        for i, nm in enumerate(NAMES):
            if nm == "-":
                continue
            v = tr[:, i] / tick_per_us / (L if i < 19 or i == 20 else 1)
            print(f"{nm:18s} {v.mean():14.2f} {v.min():8.2f} {v.max():8.2f}")


if __name__ == "__main__":
    main()
