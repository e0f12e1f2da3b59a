#!/usr/bin/env python3
"""Batched-decode GEMM probe (2 <= T <= 64 token columns): per call-site launch time over the 36 layers' weights of a
synthetic Qwen3-4B (cold weights, like a decode step) and a checksum of a seeded product per shape, so routing /
kernel variants selected through environment knobs (PEGAINFER_SKINNY_MIN_T, PEGAINFER_SKINNY_RB, PEGAINFER_SPLITK,
PEGAINFER_MID_MIN_ROWS) can be compared for speed AND bit-equality in one GPU call (tools/gpu_skinny_probe.sh):

    SKINNY_TS='8 16 32' SKINNY_VARIANTS='X=0 PEGAINFER_SPLITK=0' bash tools/gpu_skinny_probe.sh
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pegainfer_amd import ops  # noqa: E402
from pegainfer_amd.qwen3 import QWEN3_4B, Qwen3Engine  # noqa: E402

SITES = [(0, "qkv", 6144, 2560), (1, "o", 2560, 4096), (2, "gate_up", 19456, 2560), (3, "down", 2560, 9728),
         (4, "lm_head", 151936, 2560), (5, "gate_up+norm+silu", 19456, 2560), (6, "qkv+norm", 6144, 2560)]


def main():
    Ts = [int(x) for x in sys.argv[1:]] or [8, 16]
    knobs = ("PEGAINFER_SKINNY_", "PEGAINFER_SPLITK", "PEGAINFER_MID_", "PEGAINFER_STREAM_")
    tag = " ".join(f"{k[10:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith(knobs)) or "default"
    dev = torch.device("cuda:0")
    eng = Qwen3Engine(dict(QWEN3_4B), num_kv_pages=64, max_batch_size=max(Ts), enable_graph=False, device=0)
    eng.fill_synthetic(seed=42, std=0.02)
    for T in Ts:
        cells, total = [], 0.0
        for which, name, M, K in SITES:
            if which >= 5 and T > 16:      # the fused prologue / epilogue forms exist up to 16 columns
                continue
            us = eng.bench_gemv(which, 20 if which == 4 else 72, T) * 1e3
            cells.append(f"{name} {us:7.2f}us {M * K * 2 / us * 1e-6:5.2f}TB/s")
            if which in (0, 1, 2, 3):
                total += us
        print(f"[{tag}] T={T:2d} layer4={total:7.2f}us | " + " | ".join(cells), flush=True)
    eng.close()
    # bit-equality across variants (and a sanity bound against fp32 torch) on seeded products
    g = torch.Generator(device="cpu").manual_seed(11)
    for name, M, K in [("o", 2560, 4096), ("down", 2560, 9728), ("qkv", 6144, 2560), ("ragged", 1000, 2560),
                       ("gate_up", 19456, 2560), ("tall_ragged", 12300, 4096)]:
        W = (torch.randn(M, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        for T in (5, 16, 32, 64):
            X = torch.randn(T, K, generator=g).to(torch.bfloat16).to(dev)
            Y = torch.empty(T, M, dtype=torch.bfloat16, device=dev)
            ops.gemm_into(W, X, Y)
            torch.cuda.synchronize()
            ref = X.float() @ W.float().T
            err = (Y.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
            sha = hashlib.sha1(Y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
            print(f"[{tag}] check {name} T={T} sha={sha} rel_err={err:.2e}", flush=True)


if __name__ == "__main__":
    main()
