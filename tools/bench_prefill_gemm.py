#!/usr/bin/env python3
"""Prefill-shaped GEMM probe: times gemm_cuda on the Qwen3-4B projection shapes at T tokens and prints a
checksum of each result, so two runs with different PEGAINFER_GEMM modes (reg | default | w | n) can be compared
for speed AND bit-equality (all un-split variants keep the same per-element K order; PEGAINFER_SPLITK=0 turns the
3-slice split of the < 256-tile shapes off).

    PEGAINFER_GEMM=reg python tools/bench_prefill_gemm.py 1024
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pegainfer_amd import ops  # noqa: E402

SHAPES = [("qkv", 6144, 2560), ("o", 2560, 4096), ("gate_up", 19456, 2560), ("down", 2560, 9728),
          ("ragged", 1000, 2560)]
if os.environ.get("PEGAINFER_PROBE_MODEL") == "8b":     # the Qwen3-8B projections
    SHAPES = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 24576, 4096), ("down", 4096, 12288)]


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # > 1: cycle through that many weight buffers (cold W)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    mode = os.environ.get("PEGAINFER_GEMM", "default")
    total_us, total_flop = 0.0, 0.0
    for name, M, K in SHAPES:
        W = (torch.randn(M, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        Ws = [W] + [W.clone() for _ in range(copies - 1)]
        X = torch.randn(T, K, generator=g).to(torch.bfloat16).to(dev)
        Y = torch.empty(T, M, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm_into(W, X, Y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for i in range(iters):
            ops.gemm_into(Ws[i % copies], X, Y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / iters
        flop = 2.0 * M * T * K
        digest = hashlib.sha1(Y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
        ref = (X.float() @ W.float().T)
        err = (Y.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        print(f"{mode:8s} copies={copies} T={T} {name:8s} M={M:6d} K={K:5d}  {us:8.1f} us  {flop / us * 1e-6:7.1f} TF/s  sha={digest} rel_err={err:.2e}")
        if name != "ragged":
            total_us += us
            total_flop += flop
    print(f"{mode:8s} T={T} layer-total {total_us:8.1f} us  {total_flop / total_us * 1e-6:7.1f} TF/s")


if __name__ == "__main__":
    main()
