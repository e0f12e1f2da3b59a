#!/usr/bin/env python3
"""Sampling kernel probe: device time of gpu_sample_flashinfer_cuda at the Qwen3 vocabulary for the filter
combinations of ops_embedding_sampling_bench.rs:49-90, plus flashinfer_top1_cuda."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pegainfer_amd import ffi  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 151936
    L = ffi.lib()
    dev = torch.device("cuda:0")
    idx = torch.arange(V, device=dev)
    logits = (((idx % 1024).float() / 1024.0) * 6.0 - 3.0).to(torch.bfloat16)      # reference bench pattern
    logits2 = (torch.randn(V, device=dev) * 2.0).to(torch.bfloat16)
    probs = torch.zeros(V, dtype=torch.float32, device=dev)
    valid = torch.zeros(1, dtype=torch.uint8, device=dev)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    top1v = torch.zeros(1, dtype=torch.bfloat16, device=dev)
    rows = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    cases = [("temperature only", 0.8, -1, 1.0), ("top_k 50", 0.8, 50, 1.0), ("top_p 0.9", 0.8, -1, 0.9),
             ("top_k 50 + top_p 0.95", 0.8, 50, 0.95)]
    for name_l, lg in (("pattern", logits), ("randn", logits2)):
        for name, t, k, p in cases:
            def run():
                L.gpu_sample_flashinfer_cuda(lg.data_ptr(), probs.data_ptr(), valid.data_ptr(), out.data_ptr(), V,
                                             1.0 / t, k, p, 12345, s)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f"{name_l:8s} {name:24s} {e0.elapsed_time(e1) * 1000 / 20:8.1f} us   token {int(out.item())}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.flashinfer_top1_cuda(logits2.data_ptr(), top1v.data_ptr(), rows.data_ptr(), out.data_ptr(), V, s)
    e1.record()
    torch.cuda.synchronize()
    print(f"top1 {e0.elapsed_time(e1) * 1000 / 20:8.1f} us")


if __name__ == "__main__":
    main()
