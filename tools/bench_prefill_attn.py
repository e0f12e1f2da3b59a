#!/usr/bin/env python3
"""Paged causal prefill attention probe (Qwen3-4B heads: 32 q / 8 kv, head_dim 128, 16-token pages): times
batch_prefill_paged_cuda_with_cta_tile_q on ONE request of T tokens with the plan tile the model crates use (64 packed
rows) and prints a checksum, so two builds / env settings (PEGAINFER_PREFILL_DMA=0|1 ...) can be compared for speed and
bit-equality.

    python tools/bench_prefill_attn.py 10000 [tile]
"""
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pegainfer_amd import ffi, ops  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    Hq, Hkv, D, PS = 32, 8, 128, 16
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    n_pages = (T + PS - 1) // PS
    L = ops.PagedKvLayout(1, Hkv, D, PS)
    kv = (torch.randn(n_pages * L.page_stride, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    q = torch.randn(T, Hq * D, generator=g).to(torch.bfloat16).to(dev)
    out = torch.zeros(T, Hq * D, dtype=torch.bfloat16, device=dev)
    perm = torch.randperm(n_pages, generator=g).tolist()
    if os.environ.get("PF_SAMEPAGE"):      # timing experiment: every KV tile comes from the same few pages (cache-resident KV)
        k = int(os.environ["PF_SAMEPAGE"])
        perm = [perm[i % k] for i in range(n_pages)]
    if os.environ.get("PF_LINEAR"):        # pages in address order
        perm = list(range(n_pages))
    plan = ops.PrefillPagedPlan([perm], [T - (n_pages - 1) * PS], [0], [T], Hq, Hkv, D, tile)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        rc = ffi.lib().batch_prefill_paged_cuda_with_cta_tile_q(
            q.data_ptr(), out.data_ptr(), kv.data_ptr(), 0, L.kv_block_len, plan.page_indices_d.data_ptr(),
            plan.page_indptr_d.data_ptr(), plan.last_page_len_d.data_ptr(), plan.q_indptr_d.data_ptr(),
            plan.request_indices_d.data_ptr(), plan.qo_tile_indices_d.data_ptr(), plan.kv_tile_indices_d.data_ptr(),
            plan.kv_chunk_size_d.data_ptr(), plan.total_num_rows_d.data_ptr(), Hq, Hkv, D, PS, T, 1, plan.num_tiles,
            L.page_stride, 1.0 / math.sqrt(D), plan.cta_tile_q, s)
        assert rc == 0, rc

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    iters = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    flop = 4.0 * D * Hq * T * (T + 1) / 2
    sha = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    tag = " ".join(f"{k[10:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PEGAINFER_PREFILL"))
    print(f"T={T} tile={plan.cta_tile_q} [{tag}] {us:9.1f} us  {flop / us * 1e-6:7.1f} TF/s  sha={sha}")


if __name__ == "__main__":
    main()
