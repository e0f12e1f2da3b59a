#!/usr/bin/env python3
"""Cold-cache per-op report on a real MI355X - the counterpart of the reference's manifest-driven kernel report
(pegainfer-qwen3-4b/src/bin/qwen3_kernel_report.rs:34-51, kernel_bench.rs:236-282): every op of the Qwen3-4B path
through the C ABI, timed one launch at a time with hipEvents, the caches flushed before each launch by a streaming
add_cuda over 3 x 400 MB (the reference's L2CacheClear; here it also has to push the 256 MB Infinity Cache out).
Per op: mean / min microseconds, algorithmic bytes (or flops), GB/s (TFLOP/s) and the fraction of the 8 TB/s
(2.5 PFLOP/s dense bf16) roof.  Counters for the same launches come from running this script under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace` (each launch is one dispatch, kernel names identify the op).

usage: python tools/op_report.py [--iters 10] [--only decode_attn,prefill_attn,...]"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM, MFMA = 8.0e12, 2.5e15
Hq, Hkv, D, H, I, V, PS = 32, 8, 128, 2560, 9728, 151936, 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch
    from pegainfer_amd import ffi
    from pegainfer_amd import ops as P
    L = ffi.lib()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    bf = lambda *shape, scale=1.0: (torch.randn(*shape, device=dev, generator=g) * scale).to(torch.bfloat16)
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=dev)
    stream = lambda: torch.cuda.current_stream().cuda_stream

    n_clear = 200 * 1024 * 1024   # bf16 elements per buffer = 400 MB, three buffers
    ca, cb, cc = (torch.zeros(n_clear, dtype=torch.bfloat16, device=dev) for _ in range(3))

    def clear():
        L.add_cuda(ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), n_clear, stream())

    rows = []
    only = set(x for x in args.only.split(",") if x)

    def report(group, name, fn, nbytes=0, flops=0):
        if only and group not in only:
            return
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            clear()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        mean, mn = float(np.mean(ts)), float(np.min(ts))
        if flops:
            rate, frac, unit = flops / (mean * 1e-6) / 1e12, flops / (mean * 1e-6) / MFMA, "TFLOP/s"
        else:
            rate, frac, unit = nbytes / (mean * 1e-6) / 1e9, nbytes / (mean * 1e-6) / HBM, "GB/s"
        rows.append((group, name, mean, mn, nbytes, flops, rate, unit, frac))
        print(f"{group:<14} {name:<46} {mean:9.2f} us (min {mn:8.2f})  {rate:9.1f} {unit:<8} {100 * frac:5.1f} % of roof", flush=True)

    # ---------------- elementwise / norm / rope / embedding ----------------
    for T in (1, 16, 1024):
        x, r, w = bf(T, H), bf(T, H), bf(H)
        out = torch.empty_like(x)
        report("norm", f"rms_norm_batched T={T}", lambda: P.rms_norm_batch_into(x, w, 1e-6, out), nbytes=2 * T * H * 2 + H * 2)
        report("norm", f"fused_add_rms_norm_batched T={T}", lambda: P.fused_add_rms_norm_batch_into(x, r, w, 1e-6, out),
               nbytes=4 * T * H * 2 + H * 2)
        report("elementwise", f"add T={T}", lambda: P.add_batch_into(x, r, out), nbytes=3 * T * H * 2)
        gu, act = bf(T, 2 * I), torch.empty(T, I, dtype=torch.bfloat16, device=dev)
        report("elementwise", f"silu_mul_fused T={T}", lambda: P.silu_mul_fused_batch_into(gu, act), nbytes=3 * T * I * 2)
        emb = bf(4096, H)
        tok = torch.randint(0, 4096, (T,), device=dev, dtype=torch.int32, generator=g).view(torch.uint32)
        report("embedding", f"embedding_batched T={T}", lambda: P.embedding_batch(emb, tok, out), nbytes=2 * T * H * 2)
    cos, sin = bf(8192, D), bf(8192, D)
    qn, kn = bf(D), bf(D)
    for T in (1, 16, 1024):
        q, k = bf(T, Hq * D), bf(T, Hkv * D)
        pos = i32(np.arange(T) + 100)
        report("rope", f"qk_norm_rope_batched_decode T={T}",
               lambda: P.qk_norm_rope_batch_decode_into(q, k, qn, kn, cos, sin, pos, Hq, Hkv, D, 1e-6),
               nbytes=2 * T * (Hq + Hkv) * D * 2)
    q, k = bf(1024, Hq * D), bf(1024, Hkv * D)
    report("rope", "prefill_qk_norm_rope_only T=1024", lambda: P.prefill_qk_norm_rope_only(q, k, qn, kn, cos, sin, Hq, Hkv, D, 0, 1e-6),
           nbytes=2 * 1024 * (Hq + Hkv) * D * 2)

    # ---------------- paged KV: scatter, decode attention, prefill attention ----------------
    lay = P.PagedKvLayout(1, Hkv, D, PS)

    def paged(lens):
        need = [-(-n // PS) for n in lens]
        total = sum(need) + 1
        perm = np.random.default_rng(3).permutation(np.arange(1, total))
        pages, indptr, c = [], [0], 0
        for kq in need:
            pages.extend(perm[c:c + kq].tolist()); c += kq
            indptr.append(len(pages))
        last = [((n - 1) % PS) + 1 for n in lens]
        kv = bf(total * lay.page_stride)
        return kv, pages, indptr, last

    for bs, ctx in ((1, 1024), (1, 4096), (16, 1024), (32, 1024), (32, 4096)):
        lens = [ctx] * bs
        kv, pages, indptr, last = paged(lens)
        pg, ip, lp = i32(pages), i32(indptr), i32(last)
        q, k, v = bf(bs, Hq * D), bf(bs, Hkv * D), bf(bs, Hkv * D)
        out = torch.empty(bs, Hq * D, dtype=torch.bfloat16, device=dev)
        pos, ri, kti, kcs = i32(np.asarray(lens) - 1), i32(np.arange(bs)), i32(np.zeros(bs)), i32(lens)
        kv_bytes = bs * ctx * 2 * Hkv * D * 2
        report("kv_scatter", f"paged_kv_scatter bs={bs}", lambda: P.paged_kv_scatter(kv, lay, 0, pg, ip, lp, k, v, ri, pos),
               nbytes=bs * 4 * Hkv * D * 2)
        sm = 1.0 / math.sqrt(D)
        report("decode_attn", f"paged_attention_decode bs={bs} ctx={ctx}",
               lambda: L.paged_attention_decode_cuda(q.data_ptr(), out.data_ptr(), kv.data_ptr(), 0, lay.kv_block_len, pg.data_ptr(),
                                                     ip.data_ptr(), lp.data_ptr(), ri.data_ptr(), kti.data_ptr(), kcs.data_ptr(), Hq, Hkv,
                                                     D, PS, bs, lay.page_stride, sm, stream()), nbytes=kv_bytes)
        for policy in (0, 1):
            padded = bs
            maxs = 64 * padded
            sri, skt = np.zeros(maxs, np.int32), np.zeros(maxs, np.int32)
            soi, sva, chunk, use = np.zeros(padded + 1, np.int32), np.zeros(maxs, np.uint8), np.zeros(1, np.int32), np.zeros(1, np.int32)
            hl = ffi.host_lib()
            slots = hl.pegainfer_split_kv_plan(policy, bs, np.asarray(lens, np.int32).ctypes.data, padded, Hkv, sri.ctypes.data,
                                               skt.ctypes.data, soi.ctypes.data, sva.ctypes.data, chunk.ctypes.data, use.ctypes.data)
            if not use[0] or slots <= 0:
                continue
            d_sri, d_skt, d_soi, d_chunk = i32(sri[:slots]), i32(skt[:slots]), i32(soi), i32(chunk)
            d_sva = torch.tensor(sva[:slots], dtype=torch.uint8, device=dev)
            tmp_v = torch.zeros(slots * Hq * D, dtype=torch.bfloat16, device=dev)
            tmp_s = torch.zeros(slots * Hq, dtype=torch.float32, device=dev)
            report("decode_attn", f"paged_attention_decode_split_kv policy={policy} bs={bs} ctx={ctx} ({slots} slots)",
                   lambda: L.paged_attention_decode_split_kv_cuda(
                       q.data_ptr(), out.data_ptr(), kv.data_ptr(), 0, lay.kv_block_len, pg.data_ptr(), ip.data_ptr(), lp.data_ptr(),
                       d_sri.data_ptr(), d_skt.data_ptr(), d_chunk.data_ptr(), d_soi.data_ptr(), d_sva.data_ptr(), tmp_v.data_ptr(),
                       tmp_s.data_ptr(), Hq, Hkv, D, PS, bs, slots, lay.page_stride, sm, stream()), nbytes=kv_bytes)
    for T in (1024, 4096, 10000):
        kv, pages, indptr, last = paged([T])
        plan = P.PrefillPagedPlan([pages], last, [0], [T], Hq, Hkv, D, 64)
        q = bf(T, Hq * D)
        out = torch.empty_like(q)
        flops = 4.0 * Hq * D * (T * (T + 1) / 2)
        report("prefill_attn", f"batch_prefill_paged (attention core) T={T} cta 64",
               lambda: L.batch_prefill_paged_cuda_with_cta_tile_q(
                   q.data_ptr(), out.data_ptr(), kv.data_ptr(), 0, lay.kv_block_len, plan.page_indices_d.data_ptr(),
                   plan.page_indptr_d.data_ptr(), plan.last_page_len_d.data_ptr(), plan.q_indptr_d.data_ptr(),
                   plan.request_indices_d.data_ptr(), plan.qo_tile_indices_d.data_ptr(), plan.kv_tile_indices_d.data_ptr(),
                   plan.kv_chunk_size_d.data_ptr(), plan.total_num_rows_d.data_ptr(), Hq, Hkv, D, PS, T, 1, plan.num_tiles,
                   lay.page_stride, 1.0 / math.sqrt(D), plan.cta_tile_q, stream()), flops=flops)
        k, v = bf(T, Hkv * D), bf(T, Hkv * D)
        report("kv_scatter", f"paged_kv_scatter (prefill) T={T}",
               lambda: P.paged_kv_scatter(kv, lay, 0, plan.page_indices_d, plan.page_indptr_d, plan.last_page_len_d, k, v,
                                          plan.batch_indices_d, plan.positions_d), nbytes=T * 4 * Hkv * D * 2)

    # ---------------- sampling ----------------
    logits = bf(V, scale=4.0)
    report("sampling", "argmax V=151936", lambda: L.argmax_cuda(logits.data_ptr(), torch.zeros(1, dtype=torch.int32, device=dev).data_ptr(), V, stream()),
           nbytes=V * 2)
    lg32 = bf(32, V, scale=4.0)
    state = torch.zeros(P.FLASHINFER_TOPK_ROW_STATES_BYTES, dtype=torch.uint8, device=dev)
    outi = torch.zeros(32, dtype=torch.int32, device=dev)
    report("sampling", "pegainfer_batched_top1 32 x V", lambda: L.pegainfer_batched_top1(lg32.data_ptr(), V, 32, V, state.data_ptr(), outi.data_ptr(), stream()),
           nbytes=32 * V * 2)
    probs = torch.zeros(V, dtype=torch.float32, device=dev)
    valid = torch.zeros(1, dtype=torch.uint8, device=dev)
    report("sampling", "gpu_sample_flashinfer (T 0.8, top-k 50, top-p 0.95)",
           lambda: L.gpu_sample_flashinfer_cuda(logits.data_ptr(), probs.data_ptr(), valid.data_ptr(), outi.data_ptr(), V, 1.0 / 0.8, 50, 0.95,
                                                12345, stream()), nbytes=V * 2)

    # ---------------- GEMM call sites (cold weights by construction) ----------------
    for name, M, K in (("qkv", (Hq + 2 * Hkv) * D, H), ("o", H, Hq * D), ("gate_up", 2 * I, H), ("down", H, I), ("lm_head", V, H)):
        W = bf(M, K, scale=0.02)
        for T in (1, 16, 1024):
            if name == "lm_head" and T == 1024:
                continue
            x, y = bf(T, K), torch.empty(T, M, dtype=torch.bfloat16, device=dev)
            if T <= 16:
                report("gemm", f"gemm_cuda {name} [{M}x{K}] T={T}", lambda: L.gemm_cuda(W.data_ptr(), x.data_ptr(), y.data_ptr(), M, T, K, stream()),
                       nbytes=M * K * 2 + T * (K + M) * 2)
            else:
                report("gemm", f"gemm_cuda {name} [{M}x{K}] T={T}", lambda: L.gemm_cuda(W.data_ptr(), x.data_ptr(), y.data_ptr(), M, T, K, stream()),
                       flops=2.0 * M * K * T)
    print(f"\n{len(rows)} ops, {args.iters} cold launches each; flush = add_cuda over 3 x {n_clear * 2 >> 20} MB")


if __name__ == "__main__":
    main()
