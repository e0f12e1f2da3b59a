"""The MI355X fused decode kernels (include/pegainfer_kernels_ext.h) must be BIT-IDENTICAL to the sequence
of reference-named ops they replace - at op level and through the whole model (decode_mode 1 vs 0)."""
import json
import os

import numpy as np
import pytest

from conftest import from_dev, to_dev
from oracle import ops as O
from oracle.bf16 import bf16_bits, bf16_round

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rnd(rng, *shape, scale=1.0):
    return bf16_round((rng.standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("T", [1, 2, 3, 8, 11, 16])
@pytest.mark.parametrize("M,K", [(6144, 2560), (2560, 4096), (512, 256)])
def test_gemv_fused_prologues_match_unfused(built_libs, T, M, K):
    import torch
    import pegainfer_amd.ops as P
    from pegainfer_amd import ffi
    rng = np.random.default_rng(T + M + K)
    W, X, R, g = rnd(rng, M, K, scale=0.05), rnd(rng, T, K, scale=2), rnd(rng, T, K, scale=0.7), bf16_round(1 + rnd(rng, K, scale=0.2))
    Wd, s = to_dev(W), torch.cuda.current_stream().cuda_stream
    L = ffi.lib()
    # (a) plain == gemm_graphsafe
    y1 = torch.zeros((T, M), dtype=torch.bfloat16, device="cuda")
    Xd = to_dev(X)
    assert L.pegainfer_gemv_fused(Wd.data_ptr(), Xd.data_ptr(), y1.data_ptr(), M, T, K, None, None, None, 0.0, 0, s) == 0
    assert np.array_equal(bf16_bits(from_dev(y1)), bf16_bits(from_dev(P.gemm(Wd, Xd))))
    # (b) norm prologue == rms_norm_batched + gemm
    normed = torch.zeros_like(Xd)
    P.rms_norm_batch_into(Xd, to_dev(g), 1e-6, normed)
    ref = P.gemm(Wd, normed)
    gd = to_dev(g)
    assert L.pegainfer_gemv_fused(Wd.data_ptr(), Xd.data_ptr(), y1.data_ptr(), M, T, K, None, gd.data_ptr(), None, 1e-6, 0, s) == 0
    assert np.array_equal(bf16_bits(from_dev(y1)), bf16_bits(from_dev(ref)))
    # (c) add+norm prologue == fused_add_rms_norm_batched + gemm, hidden_out == updated hidden
    hid, Rd = to_dev(X), to_dev(R)
    P.fused_add_rms_norm_batch_into(hid, Rd, gd, 1e-6, normed)
    ref = P.gemm(Wd, normed)
    hout = torch.zeros_like(Xd)
    assert L.pegainfer_gemv_fused(Wd.data_ptr(), Xd.data_ptr(), y1.data_ptr(), M, T, K, Rd.data_ptr(), gd.data_ptr(),
                                  hout.data_ptr(), 1e-6, 0, s) == 0
    assert np.array_equal(bf16_bits(from_dev(y1)), bf16_bits(from_dev(ref)))
    assert np.array_equal(bf16_bits(from_dev(hout)), bf16_bits(from_dev(hid)))


@pytest.mark.parametrize("T,I,K", [(1, 9728, 2560), (4, 9728, 2560), (16, 512, 256), (2, 1000, 4096), (12, 9728, 2560)])
def test_gemv_fused_silu_epilogue_matches_unfused(built_libs, T, I, K):
    import torch
    import pegainfer_amd.ops as P
    from pegainfer_amd import ffi
    rng = np.random.default_rng(T + I)
    W, X = rnd(rng, 2 * I, K, scale=0.06), rnd(rng, T, K, scale=1.5)
    Wd, Xd = to_dev(W), to_dev(X)
    gu = P.gemm(Wd, Xd)
    ref = torch.zeros((T, I), dtype=torch.bfloat16, device="cuda")
    P.silu_mul_fused_batch_into(gu, ref)
    out = torch.zeros_like(ref)
    assert ffi.lib().pegainfer_gemv_fused(Wd.data_ptr(), Xd.data_ptr(), out.data_ptr(), 2 * I, T, K, None, None, None,
                                          0.0, I, torch.cuda.current_stream().cuda_stream) == 0
    assert np.array_equal(bf16_bits(from_dev(out)), bf16_bits(from_dev(ref)))


@pytest.mark.parametrize("T", [3, 8, 16])
def test_skinny_flush_forms_are_bit_identical(built_libs, T):
    """Round 5: skinny_resident_kernel combines its 8 waves' partial sums per row block with two barriers, one barrier, or
    LDS tickets without a barrier - waiting or lazy (the launcher's choice where a workgroup walks more than two row blocks).  Same bits in all
    three, on shapes whose workgroups walk 1, 2, 3-5 and ~40 row blocks (the last two re-use the ticket ring's buffers), for
    the plain store and for the add + RMSNorm prologue with the SwiGLU epilogue."""
    import torch
    from pegainfer_amd import ffi
    L = ffi.lib()
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(100 + T)
    try:
        for M, K in [(2560, 4096), (6144, 2560), (19456, 2560), (12300, 4096), (151936 // 4, 2560)]:
            W, X = rnd(rng, M, K, scale=0.05), rnd(rng, T, K)
            Wd, Xd = to_dev(W), to_dev(X)
            outs = {}
            for mode in (0, 1, 4, 5, -1):
                L.pegainfer_debug_skinny_flush(mode)
                y = torch.zeros((T, M), dtype=torch.bfloat16, device="cuda")
                assert L.pegainfer_gemv_fused(Wd.data_ptr(), Xd.data_ptr(), y.data_ptr(), M, T, K, None, None, None, 0.0, 0, s) == 0
                outs[mode] = bf16_bits(from_dev(y))
            assert all(np.array_equal(outs[0], outs[m]) for m in (1, 4, 5, -1)), (M, K)
            ref = X.astype(np.float64) @ W.astype(np.float64).T
            assert np.abs(from_dev(y) - ref).max() <= 2.0 ** -6 * max(1.0, np.abs(ref).max())
        I, K = 9728, 2560
        W, X, R, g = rnd(rng, 2 * I, K, scale=0.06), rnd(rng, T, K), rnd(rng, T, K), rnd(rng, K)
        Wd, Xd, Rd, gd = to_dev(W), to_dev(X), to_dev(R), to_dev(g)
        outs = {}
        for mode in (0, 1, 4, 5, -1):
            L.pegainfer_debug_skinny_flush(mode)
            y = torch.zeros((T, I), dtype=torch.bfloat16, device="cuda")
            h = torch.zeros((T, K), dtype=torch.bfloat16, device="cuda")
            assert L.pegainfer_gemv_fused(Wd.data_ptr(), Xd.data_ptr(), y.data_ptr(), 2 * I, T, K, Rd.data_ptr(), gd.data_ptr(),
                                          h.data_ptr(), 1e-6, I, s) == 0
            outs[mode] = (bf16_bits(from_dev(y)), bf16_bits(from_dev(h)))
        for m in (1, 4, 5, -1):
            assert np.array_equal(outs[0][0], outs[m][0]) and np.array_equal(outs[0][1], outs[m][1])
    finally:
        L.pegainfer_debug_skinny_flush(-1)


def test_gemv_fused_rejects_unsupported_shapes(built_libs):
    import torch
    from pegainfer_amd import ffi
    x = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    L = ffi.lib()
    assert L.pegainfer_gemv_fused(x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, 65, 8, None, None, None, 0.0, 0, s) != 0
    assert L.pegainfer_gemv_fused(x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, 1, 12, None, None, None, 0.0, 0, s) != 0
    assert L.pegainfer_gemv_fused(x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, 1, 8, x.data_ptr(), x.data_ptr(),
                                  x.data_ptr(), 0.0, 0, s) != 0   # hidden_out aliases X
    # prologue / epilogue forms are for decode batches <= 16; the plain form still takes 17..64 columns
    big = torch.zeros(64 * 64, dtype=torch.bfloat16, device="cuda")
    assert L.pegainfer_gemv_fused(big.data_ptr(), big.data_ptr(), big.data_ptr(), 8, 17, 64, None, big.data_ptr(), None,
                                  1e-6, 0, s) != 0
    assert L.pegainfer_gemv_fused(big.data_ptr(), big.data_ptr(), big.data_ptr(), 8, 17, 64, None, None, None, 0.0, 4,
                                  s) != 0
    y = torch.zeros(17 * 8, dtype=torch.bfloat16, device="cuda")
    assert L.pegainfer_gemv_fused(big.data_ptr(), big.data_ptr(), y.data_ptr(), 8, 17, 64, None, None, None, 0.0, 0,
                                  s) == 0      # 17..64 columns: the MFMA kernels take K % 64 == 0
    assert L.pegainfer_gemv_fused(big.data_ptr(), big.data_ptr(), y.data_ptr(), 8, 17, 32, None, None, None, 0.0, 0,
                                  s) != 0


@pytest.mark.parametrize("lens,split", [([1], False), ([17, 300], False), ([1024], True), ([2000, 70], True)])
def test_fused_decode_attention_matches_unfused(built_libs, lens, split):
    """qk_norm_rope + scatter + decode attention in one launch == the three reference-named calls:
    identical attention output bits AND identical KV-cache bytes."""
    import torch
    import pegainfer_amd.ops as P
    from pegainfer_amd import ffi
    from test_gpu_ops import make_paged
    rng = np.random.default_rng(sum(lens))
    bs, Hq, Hkv, D = len(lens), 32, 8, 128
    lay, kv, pages, indptr, last = make_paged(rng, bs, lens)
    qkv = rnd(rng, bs, (Hq + 2 * Hkv) * D, scale=1.5)
    qw, kw = bf16_round(1 + rnd(rng, D, scale=0.2)), bf16_round(1 + rnd(rng, D, scale=0.2))
    cos, sin = O.precompute_rope(D, 4096, 1e6)
    pos = np.asarray(lens, np.int32) - 1                   # the new token is the last position
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device="cuda")
    Lk = P.PagedKvLayout(2, Hkv, D, 16)
    plan = O.split_kv_plan(lens, bs)
    d = dict(pages=i32(pages), indptr=i32(indptr), last=i32(last), pos=i32(pos), ri=i32(np.arange(bs)),
             kti=i32(np.zeros(bs)), kcs=i32(lens), sri=i32(plan["request_indices"]), skt=i32(plan["kv_tile_indices"]),
             skc=i32([plan["kv_chunk_size"]]), soi=i32(plan["o_indptr"]),
             sva=torch.tensor(plan["block_valid_mask"], dtype=torch.uint8, device="cuda"),
             qw=to_dev(qw), kw=to_dev(kw), cos=to_dev(cos), sin=to_dev(sin))
    slots = plan["padded_slots"]
    tmp_v = torch.zeros(slots * Hq * D, dtype=torch.bfloat16, device="cuda")
    tmp_s = torch.zeros(slots * Hq, dtype=torch.float32, device="cuda")
    # ---- unfused reference-named sequence ----
    kv_a = to_dev(kv)
    qkvd = to_dev(qkv)
    q = qkvd[:, :Hq * D].clone(); k = qkvd[:, Hq * D:(Hq + Hkv) * D].clone(); v = qkvd[:, (Hq + Hkv) * D:].clone()
    P.qk_norm_rope_batch_decode_into(q, k, d["qw"], d["kw"], d["cos"], d["sin"], d["pos"], Hq, Hkv, D, 1e-6)
    out_a = torch.zeros((bs, Hq * D), dtype=torch.bfloat16, device="cuda")
    if split:
        P.paged_attention_batch_decode_split_kv_into(q, k, v, kv_a, Lk, 1, d["pages"], d["indptr"], d["last"], d["pos"],
                                                     d["ri"], d["sri"], d["skt"], d["skc"], d["soi"], d["sva"], tmp_v,
                                                     tmp_s, slots, out_a, Hq, bs)
    else:
        P.paged_attention_batch_decode_into(q, k, v, kv_a, Lk, 1, d["pages"], d["indptr"], d["last"], d["pos"], d["ri"],
                                            d["kti"], d["kcs"], out_a, Hq, bs)
    # ---- fused ----
    kv_b = to_dev(kv)
    out_b = torch.zeros_like(out_a)
    rc = ffi.lib().pegainfer_fused_decode_attention(
        qkvd.data_ptr(), out_b.data_ptr(), kv_b.data_ptr(), Lk.layer_stride, Lk.layer_stride + Lk.kv_block_len,
        d["pages"].data_ptr(), d["indptr"].data_ptr(), d["last"].data_ptr(), d["pos"].data_ptr(), d["qw"].data_ptr(),
        d["kw"].data_ptr(), d["cos"].data_ptr(), d["sin"].data_ptr(), 1e-6, int(split), d["sri"].data_ptr(),
        d["skt"].data_ptr(), d["skc"].data_ptr(), d["soi"].data_ptr(), d["sva"].data_ptr(), tmp_v.data_ptr(),
        tmp_s.data_ptr(), Hq, Hkv, D, 16, bs, slots, Lk.page_stride, 1.0 / np.sqrt(128.0), None, None,
        torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert np.array_equal(bf16_bits(from_dev(kv_b)), bf16_bits(from_dev(kv_a)))
    assert np.array_equal(bf16_bits(from_dev(out_b)), bf16_bits(from_dev(out_a)))
    # ---- fused with host-resolved slot descriptors: same bits again ----
    nslots = slots if split else bs
    desc = np.zeros((nslots, 8), np.int32)
    desc[:, 1] = -1
    if split:
        for sl in range(int(plan["o_indptr"][bs])):
            b = int(plan["request_indices"][sl]); t = int(plan["kv_tile_indices"][sl]); c = plan["kv_chunk_size"]
            desc[sl] = [b, t * c, min((t + 1) * c, lens[b]), indptr[b], pos[b], lens[b], 0, 0]
    else:
        for b in range(bs):
            desc[b] = [b, 0, lens[b], indptr[b], pos[b], lens[b], 0, 0]
    dd = i32(desc.reshape(-1))
    kv_c = to_dev(kv)
    out_c = torch.zeros_like(out_a)
    rc = ffi.lib().pegainfer_fused_decode_attention(
        qkvd.data_ptr(), out_c.data_ptr(), kv_c.data_ptr(), Lk.layer_stride, Lk.layer_stride + Lk.kv_block_len,
        d["pages"].data_ptr(), d["indptr"].data_ptr(), d["last"].data_ptr(), d["pos"].data_ptr(), d["qw"].data_ptr(),
        d["kw"].data_ptr(), d["cos"].data_ptr(), d["sin"].data_ptr(), 1e-6, int(split), d["sri"].data_ptr(),
        d["skt"].data_ptr(), d["skc"].data_ptr(), d["soi"].data_ptr(), d["sva"].data_ptr(), tmp_v.data_ptr(),
        tmp_s.data_ptr(), Hq, Hkv, D, 16, bs, slots, Lk.page_stride, 1.0 / np.sqrt(128.0), dd.data_ptr(), None,
        torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert np.array_equal(bf16_bits(from_dev(kv_c)), bf16_bits(from_dev(kv_a)))
    assert np.array_equal(bf16_bits(from_dev(out_c)), bf16_bits(from_dev(out_a)))
    # ---- partials merged by the last workgroup of each (request, kv head) instead of a merge launch: same bits,
    #      counters re-armed (run twice on the same counters) ----
    if split:
        ctr = torch.zeros(bs * Hkv * 32, dtype=torch.int32, device="cuda")   # one cache line per (request, kv head)
        for _ in range(2):
            kv_d = to_dev(kv)
            out_d = torch.zeros_like(out_a)
            rc = ffi.lib().pegainfer_fused_decode_attention(
                qkvd.data_ptr(), out_d.data_ptr(), kv_d.data_ptr(), Lk.layer_stride, Lk.layer_stride + Lk.kv_block_len,
                d["pages"].data_ptr(), d["indptr"].data_ptr(), d["last"].data_ptr(), d["pos"].data_ptr(),
                d["qw"].data_ptr(), d["kw"].data_ptr(), d["cos"].data_ptr(), d["sin"].data_ptr(), 1e-6, 1,
                d["sri"].data_ptr(), d["skt"].data_ptr(), d["skc"].data_ptr(), d["soi"].data_ptr(), d["sva"].data_ptr(),
                tmp_v.data_ptr(), tmp_s.data_ptr(), Hq, Hkv, D, 16, bs, slots, Lk.page_stride, 1.0 / np.sqrt(128.0),
                dd.data_ptr(), ctr.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            assert np.array_equal(bf16_bits(from_dev(kv_d)), bf16_bits(from_dev(kv_a)))
            assert np.array_equal(bf16_bits(from_dev(out_d)), bf16_bits(from_dev(out_a)))
            assert int(ctr.abs().sum().item()) == 0


@pytest.mark.parametrize("split_policy", [0, 1])
def test_model_fused_decode_bitwise_equals_reference_sequence(built_libs, split_policy):
    from pegainfer_amd.qwen3 import Qwen3Engine
    meta = json.load(open(os.path.join(G, "qwen3_tiny_golden.json")))
    ck = os.path.join(G, "qwen3_tiny.safetensors")
    rng = np.random.default_rng(1)
    prompts = [meta["cases"][3]["prompt_tokens"], rng.integers(0, 1024, 1100).tolist()]
    outs = []
    for mode in (0, 1):
        eng = Qwen3Engine(meta["config"], num_kv_pages=256, max_batch_size=4, decode_mode=mode,
                          split_policy=split_policy).load_safetensors(ck)
        rids = [eng.new_request() for _ in prompts]
        toks = np.array([int(eng.prefill([r], [p])[0]) for r, p in zip(rids, prompts)], np.int32)
        rows = []
        for _ in range(6):
            toks, lg = eng.decode(rids, toks, return_logits=True)
            rows.append(lg.copy())
        outs.append(np.stack(rows))
        eng.close()
    assert np.array_equal(outs[0], outs[1])


def test_model_mid_batch_fused_layer_bitwise_equals_reference_sequence(built_libs):
    """Decode batches of 17..64 (bucket 32 here): the 7-launch layer (stacked qkv GEMM, qk-norm + RoPE + KV append in
    the attention launch, SwiGLU epilogue, split-K slice sum + add + RMSNorm in one launch) == the reference op
    sequence over the same GEMM kernels (PEGAINFER_MID_BATCH_FUSED=0), logits bit for bit over 5 steps."""
    from pegainfer_amd.qwen3 import Qwen3Engine
    meta = json.load(open(os.path.join(G, "qwen3_tiny_golden.json")))
    ck = os.path.join(G, "qwen3_tiny.safetensors")
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, 1024, int(n)).tolist() for n in rng.integers(3, 90, 20)]
    outs = []
    old = os.environ.get("PEGAINFER_MID_BATCH_FUSED")
    try:
        for flag in ("0", "1"):
            os.environ["PEGAINFER_MID_BATCH_FUSED"] = flag
            eng = Qwen3Engine(meta["config"], num_kv_pages=512, max_batch_size=32, decode_mode=1).load_safetensors(ck)
            rids = [eng.new_request() for _ in prompts]
            toks = np.array([int(eng.prefill([r], [p])[0]) for r, p in zip(rids, prompts)], np.int32)
            rows = []
            for _ in range(5):
                toks, lg = eng.decode(rids, toks, return_logits=True)
                rows.append(lg.copy())
            outs.append(np.stack(rows))
            eng.close()
    finally:
        if old is None:
            os.environ.pop("PEGAINFER_MID_BATCH_FUSED", None)
        else:
            os.environ["PEGAINFER_MID_BATCH_FUSED"] = old
    assert np.array_equal(outs[0], outs[1])


# ------------------------------------------------------------------ prefill launch fusions (round 3)
@pytest.mark.parametrize("T", [1024, 333, 40, 5, 4100])   # 4100 tokens: un-split 256 x 256 tiles + the one-pass add-then-norm rows kernel
def test_gemm_add_then_rms_norm_matches_the_three_calls(built_libs, T):
    """pegainfer_gemm_add_then_rms_norm (down_proj + residual add + the next layer's input RMSNorm; on split-K shapes
    the slice sum, the add and the norm are one launch) == gemm_cuda -> add_cuda -> rms_norm_batched_cuda, every bit of
    the new hidden state and of the normalised output (the norm sees the bf16-ROUNDED sum, prefill.rs:183 + :89)."""
    import torch
    import pegainfer_amd.ops as P
    from pegainfer_amd import ffi
    M, K = 2560, 9728
    rng = np.random.default_rng(T)
    W, X, A = rnd(rng, M, K, scale=0.03), rnd(rng, T, K, scale=1.0), rnd(rng, T, M, scale=2.0)
    g = bf16_round(1 + rnd(rng, M, scale=0.2))
    L, s = ffi.lib(), torch.cuda.current_stream().cuda_stream
    Wd, Xd, Ad, gd = to_dev(W), to_dev(X), to_dev(A), to_dev(g)
    # the three reference-named calls
    y = P.gemm(Wd, Xd)
    out_ref = torch.empty_like(Ad)
    assert L.add_cuda(Ad.data_ptr(), y.data_ptr(), out_ref.data_ptr(), T * M, s) == 0
    n_ref = torch.empty_like(Ad)
    P.rms_norm_batch_into(out_ref, gd, 1e-6, n_ref)
    # the fused entry point
    scratch, out, normed = torch.empty_like(Ad), torch.empty_like(Ad), torch.empty_like(Ad)
    assert L.pegainfer_gemm_add_then_rms_norm(Wd.data_ptr(), Xd.data_ptr(), scratch.data_ptr(), Ad.data_ptr(), out.data_ptr(),
                                              gd.data_ptr(), normed.data_ptr(), M, T, K, 1e-6, s) == 0
    assert np.array_equal(bf16_bits(from_dev(out)), bf16_bits(from_dev(out_ref)))
    assert np.array_equal(bf16_bits(from_dev(normed)), bf16_bits(from_dev(n_ref)))
    # and against the oracle's op sequence (tolerance of the GEMM summation order)
    ref = O.rms_norm(O.add(A, O.gemm(W, X)), g, 1e-6)
    assert np.abs(from_dev(normed) - ref).max() <= 2.0 ** -5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("lens,starts", [([300], [0]), ([70, 3, 129], [5, 40, 0])])
def test_qk_norm_rope_scatter_matches_the_two_calls(built_libs, lens, starts):
    """pegainfer_qk_norm_rope_scatter == qk_norm_rope_batched_decode_cuda + paged_kv_scatter_cuda: q and k in place and
    every byte of the cache (32 / 8 heads x 128, ragged multi-request batch with chunked continuations)."""
    import torch
    from pegainfer_amd import ffi
    from test_gpu_ops import make_paged
    rng = np.random.default_rng(sum(lens))
    Hq, Hkv, D = 32, 8, 128
    tot = [s + n for s, n in zip(starts, lens)]
    lay, kv, pages, indptr, last = make_paged(rng, len(lens), tot, Hkv=Hkv, D=D)
    T = sum(lens)
    q, k, v = rnd(rng, T, Hq * D), rnd(rng, T, Hkv * D), rnd(rng, T, Hkv * D)
    qw, kw = bf16_round(1 + rnd(rng, D, scale=0.1)), bf16_round(1 + rnd(rng, D, scale=0.1))
    cos, sin = O.precompute_rope(D, 512, 1e6)
    bidx = np.concatenate([np.full(n, i) for i, n in enumerate(lens)]).astype(np.int32)
    pos = np.concatenate([np.arange(s, s + n) for s, n in zip(starts, lens)]).astype(np.int32)
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device="cuda")
    L, s = ffi.lib(), torch.cuda.current_stream().cuda_stream
    keep = [to_dev(qw), to_dev(kw), to_dev(cos), to_dev(sin), i32(pos), i32(bidx), i32(pages), i32(indptr), i32(last), to_dev(v)]
    outs = []
    for fused in (False, True):
        qd, kd, kvd = to_dev(q), to_dev(k), to_dev(kv)
        if fused:
            assert L.pegainfer_qk_norm_rope_scatter(qd.data_ptr(), kd.data_ptr(), keep[9].data_ptr(), keep[0].data_ptr(),
                                                    keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                                                    keep[5].data_ptr(), kvd.data_ptr(), lay.layer_stride, lay.layer_stride + lay.kv_block_len,
                                                    keep[6].data_ptr(), keep[7].data_ptr(), Hq, Hkv, D, 16, lay.page_stride, T,
                                                    1e-6, s) == 0
        else:
            L.qk_norm_rope_batched_decode_cuda(qd.data_ptr(), kd.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(),
                                               keep[3].data_ptr(), keep[4].data_ptr(), Hq, Hkv, D, T, 1e-6, s)
            assert L.paged_kv_scatter_cuda(kvd.data_ptr(), lay.layer_stride, lay.layer_stride + lay.kv_block_len, keep[6].data_ptr(),
                                           keep[7].data_ptr(), keep[8].data_ptr(), kd.data_ptr(), keep[9].data_ptr(),
                                           keep[5].data_ptr(), keep[4].data_ptr(), T, Hkv, D, 16, lay.page_stride, Hkv * D, D, s) == 0
        outs.append((bf16_bits(from_dev(qd)), bf16_bits(from_dev(kd)), bf16_bits(from_dev(kvd))))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_model_prefill_fusions_bitwise_equal_reference_sequence(built_libs, monkeypatch):
    """Whole model: prefill logits and the KV the following decode steps read are bit-identical with the prefill launch
    fusions on (default) and off (PEGAINFER_PREFILL_FUSE=0 = the reference op sequence 1:1); single prompt, ragged batch
    and a chunked continuation."""
    from pegainfer_amd.qwen3 import Qwen3Engine
    meta = json.load(open(os.path.join(G, "qwen3_tiny_golden.json")))
    prompts = [c["prompt_tokens"] for c in meta["cases"][:3]]
    runs = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("PEGAINFER_PREFILL_FUSE", fuse)
        eng = Qwen3Engine(meta["config"], num_kv_pages=96, max_batch_size=4).load_safetensors(os.path.join(G, "qwen3_tiny.safetensors"))
        rids = [eng.new_request() for _ in prompts]
        tok, lg = eng.prefill(rids, prompts, return_logits=True)
        rows = [lg.copy()]
        for _ in range(3):
            tok, lg = eng.decode(rids, tok, return_logits=True)
            rows.append(lg.copy())
        r = eng.new_request()                                   # chunked: 20 tokens, then the rest
        eng.prefill([r], [prompts[2][:20]])
        _, lg2 = eng.prefill([r], [prompts[2][20:]], return_logits=True)
        rows.append(lg2.copy())
        # short prompts (<= 16 token columns): the 6-launch layer of round 4 (stacked q|k|v GEMV with the norm in its
        # prologue, kGemvRoundSum between layers, stacked norm + RoPE + scatter) - 1 / 5 / 16 tokens (dot2 and skinny
        # kernels), a ragged pair, a 9-token continuation of a cached prefix, then decode steps over that KV
        for group in ([prompts[0][:1]], [prompts[0]], [prompts[1]], [prompts[0], prompts[2][:9]], [prompts[2][:3], prompts[1][:4], prompts[0][:2]]):
            rs = [eng.new_request() for _ in group]
            tk, lg3 = eng.prefill(rs, group, return_logits=True)
            rows.append(lg3.copy())
            tk, lg3 = eng.decode(rs, tk, return_logits=True)
            rows.append(lg3.copy())
        r2 = eng.new_request()
        eng.prefill([r2], [prompts[2][:20]])
        tk, lg4 = eng.prefill([r2], [prompts[2][20:29]], return_logits=True)
        rows.append(lg4.copy())
        # unified step with few columns: a 6-token prompt arrives while r2 decodes
        r3 = eng.new_request()
        (tp, td), (lp, ld) = eng.unified_step([r3], [prompts[1][:6]], [r2], tk, return_logits=True)
        rows += [lp.copy(), ld.copy()]
        eng.close()
        runs.append(rows)
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
