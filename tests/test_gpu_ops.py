"""GPU parity: every HIP op through the C ABI vs the CPU oracle on seeded inputs, plus the
reference's own KATs (pegainfer-server/src/ops/tests.rs) run against the HIP path.

Bars (written per test): exact bits for byte movers (embedding, scatter, argmax/top-1);
<= 1 bf16 ulp for fp32-math-then-round elementwise ops (fp32 sums may differ in the last
bit from the oracle's float64 reductions, libm exp differs by <= 2 ulp fp32);
GEMM: |diff| <= 2^-7 * sum|w||x| envelope (bf16 output rounding of an fp32 accumulation).
"""
import numpy as np
import pytest

from conftest import bf16_ulp_diff, from_dev, to_dev
from oracle import ops as O
from oracle.bf16 import bf16_bits, bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(built_libs):
    import pegainfer_amd.ops as P
    return P


def rnd(rng, *shape, scale=1.0):
    return bf16_round((rng.standard_normal(shape) * scale).astype(np.float32))


def empty_like_dev(a):
    import torch
    return torch.empty(a.shape, dtype=torch.bfloat16, device="cuda")


# ------------------------------------------------------------------ KATs from tests.rs
def test_kat_gemv(P):  # tests.rs:51-77
    y = P.linear(to_dev(np.float32([1, 2, 3])), to_dev(np.float32([[1, 2, 3], [4, 5, 6]])))
    y = from_dev(y)
    assert abs(y[0] - 14.0) < 0.1 and abs(y[1] - 32.0) < 0.1


def test_kat_argmax(P):  # tests.rs:80-86
    assert P.argmax(to_dev(np.float32([1.0, 9.0, 3.0, 8.0]))) == 1


def test_kat_rms_norm_variants(P):  # tests.rs:89-170
    x, w = np.float32([1, 2, 3, 4]), np.float32([1, 1, 1, 1])
    out = empty_like_dev(x)
    P.rms_norm_into(to_dev(x), to_dev(w), 1e-6, out)
    assert np.abs(from_dev(out) - O.rms_norm_kat_reference(x, w, 1e-6)).max() <= 0.01
    hd, T = 260, 2
    idx = np.arange(hd * T)
    xb = bf16_round(((idx % 17).astype(np.float32) - 8.0) * 0.25).reshape(T, hd)
    wb = bf16_round(0.5 + (np.arange(hd) % 11).astype(np.float32) * 0.0625)
    out = empty_like_dev(xb)
    P.rms_norm_batch_into(to_dev(xb), to_dev(wb), 1e-6, out)
    got = from_dev(out)
    for r in range(T):
        assert np.abs(got[r] - O.rms_norm_kat_reference(xb[r], wb, 1e-6)).max() <= 0.02
    x = bf16_round(np.float32([-2.0, -0.5, 0.25, 1.5, 3.0, 0.75, -1.25]))
    w = bf16_round(np.float32([0.0, 0.5, -0.25, 0.125, 1.0, -0.5, 0.25]))
    out = empty_like_dev(x)
    P.rms_norm_offset_into(to_dev(x), to_dev(w), 1e-6, out)
    assert np.abs(from_dev(out) - O.rms_norm_kat_reference(x, w, 1e-6, True)).max() <= 0.02


def test_kat_embedding_variants(P):  # tests.rs:173-226, ops/embedding.rs:92-128
    import torch
    embed = np.arange(1, 13, dtype=np.float32).reshape(3, 4)
    e = to_dev(embed)
    out = torch.zeros(4, dtype=torch.bfloat16, device="cuda")
    P.embedding_decode_into(e, torch.tensor([1], dtype=torch.int32, device="cuda"), out)
    assert from_dev(out).tolist() == [5, 6, 7, 8]
    out = torch.zeros((2, 4), dtype=torch.bfloat16, device="cuda")
    P.embedding_batch(e, torch.tensor([2, 0], dtype=torch.int32, device="cuda"), out)
    assert from_dev(out).tolist() == [[9, 10, 11, 12], [1, 2, 3, 4]]
    out = torch.full((3, 4), 7.0, dtype=torch.bfloat16, device="cuda")
    P.embedding_batch_vocab_shard(to_dev(embed[1:3]), torch.tensor([0, 1, 2], dtype=torch.int32, device="cuda"),
                                  out, 1)
    assert from_dev(out).tolist() == [[0, 0, 0, 0], [5, 6, 7, 8], [9, 10, 11, 12]]


def test_kat_gpu_sample(P):  # tests.rs:229-305
    import torch
    logits = to_dev(np.float32([1.0, 2.0, 10.0, 1.5, 0.5]))
    probs = torch.zeros(5, dtype=torch.float32, device="cuda")
    top1 = torch.zeros(1, dtype=torch.bfloat16, device="cuda")
    rows = torch.zeros(P.flashinfer_topk_row_states_bytes(), dtype=torch.uint8, device="cuda")
    assert P.gpu_sample(logits, probs, top1, rows, 0.01, -1, 1.0, 0.5) == 2
    assert P.gpu_sample(logits, probs, top1, rows, 1.0, -1, 1.0, 0.0) < 5
    assert P.gpu_sample(logits, probs, top1, rows, 1.0, 1, 1.0, 0.5) == 2
    assert float(top1.float().item()) == 10.0
    assert int(rows[:16].sum().item()) == 0  # scratch left zeroed (ops/sampling.rs:7 contract)


# ------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("n", [1, 7, 8, 2560, 2560 * 3 + 5])
def test_add(P, n):
    rng = np.random.default_rng(n)
    a, b = rnd(rng, n, scale=3), rnd(rng, n, scale=3)
    out = empty_like_dev(a)
    P.add_batch_into(to_dev(a), to_dev(b), out)
    assert np.array_equal(bf16_bits(from_dev(out)), bf16_bits(O.add(a, b)))  # exact: one fp32 add, one RNE


@pytest.mark.parametrize("I,bs", [(9728, 1), (9728, 3), (520, 2), (37, 2)])
def test_silu_mul_fused(P, I, bs):
    rng = np.random.default_rng(I + bs)
    gu = rnd(rng, bs, 2 * I, scale=2.5)
    out = empty_like_dev(gu[:, :I])
    P.silu_mul_fused_batch_into(to_dev(gu), out)
    assert bf16_ulp_diff(from_dev(out), O.silu_mul_fused(gu, I)) <= 1


def test_silu_mul_rounded_variant(P):  # elementwise.cu:28-42
    rng = np.random.default_rng(5)
    g, u = rnd(rng, 4, 9216, scale=2.5), rnd(rng, 4, 9216, scale=2.5)
    out = empty_like_dev(g)
    P.silu_mul_batch_into(to_dev(g), to_dev(u), out)
    assert bf16_ulp_diff(from_dev(out), O.silu_mul(g, u)) <= 1


@pytest.mark.parametrize("hidden,T", [(2560, 1), (2560, 9), (100, 3)])
def test_embedding_batched(P, hidden, T):
    import torch
    rng = np.random.default_rng(hidden + T)
    embed = rnd(rng, 300, hidden)
    ids = rng.integers(0, 300, T)
    out = torch.zeros((T, hidden), dtype=torch.bfloat16, device="cuda")
    P.embedding_batch(to_dev(embed), torch.tensor(ids, dtype=torch.int32, device="cuda"), out)
    assert np.array_equal(from_dev(out), O.embedding_batched(embed, ids))


def test_casts(P):
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(1)
    a = rnd(rng, 5000, scale=10)
    d = to_dev(a)
    f = torch.zeros(5000, dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert ffi.lib().deepseek_bf16_to_f32_cuda(d.data_ptr(), f.data_ptr(), 5000, s) == 0
    assert np.array_equal(f.cpu().numpy(), a)
    x = (rng.standard_normal(5000) * 10).astype(np.float32)
    fx = torch.from_numpy(x).cuda()
    assert ffi.lib().deepseek_f32_to_bf16_cuda(fx.data_ptr(), d.data_ptr(), 5000, s) == 0
    assert np.array_equal(bf16_bits(from_dev(d)), bf16_bits(x))


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("hidden,T,offset", [(2560, 1, False), (2560, 64, False), (4096, 5, True), (260, 2, False)])
def test_rms_norm(P, hidden, T, offset):
    rng = np.random.default_rng(hidden + T)
    x, w = rnd(rng, T, hidden, scale=2), rnd(rng, hidden, scale=0.5) + (0 if offset else 1)
    w = bf16_round(w)
    out = empty_like_dev(x)
    (P.rms_norm_batch_offset_into if offset else P.rms_norm_batch_into)(to_dev(x), to_dev(w), 1e-6, out)
    assert bf16_ulp_diff(from_dev(out), O.rms_norm(x, w, 1e-6, offset)) <= 1


@pytest.mark.parametrize("hidden,T", [(2560, 1), (2560, 33), (260, 2)])
def test_fused_add_rms_norm(P, hidden, T):
    rng = np.random.default_rng(hidden * T)
    h, r, w = rnd(rng, T, hidden, scale=4), rnd(rng, T, hidden, scale=0.7), bf16_round(1 + rnd(rng, hidden, scale=0.2))
    hd, out = to_dev(h), empty_like_dev(h)
    P.fused_add_rms_norm_batch_into(hd, to_dev(r), to_dev(w), 1e-6, out)
    nh, no = O.fused_add_rms_norm(h, r, w, 1e-6)
    assert np.array_equal(bf16_bits(from_dev(hd)), bf16_bits(nh))      # residual stream: exact
    assert bf16_ulp_diff(from_dev(out), no) <= 1


@pytest.mark.parametrize("hidden", [2560, 4096, 1000 * 8])
def test_fused_add_rms_norm_long_prompt_rows_equal_the_two_pass_kernel(P, hidden):
    """From 256 rows on the fused add + RMSNorm keeps a row in registers between its two passes (round 6: one memory pass, hidden <=
    2560 / 4096; wider rows stay on the two-pass kernel).  Same canonical summation order: every bit of the new hidden state and of
    the normalised output equals the two-pass kernel's, which the same call takes below 256 rows."""
    rng = np.random.default_rng(hidden)
    T = 300
    h, r, w = rnd(rng, T, hidden, scale=4), rnd(rng, T, hidden, scale=0.7), bf16_round(1 + rnd(rng, hidden, scale=0.2))
    hd, out = to_dev(h), empty_like_dev(h)
    P.fused_add_rms_norm_batch_into(hd, to_dev(r), to_dev(w), 1e-6, out)                    # 300 rows: the one-pass rows kernel
    wd = to_dev(w)
    for lo in range(0, T, 100):                                                             # 100 rows: the two-pass kernel
        h2, o2 = to_dev(h[lo:lo + 100]), empty_like_dev(h[lo:lo + 100])
        P.fused_add_rms_norm_batch_into(h2, to_dev(r[lo:lo + 100]), wd, 1e-6, o2)
        assert np.array_equal(bf16_bits(from_dev(hd)[lo:lo + 100]), bf16_bits(from_dev(h2)))
        assert np.array_equal(bf16_bits(from_dev(out)[lo:lo + 100]), bf16_bits(from_dev(o2)))


def test_rms_norm_gated(P):  # norm.cu:17-61 (Qwen3.5 linear-attention output)
    import torch
    rng = np.random.default_rng(3)
    T, heads, hd = 3, 32, 128
    x, g = rnd(rng, T, heads * hd, scale=2), rnd(rng, T, heads * hd, scale=2)
    w = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    out = empty_like_dev(x)
    P.rms_norm_gated_batch_into(to_dev(x), torch.from_numpy(w).cuda(), to_dev(g), out, heads, hd, 1e-6)
    assert bf16_ulp_diff(from_dev(out), O.rms_norm_gated(x, w, g, hd, 1e-6)) <= 1


# ------------------------------------------------------------------ qk norm + rope
@pytest.mark.parametrize("T,decode", [(1, True), (7, True), (40, False)])
def test_qk_norm_rope(P, T, decode):
    import torch
    rng = np.random.default_rng(T)
    Hq, Hkv, D = 32, 8, 128
    q, k = rnd(rng, T, Hq * D, scale=2), rnd(rng, T, Hkv * D, scale=2)
    qw, kw = bf16_round(1 + rnd(rng, D, scale=0.2)), bf16_round(1 + rnd(rng, D, scale=0.2))
    cos, sin = O.precompute_rope(D, 4096, 1e6)
    qd, kd = to_dev(q), to_dev(k)
    if decode:
        pos = rng.integers(0, 4000, T)
        P.qk_norm_rope_batch_decode_into(qd, kd, to_dev(qw), to_dev(kw), to_dev(cos), to_dev(sin),
                                         torch.tensor(pos, dtype=torch.int32, device="cuda"), Hq, Hkv, D, 1e-6)
    else:
        pos = np.arange(17, 17 + T)
        P.prefill_qk_norm_rope_only(qd, kd, to_dev(qw), to_dev(kw), to_dev(cos), to_dev(sin), Hq, Hkv, D, 17, 1e-6)
    eq, ek = O.qk_norm_rope(q, k, qw, kw, cos, sin, pos, Hq, Hkv, D, 1e-6)
    # three chained roundings: allow 2 ulp, require > 99.5 % exact
    assert bf16_ulp_diff(from_dev(qd), eq) <= 2 and bf16_ulp_diff(from_dev(kd), ek) <= 2
    exact = (bf16_bits(from_dev(qd)) == bf16_bits(eq)).mean()
    assert exact > 0.995, exact


# ------------------------------------------------------------------ GEMM
def gemm_check(P, M, T, K, seed=0):
    rng = np.random.default_rng(seed + M + T + K)
    W, X = rnd(rng, M, K, scale=0.05), rnd(rng, T, K, scale=1.0)
    got = from_dev(P.gemm(to_dev(W), to_dev(X)))
    ref64 = X.astype(np.float64) @ W.astype(np.float64).T
    env = (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64).T)
    tol = env * 2.0 ** -7 * 0.05 + np.abs(ref64) * 2.0 ** -8 + 1e-6   # bf16 store + fp32 accumulation slack
    assert np.all(np.abs(got - ref64) <= tol), float((np.abs(got - ref64) / tol).max())
    return got


@pytest.mark.parametrize("M,T,K", [(4096, 1, 2560), (1024, 1, 2560), (2560, 1, 4096), (2560, 1, 9728),
                                   (19456, 1, 2560), (512, 2, 256), (2560, 3, 9728), (1000, 5, 2560),
                                   (2560, 8, 4096), (2560, 13, 2560), (2560, 16, 9728), (19456, 24, 2560),
                                   (2560, 33, 9728), (1000, 64, 2560), (2560, 64, 4096), (512, 7, 264),
                                   (256, 17, 512), (1024, 100, 2560), (2560, 300, 4096), (300, 130, 192),
                                   (6, 3, 5), (64, 20, 72),
                                   # 128 x 256-tile kernel (gemm256.h): K-split plans, the un-split round, ragged M / T / K tiles
                                   (2560, 1024, 4096), (6144, 1024, 2560), (1000, 600, 2560), (2500, 515, 4160),
                                   (5004, 520, 1024), (10000, 530, 1024), (2560, 512, 128),
                                   # weight-streaming kernel (gemm_stream.h): 80 / 64-row tiles at 17..64 columns (64-token
                                   # tile) and 65..128 (128-token tile), last row tile ragged / almost empty
                                   (19456, 17, 2560), (19460, 40, 2560), (19460, 100, 2560), (20000, 128, 2560),
                                   (12292, 64, 1024), (16388, 96, 512),
                                   # its K-split form (o_proj / down_proj: 8 slices x 32 row tiles of 80), 64- and 128-token tile,
                                   # ragged last row tile, a short last slice
                                   (2560, 100, 9728), (2560, 128, 4096), (2564, 70, 9728), (2568, 40, 4160),
                                   # 32-row tiles x one 128-token tile (the stacked qkv at 65..128 tokens), Qwen3-8B's K, ragged M
                                   (6144, 100, 2560), (6144, 128, 4096), (6148, 65, 2560)])
def test_gemm_shapes(P, M, T, K):
    gemm_check(P, M, T, K)


def test_gemm_lm_head_shape(P):
    gemm_check(P, 151936, 1, 2560)


@pytest.mark.parametrize("T,K,ms", [(1024, 2560, (4096, 1024, 1024)), (200, 512, (256, 64, 192)),
                                    (65, 128, (132, 4, 8)), (40, 256, (128, 64, 64)), (16, 256, (128, 64, 64))])
def test_gemm_split3_equals_three_gemms(P, T, K, ms):
    """Stacked q/k/v projection in one launch == the row ranges of gemm_cuda over the stacked matrix, bit for bit, and
    == the three reference-ABI gemm_cuda calls (prefill.rs:120-129) bit for bit whenever those take the same K-split
    plan as the stacked matrix (everything here except the 1024-token case, where the q and k / v projections on their
    own have few enough tiles to be split-K shapes: GEMM tolerance); also pins the LDS-DMA tiled GEMM against the oracle
    GEMM."""
    import torch
    rng = np.random.default_rng(5)
    W, X = rnd(rng, sum(ms), K, scale=0.05), rnd(rng, T, K)
    Wd, Xd = to_dev(W), to_dev(X)
    outs = [torch.empty((T, m), dtype=torch.bfloat16, device=Xd.device) for m in ms]
    P.gemm_split3_into(Wd, Xd, *outs)
    full = bf16_bits(from_dev(P.gemm(Wd, Xd)))
    r0 = 0
    for m, o in zip(ms, outs):
        assert np.array_equal(bf16_bits(from_dev(o)), full[:, r0:r0 + m]), (ms, m)
        sep = from_dev(P.gemm(Wd[r0:r0 + m], Xd))
        if T == 1024:
            assert np.abs(from_dev(o) - sep).max() <= 2.0 ** -6 * max(1.0, np.abs(sep).max())
        else:
            assert np.array_equal(bf16_bits(from_dev(o)), bf16_bits(sep)), (ms, m)
        ref = O.gemm(W[r0:r0 + m], X)
        assert np.abs(from_dev(o) - ref).max() <= 2.0 ** -6 * max(1.0, np.abs(ref).max())
        r0 += m


@pytest.mark.parametrize("T,I,K", [(1024, 9728, 2560), (130, 512, 256), (40, 9728, 2560), (40, 512, 256), (100, 1000, 128),
                                   (7, 9728, 2560), (256, 9728, 2560), (200, 9730 - 2, 2560), (100, 8200, 1024),
                                   # weight-streaming kernel, SwiGLU form (40 + 40 rows as 3 + 3 blocks): ragged last tile,
                                   # both tile widths, the Qwen3-8B (48 + 48 rows) and Qwen3.5 (36 + 36) column counts
                                   (64, 9732, 2560), (128, 9728, 2560), (17, 12288, 1024), (90, 9216, 512)])
def test_gemm_silu_epilogue_equals_gemm_then_silu_mul_fused(P, T, I, K):
    """SwiGLU in the tiled GEMM's epilogue (64 gate rows + their 64 up rows per workgroup) == gemm_cuda +
    silu_mul_fused_cuda, bit for bit, incl. partial row tiles, the mid-batch (17..64) family and the small-shape
    fallback; and within 1 bf16 ulp + GEMM tolerance of the oracle."""
    import torch
    rng = np.random.default_rng(T + I)
    W, X = rnd(rng, 2 * I, K, scale=0.06), rnd(rng, T, K, scale=1.5)
    Wd, Xd = to_dev(W), to_dev(X)
    ref = torch.zeros((T, I), dtype=torch.bfloat16, device="cuda")
    P.silu_mul_fused_batch_into(P.gemm(Wd, Xd), ref)
    out = torch.zeros_like(ref)
    P.gemm_silu_into(Wd, Xd, out)
    from pegainfer_amd import ffi
    kind = np.zeros(3, np.int32)
    ffi.lib().pegainfer_debug_gemm_route(0, T, K, I, kind.ctypes.data)
    if kind[0] == 258:
        # round 6: the stream-K route sums K in a fixed order of ITS OWN per (tile count, K, CU count) - and a 128 + 128-row SwiGLU
        # tile and a 256-row plain tile cut K at different units - so fused == pair holds to the GEMM tolerance there, not to
        # the bit (cross-route bit identity of prefill GEMMs given up; tests/test_gpu_real_dims.py pins the
        # route against float64 and its own rerun determinism)
        d = np.abs(from_dev(out) - from_dev(ref))
        assert d.max() <= 2.0 ** -6 * max(1.0, np.abs(from_dev(ref)).max()) and (d > 0).mean() < 0.2
    else:
        assert np.array_equal(bf16_bits(from_dev(out)), bf16_bits(from_dev(ref)))
    exp = O.silu_mul_fused(O.gemm(W, X), I)
    assert np.abs(from_dev(out) - exp).max() <= 2.0 ** -6 * max(1.0, np.abs(exp).max())


@pytest.mark.parametrize("M,T,K,silu", [(4096, 9000, 256, False),     # T > M: row tile fastest, 16 row tiles in chunks of 8
                                        (9216 + 40, 4100, 256, False),  # M > T: token tile fastest, 17 token tiles in chunks of 8 + 8 + 1, ragged rows
                                        (12288, 4100, 256, True)])      # the SwiGLU form: 48 (128 + 128)-row tiles x 17 token tiles
def test_large_grid_tile_orders_cover_every_tile_once(P, M, T, K, silu):
    """Round 6: launches of more than two rounds of 256 x 256 tiles walk them in an XCD-aware CHUNKED order (and row-tile-fastest
    when T > M) - a pure permutation of the work items.  Every output element is written (the buffer starts as NaN) and equals a
    float32 reference within the GEMM tolerance; the ragged last chunk / last row tile included."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(M + T)
    W = (torch.randn(M, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    X = torch.randn(T, K, generator=g).to(torch.bfloat16).cuda()
    from pegainfer_amd import ffi
    kind = np.zeros(3, np.int32)
    ffi.lib().pegainfer_debug_gemm_route(0 if silu else M, T, K, M // 2 if silu else 0, kind.ctypes.data)
    assert kind[0] in (256, 257), kind      # the 256 x 256 kernel (257: with its thin tail on the feeder kernel)
    ref = X.float() @ W.float().T
    if silu:
        I = M // 2
        out = torch.full((T, I), float("nan"), dtype=torch.bfloat16, device="cuda")
        P.gemm_silu_into(W, X, out)
        gr, ur = ref[:, :I].to(torch.bfloat16).float(), ref[:, I:].to(torch.bfloat16).float()
        ref = torch.nn.functional.silu(gr) * ur
    else:
        out = torch.full((T, M), float("nan"), dtype=torch.bfloat16, device="cuda")
        P.gemm_into(W, X, out)
    assert not torch.isnan(out).any()
    err = (out.float() - ref).abs().max().item()
    assert err <= 2.0 ** -6 * max(1.0, ref.abs().max().item()), err


def test_gemm_split3_mid_batch_matches_oracle_and_full_gemm(P):
    """17..64 columns, stacked q|k|v of 6144 rows (the mid-batch decode path): the three outputs are the row ranges of
    ONE tiled GEMM - bit-identical to gemm_cuda over the stacked matrix, within the GEMM tolerance of the oracle."""
    import torch
    rng = np.random.default_rng(6)
    T, K, ms = 48, 2560, (4096, 1024, 1024)
    W, X = rnd(rng, sum(ms), K, scale=0.05), rnd(rng, T, K)
    Wd, Xd = to_dev(W), to_dev(X)
    outs = [torch.empty((T, m), dtype=torch.bfloat16, device=Xd.device) for m in ms]
    P.gemm_split3_into(Wd, Xd, *outs)
    full = from_dev(P.gemm(Wd, Xd))
    ref = O.gemm(W, X)
    r0 = 0
    for m, o in zip(ms, outs):
        assert np.array_equal(bf16_bits(from_dev(o)), bf16_bits(full[:, r0:r0 + m]))
        assert np.abs(from_dev(o) - ref[:, r0:r0 + m]).max() <= 2.0 ** -6 * max(1.0, np.abs(ref).max())
        r0 += m


@pytest.mark.parametrize("ms", [(8192, 4096, 32, 32), (512, 512), (64, 8, 4, 12)])
def test_gemm_split_n_outputs_equals_separate_gemms(P, ms):
    """2- and 4-output stacked GEMM (Qwen3.5 gate|up and qkv|z|b|a) == separate gemm_cuda calls, bit for bit."""
    import ctypes
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(9)
    T, K = 130, 256
    W, X = rnd(rng, sum(ms), K, scale=0.05), rnd(rng, T, K)
    Wd, Xd = to_dev(W), to_dev(X)
    outs = [torch.empty((T, m), dtype=torch.bfloat16, device=Xd.device) for m in ms]
    ys = (ctypes.c_void_p * len(ms))(*[o.data_ptr() for o in outs])
    mm = (ctypes.c_int32 * len(ms))(*ms)
    assert ffi.lib().pegainfer_gemm_split(Wd.data_ptr(), Xd.data_ptr(), len(ms), ys, mm, T, K,
                                          torch.cuda.current_stream().cuda_stream) == 0
    r0 = 0
    for m, o in zip(ms, outs):
        assert np.array_equal(bf16_bits(from_dev(o)), bf16_bits(from_dev(P.gemm(Wd[r0:r0 + m], Xd)))), (ms, m)
        r0 += m


@pytest.mark.parametrize("T,M,K", [(1024, 2560, 9728), (300, 2560, 4096), (40, 2560, 4096), (8, 2560, 4096), (100, 512, 256)])
def test_gemm_add_equals_gemm_then_add(P, T, M, K):
    """down_proj + residual add in one call == gemm_cuda + add_cuda, bit for bit, on the split-K shapes (prefill with
    fewer 128x128 tiles than CUs, decode batches of 17..64) and the fallback shapes; out aliasing a is allowed."""
    rng = np.random.default_rng(T + K)
    W, X, A = rnd(rng, M, K, scale=0.05), rnd(rng, T, K), rnd(rng, T, M)
    Wd, Xd = to_dev(W), to_dev(X)
    ref = empty_like_dev(A)
    P.add_batch_into(to_dev(A), P.gemm(Wd, Xd), ref)
    out = to_dev(A)
    P.gemm_add_into(Wd, Xd, out, out)
    assert np.array_equal(bf16_bits(from_dev(out)), bf16_bits(from_dev(ref)))
    exp = O.add(A, O.gemm(W, X))
    assert np.abs(from_dev(out) - exp).max() <= 2.0 ** -6 * max(1.0, np.abs(exp).max())


@pytest.mark.parametrize("T,M,K", [(40, 2560, 4096), (64, 2560, 9728), (17, 1000, 2560), (8, 2560, 4096), (100, 512, 256),
                                   (1024, 2560, 4096), (200, 2560, 9728)])
def test_gemm_add_rms_norm_equals_gemm_then_fused_add_rms_norm(P, T, M, K):
    """o_proj / down_proj + residual add + RMSNorm in one call == gemm_cuda + fused_add_rms_norm_batched_cuda, bit
    for bit (hidden AND normed), on the split-K shapes (one launch over the fp32 partials) and the fallback shapes;
    and within the GEMM tolerance of the oracle sequence."""
    import torch
    rng = np.random.default_rng(T + M)
    W, X, Hd, G = rnd(rng, M, K, scale=0.05), rnd(rng, T, K), rnd(rng, T, M), rnd(rng, M)
    Wd, Xd, Gd = to_dev(W), to_dev(X), to_dev(G)
    h_ref, n_ref = to_dev(Hd), torch.zeros((T, M), dtype=torch.bfloat16, device="cuda")
    P.fused_add_rms_norm_batch_into(h_ref, P.gemm(Wd, Xd), Gd, 1e-6, n_ref)
    h, n = to_dev(Hd), torch.zeros_like(n_ref)
    P.gemm_add_rms_norm_into(Wd, Xd, h, Gd, 1e-6, n)
    assert np.array_equal(bf16_bits(from_dev(h)), bf16_bits(from_dev(h_ref)))
    assert np.array_equal(bf16_bits(from_dev(n)), bf16_bits(from_dev(n_ref)))
    eh, en = O.fused_add_rms_norm(Hd, O.gemm(W, X), G, 1e-6)
    assert np.abs(from_dev(h) - eh).max() <= 2.0 ** -6 * max(1.0, np.abs(eh).max())
    assert np.abs(from_dev(n) - en).max() <= 2.0 ** -5 * max(1.0, np.abs(en).max())


def test_decode_gemm_batch_invariance_and_row_slices(P):
    """Decode GEMM: within a kernel family column t of a batched call == the same column in any other batch size,
    bit for bit.  Families by token columns: 1..2 (dot2 GEMV), 3..16 (skinny MFMA), 17..64 (tiled LDS-DMA GEMM, with
    K split over workgroups - by shape only - when the matrix has < 16384 rows).  A row slice of the fused matrix ==
    the fused call's rows in the decode families (reference relies on this: batch_decode.rs:160-163).  Across
    families the summation order differs: equal within the GEMM tolerance (the reference's cuBLAS also switches
    kernels with N)."""
    rng = np.random.default_rng(11)
    for K in (2560, 9728):
        W, X = rnd(rng, 6144, K, scale=0.05), rnd(rng, 64, K)
        Wd = to_dev(W)
        full = bf16_bits(from_dev(P.gemm(Wd, to_dev(X))))
        for T in (17, 32, 40):
            assert np.array_equal(bf16_bits(from_dev(P.gemm(Wd, to_dev(X[:T])))), full[:T]), (K, T)
        small = bf16_bits(from_dev(P.gemm(Wd[:2560], to_dev(X))))             # 20 row tiles: the split-K form
        for T in (17, 40):
            assert np.array_equal(bf16_bits(from_dev(P.gemm(Wd[:2560], to_dev(X[:T])))), small[:T]), (K, T)
        small16 = bf16_bits(from_dev(P.gemm(Wd[:2560], to_dev(X[:16]))))
        for T in (5, 9):
            assert np.array_equal(bf16_bits(from_dev(P.gemm(Wd[:2560], to_dev(X[:T])))), small16[:T]), (K, T)
        sixteen = bf16_bits(from_dev(P.gemm(Wd, to_dev(X[:16]))))
        for T in (3, 4, 8, 11):
            assert np.array_equal(bf16_bits(from_dev(P.gemm(Wd, to_dev(X[:T])))), sixteen[:T]), (K, T)
        two = bf16_bits(from_dev(P.gemm(Wd, to_dev(X[:2]))))
        assert np.array_equal(bf16_bits(from_dev(P.gemm(Wd, to_dev(X[:1])))), two[:1]), K
        for T in (1, 8):                                                # row slices, both decode families
            sl = bf16_bits(from_dev(P.gemm(Wd[4096:5120], to_dev(X[:T]))))
            ref = two[:1] if T == 1 else sixteen[:8]
            assert np.array_equal(sl, ref[:, 4096:5120])
        for lo, hi in ((2, 8), (16, 32)):                               # family boundaries: tolerance
            a, b = from_dev(P.gemm(Wd, to_dev(X[:lo]))), from_dev(P.gemm(Wd, to_dev(X[:hi])))[:lo]
            assert np.abs(a - b).max() <= 2.0 ** -6 * max(1.0, np.abs(b).max())


# ------------------------------------------------------------------ paged KV + attention
def make_paged(rng, bs, lens, Hkv=8, D=128, layers=2, ps=16, extra_pages=3):  # noqa: E501
    lay = O.PagedKvLayout(layers, Hkv, D, ps)
    need = [-(-n // ps) for n in lens]
    total = sum(need) + extra_pages
    perm = rng.permutation(total)
    pages, indptr, last, c = [], [0], [], 0
    for n, k in zip(lens, need):
        pages.extend(perm[c:c + k].tolist()); c += k
        indptr.append(len(pages))
        last.append(0 if n == 0 else ((n - 1) % ps) + 1)
    kv = bf16_round(rng.standard_normal(total * lay.page_stride).astype(np.float32))
    return lay, kv, np.int32(pages), np.int32(indptr), np.int32(last)


def test_paged_kv_scatter(P):
    import torch
    rng = np.random.default_rng(2)
    lens = [5, 16, 37]
    lay, kv, pages, indptr, last = make_paged(rng, 3, lens)
    nnz = sum(lens)
    k, v = rnd(rng, nnz, 8 * 128), rnd(rng, nnz, 8 * 128)
    bidx = np.concatenate([np.full(n, i) for i, n in enumerate(lens)]).astype(np.int32)
    pos = np.concatenate([np.arange(n) for n in lens]).astype(np.int32)
    kvd = to_dev(kv)
    L = P.PagedKvLayout(2, 8, 128, 16)
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device="cuda")
    P.paged_kv_scatter(kvd, L, 1, i32(pages), i32(indptr), i32(last), to_dev(k), to_dev(v), i32(bidx), i32(pos))
    O.paged_kv_scatter(kv, lay, 1, pages, indptr, k, v, bidx, pos)
    assert np.array_equal(bf16_bits(from_dev(kvd)), bf16_bits(kv))  # exact bytes


def attn_tol(ref):
    return 2.0 ** -7 * np.abs(ref).max() + 1e-3   # bf16 output rounding of O(1) values + p*V accumulation


@pytest.mark.parametrize("lens", [[1], [16], [17], [300], [5, 64, 1, 130], [1024]])
def test_paged_decode_attention(P, lens):
    import torch
    rng = np.random.default_rng(sum(lens))
    bs = len(lens)
    lay, kv, pages, indptr, last = make_paged(rng, bs, lens)
    q = rnd(rng, bs, 32 * 128)
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device="cuda")
    out = torch.zeros((bs, 32 * 128), dtype=torch.bfloat16, device="cuda")
    from pegainfer_amd import ffi
    sm = 1.0 / np.sqrt(128.0)
    L = P.PagedKvLayout(2, 8, 128, 16)
    qd = to_dev(q)
    rc = ffi.lib().paged_attention_decode_cuda(
        qd.data_ptr(), out.data_ptr(), (kvd := to_dev(kv)).data_ptr(), L.layer_stride,
        L.layer_stride + L.kv_block_len, (a := i32(pages)).data_ptr(), (b := i32(indptr)).data_ptr(),
        (c := i32(last)).data_ptr(), (d := i32(np.arange(bs))).data_ptr(), (e := i32(np.zeros(bs))).data_ptr(),
        (f := i32(lens)).data_ptr(), 32, 8, 128, 16, bs, L.page_stride, sm, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = O.paged_attention_decode(q, kv, lay, 1, pages, indptr, last, 32, sm)
    assert np.abs(from_dev(out) - ref).max() <= attn_tol(ref)


@pytest.mark.parametrize("lens,padded", [([1024], 1), ([1500, 40], 2), ([5000], 1)])
def test_split_kv_decode_equals_non_partition(P, lens, padded):
    """split-KV == non-partition within bf16 rounding (reference sanity check, model-crate.md:205),
    using the reference's own plan (batch_decode_buffers.rs:229-279) incl. masked padding slots."""
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(len(lens) + lens[0])
    bs = len(lens)
    lay, kv, pages, indptr, last = make_paged(rng, bs, lens)
    # padding request slots point at one page with seq_len 1 (kv_pool.rs:60-63)
    for _ in range(bs, padded):
        pages = np.append(pages, 0); indptr = np.append(indptr, len(pages)); last = np.append(last, 1)
    q = rnd(rng, padded, 32 * 128)
    plan = O.split_kv_plan(lens, padded)
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device="cuda")
    L = P.PagedKvLayout(2, 8, 128, 16)
    sm = 1.0 / np.sqrt(128.0)
    slots = plan["padded_slots"]
    out = torch.zeros((padded, 32 * 128), dtype=torch.bfloat16, device="cuda")
    tmp_v = torch.zeros(slots * 32 * 128, dtype=torch.bfloat16, device="cuda")
    tmp_s = torch.zeros(slots * 32, dtype=torch.float32, device="cuda")
    keep = [to_dev(q), to_dev(kv), i32(pages), i32(indptr), i32(last), i32(plan["request_indices"]),
            i32(plan["kv_tile_indices"]), i32([plan["kv_chunk_size"]]), i32(plan["o_indptr"]),
            torch.tensor(plan["block_valid_mask"], dtype=torch.uint8, device="cuda")]
    rc = ffi.lib().paged_attention_decode_split_kv_cuda(
        keep[0].data_ptr(), out.data_ptr(), keep[1].data_ptr(), 0, L.kv_block_len, keep[2].data_ptr(),
        keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(), keep[7].data_ptr(),
        keep[8].data_ptr(), keep[9].data_ptr(), tmp_v.data_ptr(), tmp_s.data_ptr(), 32, 8, 128, 16, padded, slots,
        L.page_stride, sm, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = O.paged_attention_decode(q[:bs], kv, lay, 0, pages, indptr, last, 32, sm)
    got = from_dev(out)[:bs]
    assert np.abs(got - ref).max() <= 2 * attn_tol(ref)   # partials are bf16-rounded before the merge
    spl = O.paged_attention_decode_split_kv(q[:bs], kv, lay, 0, pages, indptr, last, plan["request_indices"],
                                            plan["kv_tile_indices"], plan["kv_chunk_size"], plan["o_indptr"],
                                            plan["block_valid_mask"], 32, sm)
    assert np.abs(got - spl).max() <= attn_tol(ref)


@pytest.mark.parametrize("seq_lens,starts,tile", [([5], [0], 0), ([70], [0], 64), ([64], [0], 128), ([33, 1, 100], [0, 0, 0], 64),
                                                  ([20], [50], 64), ([200, 17], [0, 30], 0), ([3], [0], 16)])
def test_batch_prefill_paged(P, seq_lens, starts, tile):
    """causal varlen prefill over paged KV incl. chunked prefill (start_pos > 0), every CTA tile size."""
    import torch
    rng = np.random.default_rng(sum(seq_lens) + tile)
    lens = [s + n for s, n in zip(starts, seq_lens)]
    lay, kv, pages, indptr, last = make_paged(rng, len(lens), lens)
    T = sum(seq_lens)
    q = rnd(rng, T, 32 * 128)
    page_lists = [pages[indptr[i]:indptr[i + 1]].tolist() for i in range(len(lens))]
    plan = P.PrefillPagedPlan(page_lists, last.tolist(), starts, seq_lens, 32, 8, 128, tile)
    oplan = O.prefill_paged_plan(page_lists, last.tolist(), starts, seq_lens, 32, 8, 128, tile)
    for key in ("page_indices", "page_indptr", "last_page_len", "batch_indices", "positions", "q_indptr",
                "request_indices", "qo_tile_indices", "kv_tile_indices", "kv_chunk_size"):
        assert np.array_equal(getattr(plan, key + "_d").cpu().numpy(), oplan[key]), key   # integer plan: exact
    assert (plan.num_tiles, plan.cta_tile_q) == (oplan["num_tiles"], oplan["cta_tile_q"])
    from pegainfer_amd import ffi
    L = P.PagedKvLayout(2, 8, 128, 16)
    out = torch.zeros((T, 32 * 128), dtype=torch.bfloat16, device="cuda")
    qd, kvd = to_dev(q), to_dev(kv)
    sm = 1.0 / np.sqrt(128.0)
    rc = ffi.lib().batch_prefill_paged_cuda_with_cta_tile_q(
        qd.data_ptr(), out.data_ptr(), kvd.data_ptr(), L.layer_stride, L.layer_stride + L.kv_block_len,
        plan.page_indices_d.data_ptr(), plan.page_indptr_d.data_ptr(), plan.last_page_len_d.data_ptr(),
        plan.q_indptr_d.data_ptr(), plan.request_indices_d.data_ptr(), plan.qo_tile_indices_d.data_ptr(),
        plan.kv_tile_indices_d.data_ptr(), plan.kv_chunk_size_d.data_ptr(), plan.total_num_rows_d.data_ptr(),
        32, 8, 128, 16, T, len(lens), plan.num_tiles, L.page_stride, sm, plan.cta_tile_q,
        torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = O.batch_prefill_paged(q, kv, lay, 1, pages, indptr, last, oplan["q_indptr"], 32, sm)
    assert np.abs(from_dev(out) - ref).max() <= 2 * attn_tol(ref)   # P is bf16 before the PV MFMA


def test_prefill_invalid_tile_override(P):
    from pegainfer_amd import ffi
    assert ffi.lib().batch_prefill_paged_num_tiles_with_cta_tile_q(100, 32, 8, 128, 7) == -1
    assert ffi.lib().batch_prefill_cta_tile_q_with_override(100, 32, 8, 128, 7) == 0


# ------------------------------------------------------------------ sampling
@pytest.mark.parametrize("rows,n", [(1, 151936), (64, 151936), (7, 1001)])
def test_batched_top1_every_row_lowest_index_ties_and_self_resetting_state(P, rows, n):
    """Greedy token of all batch columns in one launch (64 workgroups per row meet through two atomics per row):
    == the oracle argmax per row incl. exact ties, over 25 back-to-back launches on the same scratch, which the
    kernel must leave zero (no ordering fence around the ticket: the atomics alone carry the hand-off)."""
    import torch
    rng = np.random.default_rng(rows + n)
    x = rnd(rng, rows, n, scale=3)
    for r in range(rows):
        pos = sorted(rng.choice(n, size=3, replace=False).tolist())
        x[r, pos] = x[r].max() + 1
    xd = to_dev(x)
    want = np.array([O.argmax(x[r]) for r in range(rows)])
    state = torch.zeros(16 * rows, dtype=torch.uint8, device="cuda")
    outs = [P.batched_top1(xd, state) for _ in range(25)]
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), want)
    assert int(state.sum().item()) == 0


@pytest.mark.parametrize("n", [5, 1000, 151936])
def test_argmax_and_top1_lowest_index_ties(P, n):
    import torch
    rng = np.random.default_rng(n)
    x = rnd(rng, n, scale=3)
    mx = x.max() + 1
    pos = sorted(rng.choice(n, size=min(3, n), replace=False).tolist())
    x[pos] = mx                                             # exact ties: lowest index must win
    xd = to_dev(x)
    assert P.argmax(xd) == pos[0] == O.argmax(x)
    rows = torch.zeros(P.flashinfer_topk_row_states_bytes(), dtype=torch.uint8, device="cuda")
    top1 = torch.zeros(1, dtype=torch.bfloat16, device="cuda")
    probs = torch.zeros(n, dtype=torch.float32, device="cuda")
    for _ in range(3):                                      # scratch is reused across calls
        assert P.gpu_sample(xd, probs, top1, rows, 0.0, -1, 1.0, 0.1) == pos[0]
    assert float(top1.float().item()) == float(mx)


def test_sampling_distribution(P):
    """Random branch: parity is distributional only (FlashInfer's Philox stream is un-vendored).
    Every draw must fall in the joint top-k/top-p support; empirical frequencies must match the
    renormalised probabilities (chi-square-style bound); probs scratch == fp32 softmax."""
    import torch
    rng = np.random.default_rng(0)
    n = 4096
    logits = rnd(rng, n, scale=2.0)
    ld = to_dev(logits)
    rows = torch.zeros(P.flashinfer_topk_row_states_bytes(), dtype=torch.uint8, device="cuda")
    top1 = torch.zeros(1, dtype=torch.bfloat16, device="cuda")
    probs = torch.zeros(n, dtype=torch.float32, device="cuda")
    for (T, k, p) in [(0.8, 50, 0.95), (1.0, -1, 0.9), (0.7, 20, 1.0), (1.0, -1, 1.0)]:
        ref = O.logits_to_probs(logits, 1.0 / T)
        keep = O.top_k_top_p_support(ref, k, p)
        draws = np.array([P.gpu_sample(ld, probs, top1, rows, T, k, p, float(r)) for r in rng.random(600)])
        assert np.allclose(probs.cpu().numpy(), ref, rtol=2e-5, atol=1e-9)
        assert keep[draws].all(), (T, k, p)
        if keep.sum() <= 60:
            renorm = np.where(keep, ref, 0) / ref[keep].sum()
            freq = np.bincount(draws, minlength=n) / len(draws)
            assert np.abs(freq - renorm).max() < 0.08
