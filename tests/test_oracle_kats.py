"""Oracle vs the reference's own known-answer tests (CPU, no GPU).

Golden values restated from pegainfer-server/src/ops/tests.rs (line cited per test) and the
synthetic generators of pegainfer-server/benches/ops/common/mod.rs:54-82.
"""
import numpy as np

from oracle import ops
from oracle.bf16 import bf16_round, bf16_bits, bf16_from_bits


def bf(x):
    return bf16_round(np.asarray(x, dtype=np.float32))


def test_bf16_round_trip_and_rne():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.14159, 1e-30, 65504.0], dtype=np.float32)
    r = bf16_round(x)
    assert np.array_equal(bf16_from_bits(bf16_bits(r)), r)
    # 1 + 2^-8 is a tie between 1.0 and 1.0078125 -> even mantissa (1.0)
    assert r[1] == np.float32(1.0)
    # 1 + 3*2^-8 ties between 1.0078125 and 1.015625 -> even (1.015625)
    assert r[2] == np.float32(1.015625)


def test_gemv_kat():  # tests.rs:51-77
    y = ops.gemm(bf([[1, 2, 3], [4, 5, 6]]), bf([[1, 2, 3]]))
    assert abs(y[0, 0] - 14.0) < 0.1 and abs(y[0, 1] - 32.0) < 0.1


def test_argmax_kat():  # tests.rs:80-86
    assert ops.argmax(bf([1.0, 9.0, 3.0, 8.0])) == 1
    assert ops.argmax(bf([5.0, 9.0, 9.0, 8.0])) == 1  # lowest index wins ties (argmax.cu:18)


def test_rms_norm_kat():  # tests.rs:89-103, tol 0.01
    x, w = bf([1.0, 2.0, 3.0, 4.0]), bf([1.0, 1.0, 1.0, 1.0])
    got = ops.rms_norm(x, w, 1e-6)[0]
    assert np.abs(got - ops.rms_norm_kat_reference(x, w, 1e-6)).max() <= 0.01


def test_rms_norm_batch_multi_tile_kat():  # tests.rs:106-150, tol 0.02
    hd, T = 260, 2
    idx = np.arange(hd * T)
    x = bf(((idx % 17).astype(np.float32) - 8.0) * 0.25).reshape(T, hd)
    w = bf(0.5 + (np.arange(hd) % 11).astype(np.float32) * 0.0625)
    got = ops.rms_norm(x, w, 1e-6)
    for r in range(T):
        assert np.abs(got[r] - ops.rms_norm_kat_reference(x[r], w, 1e-6)).max() <= 0.02


def test_rms_norm_offset_kat():  # tests.rs:153-170, tol 0.02
    x = bf([-2.0, -0.5, 0.25, 1.5, 3.0, 0.75, -1.25])
    w = bf([0.0, 0.5, -0.25, 0.125, 1.0, -0.5, 0.25])
    got = ops.rms_norm(x, w, 1e-6, offset=True)[0]
    assert np.abs(got - ops.rms_norm_kat_reference(x, w, 1e-6, offset=True)).max() <= 0.02


def test_embedding_variants_kat():  # tests.rs:173-226
    embed = bf(np.arange(1, 13)).reshape(3, 4)
    assert np.array_equal(ops.embedding_batched(embed, [1])[0], bf([5, 6, 7, 8]))
    out = ops.embedding_batched(embed, [2, 0])
    assert out[0, 0] == 9 and out[0, 3] == 12 and out[1, 0] == 1 and out[1, 3] == 4
    # vocab shard masks to zero outside the shard (ops/embedding.rs:92-128)
    sh = ops.embedding_batched_vocab_shard(embed[1:3], [0, 1, 2], 1, 2)
    assert np.all(sh[0] == 0) and np.array_equal(sh[1], embed[1]) and np.array_equal(sh[2], embed[2])


def test_gpu_sample_kat_distribution():  # tests.rs:229-305
    logits = bf([1.0, 2.0, 10.0, 1.5, 0.5])
    p = ops.logits_to_probs(logits, 1.0 / 0.01)
    assert p.argmax() == 2 and p[2] > 0.999999          # T=0.01 -> idx 2
    assert ops.argmax(logits) == 2                       # top_k=1 -> greedy branch
    keep = ops.top_k_top_p_support(ops.logits_to_probs(logits, 1.0), 2, 1.0)
    assert keep.tolist() == [False, True, True, False, False]


def test_rope_table_matches_formula():  # weight_loader.rs:210-244
    cos, sin = ops.precompute_rope(128, 64, 1e6)
    assert cos.shape == (64, 128) and np.array_equal(cos[:, :64], cos[:, 64:])
    assert np.all(cos[0] == 1.0) and np.all(sin[0] == 0.0)
    f = np.float32(5) * (np.float32(1.0) / np.power(np.float32(1e6), np.float32(6.0 / 128.0), dtype=np.float32))
    assert cos[5, 3] == bf16_round(np.cos(f, dtype=np.float32))


def test_silu_variants_rounding_points():  # fused_proj.cu:57-62 vs elementwise.cu:36-41
    g = bf(np.linspace(-6, 6, 64)); u = bf(np.linspace(2, -2, 64))
    fused = ops.silu_mul_fused(np.concatenate([g, u])[None, :], 64)[0]
    two = ops.silu_mul(g, u)
    assert np.abs(fused - two).max() <= 0.0625 and not np.array_equal(fused, two)  # <= 1 bf16 ulp at |x|<16


def test_fused_add_norm_uses_unrounded_sum():
    h = bf(np.full((1, 8), 1.0)); r = bf(np.full((1, 8), 2.0 ** -9))  # sum not bf16-representable
    nh, out = ops.fused_add_rms_norm(h, r, bf(np.ones(8)), 1e-6)
    assert np.all(nh == 1.0) and np.all(out == 1.0)


def test_plan_helpers():  # paged_attention.cu:312-397, ops/attention.rs:208-302
    assert ops.batch_prefill_cta_tile_q(1, 32, 8, 128) == 16
    assert ops.batch_prefill_cta_tile_q(16, 32, 8, 128) == 64
    assert ops.batch_prefill_cta_tile_q(17, 32, 8, 128) == 128
    assert ops.batch_prefill_paged_num_tiles(10000, 32, 8, 128) == 313
    assert ops.batch_prefill_paged_num_tiles(100, 32, 8, 128, 7) == -1
    plan = ops.prefill_paged_plan([[3, 4], [5]], [4, 7], [0, 0], [20, 7], 32, 8, 128, 64)
    assert plan["request_indices"].tolist() == [0, 0, 1] and plan["qo_tile_indices"].tolist() == [0, 1, 0]
    assert plan["kv_chunk_size"].tolist() == [20, 7] and plan["q_indptr"].tolist() == [0, 20, 27]


def test_split_kv_plan():  # batch_decode_buffers.rs:229-287
    p = ops.split_kv_plan([1024], 1)
    assert p["kv_chunk_size"] == 256 and p["o_indptr"].tolist() == [0, 4] and p["padded_slots"] == 64
    assert p["block_valid_mask"][:4].tolist() == [1, 1, 1, 1] and p["block_valid_mask"][4:].sum() == 0
    p = ops.split_kv_plan([20000, 5], 2)
    assert p["kv_chunk_size"] == 313 and p["o_indptr"].tolist() == [0, 64, 65]
    assert ops.attention_path_is_split(2, 1024) and not ops.attention_path_is_split(4, 5000)
    assert ops.bucket_for(3) == 4 and ops.bucket_for(64) == 64


def test_split_kv_equals_non_partition_oracle():  # model-crate.md:205 invariant
    rng = np.random.default_rng(0)
    lay = ops.PagedKvLayout(1, 2, 128, 16)
    kv = bf16_round(rng.standard_normal(8 * lay.page_stride).astype(np.float32))
    q = bf16_round(rng.standard_normal((1, 8 * 128)).astype(np.float32))
    pages, indptr, last = np.array([3, 1, 5, 2, 7]), np.array([0, 5]), np.array([9])
    a = ops.paged_attention_decode(q, kv, lay, 0, pages, indptr, last, 8, 0.0884)
    b = ops.paged_attention_decode_split_kv(q, kv, lay, 0, pages, indptr, last, [0, 0, 0], [0, 1, 2], 32,
                                            [0, 3], [1, 1, 1], 8, 0.0884)
    assert np.abs(a - b).max() <= 0.02


def test_vectorised_prefill_attention_equals_rowwise_restatement():
    """oracle.ops.batch_prefill_paged (block form, used for the 1-4k-token GPU parity cases) ==
    batch_prefill_paged_rowwise (one _attend per query row, the literal restatement of
    csrc/paged_attention.cu:399-535): two requests, one of them chunked (kv_len > qo_len), ragged
    last pages.  Both are float64; the only difference is the BLAS summation order, so after the bf16
    store they agree to <= 1 ulp with > 99.9 % of the elements bit-equal."""
    from conftest import bf16_ulp_diff
    from oracle.bf16 import bf16_bits
    rng = np.random.default_rng(11)
    Hq, Hkv, D, ps = 8, 2, 128, 16
    layout = ops.PagedKvLayout(2, Hkv, D, ps)
    kv_lens, qo_lens = [150, 77], [150, 30]
    pages, indptr, last, nxt = [], [0], [], 1
    for L in kv_lens:
        n = -(-L // ps)
        pages += list(range(nxt, nxt + n)); nxt += n
        indptr.append(len(pages)); last.append(L - (n - 1) * ps)
    kv = np.zeros(nxt * layout.page_stride, np.float32)
    for b, L in enumerate(kv_lens):
        k = bf16_round(rng.standard_normal((L, Hkv * D)).astype(np.float32))
        v = bf16_round(rng.standard_normal((L, Hkv * D)).astype(np.float32))
        ops.paged_kv_scatter(kv, layout, 1, np.asarray(pages), np.asarray(indptr), k, v, np.full(L, b), np.arange(L))
    q = bf16_round(rng.standard_normal((sum(qo_lens), Hq * D)).astype(np.float32) * 2)
    qind = np.concatenate([[0], np.cumsum(qo_lens)])
    args = (q, kv, layout, 1, np.asarray(pages), np.asarray(indptr), np.asarray(last), qind, Hq, np.float32(D ** -0.5))
    a = ops.batch_prefill_paged(*args, row_block=64)
    b_ = ops.batch_prefill_paged_rowwise(*args)
    assert bf16_ulp_diff(a, b_) <= 1
    assert (bf16_bits(a) == bf16_bits(b_)).mean() > 0.999


def test_bf16_round_c_helper_equals_the_numpy_statement():
    """oracle/bf16_round.c (one pass, OpenMP; used for arrays >= 64 K elements when built) == the numpy statement of
    __float2bfloat16 on random bit patterns, ties, infinities, NaNs of both signs and denormals."""
    from oracle import bf16 as B
    if B._native_round() is None:
        pytest.skip("oracle/_native/libpegainfer_oracle.so not built (make -C oracle)")
    rng = np.random.default_rng(3)
    u = rng.integers(0, 2 ** 32, B._NATIVE_MIN * 8, dtype=np.uint64).astype(np.uint32)
    u[:8] = [0x3F808000, 0x3F818000, 0x7F800000, 0xFF800000, 0x7FC12345, 0xFFC00001, 0x00000001, 0x80008000]
    x = u.view(np.float32)
    a, b = B.bf16_round(x), B._bf16_round_numpy(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(B.bf16_round(x[:1000]).view(np.uint32), b[:1000].view(np.uint32))       # small arrays: numpy path
