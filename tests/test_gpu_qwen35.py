"""GPU parity of the Qwen3.5-4B hybrid extras against the oracle: causal conv1d (+ state),
gated-delta-rule decode (+ fp32 state), HD256 prep / partial RoPE / gate, HD256 paged decode and prefill
attention.  Shapes from docs/models/qwen35/optimization.md:56-76: 16 q / 4 kv heads x 256, rotary 64,
linear attention 16 k-heads x 128, 32 v-heads x 128, conv k=4 over 8192 channels."""
import numpy as np
import pytest

from conftest import bf16_ulp_diff, from_dev, to_dev
from oracle import ops as O
from oracle.bf16 import bf16_bits, bf16_round

pytestmark = pytest.mark.gpu


def rnd(rng, *shape, scale=1.0):
    return bf16_round((rng.standard_normal(shape) * scale).astype(np.float32))


def S():
    import torch
    return torch.cuda.current_stream().cuda_stream


def test_conv1d_long_prompt_kernel_equals_the_short_one(built_libs):
    """From 256 tokens on a lane walks 8 consecutive tokens with a sliding window (round 6); per output the same taps in the same
    order: the first 200 outputs of a 300-token call equal a 200-token call (one token per lane) bit for bit, state included."""
    from pegainfer_amd import ffi
    rng = np.random.default_rng(11)
    C, K = 8192, 4
    x, w, st = rnd(rng, 300, C), rnd(rng, C, K, scale=0.5), rnd(rng, C, K - 1)
    wd = to_dev(w)
    o300, o200 = to_dev(np.zeros((300, C), np.float32)), to_dev(np.zeros((200, C), np.float32))
    s300, s200 = to_dev(st), to_dev(st)
    ffi.lib().conv1d_prefill_cuda(to_dev(x).data_ptr(), wd.data_ptr(), s300.data_ptr(), o300.data_ptr(), C, 300, K, S())
    ffi.lib().conv1d_prefill_cuda(to_dev(x[:200]).data_ptr(), wd.data_ptr(), s200.data_ptr(), o200.data_ptr(), C, 200, K, S())
    assert np.array_equal(bf16_bits(from_dev(o300)[:200]), bf16_bits(from_dev(o200)))
    eo, es = O.conv1d_prefill(x, w, st)
    assert bf16_ulp_diff(from_dev(o300), eo) <= 1 and np.array_equal(bf16_bits(from_dev(s300)), bf16_bits(es))


@pytest.mark.parametrize("T,C,K", [(1, 8192, 4), (37, 8192, 4), (2, 100, 4), (5, 64, 2), (300, 8192, 4)])
def test_conv1d_prefill_and_state(built_libs, T, C, K):
    from pegainfer_amd import ffi
    rng = np.random.default_rng(T + C)
    x, w, st = rnd(rng, T, C), rnd(rng, C, K, scale=0.5), rnd(rng, C, K - 1)
    xd, wd, sd = to_dev(x), to_dev(w), to_dev(st)
    out = to_dev(np.zeros((T, C), np.float32))
    ffi.lib().conv1d_prefill_cuda(xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), out.data_ptr(), C, T, K, S())
    eo, es = O.conv1d_prefill(x, w, st)
    assert bf16_ulp_diff(from_dev(out), eo) <= 1
    assert np.array_equal(bf16_bits(from_dev(sd)), bf16_bits(es))          # state = raw inputs: exact
    # decode continues from the state (recurrent.rs:49-79): one more token
    x2 = rnd(rng, 1, C)
    x2d = to_dev(x2)
    ffi.lib().conv1d_prefill_cuda(x2d.data_ptr(), wd.data_ptr(), sd.data_ptr(), out.data_ptr(), C, 1, K, S())
    eo2, es2 = O.conv1d_prefill(x2, w, es)
    assert bf16_ulp_diff(from_dev(out)[:1], eo2) <= 1 and np.array_equal(bf16_bits(from_dev(sd)), bf16_bits(es2))


def test_gated_delta_rule_decode(built_libs):
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(7)
    kh, vh, kd, vd = 16, 32, 128, 128
    qkv = rnd(rng, 2 * kh * kd + vh * vd)
    b, a, dtb = rnd(rng, vh), rnd(rng, vh), rnd(rng, vh, scale=0.5)
    alog = (rng.standard_normal(vh) * 0.5).astype(np.float32)
    state = (rng.standard_normal((vh, kd, vd)) * 0.1).astype(np.float32)
    sd = torch.from_numpy(state.copy()).cuda()
    out = to_dev(np.zeros(vh * vd, np.float32))
    ref_state = state
    bd, ad, dd, ald = to_dev(b), to_dev(a), to_dev(dtb), torch.from_numpy(alog).cuda()   # keep alive
    for step in range(3):                                   # the state carries across steps
        qd = to_dev(qkv)
        ffi.lib().gated_delta_rule_decode_cuda(qd.data_ptr(), bd.data_ptr(), ad.data_ptr(), dd.data_ptr(),
                                               ald.data_ptr(), sd.data_ptr(), out.data_ptr(), kh, vh, kd, vd, S())
        eo, ref_state = O.gated_delta_rule_decode(qkv, b, a, dtb, alog, ref_state, kh, vh, kd, vd)
        got = from_dev(out)
        assert np.abs(got - eo).max() <= 2.0 ** -7 * max(1.0, np.abs(eo).max())    # bf16 output of an fp32 recurrence
        assert np.allclose(sd.cpu().numpy(), ref_state, rtol=2e-5, atol=2e-6)
        qkv = rnd(rng, 2 * kh * kd + vh * vd)


@pytest.mark.parametrize("kh,vh", [(16, 32), (4, 4), (2, 8)])
def test_linear_attn_decode_fused_equals_the_three_calls(built_libs, kh, vh):
    """pegainfer_linear_attn_decode_fused (round 6: one workgroup per value head, the shared q / k conv windows shifted by the last arriver of their key head) == conv1d_prefill_cuda(seq_len 1) ->
    gated_delta_rule_decode_cuda -> rms_norm_gated_cuda: every bit of the output, of the conv window and of the fp32 recurrent
    state, over three steps that carry both states (recurrent.rs:49-79); and the output against the oracle's op sequence."""
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(kh)
    kd = vd = 128
    K, C, Z = 4, 2 * kh * kd + vh * vd, vh * vd
    w, conv0 = rnd(rng, C, K, scale=0.5), rnd(rng, C, K - 1)
    dtb = rnd(rng, vh, scale=0.5)
    alog = (rng.standard_normal(vh) * 0.5).astype(np.float32)
    nw = (1 + rng.standard_normal(vd) * 0.1).astype(np.float32)
    state0 = (rng.standard_normal((vh, kd, vd)) * 0.1).astype(np.float32)
    L = ffi.lib()
    wd, dd, ald, nwd = to_dev(w), to_dev(dtb), torch.from_numpy(alog).cuda(), torch.from_numpy(nw).cuda()
    cs = [to_dev(conv0), to_dev(conv0)]                                   # [three calls, fused]
    st = [torch.from_numpy(state0.copy()).cuda(), torch.from_numpy(state0.copy()).cuda()]
    tmp, tmp2 = to_dev(np.zeros(C, np.float32)), to_dev(np.zeros(Z, np.float32))
    outs = [to_dev(np.zeros(Z, np.float32)), to_dev(np.zeros(Z, np.float32))]
    tick = torch.zeros(kh, dtype=torch.int32, device="cuda")               # zero once: the words reset themselves
    o_conv, o_state = conv0, state0
    for step in range(3):
        x, z, b, a = rnd(rng, C), rnd(rng, Z), rnd(rng, vh), rnd(rng, vh)
        xd, zd, bd, ad = to_dev(x), to_dev(z), to_dev(b), to_dev(a)
        L.conv1d_prefill_cuda(xd.data_ptr(), wd.data_ptr(), cs[0].data_ptr(), tmp.data_ptr(), C, 1, K, S())
        L.gated_delta_rule_decode_cuda(tmp.data_ptr(), bd.data_ptr(), ad.data_ptr(), dd.data_ptr(), ald.data_ptr(),
                                       st[0].data_ptr(), tmp2.data_ptr(), kh, vh, kd, vd, S())
        L.rms_norm_gated_cuda(tmp2.data_ptr(), nwd.data_ptr(), zd.data_ptr(), outs[0].data_ptr(), vh, vd, 1e-6, S())
        rc = L.pegainfer_linear_attn_decode_fused(xd.data_ptr(), wd.data_ptr(), cs[1].data_ptr(), bd.data_ptr(), ad.data_ptr(),
                                                  dd.data_ptr(), ald.data_ptr(), st[1].data_ptr(), nwd.data_ptr(), zd.data_ptr(),
                                                  outs[1].data_ptr(), kh, vh, kd, vd, K, 1e-6, tick.data_ptr(), S())
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(bf16_bits(from_dev(outs[0])), bf16_bits(from_dev(outs[1]))), step
        assert np.array_equal(bf16_bits(from_dev(cs[0])), bf16_bits(from_dev(cs[1]))), step
        assert np.array_equal(st[0].cpu().numpy().view(np.uint32), st[1].cpu().numpy().view(np.uint32)), step
        eo, o_conv = O.conv1d_prefill(x[None], w, o_conv)
        eg, o_state = O.gated_delta_rule_decode(eo[0], b, a, dtb, alog, o_state, kh, vh, kd, vd)
        o_state = o_state.astype(np.float32)
        en = O.rms_norm_gated(eg, nw, z, vd, 1e-6)
        assert np.abs(from_dev(outs[1]) - en).max() <= 2.0 ** -6 * max(1.0, np.abs(en).max())
    # a shape the kernel does not take launches nothing and says so
    assert L.pegainfer_linear_attn_decode_fused(xd.data_ptr(), wd.data_ptr(), cs[1].data_ptr(), bd.data_ptr(), ad.data_ptr(),
                                                dd.data_ptr(), ald.data_ptr(), st[1].data_ptr(), nwd.data_ptr(), zd.data_ptr(),
                                                outs[1].data_ptr(), kh, vh, 64, vd, K, 1e-6, tick.data_ptr(), S()) == 1
    assert int(tick.abs().sum()) == 0


@pytest.mark.parametrize("T,prefill", [(1, False), (5, False), (40, True)])
def test_hd256_prep_and_gate(built_libs, T, prefill):
    import torch
    from pegainfer_amd import ffi
    rng = np.random.default_rng(T)
    Hq, Hkv, rot, max_seq = 16, 4, 64, 128
    q_full, k, v = rnd(rng, T, Hq * 512, scale=2), rnd(rng, T, Hkv * 256, scale=2), rnd(rng, T, Hkv * 256)
    qw, kw = rnd(rng, 256, scale=0.2), rnd(rng, 256, scale=0.2)
    half = rot // 2
    inv = (1.0 / np.power(np.float32(1e7), np.arange(half, dtype=np.float32) * 2 / rot)).astype(np.float32)
    fr = np.arange(256, dtype=np.float32)[:, None] * inv[None, :]
    cos = bf16_round(np.concatenate([np.cos(fr), np.cos(fr)], 1).astype(np.float32))     # [pos, rotary_dim]
    sin = bf16_round(np.concatenate([np.sin(fr), np.sin(fr)], 1).astype(np.float32))
    q_src = q_full.reshape(T, Hq, 2, 256)[:, :, 0, :]
    L = ffi.lib()
    q_out = to_dev(np.zeros((T, Hq * 256), np.float32))
    qd, kd_, vd_ = to_dev(q_full), to_dev(k), to_dev(v)
    args = (to_dev(qw), to_dev(kw), to_dev(cos), to_dev(sin))
    if prefill:
        start = 17
        pos = np.arange(start, start + T)
        kc = to_dev(np.zeros((Hkv, max_seq, 256), np.float32)); vc = to_dev(np.zeros((Hkv, max_seq, 256), np.float32))
        sp = torch.tensor([start], dtype=torch.int32, device="cuda")
        L.prefill_attention_hd256_prep_cuda(qd.data_ptr(), kd_.data_ptr(), vd_.data_ptr(), args[0].data_ptr(),
                                            args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(),
                                            q_out.data_ptr(), kc.data_ptr(), vc.data_ptr(), Hq, Hkv, T,
                                            sp.data_ptr(), rot, 1e-6, max_seq, S())
        ek = O.hd256_norm_partial_rope(k.reshape(T, Hkv, 256), kw, cos, sin, pos, rot, 1e-6)
        gk = from_dev(kc)[:, start:start + T].transpose(1, 0, 2)
        assert bf16_ulp_diff(gk, ek) <= 1
        assert np.array_equal(from_dev(vc)[:, start:start + T].transpose(1, 0, 2), v.reshape(T, Hkv, 256))
        assert np.all(from_dev(kc)[:, :start] == 0) and np.all(from_dev(kc)[:, start + T:] == 0)
    else:
        pos = rng.integers(0, 200, T)
        L.qk_norm_partial_rope_batched_decode_hd256_cuda(qd.data_ptr(), kd_.data_ptr(), args[0].data_ptr(),
                                                         args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(),
                                                         (posd := torch.tensor(pos, dtype=torch.int32, device="cuda")).data_ptr(),
                                                         q_out.data_ptr(), Hq, Hkv, T, rot, 1e-6, S())
        ek = O.hd256_norm_partial_rope(k.reshape(T, Hkv, 256), kw, cos, sin, pos, rot, 1e-6)
        assert bf16_ulp_diff(from_dev(kd_).reshape(T, Hkv, 256), ek) <= 1
    eq = O.hd256_norm_partial_rope(q_src, qw, cos, sin, pos, rot, 1e-6)
    assert bf16_ulp_diff(from_dev(q_out).reshape(T, Hq, 256), eq) <= 1
    assert np.array_equal(from_dev(qd), q_full)                                   # q_full itself is read-only
    attn = rnd(rng, T, Hq * 256)
    ad = to_dev(attn)
    L.attention_gate_batch_hd256_cuda(qd.data_ptr(), ad.data_ptr(), Hq, T, S())
    assert bf16_ulp_diff(from_dev(ad), O.attention_gate_hd256(q_full, attn, Hq)) <= 1


@pytest.mark.parametrize("lens", [[1], [300, 17], [1024], [4096, 3000, 2049, 4096, 17, 1024, 4000, 4095]])   # last: bs 8 / ctx 4096
def test_paged_decode_attention_hd256(built_libs, lens):
    import torch
    from pegainfer_amd import ffi
    from test_gpu_ops import attn_tol, make_paged
    rng = np.random.default_rng(sum(lens))
    bs, Hq, Hkv, D = len(lens), 16, 4, 256
    lay, kv, pages, indptr, last = make_paged(rng, bs, lens, Hkv=Hkv, D=D)
    q = rnd(rng, bs, Hq * D)
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device="cuda")
    keep = [to_dev(q), to_dev(kv), i32(pages), i32(indptr), i32(last), i32(np.arange(bs)), i32(np.zeros(bs)), i32(lens)]
    out = torch.zeros((bs, Hq * D), dtype=torch.bfloat16, device="cuda")
    sm = 1.0 / np.sqrt(256.0)
    rc = ffi.lib().paged_attention_decode_cuda_hd256(keep[0].data_ptr(), out.data_ptr(), keep[1].data_ptr(),
                                                     lay.layer_stride, lay.layer_stride + lay.kv_block_len,
                                                     keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                                                     keep[5].data_ptr(), keep[6].data_ptr(), keep[7].data_ptr(), Hq, Hkv,
                                                     D, 16, bs, lay.page_stride, sm, S())
    assert rc == 0
    ref = O.paged_attention_decode(q, kv, lay, 1, pages, indptr, last, Hq, sm)
    assert np.abs(from_dev(out) - ref).max() <= attn_tol(ref)


@pytest.mark.parametrize("seq_lens,starts", [([5], [0]), ([70, 3], [0, 20]), ([130], [0]),
                                             # the lengths TTFT is quoted at: 1024 / 4096 tokens, a ragged chunked batch
                                             ([1024], [0]), ([4096], [0]), ([1500, 548], [0, 300])])
def test_batch_prefill_paged_hd256(built_libs, seq_lens, starts):
    import torch
    from pegainfer_amd import ffi
    from test_gpu_ops import attn_tol, make_paged
    rng = np.random.default_rng(sum(seq_lens))
    Hq, Hkv, D = 16, 4, 256
    lens = [s + n for s, n in zip(starts, seq_lens)]
    lay, kv, pages, indptr, last = make_paged(rng, len(lens), lens, Hkv=Hkv, D=D)
    T = sum(seq_lens)
    q = rnd(rng, T, Hq * D)
    page_lists = [pages[indptr[i]:indptr[i + 1]].tolist() for i in range(len(lens))]
    pl = O.prefill_paged_plan(page_lists, last.tolist(), starts, seq_lens, Hq, Hkv, D, 0)
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device="cuda")
    keep = {k: i32(pl[k]) for k in ("page_indices", "page_indptr", "last_page_len", "q_indptr", "request_indices",
                                    "qo_tile_indices", "kv_tile_indices", "kv_chunk_size")}
    tot = i32([T])
    qd, kvd = to_dev(q), to_dev(kv)
    out = torch.zeros((T, Hq * D), dtype=torch.bfloat16, device="cuda")
    sm = 1.0 / np.sqrt(256.0)
    rc = ffi.lib().batch_prefill_paged_cuda_hd256(
        qd.data_ptr(), out.data_ptr(), kvd.data_ptr(), lay.layer_stride, lay.layer_stride + lay.kv_block_len,
        keep["page_indices"].data_ptr(), keep["page_indptr"].data_ptr(), keep["last_page_len"].data_ptr(),
        keep["q_indptr"].data_ptr(), keep["request_indices"].data_ptr(), keep["qo_tile_indices"].data_ptr(),
        keep["kv_tile_indices"].data_ptr(), keep["kv_chunk_size"].data_ptr(), tot.data_ptr(), Hq, Hkv, D, 16, T,
        len(lens), pl["num_tiles"], lay.page_stride, sm, S())
    assert rc == 0
    ref = O.batch_prefill_paged(q, kv, lay, 1, pages, indptr, last, pl["q_indptr"], Hq, sm)
    assert np.abs(from_dev(out) - ref).max() <= 2 * attn_tol(ref)


def _gdr_inputs(rng, T, kh, vh):
    import torch
    qkv = rnd(rng, T, 2 * kh * 128 + vh * 128)
    b, a, dtb = rnd(rng, T, vh), rnd(rng, T, vh), rnd(rng, vh, scale=0.5)
    alog = (rng.standard_normal(vh) * 0.5).astype(np.float32)
    state = (rng.standard_normal((vh, 128, 128)) * 0.1).astype(np.float32)
    return qkv, b, a, dtb, alog, state


@pytest.mark.parametrize("T,kh,vh", [(64, 2, 4), (150, 2, 4), (1, 1, 2), (333, 16, 32),
                                     (1024, 16, 32), (4096, 16, 32), (1000, 16, 32)])   # Qwen3.5-4B heads at the bench lengths; a ragged last chunk
def test_gdr_chunkwise_prefill_stages_and_operator(built_libs, T, kh, vh):
    """The seven chunk-wise stages (ffi.rs:1041-1137) against the oracle restatement of the Triton kernels, stage by
    stage on the oracle's own inputs (so one stage's rounding noise is not amplified by the next), then the whole
    operator, then chunk-wise == token-by-token decode recurrence (the property the reference relies on when a
    prefilled state is handed to decode).  Tolerances: fp32 stages 1e-5 relative; bf16 outputs 2^-7 of the tensor's
    max (the GPU accumulates in fp32, the oracle in fp64, both round at the same points)."""
    import torch
    import pegainfer_amd.ops as P
    from pegainfer_amd import ffi
    L = ffi.lib()
    rng = np.random.default_rng(T * 7 + vh)
    qkv, b, a, dtb, alog, state = _gdr_inputs(rng, T, kh, vh)
    f32d = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    bfd = lambda x: to_dev(np.ascontiguousarray(x).reshape(x.shape[0], -1))
    close = lambda got, ref, tol: np.abs(got - ref).max() <= tol * max(1e-6, np.abs(ref).max())
    # oracle stages
    q, k, v, g, beta = O.gdr_chunk_prepare(qkv, b, a, dtb, alog, kh, vh, 128, 128)
    gc = O.gdr_chunk_cumsum(g)
    A = O.gdr_chunk_a(k, gc, beta)
    Ai = O.gdr_chunk_solve(A)
    w, u = O.gdr_chunk_recompute(k, v, beta, Ai, gc)
    cs, vn, fs = O.gdr_chunk_state(k, w, u, gc, state)
    out = O.gdr_chunk_o(q, k, vn, cs, gc, 1.0 / np.sqrt(128.0))
    sc = P.GdrChunkwiseScratch35(vh, 128, 128, T)
    qkvd, bd, ad, dd, ald = to_dev(qkv), to_dev(b), to_dev(a), to_dev(dtb), f32d(alog)
    # 1 prepare
    assert L.gated_delta_rule_prefill_chunk_prepare_cuda(
        qkvd.data_ptr(), bd.data_ptr(), ad.data_ptr(), dd.data_ptr(), ald.data_ptr(), sc.q_expanded.data_ptr(),
        sc.k_expanded.data_ptr(), sc.v_raw.data_ptr(), sc.g_cumsum.data_ptr(), sc.beta.data_ptr(), kh, vh,
        qkv.shape[1], T, S()) == 0
    assert bf16_ulp_diff(from_dev(sc.q_expanded), q.reshape(T, -1)) <= 1
    assert bf16_ulp_diff(from_dev(sc.k_expanded), k.reshape(T, -1)) <= 1
    assert np.array_equal(bf16_bits(from_dev(sc.v_raw)), bf16_bits(v.reshape(T, -1)))
    assert close(sc.g_cumsum.cpu().numpy().reshape(T, vh), g, 1e-5)
    assert close(sc.beta.cpu().numpy().reshape(T, vh), beta, 1e-5)
    # 2 cumsum (in place, like recurrent.rs:165-183)
    gd = f32d(g)
    assert L.gated_delta_rule_prefill_chunk_cumsum_cuda(gd.data_ptr(), gd.data_ptr(), T, vh, S()) == 0
    assert close(gd.cpu().numpy(), gc, 1e-5)
    # 3 A
    kd, gcd, betad = bfd(k), f32d(gc), f32d(beta)
    assert L.gated_delta_rule_prefill_chunk_a_cuda(kd.data_ptr(), gcd.data_ptr(), betad.data_ptr(),
                                                   sc.a_tril.data_ptr(), T, vh, S()) == 0
    assert close(sc.a_tril.cpu().numpy().reshape(T, vh, 64), A, 1e-4)
    # 4 solve
    Ad = f32d(A)
    assert L.gated_delta_rule_prefill_chunk_solve_cuda(Ad.data_ptr(), sc.a_inv.data_ptr(), T, vh, S()) == 0
    got_ai = from_dev(sc.a_inv).reshape(T, vh, 64)
    assert close(got_ai, Ai, 2.0 ** -7)
    # 5 recompute
    vd_, aid = bfd(v), bfd(Ai)
    assert L.gated_delta_rule_prefill_chunk_recompute_cuda(kd.data_ptr(), vd_.data_ptr(), betad.data_ptr(),
                                                           sc.w.data_ptr(), sc.u.data_ptr(), aid.data_ptr(),
                                                           gcd.data_ptr(), T, vh, S()) == 0
    assert close(from_dev(sc.w), w.reshape(T, -1), 2.0 ** -7) and close(from_dev(sc.u), u.reshape(T, -1), 2.0 ** -7)
    # 6 state (in place on the state buffer, recurrent.rs:290-330)
    wd, ud, std = bfd(w), bfd(u), f32d(state)
    assert L.gated_delta_rule_prefill_chunk_state_cuda(kd.data_ptr(), wd.data_ptr(), ud.data_ptr(), gcd.data_ptr(),
                                                       std.data_ptr(), sc.chunk_state.data_ptr(),
                                                       sc.v_new.data_ptr(), std.data_ptr(), T, vh, S()) == 0
    assert close(from_dev(sc.v_new), vn.reshape(T, -1), 2.0 ** -6)
    assert close(sc.chunk_state.cpu().numpy().reshape(cs.shape), cs, 2.0 ** -6)
    assert close(std.cpu().numpy(), fs, 2.0 ** -6)
    # 7 output
    qd, vnd, csd = bfd(q), bfd(vn), f32d(cs)
    outd = torch.zeros((T, vh * 128), dtype=torch.bfloat16, device="cuda")
    assert L.gated_delta_rule_prefill_chunk_o_cuda(qd.data_ptr(), kd.data_ptr(), vnd.data_ptr(), csd.data_ptr(),
                                                   gcd.data_ptr(), outd.data_ptr(), T, vh,
                                                   1.0 / float(np.sqrt(128.0)), S()) == 0
    assert close(from_dev(outd), out, 2.0 ** -6)
    # whole operator through the Rust-shaped wrapper, vs the oracle operator and vs the decode recurrence
    std2 = f32d(state)
    out2 = torch.zeros((T, vh * 128), dtype=torch.bfloat16, device="cuda")
    P.gated_delta_rule_prefill_chunkwise_into(qkvd, bd, ad, dd, ald, std2, sc, out2, kh, vh, 128, 128)
    assert close(from_dev(out2), out, 2.0 ** -5) and close(std2.cpu().numpy(), fs, 2.0 ** -5)
    if T <= 150:
        s_seq, rows = state.copy(), []
        for i in range(T):
            o, s_seq = O.gated_delta_rule_decode(qkv[i], b[i], a[i], dtb, alog, s_seq, kh, vh, 128, 128)
            rows.append(o)
        assert close(from_dev(out2), np.stack(rows), 2.0 ** -4) and close(std2.cpu().numpy(), s_seq, 2.0 ** -4)
