"""GPU: the native collective layer (include/pegainfer_comm.h, csrc/host/comm.cpp).

A 1-GPU box can run (a) every MP8 verb on a world-size-1 communicator, where each must degenerate to the exact
copy / cast it is built around, and (b) the expert-parallel dispatch / combine at world size 4 through the LOOPBACK
hub (virtual ranks in one process: same routing / packing / combine kernels, device memcpy as the transport) against
oracle/ep_ref.py, plus the same path over a world-size-1 RCCL-style communicator.  The N > 1 RCCL transport itself is
exercised by bench.py --gpus N (mp8_collectives_us) on the driver's multi-GPU runs."""
import numpy as np
import pytest

from oracle import ep_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(built_libs):
    from pegainfer_amd.parallel import NativeComm
    c = NativeComm()
    yield c
    c.close()


def test_world1_verbs_are_exact_copies_and_casts(comm):
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(37, 4096, device="cuda", generator=g).to(torch.bfloat16)
    bits = x.view(torch.int16).clone()
    assert comm.world == 1 and comm.rank == 0
    assert torch.equal(comm.all_reduce_in_place(x.clone()).view(torch.int16), bits)
    f = x.float()
    assert torch.equal(comm.all_reduce_in_place(f.clone()), f)
    assert torch.equal(comm.all_reduce_hidden_fp32_in_place(x.clone()).view(torch.int16), bits)   # bf16 -> f32 -> bf16
    out = torch.empty_like(f)
    assert torch.equal(comm.all_reduce_hidden_to_f32(x, out), f)                                   # exact widening
    assert torch.equal(comm.all_gather(x).view(torch.int16), bits)
    u = torch.arange(1000, dtype=torch.int32, device="cuda")
    assert torch.equal(comm.all_gather(u), u)
    assert torch.equal(comm.reduce_scatter(f), f)
    assert torch.equal(comm.all_to_all(x).view(torch.int16), bits)
    assert torch.equal(comm.all_gather_logits(f[0]), f[0])


def test_moe_all_gather_reduce_scatter_overlap_path(comm):
    """moe.rs:1327-1461 on the comm stream with the two event fences: at world size 1 the result is
    expert_fn(hidden) + shared_fn(hidden) in f32."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(6)
    h = torch.randn(8, 4096, device="cuda", generator=g).to(torch.bfloat16)
    got = comm.moe_all_gather_reduce_scatter(h, lambda a: a.float() * 2.0, lambda a: (a.float() + 1.0).to(torch.bfloat16))
    want = h.float() * 2.0 + (h.float() + 1.0).to(torch.bfloat16).float()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def _mk_eps(world, H, T, cap, E, topk, use_comm=None, **kw):
    from pegainfer_amd.parallel import EpEndpoint
    hub = None if use_comm is not None else EpEndpoint.hub(world)
    eps = [EpEndpoint(H, max(T, 1), cap, E, topk, comm=use_comm, hub=hub, rank=r, world=world, **kw) for r in range(world)]
    return eps, hub


def _close_eps(eps, hub):
    from pegainfer_amd import ffi
    for e in eps:
        e.close()
    if hub:
        ffi.host_lib().pegainfer_ep_hub_destroy(hub)


def _run_ep(world, E, topk, H, Ts, seed, use_comm=None, pad=1, bounds=None):
    """random ragged case vs oracle/ep_ref.py: payload rows exact and in the oracle's order, counts exact (device
    array AND host mirror), combine bit-exact against the reference's f32 fma chain (one bf16 rounding)"""
    import torch
    from oracle.bf16 import bf16_round
    rng = np.random.default_rng(seed)
    xs = [torch.tensor(rng.standard_normal((t, H)), dtype=torch.float32).to(torch.bfloat16).cuda() for t in Ts]
    idx = [torch.tensor(np.stack([rng.permutation(E)[:topk] for _ in range(t)]) if t else np.zeros((0, topk)),
                        dtype=torch.int32).cuda() for t in Ts]
    ws = [torch.tensor(rng.random((t, topk)), dtype=torch.float32).cuda() for t in Ts]
    bm = [None] * world if bounds is None else [torch.tensor([b], dtype=torch.int32).cuda() for b in bounds]
    cap = sum(Ts) * topk + (E // world) * pad + 1
    eps, hub = _mk_eps(world, H, max(Ts), cap, E, topk, use_comm=use_comm, expert_padding=pad)
    try:
        for r in range(world):
            eps[r].dispatch_send(xs[r], idx[r], ws[r], bound_m=bm[r])
        recv = [eps[r].dispatch_recv() for r in range(world)]
        ref = ep_ref.dispatch([x.float().cpu().numpy() for x in xs], [i.cpu().numpy() for i in idx], E,
                              expert_padding=pad, bound_m=bounds)
        ys = []
        for r in range(world):
            rows, cnt = recv[r]
            n = len(ref[r][0])
            assert eps[r].num_padded_recv_tokens() == n and eps[r].num_recv_tokens() == int(ref[r][1].sum())
            live = ref[r][2][:, 0] >= 0
            assert np.array_equal(rows[:n].float().cpu().numpy()[live], ref[r][0][live])   # payload rows: exact, in order
            assert np.array_equal(rows[:n].float().cpu().numpy()[~live], np.zeros_like(ref[r][0][~live]))   # gaps untouched
            assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), ref[r][1])          # device counts (ABI output)
            assert np.array_equal(eps[r].tokens_per_expert(), ref[r][1])
            ys.append((rows.float() * 2.0 + 1.0).to(torch.bfloat16))                       # the "expert"
        for r in range(world):
            eps[r].combine_send(ys[r])
        prev = [torch.full((t, H), 0.5, dtype=torch.bfloat16, device="cuda") for t in Ts]
        for acc in (False, True):
            outs = [eps[r].combine_recv(prev[r].clone(), idx[r], ws[r], accumulate=acc, bound_m=bm[r]) for r in range(world)]
            want = ep_ref.combine_f32([y.float().cpu().numpy()[:len(ref[r][0])] for r, y in enumerate(ys)],
                                      [ref[r][2] for r in range(world)], [w.cpu().numpy() for w in ws], list(Ts), H,
                                      prev=[p.float().cpu().numpy() for p in prev] if acc else None, bound_m=bounds)
            for r in range(world):
                got = outs[r].float().cpu().numpy()
                exp = bf16_round(want[r])
                if not acc and bounds is not None:      # rows at or beyond the bound are not written: they keep `prev`
                    exp[bounds[r]:] = 0.5
                assert np.array_equal(got, exp), (r, acc, np.abs(got - exp).max())
    finally:
        _close_eps(eps, hub)


@pytest.mark.parametrize("world,E,topk,H,Ts", [(4, 16, 2, 256, (5, 0, 7, 3)), (8, 64, 6, 512, (9, 1, 0, 4, 16, 2, 3, 5)),
                                               (2, 8, 8, 128, (3, 3)), (1, 4, 2, 64, (6,))])
def test_ep_dispatch_combine_loopback_matches_oracle(built_libs, world, E, topk, H, Ts):
    _run_ep(world, E, topk, H, Ts, seed=world * 100 + E)


def test_ep_expert_padding_and_device_token_bound(built_libs):
    """EpTopology.expert_padding (a2a_worker.rs:598-606) and bound_m_ptr (a2a_dispatch_send.cu:172): groups start at
    multiples of the padding; only the first *bound_m tokens of a rank are dispatched / combined"""
    _run_ep(4, 16, 3, 128, (9, 4, 0, 6), seed=11, pad=8)
    _run_ep(4, 16, 3, 128, (9, 4, 3, 6), seed=12, pad=4, bounds=[5, 0, 3, 6])


def test_ep_dispatch_combine_over_comm_world1(comm):
    _run_ep(1, 8, 2, 256, (11,), seed=3, use_comm=comm)


@pytest.mark.parametrize("cid", __import__("ep_golden").ids())
def test_ep_reproduces_the_reference_a2a_test(built_libs, cid):
    """The reference's own all-to-all test (pegainfer-comm/tests/p2p_all_to_all/test_p2p_all_to_all.py:95-232) with its
    own inputs (RankTestData.create; fixture tests/golden/ep_a2a_golden.npz made by tests/golden/make_ep_golden.py) on
    the HIP path through the C ABI, world_size virtual ranks on the loopback hub: dispatch -> expert = _act -> combine,
    then the reference's checks (per-expert counts, padded groups, token membership, out == ref_out_tokens within
    torch's assert_close tolerance) - and the same outputs bit-for-bit against oracle/ep_ref.py."""
    import ep_golden
    import torch
    from oracle.bf16 import bf16_round
    case = ep_golden.load(cid)
    W, E, H, T, topk, pad, Hs = (case[k] for k in ("world", "E", "H", "T", "topk", "pad", "Hs"))
    in_dt = torch.float32 if case["in_el"] == 4 else torch.bfloat16
    out_dt = torch.float32 if case["out_el"] == 4 else torch.bfloat16
    epr = E // W
    # test_p2p_all_to_all.py:86 sizes it max_num_tokens * num_local_experts * num_dp_groups; + one padding tail per expert
    max_recv = max(T * epr * W, T * topk * W + epr * pad)

    def dev(a, dt):
        return torch.from_numpy(ep_golden.as_f32(a)).to(dt).cuda()

    xs = [dev(d["dp_x"], in_dt) for d in case["ranks"]]
    idx = [torch.from_numpy(d["indices"].astype(np.int32)).cuda() for d in case["ranks"]]
    ws = [torch.from_numpy(d["weights"]).cuda() for d in case["ranks"]]
    sc = [torch.from_numpy(d["dp_x_scale"]).cuda() for d in case["ranks"]] if Hs else [None] * W
    eps, hub = _mk_eps(W, H, T, max_recv, E, topk, expert_padding=pad, hidden_scale=Hs, in_elemsize=case["in_el"],
                       out_elemsize=case["out_el"])
    try:
        for r in range(W):
            eps[r].dispatch_send(xs[r], idx[r], ws[r], x_scale=sc[r])
        ref = ep_ref.dispatch([ep_golden.as_f32(d["dp_x"]) for d in case["ranks"]], [d["indices"] for d in case["ranks"]], E,
                              expert_padding=pad, scales=[d["dp_x_scale"] for d in case["ranks"]] if Hs else None)
        ys = []
        for r in range(W):
            got = eps[r].dispatch_recv(dtype=in_dt, with_scale=bool(Hs))
            rows, cnt = got[0].float().cpu().numpy(), got[1].cpu().numpy()
            extent = ep_golden.check_dispatch(case, r, cnt, rows)                          # the reference's checks
            assert extent == eps[r].num_padded_recv_tokens() == len(ref[r][0])
            assert np.array_equal(rows[:extent], ref[r][0]) and np.array_equal(cnt.astype(np.uint32), ref[r][1])
            srows = None
            if Hs:
                srows = got[2].cpu().numpy()
                assert np.array_equal(srows[:extent], ref[r][3])                            # scale planes rode along
            y = ep_golden.act(rows, srows)                                                  # expert_y = _act(...).to(out_dtype)
            ys.append(torch.from_numpy(bf16_round(y) if case["out_el"] == 2 else y).to(out_dt).cuda())
        for r in range(W):
            eps[r].combine_send(ys[r])
        want = ep_ref.combine_f32([y.float().cpu().numpy()[:len(ref[r][0])] for r, y in enumerate(ys)],
                                  [ref[r][2] for r in range(W)], [d["weights"] for d in case["ranks"]], [T] * W, H)
        for r in range(W):
            out = torch.full((T, H), 7.0, dtype=out_dt, device="cuda")
            eps[r].combine_recv(out, idx[r], ws[r])
            got = out.float().cpu().numpy()
            ep_golden.check_combine(case, r, got)                                           # == ref_out_tokens (reference bar)
            exp = bf16_round(want[r]) if case["out_el"] == 2 else want[r]
            assert np.array_equal(got, exp), (r, np.abs(got - exp).max())                   # == the oracle, every bit
    finally:
        _close_eps(eps, hub)


def test_ep_errors_are_reported_not_hung(built_libs):
    """protocol order and capacity errors return -1 with a message (the loopback combine_recv must see every
    peer's combine_send of THIS round; an overflow of max_recv_tokens is an error, not a hang)"""
    import torch
    eps, hub = _mk_eps(2, 64, 4, 3, 4, 2)                       # max_recv 3 rows: rank 0 will overflow
    try:
        x = torch.ones((4, 64), dtype=torch.bfloat16, device="cuda")
        idx = torch.zeros((4, 2), dtype=torch.int32, device="cuda")
        idx[:, 1] = 1                                             # every pair -> experts 0 / 1 = rank 0
        w = torch.ones((4, 2), dtype=torch.float32, device="cuda")
        with pytest.raises(RuntimeError, match="before dispatch_send"):
            eps[0].dispatch_recv()
        eps[0].dispatch_send(x, idx, w)
        with pytest.raises(RuntimeError, match="every virtual rank"):
            eps[0].dispatch_recv()
        eps[1].dispatch_send(x[:0], idx[:0], w[:0])
        with pytest.raises(RuntimeError, match="exceed max_recv_tokens"):
            eps[0].dispatch_recv()
        out, cnt = eps[1].dispatch_recv()
        assert int(cnt.sum()) == 0
        with pytest.raises(RuntimeError, match="before dispatch_recv"):
            eps[0].combine_send(out)
        eps[1].combine_send(out)
        with pytest.raises(RuntimeError, match="before combine_send"):
            eps[0].combine_recv(x.clone(), idx, w)
    finally:
        _close_eps(eps, hub)
