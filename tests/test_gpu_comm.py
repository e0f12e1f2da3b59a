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


def _run_ep(world, E, topk, H, Ts, seed, use_comm=None):
    import torch
    from pegainfer_amd import ffi
    from pegainfer_amd.parallel import EpEndpoint
    rng = np.random.default_rng(seed)
    xs = [torch.tensor(rng.standard_normal((t, H)), dtype=torch.float32).to(torch.bfloat16).cuda() for t in Ts]
    idx = [torch.tensor(np.stack([rng.permutation(E)[:topk] for _ in range(t)]) if t else np.zeros((0, topk)),
                        dtype=torch.int32).cuda() for t in Ts]
    ws = [torch.tensor(rng.random((t, topk)), dtype=torch.float32).cuda() for t in Ts]
    cap = sum(Ts) * topk + 1
    hub = None if use_comm is not None else EpEndpoint.hub(world)
    eps = [EpEndpoint(H, max(max(Ts), 1), cap, E, topk, comm=use_comm, hub=hub, rank=r) for r in range(world)]
    try:
        for r in range(world):
            eps[r].dispatch_send(xs[r], idx[r], ws[r])
        recv = [eps[r].dispatch_recv() for r in range(world)]
        ref = ep_ref.dispatch([x.float().cpu().numpy() for x in xs], [i.cpu().numpy() for i in idx], E)
        ys = []
        for r in range(world):
            rows, n = recv[r]
            assert n == len(ref[r][0])
            assert np.array_equal(rows.float().cpu().numpy(), ref[r][0])                   # payload rows: exact, in order
            assert np.array_equal(eps[r].tokens_per_expert(E // world), ref[r][1])
            ys.append((rows.float() * 2.0 + 1.0).to(torch.bfloat16))                       # the "expert"
        for r in range(world):
            eps[r].combine_send(ys[r])
        prev = [torch.full((t, H), 0.5, dtype=torch.bfloat16, device="cuda") for t in Ts]
        for acc in (False, True):
            outs = [eps[r].combine_recv(prev[r].clone(), idx[r], ws[r], accumulate=acc) for r in range(world)]
            want = ep_ref.combine([y.float().cpu().numpy() for y in ys], [ref[r][2] for r in range(world)],
                                  [w.cpu().numpy() for w in ws], list(Ts), H,
                                  prev=[p.float().cpu().numpy() for p in prev] if acc else None)
            for r in range(world):
                got = outs[r].float().cpu().numpy()
                tol = 2.0 ** -8 * np.abs(want[r]) + 1e-6                                   # one bf16 rounding of an f32 sum
                assert np.all(np.abs(got - want[r]) <= tol), (r, acc, np.abs(got - want[r]).max())
    finally:
        for e in eps:
            e.close()
        if hub:
            ffi.host_lib().pegainfer_ep_hub_destroy(hub)


@pytest.mark.parametrize("world,E,topk,H,Ts", [(4, 16, 2, 256, (5, 0, 7, 3)), (8, 64, 6, 512, (9, 1, 0, 4, 16, 2, 3, 5)),
                                               (2, 8, 8, 128, (3, 3)), (1, 4, 2, 64, (6,))])
def test_ep_dispatch_combine_loopback_matches_oracle(built_libs, world, E, topk, H, Ts):
    _run_ep(world, E, topk, H, Ts, seed=world * 100 + E)


def test_ep_dispatch_combine_over_comm_world1(comm):
    _run_ep(1, 8, 2, 256, (11,), seed=3, use_comm=comm)
