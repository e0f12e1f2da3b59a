"""Multi-GPU pieces of the hot path, one process per GPU over ``torch.distributed``
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What the reference does and where it lives here:

* **Replicas** - independent requests need no data-path collective: ``shard_requests`` deals request
  indices to ranks, ``max_over_ranks`` is the only collective (timing).  This is what ``bench.py --gpus N`` uses.
* **Qwen3 tensor parallel** (pegainfer-qwen3-4b/src/weights.rs:121-291): q/k/v/gate/up row-sharded by head /
  intermediate, o/down column-sharded, embeddings + norms + lm_head replicated, one bf16 sum all-reduce of
  ``[hidden x T]`` after O-proj and after down-proj (weights.rs:396-405).  ``shard_range`` / ``tp_local_config``
  / ``shard_qwen3_state`` produce the rank's weights; ``attach_tp`` hands an RCCL communicator to the C++
  runtime, which issues the all-reduces on the model stream (capturable in the decode hipGraph).
* **DeepSeek-V4 MP8 collective set** (pegainfer-deepseek-v4/src/runtime/collectives.rs:8-287,
  moe.rs:1327-1461, core.rs:560-609): ``Comm`` offers the same verbs - in-place all-reduce, the
  bf16->f32->all-reduce->bf16 hidden reduce, all-gather, reduce-scatter, the decode-MoE AG/RS pair on a
  separate comm stream fenced by events so shared-expert compute overlaps.  Payloads are tiny (16 KB - 128 KB
  at bs=1): latency-bound, so each verb is ONE collective call (no bucketing), and xGMI is point-to-point, so
  nothing here assumes a switch.
"""
import ctypes
import numpy as np


def shard_range(total, rank, world):
    """TensorParallel::shard_range (pegainfer-qwen3-4b/src/config.rs): equal contiguous shards."""
    if total % world != 0:
        raise ValueError(f"{total} is not divisible by tensor-parallel size {world}")
    n = total // world
    return rank * n, n


def tp_local_config(cfg, world):
    """Per-rank model shape: heads and MLP width divide by the TP size, hidden / vocab do not
    (config.rs local_* helpers).  8 kv heads => world in {1, 2, 4, 8}."""
    c = dict(cfg)
    for k in ("num_attention_heads", "num_key_value_heads", "intermediate_size"):
        if c[k] % world != 0:
            raise ValueError(f"{k}={c[k]} is not divisible by tensor-parallel size {world}")
        c[k] = c[k] // world
    return c


def shard_qwen3_state(tensors, cfg, rank, world):
    """Rank-local weights from a full HF state dict {name: 2-D/1-D array} (any dtype; slicing only).
    Row shard: q/k/v/gate/up.  Column shard: o/down.  Replicated: everything else (weights.rs:121-291)."""
    hd = cfg["head_dim"]
    q0, qn = shard_range(cfg["num_attention_heads"] * hd, rank, world)
    k0, kn = shard_range(cfg["num_key_value_heads"] * hd, rank, world)
    i0, inn = shard_range(cfg["intermediate_size"], rank, world)
    out = {}
    for name, w in tensors.items():
        if name.endswith("self_attn.q_proj.weight"):
            w = w[q0:q0 + qn]
        elif name.endswith("self_attn.k_proj.weight") or name.endswith("self_attn.v_proj.weight"):
            w = w[k0:k0 + kn]
        elif name.endswith("mlp.gate_proj.weight") or name.endswith("mlp.up_proj.weight"):
            w = w[i0:i0 + inn]
        elif name.endswith("self_attn.o_proj.weight"):
            w = w[:, q0:q0 + qn]
        elif name.endswith("mlp.down_proj.weight"):
            w = w[:, i0:i0 + inn]
        out[name] = np.ascontiguousarray(w)
    return out


def shard_requests(n_requests, rank, world):
    """Replica mode: request i runs on rank i % world (no collective on the data path)."""
    return [i for i in range(n_requests) if i % world == rank]


def max_over_ranks(seconds, device=None):
    """The bench contract's only collective: elapsed = MAX over ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def attach_tp(engine, group=None, force_comm=False):
    """Create the RCCL communicator of the C++ runtime: rank 0 draws the unique id, torch.distributed
    broadcasts its 128 bytes, every rank calls ncclCommInitRank inside pegainfer_qwen3_attach_tp.
    world == 1 attaches nothing unless force_comm (used by the single-GPU test to exercise RCCL + capture)."""
    import ctypes

    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    uid = (ctypes.c_ubyte * 128)()
    if world > 1:
        buf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            if engine.lib.pegainfer_qwen3_rccl_unique_id(ctypes.addressof(uid)) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
            buf = torch.tensor(list(uid), dtype=torch.uint8)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
        buf = buf.to(dev)
        dist.broadcast(buf, src=0, group=group)
        for i, b in enumerate(buf.cpu().tolist()):
            uid[i] = b
    elif force_comm:
        if engine.lib.pegainfer_qwen3_rccl_unique_id(ctypes.addressof(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
    ptr = ctypes.addressof(uid) if (world > 1 or force_comm) else None
    engine._chk(engine.lib.pegainfer_qwen3_attach_tp(engine.h, rank, world, ptr), "attach_tp")
    return rank, world


class Comm:
    """The reference's collective verbs (cudarc ``Comm`` + deepseek-v4 runtime/collectives.rs) over a
    torch.distributed process group.  Tensors are torch tensors on the group's device."""

    def __init__(self, group=None, comm_stream=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.comm_stream = comm_stream  # torch.cuda.Stream for the overlapped MoE AG/RS (state.rs:129)

    # -- Comm::all_reduce_in_place (TP hidden reduce, indexer scores, prefill MoE) --
    def all_reduce_in_place(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    # -- all_reduce_hidden_fp32_in_place (collectives.rs:123-184): bf16 -> f32, sum in f32, back to bf16 --
    def all_reduce_hidden_fp32_in_place(self, hidden_bf16):
        import torch
        f32 = self._cast(hidden_bf16, torch.float32)
        self.dist.all_reduce(f32, op=self.dist.ReduceOp.SUM, group=self.group)
        hidden_bf16.copy_(self._cast(f32, torch.bfloat16))
        return hidden_bf16

    # -- Comm::all_gather: [n] per rank -> [world * n], rank-major --
    def all_gather(self, t):
        import torch
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group)
        return out.view((self.world * t.shape[0],) + tuple(t.shape[1:]))

    # -- Comm::reduce_scatter: [world * n] per rank -> this rank's summed [n] --
    def reduce_scatter(self, t):
        import torch
        n = t.shape[0] // self.world
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.dist.reduce_scatter_tensor(out.view(-1), t.contiguous().view(-1), op=self.dist.ReduceOp.SUM,
                                        group=self.group)
        return out

    # -- expert-parallel token exchange (pegainfer-comm 4-stage dispatch/combine, a2a_*.cu, re-expressed as one
    #    RCCL all-to-all): `send` holds world equal slabs, slab r goes to rank r; returns what every rank sent us --
    def all_to_all(self, send):
        import torch
        out = torch.empty_like(send)
        self.dist.all_to_all_single(out.view(-1), send.contiguous().view(-1), group=self.group)
        return out

    # ragged form (tokens routed per expert rank): send_counts[r] rows go to rank r; returns (rows, recv_counts)
    def all_to_allv(self, send, send_counts):
        import torch
        sc = torch.tensor(list(send_counts), dtype=torch.int64, device=send.device)
        rc = torch.empty_like(sc)
        self.dist.all_to_all_single(rc, sc, group=self.group)
        recv_counts = [int(x) for x in rc.tolist()]
        width = 1
        for d in send.shape[1:]:
            width *= int(d)
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        self.dist.all_to_all_single(out.view(-1), send.contiguous().view(-1),
                                    output_split_sizes=[c * width for c in recv_counts],
                                    input_split_sizes=[int(c) * width for c in send_counts], group=self.group)
        return out, recv_counts

    # -- all_gather_logits (core.rs:560-609): vocab/world f32 per rank -> full vocab --
    def all_gather_logits(self, local_logits_f32):
        return self.all_gather(local_logits_f32)

    # -- decode MoE AG/RS (moe.rs:1327-1461): gather every rank's tokens, run local experts on all of them
    #    (`expert_fn`), reduce-scatter the f32 partials; `shared_fn` (shared expert) overlaps with the gather --
    def moe_all_gather_reduce_scatter(self, hidden_bf16, expert_fn, shared_fn=None):
        import torch
        if self.comm_stream is not None and hidden_bf16.is_cuda:
            ready = torch.cuda.Event()
            ready.record()                                   # compute stream -> comm stream fence
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                gathered = self.all_gather(hidden_bf16)
                done = torch.cuda.Event()
                done.record()
            # both tensors cross streams: tell the caching allocator, or a block could be handed out again while the
            # other stream still reads it
            hidden_bf16.record_stream(self.comm_stream)
            gathered.record_stream(torch.cuda.current_stream())
            shared = shared_fn(hidden_bf16) if shared_fn else None   # overlaps with the all-gather
            torch.cuda.current_stream().wait_event(done)
        else:
            gathered = self.all_gather(hidden_bf16)
            shared = shared_fn(hidden_bf16) if shared_fn else None
        partial_f32 = expert_fn(gathered)                    # [world*T, hidden] f32: this rank's experts only
        routed = self.reduce_scatter(partial_f32)            # [T, hidden] f32
        return routed if shared is None else routed + shared.float()

    @staticmethod
    def _cast(t, dtype):
        """bf16 <-> f32 around the collectives: the HIP cast kernels (deepseek_bf16_to_f32_cuda /
        deepseek_f32_to_bf16_cuda, ffi.rs:8-20) on the GPU; plain torch on the CPU test backend."""
        import torch
        if not t.is_cuda:
            return t.to(dtype)
        from . import ffi
        out = torch.empty(t.shape, dtype=dtype, device=t.device)
        s = torch.cuda.current_stream().cuda_stream
        if dtype == torch.float32:
            rc = ffi.lib().deepseek_bf16_to_f32_cuda(t.data_ptr(), out.data_ptr(), t.numel(), s)
        else:
            rc = ffi.lib().deepseek_f32_to_bf16_cuda(t.data_ptr(), out.data_ptr(), t.numel(), s)
        if rc != 0:
            raise RuntimeError(f"cast kernel failed with error {rc}")
        return out


class NativeComm:
    """The MP8 verbs on the C ABI of include/pegainfer_comm.h (csrc/host/comm.cpp): RCCL on the caller's HIP stream,
    fused bf16 <-> f32 casts around the f32 reduce, a comm stream with event fences.  Tensors are CUDA torch tensors
    (device memory only: torch is plumbing here).  The unique id is drawn on rank 0 and broadcast over the
    torch.distributed group that launched the ranks (collectives.rs:8-287, moe.rs:1327-1461, core.rs:560-609)."""

    def __init__(self, group=None, device=None, oneshot=True, peer_only=False):
        """oneshot: enable the <= 64 KB peer-access all-reduce (handles travel over RCCL itself).
        peer_only: NO RCCL communicator - only the one-shot path exists; the IPC handles travel over `group` (any
        backend, e.g. gloo).  That form also runs with several ranks on ONE device (RCCL refuses duplicate GPUs), which
        is how a single-GPU box exercises the cross-process protocol."""
        import torch
        import torch.distributed as dist
        from . import ffi
        self.lib = ffi.host_lib()
        self.dist, self.group = dist, group     # bootstrap / barrier only; no payload travels through torch
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.oneshot = False
        if peer_only:
            self.h = self.lib.pegainfer_comm_create_peer_only(self.device, self.rank, self.world)
            if not self.h:
                raise RuntimeError("pegainfer_comm_create_peer_only failed")
            if self.world > 1:
                mine = (ctypes.c_ubyte * 64)()
                self._chk(self.lib.pegainfer_comm_oneshot_handle(self.h, ctypes.addressof(mine)), "oneshot_handle")
                cpu = dist.get_backend(group) != "nccl"
                dev = "cpu" if cpu else torch.device("cuda", self.device)
                bufs = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(self.world)]
                dist.all_gather(bufs, torch.tensor(list(mine), dtype=torch.uint8, device=dev), group=group)
                allh = (ctypes.c_ubyte * (64 * self.world))(*[b for t in bufs for b in t.cpu().tolist()])
                self._chk(self.lib.pegainfer_comm_oneshot_attach(self.h, ctypes.addressof(allh)), "oneshot_attach")
                dist.barrier(group=group)       # every slab zeroed and mapped everywhere before the first push
                self.oneshot = bool(self.lib.pegainfer_comm_oneshot_active(self.h))
            return
        uid = (ctypes.c_ubyte * 128)()
        if self.world > 1:
            buf = torch.zeros(128, dtype=torch.uint8)
            if self.rank == 0:
                if self.lib.pegainfer_comm_unique_id(ctypes.addressof(uid)) != 0:
                    raise RuntimeError("ncclGetUniqueId failed")
                buf = torch.tensor(list(uid), dtype=torch.uint8)
            dev = torch.device("cuda", self.device) if dist.get_backend(group) == "nccl" else "cpu"
            buf = buf.to(dev)
            dist.broadcast(buf, src=0, group=group)
            for i, b in enumerate(buf.cpu().tolist()):
                uid[i] = b
        self.h = self.lib.pegainfer_comm_create(self.device, self.rank, self.world,
                                                ctypes.addressof(uid) if self.world > 1 else None)
        if not self.h:
            raise RuntimeError("pegainfer_comm_create failed")
        if oneshot and self.world > 1:
            # best effort: without peer access (or IPC) the verbs simply stay on RCCL; the reason is kept
            if self.lib.pegainfer_comm_oneshot_enable(self.h) == 0:
                self.oneshot = bool(self.lib.pegainfer_comm_oneshot_active(self.h))
            else:
                self.oneshot_error = self.lib.pegainfer_comm_last_error(self.h).decode()

    def oneshot_status(self):
        """0, or 0x100 | mask of the ranks whose flag never arrived within the bound (synchronises the device)"""
        return self.lib.pegainfer_comm_oneshot_status(self.h)

    def close(self):
        if self.h:
            self.lib.pegainfer_comm_destroy(self.h)
            self.h = None

    def _s(self, stream=None):
        import torch
        return (stream or torch.cuda.current_stream()).cuda_stream

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.pegainfer_comm_last_error(self.h).decode()}")

    def all_reduce_in_place(self, t, stream=None):
        import torch
        if t.dtype == torch.bfloat16:
            self._chk(self.lib.pegainfer_comm_all_reduce_bf16(self.h, t.data_ptr(), t.numel(), self._s(stream)), "all_reduce_bf16")
        elif t.dtype == torch.float32:
            self._chk(self.lib.pegainfer_comm_all_reduce_f32(self.h, t.data_ptr(), t.numel(), self._s(stream)), "all_reduce_f32")
        else:
            raise TypeError(t.dtype)
        return t

    def all_reduce_hidden_fp32_in_place(self, hidden_bf16, stream=None):
        self._chk(self.lib.pegainfer_comm_all_reduce_bf16_via_f32(self.h, hidden_bf16.data_ptr(), hidden_bf16.numel(),
                                                                  self._s(stream)), "all_reduce_bf16_via_f32")
        return hidden_bf16

    def all_reduce_hidden_to_f32(self, hidden_bf16, out_f32, stream=None):
        self._chk(self.lib.pegainfer_comm_all_reduce_bf16_to_f32(self.h, hidden_bf16.data_ptr(), out_f32.data_ptr(),
                                                                 hidden_bf16.numel(), self._s(stream)), "all_reduce_bf16_to_f32")
        return out_f32

    def all_gather(self, t, stream=None):
        import torch
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._chk(self.lib.pegainfer_comm_all_gather(self.h, t.data_ptr(), out.data_ptr(), t.numel(), t.element_size(),
                                                     self._s(stream)), "all_gather")
        return out

    def reduce_scatter(self, t, stream=None):
        import torch
        n = t.shape[0] // self.world
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
        self._chk(self.lib.pegainfer_comm_reduce_scatter_f32(self.h, t.data_ptr(), out.data_ptr(), out.numel(),
                                                             self._s(stream)), "reduce_scatter_f32")
        return out

    def all_to_all(self, send, stream=None):
        import torch
        out = torch.empty_like(send)
        self._chk(self.lib.pegainfer_comm_all_to_all(self.h, send.data_ptr(), out.data_ptr(), send.numel() // self.world,
                                                     send.element_size(), self._s(stream)), "all_to_all")
        return out

    def all_gather_logits(self, local_logits_f32, stream=None):
        return self.all_gather(local_logits_f32, stream)

    def moe_all_gather_reduce_scatter(self, hidden_bf16, expert_fn, shared_fn=None):
        """moe.rs:1327-1461: the all-gather runs on the comm stream between two event fences while the compute
        stream runs the shared expert; the reduce-scatter of the f32 partials follows on the compute stream."""
        import torch
        cur = torch.cuda.current_stream()
        comm_stream = torch.cuda.ExternalStream(self.lib.pegainfer_comm_stream(self.h))
        gathered = torch.empty((self.world * hidden_bf16.shape[0],) + tuple(hidden_bf16.shape[1:]), dtype=hidden_bf16.dtype,
                               device=hidden_bf16.device)   # allocated on the compute stream: no cross-stream reuse
        self._chk(self.lib.pegainfer_comm_fence_in(self.h, cur.cuda_stream), "fence_in")
        self._chk(self.lib.pegainfer_comm_all_gather(self.h, hidden_bf16.data_ptr(), gathered.data_ptr(), hidden_bf16.numel(),
                                                     2, comm_stream.cuda_stream), "all_gather")
        hidden_bf16.record_stream(comm_stream)
        shared = shared_fn(hidden_bf16) if shared_fn else None
        self._chk(self.lib.pegainfer_comm_fence_out(self.h, cur.cuda_stream), "fence_out")
        routed = self.reduce_scatter(expert_fn(gathered))
        return routed if shared is None else routed + shared.float()


class EpTopology(ctypes.Structure):
    """pegainfer_ep_topology_t = EpTopology (ep_backend.rs:24-51), field for field"""
    _fields_ = [(n, ctypes.c_size_t) for n in ("world_size", "rank", "node_size", "dp_size", "num_experts",
                                                "num_experts_per_token", "hidden_dim", "hidden_dim_scale",
                                                "max_num_tokens", "max_recv_tokens", "max_private_tokens",
                                                "expert_padding")]


class EpDtypes(ctypes.Structure):
    """pegainfer_ep_dtypes_t = EpDtypes (ep_backend.rs:53-66)"""
    _fields_ = [("in_elemsize", ctypes.c_size_t), ("out_elemsize", ctypes.c_size_t), ("out_dtype", ctypes.c_int32),
                ("scale_elemsize", ctypes.c_size_t)]


SCALAR_BF16, SCALAR_F16, SCALAR_F32 = 0, 1, 2


class EpEndpoint:
    """One rank of the expert-parallel dispatch / combine on include/pegainfer_comm.h - the EpBackend surface of the
    reference (ep_backend.rs:213-331), argument for argument.  comm = NativeComm (RCCL transport) or hub = handle from
    EpEndpoint.hub(world) (loopback: virtual ranks in one process, for single-GPU tests of the multi-rank routing).
    Tensors are torch CUDA tensors; payload dtype bf16 or f32 (rows travel as opaque bytes)."""

    @staticmethod
    def hub(world):
        from . import ffi
        h = ffi.host_lib().pegainfer_ep_hub_create(world)
        if not h:
            raise RuntimeError("pegainfer_ep_hub_create failed")
        return h

    def __init__(self, hidden, max_tokens, max_recv, num_experts, topk, comm=None, hub=None, rank=0, world=None,
                 expert_padding=1, hidden_scale=0, in_elemsize=2, out_elemsize=2, max_private_tokens=0):
        from . import ffi
        self.lib = ffi.host_lib()
        self.hidden, self.topk, self.max_recv, self.hidden_scale = hidden, topk, max_recv, hidden_scale
        self.world = comm.world if comm is not None else world
        if self.world is None:
            raise ValueError("loopback endpoints need world=")
        self.rank = comm.rank if comm is not None else rank
        self.epr = num_experts // self.world
        self.in_el, self.out_el = in_elemsize, out_elemsize
        topo = EpTopology(self.world, self.rank, self.world, 1, num_experts, topk, hidden, hidden_scale, max_tokens, max_recv,
                          max_private_tokens, expert_padding)
        dt = EpDtypes(in_elemsize, out_elemsize, SCALAR_F32 if out_elemsize == 4 else SCALAR_BF16, 4 if hidden_scale else 0)
        self.h = self.lib.pegainfer_ep_create(comm.h if comm is not None else None, hub, ctypes.addressof(topo),
                                              ctypes.addressof(dt))
        if not self.h:
            raise RuntimeError("pegainfer_ep_create failed")

    def close(self):
        if self.h:
            self.lib.pegainfer_ep_destroy(self.h)
            self.h = None

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.pegainfer_ep_last_error(self.h).decode()}")

    def _s(self):
        import torch
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def _p(t):
        return t.data_ptr() if t is not None else None

    def dispatch_send(self, x, indices, weights, x_scale=None, bound_m=None):
        """x [T, hidden] (row stride in BYTES goes over the ABI), indices i32 [T, topk], weights f32 [T, topk],
        x_scale f32 [T, hidden_scale] or None, bound_m = device i32[1] or None"""
        T = x.shape[0]
        self._chk(self.lib.pegainfer_ep_dispatch_send(
            self.h, T, x.data_ptr(), (x.stride(0) if T else self.hidden) * x.element_size(),
            self._p(x_scale), x_scale.stride(1) if x_scale is not None else 0, x_scale.stride(0) if x_scale is not None else 0,
            indices.data_ptr(), indices.stride(0) if T else self.topk, weights.data_ptr(), weights.stride(0) if T else self.topk,
            self._p(bound_m), self._s()), "dispatch_send")

    def dispatch_recv(self, dtype=None, with_scale=False):
        """-> (out_x [max_recv, hidden] in the padded expert-major layout, tokens_per_expert device i32 [E / world]
        [, out_x_scale f32 [max_recv, hidden_scale]])"""
        import torch
        dtype = dtype or (torch.float32 if self.in_el == 4 else torch.bfloat16)
        out = torch.zeros((self.max_recv, self.hidden), dtype=dtype, device="cuda")
        cnt = torch.full((self.epr,), -1, dtype=torch.int32, device="cuda")
        sc = torch.zeros((self.max_recv, self.hidden_scale), dtype=torch.float32, device="cuda") if with_scale else None
        self._chk(self.lib.pegainfer_ep_dispatch_recv(self.h, cnt.data_ptr(), out.data_ptr(), self.hidden * out.element_size(),
                                                      self._p(sc), 1 if with_scale else 0, self.hidden_scale if with_scale else 0,
                                                      self._s()), "dispatch_recv")
        return (out, cnt, sc) if with_scale else (out, cnt)

    def num_recv_tokens(self):
        return self.lib.pegainfer_ep_num_recv_tokens(self.h)

    def num_padded_recv_tokens(self):
        return self.lib.pegainfer_ep_num_padded_recv_tokens(self.h)

    def tokens_per_expert(self):
        """host copy of the per-local-expert row counts of the last dispatch_recv (the reference reads its device
        counter with a D2H copy; the device pointer is pegainfer_ep_tokens_per_expert_ptr)"""
        import numpy as np
        buf = (ctypes.c_uint32 * self.epr)()
        self._chk(self.lib.pegainfer_ep_tokens_per_expert_host(self.h, ctypes.addressof(buf), self.epr), "tokens_per_expert")
        return np.frombuffer(buf, dtype=np.uint32).copy()

    def combine_send(self, expert_x):
        self._chk(self.lib.pegainfer_ep_combine_send(self.h, expert_x.data_ptr(), expert_x.stride(0) * expert_x.element_size(),
                                                     self._s()), "combine_send")

    def combine_recv(self, out_tokens, indices, weights, accumulate=False, bound_m=None, in_dtype=None):
        T = out_tokens.shape[0]
        if in_dtype is None:
            in_dtype = SCALAR_F32 if self.out_el == 4 else SCALAR_BF16
        self._chk(self.lib.pegainfer_ep_combine_recv(
            self.h, T, 0, in_dtype, out_tokens.data_ptr(), out_tokens.stride(0) if T else self.hidden, indices.data_ptr(),
            indices.stride(0) if T else self.topk, weights.data_ptr(), weights.stride(0) if T else self.topk, self._p(bound_m),
            int(bool(accumulate)), self._s()), "combine_recv")
        return out_tokens


def bench_mp8_collectives(comm, hidden=4096, token_counts=(1, 8, 32, 256, 4096), iters=10, device="cuda"):
    """BASELINE.json configs[4] (DeepSeek-V4 MP8, collectives only, synthetic): per-layer f32 all-reduce of
    [T, hidden], bf16 all-gather [T, hidden] -> x world, f32 reduce-scatter of [world*T, hidden]; average us per
    collective over the group (RCCL over xGMI when the group is nccl).  Returns {verb: {T: us}}."""
    import torch
    out = {"all_reduce_f32": {}, "all_gather_bf16": {}, "reduce_scatter_f32": {}, "all_to_all_bf16": {}}
    cuda = str(device).startswith("cuda")

    def timed(fn):
        fn()
        if cuda:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if comm.world > 1:
                comm.dist.barrier(group=comm.group)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters
        import time
        comm.dist.barrier(group=comm.group)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        return (time.perf_counter() - t0) * 1e6 / iters

    for T in token_counts:
        ar = torch.ones((T, hidden), dtype=torch.float32, device=device)
        ag = torch.ones((T, hidden), dtype=torch.bfloat16, device=device)
        rs = torch.ones((comm.world * T, hidden), dtype=torch.float32, device=device)
        out["all_reduce_f32"][str(T)] = round(timed(lambda: comm.all_reduce_in_place(ar)), 1)
        out["all_gather_bf16"][str(T)] = round(timed(lambda: comm.all_gather(ag)), 1)
        out["reduce_scatter_f32"][str(T)] = round(timed(lambda: comm.reduce_scatter(rs)), 1)
        a2a = torch.ones((comm.world * T, hidden), dtype=torch.bfloat16, device=device)   # EP dispatch: T rows per peer
        out["all_to_all_bf16"][str(T)] = round(timed(lambda: comm.all_to_all(a2a)), 1)
    # the small-message regime: Qwen3 TP decode 5 KB bf16 (weights.rs:396-405), DSV4 16 KB f32
    # (moe-tilelang-review.md:12), 64 KB = the one-shot path's upper edge, and the same payloads forced onto RCCL
    if cuda and hasattr(comm, "lib"):
        small = {}
        for name, n, dt in (("bf16_5KB", 2560, torch.bfloat16), ("f32_16KB", 4096, torch.float32),
                            ("bf16_64KB", 32768, torch.bfloat16)):
            t = torch.ones(n, dtype=dt, device=device)
            small[name] = round(timed(lambda: comm.all_reduce_in_place(t)), 1)
        out["all_reduce_small"] = small
        out["oneshot_active"] = bool(getattr(comm, "oneshot", False))
        if getattr(comm, "oneshot", False):
            out["oneshot_status"] = comm.oneshot_status()
        if getattr(comm, "oneshot_error", None):
            out["oneshot_error"] = comm.oneshot_error[:200]
    return out
