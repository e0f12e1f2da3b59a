"""GPU, >= 2 devices: Qwen3 tensor parallel INSIDE the C++ runtime (weights.rs:121-291 sharding, the two bf16
sum all-reduces per layer issued by csrc/host/qwen3_runtime.cpp on the model stream over RCCL) against the unsharded
engine on one device.  One process per GPU (torch.multiprocessing, 127.0.0.1 rendezvous, backend nccl == RCCL);
the unique id of the runtime's own communicator travels through torch.distributed (parallel.attach_tp).
Skips on a 1-GPU box; the driver's multi-GPU runs exercise it."""
import os
import socket
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
           intermediate_size=512, vocab_size=1024, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True,
           max_position_embeddings=4096)
PROMPT = [3 + (7 * i) % 1000 for i in range(90)]
N_DECODE = 4


def _bf16_bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _state(seed=11):
    rng = np.random.default_rng(seed)
    c, hd = CFG, CFG["head_dim"]
    H, I, V = c["hidden_size"], c["intermediate_size"], c["vocab_size"]
    t = {"model.embed_tokens.weight": rng.standard_normal((V, H)) * 0.05, "model.norm.weight": 1 + 0.1 * rng.standard_normal(H)}
    for l in range(c["num_hidden_layers"]):
        p = f"model.layers.{l}."
        t[p + "self_attn.q_proj.weight"] = rng.standard_normal((c["num_attention_heads"] * hd, H)) * 0.05
        t[p + "self_attn.k_proj.weight"] = rng.standard_normal((c["num_key_value_heads"] * hd, H)) * 0.05
        t[p + "self_attn.v_proj.weight"] = rng.standard_normal((c["num_key_value_heads"] * hd, H)) * 0.05
        t[p + "self_attn.o_proj.weight"] = rng.standard_normal((H, c["num_attention_heads"] * hd)) * 0.05
        t[p + "self_attn.q_norm.weight"] = 1 + 0.1 * rng.standard_normal(hd)
        t[p + "self_attn.k_norm.weight"] = 1 + 0.1 * rng.standard_normal(hd)
        t[p + "mlp.gate_proj.weight"] = rng.standard_normal((I, H)) * 0.05
        t[p + "mlp.up_proj.weight"] = rng.standard_normal((I, H)) * 0.05
        t[p + "mlp.down_proj.weight"] = rng.standard_normal((H, I)) * 0.05
        t[p + "input_layernorm.weight"] = 1 + 0.1 * rng.standard_normal(H)
        t[p + "post_attention_layernorm.weight"] = 1 + 0.1 * rng.standard_normal(H)
    return {k: _bf16_bits(v) for k, v in t.items()}


def _run(eng):
    from oracle.bf16 import bf16_from_bits
    rid = eng.new_request()
    tok, lg = eng.prefill([rid], [PROMPT], return_logits=True)
    rows = [bf16_from_bits(lg[0])]
    for _ in range(N_DECODE):
        tok, lg = eng.decode([rid], tok, return_logits=True)
        rows.append(bf16_from_bits(lg[0]))
    return np.stack(rows)


def _tp_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    from pegainfer_amd import parallel
    from pegainfer_amd.qwen3 import Qwen3Engine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        for mode in (0, 1):
            eng = Qwen3Engine(parallel.tp_local_config(CFG, world), num_kv_pages=64, max_batch_size=2, decode_mode=mode,
                              device=rank)
            eng.load_state(parallel.shard_qwen3_state(_state(), CFG, rank, world))
            assert parallel.attach_tp(eng) == (rank, world)
            rows = _run(eng)
            eng.close()
            if rank == 0:
                np.save(out_path + f".mode{mode}.npy", rows)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_tp2_runtime_matches_unsharded_engine(built_libs):
    import torch
    import torch.multiprocessing as mp
    from pegainfer_amd.qwen3 import Qwen3Engine
    eng = Qwen3Engine(CFG, num_kv_pages=64, max_batch_size=2, decode_mode=1).load_state(_state())
    ref = _run(eng)            # also on a 1-GPU box: the synthetic checkpoint loads and steps
    eng.close()
    assert np.isfinite(ref).all() and ref.shape == (1 + N_DECODE, CFG["vocab_size"])
    if torch.cuda.device_count() < 2:
        pytest.skip("tensor parallel needs >= 2 visible GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "tp2")
        mp.spawn(_tp_worker, args=(2, port, path), nprocs=2, join=True)
        for mode in (0, 1):
            got = np.load(path + f".mode{mode}.npy")
            cos = (got * ref).sum(-1) / np.linalg.norm(got, axis=-1) / np.linalg.norm(ref, axis=-1)
            # partial sums are rounded to bf16 before the all-reduce: a few ulp on the logits (same bar as the
            # world-size-2 oracle test, tests/test_parallel_gloo.py)
            assert cos.min() > 0.9995 and np.abs(got - ref).max() <= 0.5, (mode, cos.min(), np.abs(got - ref).max())
