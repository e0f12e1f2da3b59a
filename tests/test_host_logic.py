"""CPU-only tests of the pure-host pieces of libpegainfer_qwen3.so against the oracle / the
reference's own unit tests (pegainfer-core/src/page_pool.rs:124-199, kv_pool.rs:290-310,
batch_decode_buffers.rs)."""
import os

import numpy as np
import pytest

from oracle import ops as O
from pegainfer_amd import ffi


@pytest.fixture(scope="module")
def H(built_libs):
    return ffi.host_lib()


def test_host_library_exports_every_declared_symbol(built_libs):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", built_libs[1]], stdout=subprocess.PIPE, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = []
    for header in ("pegainfer_qwen3.h", "pegainfer_qwen35.h", "pegainfer_scheduler.h", "pegainfer_comm.h"):
        declared += ffi.declared_symbols(header)
    assert "pegainfer_comm_all_reduce_bf16_via_f32" in declared and "pegainfer_ep_dispatch_send" in declared
    missing = sorted(set(declared) - exported)
    assert not missing, missing


def test_page_pool_order_and_all_or_nothing(H):
    pool = H.pegainfer_pagepool_create(6)
    try:
        buf = np.zeros(8, dtype=np.int32)
        assert H.pegainfer_pagepool_acquire(pool, 3, buf.ctypes.data) == 3
        assert buf[:3].tolist() == [0, 1, 2]                      # page_pool.rs:36: pops yield 0,1,2,...
        assert H.pegainfer_pagepool_available(pool) == 3
        assert H.pegainfer_pagepool_acquire(pool, 4, buf.ctypes.data) == -1   # all-or-nothing
        assert H.pegainfer_pagepool_available(pool) == 3
        rel = np.array([0, 1, 2], dtype=np.int32)
        H.pegainfer_pagepool_release(pool, rel.ctypes.data, 3)
        assert H.pegainfer_pagepool_available(pool) == 6
        assert H.pegainfer_pagepool_acquire(pool, 6, buf.ctypes.data) == 6
        assert buf[:6].tolist() == [0, 1, 2, 3, 4, 5]             # released pages come back in order
        assert H.pegainfer_pagepool_acquire(pool, 0, buf.ctypes.data) == 0
    finally:
        H.pegainfer_pagepool_destroy(pool)


def test_bucket_for(H):
    for bs in range(1, 65):
        assert H.pegainfer_bucket_for(bs) == O.bucket_for(bs)
    assert H.pegainfer_bucket_for(65) == -1


@pytest.mark.parametrize("lens,padded", [([1024], 1), ([20000, 5], 2), ([100], 1), ([1500, 40, 7, 3000], 4),
                                         ([4096] * 8, 8), ([1], 1)])
def test_reference_split_plan_matches_oracle(H, lens, padded):
    """policy 0 == BatchDecodeBuffers::sync_split_kv_meta + attention_path, bit for bit."""
    slots = padded * 64
    ri, kt = np.zeros(slots, np.int32), np.zeros(slots, np.int32)
    oi, va = np.zeros(padded + 1, np.int32), np.zeros(slots, np.uint8)
    chunk, use = np.zeros(1, np.int32), np.zeros(1, np.int32)
    L = np.asarray(lens, np.int32)
    n = H.pegainfer_split_kv_plan(0, len(lens), L.ctypes.data, padded, 8, ri.ctypes.data, kt.ctypes.data,
                                  oi.ctypes.data, va.ctypes.data, chunk.ctypes.data, use.ctypes.data)
    ref = O.split_kv_plan(lens, padded)
    assert n == ref["padded_slots"] and chunk[0] == ref["kv_chunk_size"]
    assert np.array_equal(ri, ref["request_indices"]) and np.array_equal(kt, ref["kv_tile_indices"])
    assert np.array_equal(oi, ref["o_indptr"]) and np.array_equal(va, ref["block_valid_mask"])
    assert bool(use[0]) == O.attention_path_is_split(padded, max(lens))


@pytest.mark.parametrize("lens,padded", [([1024], 1), ([10000], 1), ([100], 1), ([4096] * 8, 8), ([2000, 30], 2)])
def test_mi355x_split_policy_invariants(H, lens, padded):
    """policy 1 may pick any chunking, but must cover every token exactly once with <= 64 chunks."""
    slots = padded * 64
    ri, kt = np.zeros(slots, np.int32), np.zeros(slots, np.int32)
    oi, va = np.zeros(padded + 1, np.int32), np.zeros(slots, np.uint8)
    chunk, use = np.zeros(1, np.int32), np.zeros(1, np.int32)
    L = np.asarray(lens, np.int32)
    H.pegainfer_split_kv_plan(1, len(lens), L.ctypes.data, padded, 8, ri.ctypes.data, kt.ctypes.data,
                              oi.ctypes.data, va.ctypes.data, chunk.ctypes.data, use.ctypes.data)
    c = int(chunk[0])
    assert c % 16 == 0 and c >= 64
    for r, n in enumerate(lens):
        tiles = kt[oi[r]:oi[r + 1]]
        assert np.all(ri[oi[r]:oi[r + 1]] == r) and tiles.tolist() == list(range(len(tiles)))
        assert len(tiles) <= 64 and (len(tiles) - 1) * c < n <= len(tiles) * c
    assert va[:oi[len(lens)]].all() and not va[oi[len(lens)]:].any()


@pytest.mark.parametrize("lens,padded,chunk,chunks", [
    ([1024], 1, 64, [16]),            # 64-token chunks (the floor): unchanged by round 5
    ([2000], 1, 112, [18]),           # 2000 / 20 = 100 -> 112: one round of the 8 waves, 16-aligned as before
    ([4096], 1, 256, [16]),           # 4096 / 20 = 205 -> 208 -> whole rounds of 8 waves: 256
    ([10000], 1, 512, [20]),          # 10 000 / 20 = 500 -> 512: four balanced rounds (18 x 560 before)
    ([1030, 777], 2, 112, [10, 7]),   # two requests share the budget: 10 chunks each at most
    ([540, 300], 2, 64, [9, 5]),      # ... and below 10 x 64 tokens the chunks are the un-capped plan's 64
])
def test_mi355x_fused_form_plan_of_round_5(H, lens, padded, chunk, chunks):
    """the fused attention + o_proj plan (pegainfer_split_kv_plan's default): at most 20 chunks shared by <= 2 requests,
    chunks above 128 tokens rounded up to whole rounds of the 8-wave workgroup (profiles/r5_long_ctx_chunks_ab.txt)"""
    slots = padded * 64
    ri, kt = np.zeros(slots, np.int32), np.zeros(slots, np.int32)
    oi, va = np.zeros(padded + 1, np.int32), np.zeros(slots, np.uint8)
    c, use = np.zeros(1, np.int32), np.zeros(1, np.int32)
    L = np.asarray(lens, np.int32)
    n = H.pegainfer_split_kv_plan(1, len(lens), L.ctypes.data, padded, 8, ri.ctypes.data, kt.ctypes.data,
                                  oi.ctypes.data, va.ctypes.data, c.ctypes.data, use.ctypes.data)
    got = [int(oi[r + 1] - oi[r]) for r in range(len(lens))]
    assert int(c[0]) == chunk and got == chunks, (int(c[0]), got)
    assert sum(got) <= 20 and n == padded * (32 // padded) and use[0] == 1


def test_native_safetensors_reader_matches_python(tmp_path):
    """The C++ mmap reader (csrc/host/safetensors_loader.h: JSON header parser, single file / HF directory with
    model.safetensors.index.json shards) sees the same tensors, shapes, dtypes and bytes as a Python parse."""
    import ctypes
    import json
    import struct
    from pegainfer_amd import ffi
    lib = ffi.host_lib()
    golden = os.path.join(os.path.dirname(__file__), "golden")

    def py_parse(fp):
        with open(fp, "rb") as f:
            n = struct.unpack("<Q", f.read(8))[0]
            header = json.loads(f.read(n))
            blob = np.frombuffer(f.read(), dtype=np.uint8)
        return {k: (v["dtype"], v["shape"], blob[v["data_offsets"][0]:v["data_offsets"][1]])
                for k, v in header.items() if k != "__metadata__"}

    def probe(path, name):
        shape, ndim, dt = (ctypes.c_int64 * 4)(), ctypes.c_int32(), ctypes.create_string_buffer(8)
        bsum, nt = ctypes.c_uint64(), ctypes.c_int32()
        rc = lib.pegainfer_safetensors_probe(os.fsencode(path), name.encode(), ctypes.addressof(shape),
                                             ctypes.addressof(ndim), ctypes.addressof(dt), ctypes.addressof(bsum),
                                             ctypes.addressof(nt))
        return rc, list(shape[:ndim.value]), dt.value.decode(), bsum.value, nt.value

    def checksum(b):
        return int((b.astype(np.uint64) * (1 + (np.arange(b.size, dtype=np.uint64) & 0xFF))).sum(dtype=np.uint64))

    for fn in ("qwen3_tiny.safetensors", "qwen35_tiny.safetensors"):
        fp = os.path.join(golden, fn)
        ref = py_parse(fp)
        for name, (dtype, shape, data) in ref.items():
            rc, shp, dt, bs, nt = probe(fp, name)
            assert (rc, shp, dt, nt) == (0, shape, dtype, len(ref)) and bs == checksum(data), name
    assert probe(os.path.join(golden, "qwen3_tiny.safetensors"), "no.such.tensor")[0] == -2
    assert probe(str(tmp_path / "missing.safetensors"), "x")[0] == -1
    # sharded HF directory: two shard files + index
    ref = py_parse(os.path.join(golden, "qwen3_tiny.safetensors"))
    names = sorted(ref)
    d = tmp_path / "model"
    d.mkdir()
    weight_map = {}
    for si, part in enumerate((names[:5], names[5:])):
        header, blobs, off = {}, [], 0
        for nme in part:
            dtype, shape, data = ref[nme]
            header[nme] = {"dtype": dtype, "shape": shape, "data_offsets": [off, off + data.size]}
            blobs.append(data.tobytes())
            off += data.size
            weight_map[nme] = f"model-{si:05d}-of-00002.safetensors"
        hj = json.dumps(header, separators=(",", ":")).encode()
        with open(d / f"model-{si:05d}-of-00002.safetensors", "wb") as f:
            f.write(struct.pack("<Q", len(hj)) + hj + b"".join(blobs))
    (d / "model.safetensors.index.json").write_text(json.dumps({"metadata": {"total_size": 1}, "weight_map": weight_map}))
    for nme in (names[0], names[-1]):
        rc, shp, dt, bs, nt = probe(str(d), nme)
        assert (rc, shp, dt, nt) == (0, ref[nme][1], ref[nme][0], len(ref)) and bs == checksum(ref[nme][2])
