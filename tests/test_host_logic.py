"""CPU-only tests of the pure-host pieces of libpegainfer_qwen3.so against the oracle / the
reference's own unit tests (pegainfer-core/src/page_pool.rs:124-199, kv_pool.rs:290-310,
batch_decode_buffers.rs)."""
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
    missing = sorted(set(ffi.declared_symbols("pegainfer_qwen3.h")) - exported)
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
