"""CPU-only checks of the drop-in boundary: the shared objects build for gfx950, load, and
export every symbol include/*.h declares (no device compute is launched here); the host-side
plan helpers (pure integer) must equal the oracle bit for bit."""
import ctypes
import os
import subprocess

import pytest

from oracle import ops as O
from pegainfer_amd import ffi


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_kernel_library_exports_every_declared_symbol(built_libs):
    klib, _ = built_libs
    declared = ffi.declared_symbols("pegainfer_kernels.h") + ffi.declared_symbols("pegainfer_kernels_ext.h")
    assert len(declared) >= 38
    missing = sorted(set(declared) - _exported(klib))
    assert not missing, f"declared in include/pegainfer_kernels.h but not exported: {missing}"
    lib = ffi.lib()                      # ctypes load + prototype binding of every symbol
    assert all(hasattr(lib, s) for s in declared)


def test_reference_ffi_names_are_kept(built_libs):
    """The Qwen-path names of pegainfer-kernels/src/ffi.rs must exist verbatim (drop-in)."""
    must = ["rms_norm_cuda", "rms_norm_batched_cuda", "add_cuda", "fused_add_rms_norm_cuda",
            "fused_add_rms_norm_batched_cuda", "silu_mul_triton_aot_cuda", "embedding_batched_cuda",
            "embedding_batched_vocab_shard_cuda", "argmax_cuda", "flashinfer_top1_cuda",
            "gpu_sample_flashinfer_cuda", "gemm_cuda", "gemm_graphsafe_cuda", "embedding_decode_cuda",
            "silu_mul_fused_cuda", "cublas_init", "cublas_destroy", "cuda_set_device",
            "prefill_qk_norm_rope_only_cuda", "qk_norm_rope_batched_decode_cuda", "paged_kv_scatter_cuda",
            "batch_prefill_paged_num_tiles", "batch_prefill_paged_num_tiles_with_cta_tile_q",
            "batch_prefill_cta_tile_q", "batch_prefill_cta_tile_q_with_override", "batch_prefill_paged_cuda",
            "batch_prefill_paged_cuda_with_cta_tile_q", "single_prefill_cuda", "paged_attention_decode_cuda",
            "paged_attention_decode_split_kv_cuda", "rms_norm_batched_offset_cuda", "rms_norm_offset_cuda",
            "rms_norm_gated_cuda", "deepseek_bf16_to_f32_cuda", "deepseek_f32_to_bf16_cuda"]
    exp = _exported(built_libs[0])
    assert not [s for s in must if s not in exp]


def test_gfx950_code_object_present(built_libs):
    """The .so must carry a gfx950 code object (single target arch, no dual build)."""
    data = open(built_libs[0], "rb").read()
    assert b"gfx950" in data and b"sm_" not in data[:0]  # offload bundle names the target


@pytest.mark.parametrize("seq,hq,hkv,hd,ov", [(1, 32, 8, 128, 0), (4, 32, 8, 128, 0), (5, 32, 8, 128, 0),
                                             (16, 32, 8, 128, 0), (17, 32, 8, 128, 0), (10000, 32, 8, 128, 0),
                                             (10000, 32, 8, 128, 64), (100, 16, 4, 256, 0), (100, 32, 8, 128, 16),
                                             (100, 32, 8, 128, 128), (100, 32, 8, 128, 7), (1, 32, 32, 128, 0)])
def test_plan_helpers_match_oracle(built_libs, seq, hq, hkv, hd, ov):
    L = ffi.lib()
    assert L.batch_prefill_cta_tile_q_with_override(seq, hq, hkv, hd, ov) == O.batch_prefill_cta_tile_q(seq, hq, hkv, hd, ov)
    assert L.batch_prefill_paged_num_tiles_with_cta_tile_q(seq, hq, hkv, hd, ov) == O.batch_prefill_paged_num_tiles(seq, hq, hkv, hd, ov)
    if ov == 0:
        assert L.batch_prefill_cta_tile_q(seq, hq, hkv, hd) == O.batch_prefill_cta_tile_q(seq, hq, hkv, hd)
        assert L.batch_prefill_paged_num_tiles(seq, hq, hkv, hd) == O.batch_prefill_paged_num_tiles(seq, hq, hkv, hd)
