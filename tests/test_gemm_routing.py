"""Host logic of the prefill GEMM launcher (linear.hip: splitk_plan + tiled_route), pinned on CPU through
pegainfer_debug_gemm_route: which kernel each Qwen3-4B projection takes at the prompt lengths the bench and the GPU
parity tests run.  A threshold edit that silently moved a shape to another kernel would otherwise only show up as a
different (still correct) number - and as GPU tests that no longer exercise the kernel they were written for
(tests/test_gpu_ops.py::test_gemm_shapes, test_gpu_real_dims.py)."""
import ctypes

import pytest

QKV, O, DOWN, I = (6144, 2560), (2560, 4096), (2560, 9728), 9728


@pytest.fixture(scope="module")
def route():
    from pegainfer_amd import ffi
    lib = ffi.lib()

    def f(M, T, K, silu_I=0):
        out = (ctypes.c_int32 * 3)()
        assert lib.pegainfer_debug_gemm_route(M, T, K, silu_I, out) == 0
        return tuple(out)
    return f


def test_decode_family_is_not_routed_here(route):
    assert route(2560, 16, 4096)[0] == 0 and route(2560, 1, 4096)[0] == 0
    assert route(512, 40, 256)[0] == 0         # a small, short-K matrix at 17..64 columns stays on the skinny kernel


@pytest.mark.parametrize("T,qkv,o,down,gate_up", [
    # kind 1000 + tt = K-split plan on tt-token tiles (129: 128 x 256 kernel, 256: 256 x 256 kernel), (slices, K tiles per slice);
    # 2000 + row blocks = the same plan with its GEMM half on the stream kernel; 3000 + row blocks = the stream kernel un-split
    (17, (3002, 1, 0), (2005, 8, 8), (2005, 8, 19), (3006, 1, 0)),      # round 4: the whole 17..64-column layer on gemm_stream.h
    (64, (3002, 1, 0), (2005, 8, 8), (2005, 8, 19), (3006, 1, 0)),
    (65, (3002, 1, 0), (2005, 8, 8), (2005, 8, 19), (3006, 1, 0)),      # qkv un-split on 32-row tiles x ONE 128-token tile
    (128, (3002, 1, 0), (2005, 8, 8), (2005, 8, 19), (3006, 1, 0)),     # gate_up: 40 + 40-row SwiGLU tiles
    (129, (1064, 3, 14), (1064, 8, 8), (1064, 8, 19), (1281, 1, 0)),    # from 129 tokens on: the round-3 kernels
    (256, (23, 1, 0), (1064, 6, 11), (1064, 6, 26), (1281, 1, 0)),      # gate_up: SwiGLU form of the 128 x 256 kernel
    (512, (1129, 2, 20), (1129, 6, 11), (1129, 6, 26), (256, 1, 0)),    # gate_up: 152 tiles of 256 x 256, one round
    (1024, (1283, 1, 0), (1129, 3, 22), (1129, 3, 51), (257, 1, 64)),   # the headline TTFT shape; qkv: 256 tiles of 96 x 256 (round 6)
    (2048, (256, 1, 0), (1256, 3, 22), (1256, 3, 52), (257, 1, 64)),
    (768, (1283, 1, 0), (1129, 4, 16), (1129, 4, 38), (256, 1, 0)),     # round 6: qkv 192 tiles of 96 x 256 instead of 144 of 128 x 256
    (3072, (256, 1, 0), (1256, 2, 32), (1256, 2, 76), (256, 1, 0)),
    (4096, (256, 1, 0), (256, 1, 0), (256, 1, 0), (256, 1, 0)),
    (10000, (256, 1, 0), (256, 1, 0), (256, 1, 0), (256, 1, 0)),
])
def test_qwen3_4b_projection_routes(route, T, qkv, o, down, gate_up):
    assert route(*QKV[:1], T, QKV[1]) == qkv
    assert route(O[0], T, O[1]) == o
    assert route(DOWN[0], T, DOWN[1]) == down
    assert route(0, T, 2560, I) == gate_up


@pytest.mark.parametrize("T,qkv,o,down,gate_up", [
    # round 6, behind pegainfer_debug_streamk(1) / PEGAINFER_STREAMK=1 (default OFF: measured slower, gemm256.h): kind 258 = stream-K
    # over the 256 x 256 tiles wherever a round of tiles would leave >= 8 % of the CU-rounds idle and a team of t_tiles fits an XCD
    (512, (1129, 2, 20), (1129, 6, 11), (1129, 6, 26), (258, 1, 0)),    # gate_up: 152 tiles for 256 CUs
    (1024, (1283, 1, 0), (1129, 3, 22), (1129, 3, 51), (258, 1, 0)),    # 304 tiles (1.19 rounds)
    (2048, (258, 1, 0), (1256, 3, 22), (1256, 3, 52), (258, 1, 0)),     # qkv 192 tiles, gate_up 608 (2.4 rounds)
    (4096, (258, 1, 0), (258, 1, 0), (258, 1, 0), (256, 1, 0)),         # qkv 384 (1.5 rounds), o / down 160; gate_up 1216 = 4.75
    (10000, (256, 1, 0), (256, 1, 0), (256, 1, 0), (256, 1, 0)),        # 40 token tiles: a team would not fit an XCD (32 CUs)
])
def test_streamk_routes_behind_the_knob(route, T, qkv, o, down, gate_up):
    from pegainfer_amd import ffi
    ffi.lib().pegainfer_debug_streamk(1)
    try:
        assert route(*QKV[:1], T, QKV[1]) == qkv
        assert route(O[0], T, O[1]) == o
        assert route(DOWN[0], T, DOWN[1]) == down
        assert route(0, T, 2560, I) == gate_up
    finally:
        ffi.lib().pegainfer_debug_streamk(-1)
    assert route(0, T, 2560, I)[0] != 258          # off again: the default


def test_split_plans_cover_k_exactly(route):
    """every K-split plan walks all K tiles: (slices - 1) * per_slice < K / 64 <= slices * per_slice"""
    for (M, K) in (QKV, O, DOWN, (4096, 4096), (4096, 12288), (1024, 2560)):
        for T in (17, 40, 64, 65, 128, 200, 256, 512, 777, 1024, 1536, 2048, 3000):
            kind, ks, per = route(M, T, K)
            if kind in (1064, 1128, 1129, 1256) or 2000 < kind < 3000:   # the K-split plans
                nk = K // 64
                assert ks >= 2 and (ks - 1) * per < nk <= ks * per, (M, K, T, kind, ks, per)


def test_plain_gate_up_and_swiglu_gate_up_are_both_unsplit_stream_shapes(route):
    """gemm_silu fuses at 17..64 columns only where the plain GEMM over the same matrix is un-split too (same K order):
    the model shapes are; a narrow matrix whose plain form is K-split is not (linear.hip gemm_silu_impl)"""
    for I in (9728, 12288, 9216):
        for T in (17, 40, 64):
            assert route(0, T, 2560, I)[0] // 1000 == 3 and route(2 * I, T, 2560)[0] // 1000 == 3, (I, T)
    assert route(0, 40, 256, 512)[0] // 1000 == 3 and route(1024, 40, 256)[0] // 1000 != 3   # the case that needs the guard


def test_invalid_arguments(route):
    from pegainfer_amd import ffi
    out = (ctypes.c_int32 * 3)()
    assert ffi.lib().pegainfer_debug_gemm_route(0, 128, 2560, 0, out) != 0
    assert ffi.lib().pegainfer_debug_gemm_route(2560, 128, 2560, 0, None) != 0


@pytest.mark.parametrize("T", [3, 4, 8, 16])
def test_how_the_waves_of_the_3_to_16_column_gemm_meet(route, T):
    """Round 5 (gemm_skinny.h, skinny_flush_plan): on the resident-x kernel the 8 waves of a workgroup combine their partial
    sums per row block through lazy LDS tickets where a workgroup walks more than two row blocks (gate_up in both forms,
    lm_head) and with one barrier where it walks one or two (qkv, o_proj); down_proj's x (16 x 9728) does not fit in LDS from
    7 columns on and takes the tiled kernel.  out[1] = form, out[2] = partial buffers x 100 + rows per row block."""
    assert route(0, T, 2560, I)[:2] == (0, 5) and route(0, T, 2560, I)[2] == 413           # 749 row blocks of 13, ring of 4
    assert route(2 * I, T, 2560)[:2] == (0, 5) and route(2 * I, T, 2560)[2] == 416         # the unfused stacked launch
    assert route(151936, T, 2560)[:2] == (0, 5)                                            # lm_head
    assert route(*((QKV[0], T, QKV[1])))[1:] == (1, 212)                                   # 512 row blocks of 12: two per workgroup
    assert route(O[0], T, O[1])[1:] == (1, 210)                                            # 256 row blocks of 10: one per workgroup
    if T * DOWN[1] * 2 <= 128 * 1024:
        assert route(DOWN[0], T, DOWN[1])[1:] == (1, 210)
    else:
        assert route(DOWN[0], T, DOWN[1])[1] == -1                                         # tiled kernel
    # Qwen3-8B widths (hidden 4096): up to 15 columns a ring fits beside x; at 16 columns x is 128 KB and not even the second
    # 16 KB buffer (two weight sets) of the one-barrier form does: gate_up keeps the two-barrier form there, o_proj-like
    # single-set launches (8 KB buffers) get a ring of 3
    assert route(0, T, 4096, 12288)[1:] == ((5, 416) if T < 16 else (0, 116))
    assert route(40000, 16, 4096)[1:] == (5, 316)
    assert route(1, 1, 2560)[1] == -1 and route(6144, 2, 2560)[1] == -1                    # dot2 GEMV


def test_flush_form_override_hook(route):
    from pegainfer_amd import ffi
    lib = ffi.lib()
    try:
        for mode in (0, 1, 4, 5):
            lib.pegainfer_debug_skinny_flush(mode)
            assert route(0, 16, 2560, I)[1] == mode
        lib.pegainfer_debug_skinny_flush(4)
        assert route(0, 16, 4096, 12288)[1:] == (0, 116)       # no room for a ring nor for a second buffer: two barriers
        assert route(0, 15, 4096, 12288)[1:] == (4, 216)       # a ring of 2
    finally:
        lib.pegainfer_debug_skinny_flush(-1)
    assert route(0, 16, 2560, I)[1] == 5
