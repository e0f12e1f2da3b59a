"""bf16 helpers for the oracle (test infrastructure only).

The reference stores every activation / weight as bf16 (``Half = u16``,
pegainfer-kernels/src/ffi.rs:4) and converts with ``__float2bfloat16``
(round-to-nearest-even).  Oracle arrays are float32 arrays whose values are
exactly bf16-representable; ``bf16_bits`` gives the u16 image that crosses the
C ABI.
"""
import contextlib

import numpy as np

import threading

_EXACT = threading.local()   # per THREAD (round 6): the bf16 and the fp32-truth pass of one model run side by side (oracle/parity.py)


def _exact_on():
    return getattr(_EXACT, "on", False)


@contextlib.contextmanager
def exact_activations():
    """The "fp32 truth" pass of the parity tests: inside this context - in THIS thread - ``bf16_round`` is
    the identity, so an oracle built AND run here evaluates the same DAG on the same bf16 weights with no activation rounding at
    all (fp32 storage, fp32 / fp64 accumulation, unrounded RoPE tables).  It is what both the bf16 oracle and the HIP path
    approximate; ``err(engine vs truth) <= c * err(bf16 oracle vs truth)`` separates summation-order noise (c ~ 1) from a
    wrong or missing rounding point / a wrong kernel route (c >> 1) without a hand-picked tolerance.  Checkpoints must be
    generated OUTSIDE the context (weights stay bf16-valued)."""
    old = _exact_on()
    _EXACT.on = True
    try:
        yield
    finally:
        _EXACT.on = old


_NATIVE = [None, False]   # (ctypes function or None, looked for already)
_NATIVE_MIN = 1 << 16     # elements from which the C helper pays


def _native_round():
    """oracle/_native/libpegainfer_oracle.so (oracle/bf16_round.c, built by `make -C oracle`): one-pass, 16-thread rounding.
    Absent library -> None (the numpy statement below is then used for every size)."""
    if not _NATIVE[1]:
        _NATIVE[1] = True
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_native", "libpegainfer_oracle.so")
        if os.path.exists(path):
            try:
                fn = ctypes.CDLL(path).pegainfer_oracle_bf16_round
                fn.restype = None
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
                _NATIVE[0] = (fn, max(1, min(16, os.cpu_count() or 1)))
            except (OSError, AttributeError):
                _NATIVE[0] = None
    return _NATIVE[0]


def _bf16_round_numpy(x):
    u = x.view(np.uint32)
    bias = np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    r = ((u + bias) & np.uint32(0xFFFF0000)).astype(np.uint32)
    nan = np.isnan(x)
    if nan.any():
        r = np.where(nan, np.uint32(0x7FC00000), r)
    return r.view(np.float32).reshape(x.shape)


def bf16_round(x):
    """float32 -> nearest-even bf16, returned as float32 (``__float2bfloat16``).  The statement is `_bf16_round_numpy`; arrays
    of >= 64 K elements go through the one-pass C helper (oracle/bf16_round.c, same arithmetic, pinned against the numpy form
    by tests/test_oracle_kats.py) when it has been built - a full-depth oracle pass rounds ~3 G elements.  (Round 6 first tried
    torch's threaded conversion: 5 x faster on 8 cores, 30-75 % SLOWER passes on the 256-core GPU hosts; and a chunked numpy
    thread pool: no faster, the five numpy passes are memory-bound.)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if _exact_on():
        return x
    if x.size >= _NATIVE_MIN:
        nat = _native_round()
        if nat is not None:
            out = np.empty_like(x)
            nat[0](x.ctypes.data, out.ctypes.data, x.size, nat[1])
            return out
    return _bf16_round_numpy(x)


def bf16_bits(x):
    """float32 (already bf16-representable, or to be rounded) -> uint16 image."""
    r = bf16_round(x)
    return (r.view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def bf16_from_bits(u):
    """uint16 image -> float32."""
    u = np.ascontiguousarray(u, dtype=np.uint16)
    return (u.astype(np.uint32) << np.uint32(16)).view(np.float32).reshape(u.shape)


def bf16_bits_exact(x):
    """uint16 image of a float32 array whose values are ALREADY bf16-representable (weights out of synthetic_weights,
    oracle activations): a shift, no rounding pass - 4 G elements in seconds instead of minutes."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return (x.view(np.uint32) >> np.uint32(16)).astype(np.uint16).reshape(x.shape)
