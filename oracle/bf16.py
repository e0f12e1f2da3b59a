"""bf16 helpers for the oracle (test infrastructure only).

The reference stores every activation / weight as bf16 (``Half = u16``,
pegainfer-kernels/src/ffi.rs:4) and converts with ``__float2bfloat16``
(round-to-nearest-even).  Oracle arrays are float32 arrays whose values are
exactly bf16-representable; ``bf16_bits`` gives the u16 image that crosses the
C ABI.
"""
import numpy as np


def bf16_round(x):
    """float32 -> nearest-even bf16, returned as float32 (``__float2bfloat16``)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    bias = np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    r = ((u + bias) & np.uint32(0xFFFF0000)).astype(np.uint32)
    nan = np.isnan(x)
    if nan.any():
        r = np.where(nan, np.uint32(0x7FC00000), r)
    return r.view(np.float32).reshape(x.shape)


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
