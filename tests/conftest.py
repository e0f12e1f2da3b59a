import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_libs():
    """Build (or reuse) the native libraries; CPU-only containers cross-compile gfx950."""
    from pegainfer_amd import build
    return build.build()


# ---- oracle <-> device helpers (the oracle works on float32 arrays holding bf16 values) ----
def to_dev(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is None and a.dtype == np.float32:
        t = t.to(torch.bfloat16)
    elif dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def from_dev(t):
    import torch
    if t.dtype == torch.bfloat16:
        return t.float().cpu().numpy()
    return t.cpu().numpy()


def bf16_ulp_diff(a, b):
    """max |a-b| measured in bf16 ulps of max(|a|,|b|) (0 where equal)."""
    from oracle.bf16 import bf16_bits
    ab, bb = bf16_bits(a).astype(np.int32), bf16_bits(b).astype(np.int32)
    # map sign-magnitude to a monotone integer line
    ab = np.where(ab & 0x8000, -(ab & 0x7FFF), ab)
    bb = np.where(bb & 0x8000, -(bb & 0x7FFF), bb)
    return int(np.abs(ab - bb).max()) if ab.size else 0
