"""Shared machinery of the full-depth parity tests (tests/test_gpu_full_depth*.py on the GPU, tests/test_depth_harness.py on
the CPU): the derived-bar arithmetic and the oracle / truth passes live in oracle/parity.py (bench.py's parity legs use
them too); here are the report files and the short-prompt workload.  Nothing here touches the device.
"""
import json
import os

import numpy as np

from oracle.parity import (AGG_MAX, STEP_MAX, assert_derived, cos_rows, derived, layer_curve, near_tie_ok,  # noqa: F401
                           oracle_mode, qwen3_pass, qwen3_pass_pair, qwen35_pass, qwen35_pass_pair, rms_rows)

HERE = os.path.dirname(os.path.abspath(__file__))


def report(name, payload, fname="full_depth_parity.json"):
    """measured numbers -> gpurun_out/<fname> (when that directory exists), keyed by test name"""
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(HERE)
    d = os.path.join(root, "gpurun_out")
    if not os.path.isdir(d):
        return
    fp = os.path.join(d, fname)
    cur = json.load(open(fp)) if os.path.exists(fp) else {}
    cur[name] = payload
    json.dump(cur, open(fp, "w"), indent=1)


def report_kv(name, key, value, fname="full_depth_parity.json"):
    """one key of a dict-valued entry (test durations etc.), merged into what is already there"""
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(HERE)
    fp = os.path.join(root, "gpurun_out", fname)
    cur = json.load(open(fp)) if os.path.exists(fp) else {}
    report(name, dict(cur.get(name, {}), **{key: value}), fname)


# ---------------------------------------------------------------- the short-prompt workload of the full-depth tests
SHORT_LENS = (1, 8, 16, 17, 48, 100, 128)   # one per prefill GEMM route of the round-4 runtime: dot2 GEMV (1), skinny MFMA
#   (8, 16: the 6-launch short-prompt layer), stream GEMM at 17..64 and 65..128 columns (17, 48 / 100, 128: split-K pairs,
#   the un-split stacked qkv at 128); the reference's golden prompts (test_data/Qwen3-4B.json) are all below 80 tokens
SHORT_HF = (48, 100)                        # the two with a committed HF Transformers fixture (30 new tokens each)
SHORT_SEED = 20260927


def short_prompts(vocab):
    """seeded random token ids, one prompt per length of SHORT_LENS (the same in the fixture generator and the tests)"""
    rng = np.random.default_rng(SHORT_SEED)
    return {n: [int(x) for x in rng.integers(0, vocab, n)] for n in SHORT_LENS}
