#!/usr/bin/env python3
"""CPU dry run of the GPU full-depth parity tests (tests/test_gpu_full_depth*.py) with STAND-IN engines.

A GPU minute is the scarce resource of this project (one pool box per call, a fixed budget per round): a shape slip or an
indexing bug in 500 lines of test harness must not cost one.  This script runs the very same test modules here, where there
is no GPU, against engines made of the ORACLE with every GEMM summed in two K halves (the reference's rounding points, another
summation order - "an honest engine"), on reduced-depth fixtures generated into a scratch directory:

    python tools/dry_run_full_depth.py [--layers 2] [--layers35 4] [--keep DIR]

It checks the harness - fixtures, indexing, feeds, taps, reports - NOT the HIP path; nothing here is collected by pytest
(tests/ never imports it) and the product path never sees the stand-ins.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_fixtures(out, layers, layers35):
    g = os.path.join(ROOT, "tests", "golden")
    run = lambda *a: subprocess.run([sys.executable, *a], check=True, cwd=ROOT)
    run(os.path.join(g, "make_qwen3_4b_depth_golden.py"), "--layers", str(layers), "--out", out, "--no-oracle")
    for ext in (".json", ".npz"):
        os.replace(os.path.join(out, "qwen3_4b_depth%d_hf%s" % (layers, ext)), os.path.join(out, "qwen3_4b_depth36_hf" + ext))
    run(os.path.join(g, "make_qwen3_4b_short_golden.py"), "--dir", out)
    run(os.path.join(g, "make_qwen35_4b_depth_golden.py"), "--layers", str(layers35), "--out", out, "--prompt", "200")
    for ext in (".json", ".npz"):
        os.replace(os.path.join(out, "qwen35_4b_depth%d_hf%s" % (layers35, ext)), os.path.join(out, "qwen35_4b_depth32_hf" + ext))


def install_stand_ins():
    from oracle import ops as O
    from oracle.bf16 import bf16_bits, bf16_from_bits, bf16_round
    from oracle.qwen3_ref import KvState, Qwen3Config, Qwen3Oracle
    from oracle.qwen3_ref import synthetic_weights as w3
    from oracle.qwen35_ref import Qwen35Config, Qwen35Oracle

    def gemm_two_halves(W, X):
        """another summation order at sgemm cost: the two K halves summed separately, then added in fp32"""
        h = W.shape[1] // 2
        return bf16_round(X[:, :h] @ W[:, :h].T + X[:, h:] @ W[:, h:].T)

    class fp64:   # (historical name) the stand-in's arithmetic: the reference's rounding points, a different summation order
        def __enter__(self):
            self.old, self.old_acc = O.gemm, O.GEMM_ACCUM
            O.gemm, O.GEMM_ACCUM = gemm_two_halves, np.float32

        def __exit__(self, *a):
            O.gemm, O.GEMM_ACCUM = self.old, self.old_acc

    class _Lib:
        def __init__(self, eng):
            self.eng = eng

        def pegainfer_qwen3_export_tensor(self, h, name, ptr, n):
            import ctypes
            src = self.eng.bits[name.decode()].ravel()
            ctypes.memmove(ptr, src.ctypes.data, n * 2)
            return 0

    class Qwen3Engine:
        def __init__(self, config, **kw):
            keys = ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
                    "intermediate_size", "vocab_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings"]
            self.cfgd = dict(config)
            self.cfg = Qwen3Config(**{k: config[k] for k in keys})
            self.h, self.lib, self.tap_on, self.last_taps, self.reqs = 1, _Lib(self), False, None, {}

        def load_state(self, bits):
            self.bits = bits
            self.orc = Qwen3Oracle(self.cfg, {k: bf16_from_bits(v) for k, v in bits.items()}, num_pages=512, rope_positions=4096)
            return self

        def fill_synthetic(self, seed=42, std=0.02):
            w = w3(self.cfg, seed=seed, std=std)
            return self.load_state({k: bf16_bits(v) for k, v in w.items()})

        def export_state(self):
            return self.bits

        def _chk(self, rc, what):
            assert rc == 0, what

        def new_request(self):
            rid = len(self.reqs)
            self.reqs[rid] = KvState()
            return rid

        def drop_request(self, rid):
            self.orc.free_pages = self.reqs[rid].pages + self.orc.free_pages
            del self.reqs[rid]

        def _run(self, fn):
            self.orc.taps = [] if self.tap_on else None
            with fp64():
                lg = fn()
            if self.tap_on:
                self.last_taps = np.stack(self.orc.taps)
            self.last = np.stack(lg)
            return self.last.argmax(-1).astype(np.int32), bf16_bits(self.last)

        def prefill(self, rids, prompts, return_logits=False):
            t, b = self._run(lambda: self.orc.batch_prefill([list(p) for p in prompts], [self.reqs[r] for r in rids]))
            return (t, b) if return_logits else t

        def decode(self, rids, toks, return_logits=False):
            t, b = self._run(lambda: list(self.orc.batch_decode([int(x) for x in toks], [self.reqs[r] for r in rids])))
            return (t, b) if return_logits else t

        def generate_greedy(self, prompt, n):
            rid = self.new_request()
            out = [int(self.prefill([rid], [prompt])[0])]
            for _ in range(n - 1):
                out.append(int(self.decode([rid], [out[-1]])[0]))
            self.drop_request(rid)
            return out

        def sample(self, column, T, k, p, r):
            keep = O.top_k_top_p_support(O.logits_to_probs(self.last[column], 1.0 / T), k, p)
            return int(np.flatnonzero(keep)[int(r * keep.sum()) % int(keep.sum())])

        def debug_hidden_enable(self, on=True):
            self.tap_on = bool(on)

        def debug_hidden(self, max_rows=64):
            return bf16_bits(self.last_taps[:, :max_rows])

        def close(self):
            self.orc = None

    class Qwen35Engine:
        def __init__(self, config, **kw):
            c = {k: v for k, v in config.items() if k != "max_position_embeddings"}
            self.cfg = Qwen35Config(**c)
            self.tap_on, self.reqs = False, {}

        def load_state(self, state):
            w = {k: (v if v.dtype == np.float32 else bf16_from_bits(v)) for k, v in state.items()}
            self.orc = Qwen35Oracle(self.cfg, w, num_pages=256, rope_positions=2048)
            return self

        def new_request(self):
            rid = len(self.reqs)
            self.reqs[rid] = self.orc.new_request()
            return rid

        def drop_request(self, rid):
            self.orc.free_pages = self.reqs[rid].pages + self.orc.free_pages
            del self.reqs[rid]

        def _run(self, fn):
            self.orc.taps = [] if self.tap_on else None
            with fp64():
                lg = fn()
            if self.tap_on:
                self.last_taps = np.stack(self.orc.taps)
            return lg

        def prefill(self, rid, tokens, want_logits=False):
            lg = self._run(lambda: self.orc.prefill(list(tokens), self.reqs[rid]))
            return (int(lg.argmax()), lg) if want_logits else int(lg.argmax())

        def decode(self, rids, tokens, want_logits=False):
            lg = self._run(lambda: self.orc.batch_decode([int(t) for t in tokens], [self.reqs[r] for r in rids]))
            out = lg.argmax(-1).astype(np.int32)
            return (out, lg) if want_logits else out

        def debug_hidden_enable(self, on=True):
            self.tap_on = bool(on)

        def debug_hidden(self, max_rows=8):
            return bf16_bits(self.last_taps[:, :max_rows])

        def close(self):
            self.orc = None

    import pegainfer_amd  # noqa: F401 - the package itself (no device code runs at import)
    for name, cls in (("pegainfer_amd.qwen3", Qwen3Engine), ("pegainfer_amd.qwen35", Qwen35Engine)):
        m = types.ModuleType(name)
        setattr(m, cls.__name__, cls)
        sys.modules[name] = m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--layers35", type=int, default=4)
    ap.add_argument("--keep", default=None, help="fixture directory to (re)use instead of a temporary one")
    ap.add_argument("-k", default=None)
    args = ap.parse_args()
    out = args.keep or tempfile.mkdtemp(prefix="depth_dry_")
    os.makedirs(out, exist_ok=True)
    if not os.path.exists(os.path.join(out, "qwen35_4b_depth32_hf.json")):
        make_fixtures(out, args.layers, args.layers35)
    os.environ["PEGAINFER_DEPTH_GOLD_DIR"] = out
    os.environ["PEGAINFER_DEPTH_DRY_LAYERS"] = str(args.layers)
    os.environ["GRAFT_REPO_ROOT"] = out                      # reports land in <out>/gpurun_out
    os.makedirs(os.path.join(out, "gpurun_out"), exist_ok=True)
    import torch
    torch.cuda.is_available = lambda: True                   # conftest's skip rule
    install_stand_ins()
    import pytest
    t = os.path.join(ROOT, "tests")
    argv = [os.path.join(t, "test_gpu_full_depth.py"), os.path.join(t, "test_gpu_full_depth_qwen35.py"),
            os.path.join(t, "test_gpu_full_depth_8b.py"), os.path.join(t, "test_gpu_full_depth_batch.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--durations=12"]
    if args.k:
        argv += ["-k", args.k]
    rc = pytest.main(argv)
    print("reports:", os.path.join(out, "gpurun_out", "full_depth_parity.json"))
    if not args.keep:
        shutil.rmtree(out, ignore_errors=True)
    return int(rc)


if __name__ == "__main__":
    sys.exit(main())
