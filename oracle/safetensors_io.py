"""Minimal safetensors reader for the oracle/tests (TEST INFRASTRUCTURE ONLY).

Mirrors what the reference's loader does with the file
(pegainfer-core/src/weight_loader.rs:16-86: mmap, parse the JSON header, hand LE bf16
bytes to the device verbatim).  Returns float32 arrays (bf16-valued) or raw u16 bits.
"""
import json
import struct

import numpy as np

from .bf16 import bf16_from_bits


def load_safetensors(path, as_bits=False):
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        header = json.loads(f.read(n))
        blob = f.read()
    out = {}
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        lo, hi = meta["data_offsets"]
        raw = np.frombuffer(blob[lo:hi], dtype=np.uint8)
        if meta["dtype"] == "BF16":
            bits = raw.view(np.uint16).reshape(meta["shape"])
            out[name] = bits.copy() if as_bits else bf16_from_bits(bits)
        elif meta["dtype"] == "F32":
            out[name] = raw.view(np.float32).reshape(meta["shape"]).copy()
        else:
            raise ValueError(f"unsupported dtype {meta['dtype']} for {name}")
    return out
