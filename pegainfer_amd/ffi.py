"""ctypes binding of the C ABI - the Python twin of the reference's
``pegainfer-kernels/src/ffi.rs`` (one ``extern "C"`` block, raw device pointers, stream last).

Prototypes are parsed from ``include/*.h`` so the header stays the single source of truth;
``declared_symbols()`` is what the CPU test suite checks the shared objects against.
There is NO fallback: if the library cannot be loaded the import of ``lib()`` raises.
"""
import ctypes
import os
import re

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
INCLUDE = os.path.join(ROOT, "include")
# PEGAINFER_LIB_DIR: an alternative build of the two libraries (python -m pegainfer_amd.build --variant NAME -> pegainfer_amd/lib_NAME:
# same sources, extra hipcc flags) for same-box A/B runs; default pegainfer_amd/lib
LIBDIR = os.environ.get("PEGAINFER_LIB_DIR") or os.path.join(PKG, "lib")

_SCALARS = {
    "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint32_t": ctypes.c_uint32,
    "uint64_t": ctypes.c_uint64, "float": ctypes.c_float, "int": ctypes.c_int,
    "pegainfer_stream_t": ctypes.c_void_p, "pegainfer_status_t": ctypes.c_int32,
    "pegainfer_qwen3_t": ctypes.c_void_p, "pegainfer_qwen35_t": ctypes.c_void_p, "size_t": ctypes.c_size_t,
    "pegainfer_sched_t": ctypes.c_void_p, "pegainfer_comm_t": ctypes.c_void_p, "pegainfer_ep_hub_t": ctypes.c_void_p,
    "pegainfer_ep_t": ctypes.c_void_p,
    "double": ctypes.c_double,
}

_PROTO = re.compile(r"^\s*([A-Za-z_][\w\s\*]*?)\s*\b([a-z_][a-z0-9_]*)\s*\(([^;{}]*)\)\s*;", re.M)


def _ctype(decl):
    decl = decl.strip()
    if decl == "void":
        return None
    if "char" in decl and "*" in decl and len(decl.replace("const", "").split()) == 1:
        return ctypes.c_char_p  # `const char*` return value (argument strings stay c_void_p-compatible)
    if "*" in decl:
        return ctypes.c_void_p
    base = decl.replace("const", "").split()
    # "int32_t name" or "int32_t"
    for tok in base:
        if tok in _SCALARS:
            return _SCALARS[tok]
    raise ValueError(f"unknown C type in header: {decl!r}")


def parse_header(path):
    """-> {symbol: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in _PROTO.finditer(text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or name in ("defined",):
            continue
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        out[name] = (_ctype(ret), argtypes)
    return out


def declared_symbols(header):
    return sorted(parse_header(os.path.join(INCLUDE, header)))


class _Lib:
    def __init__(self, so_name, *headers):
        path = os.path.join(LIBDIR, so_name)
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing - build it with `python -m pegainfer_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback for the product path.")
        self.path = path
        # PyTorch-ROCm bundles its own libamdhip64 / librccl.  If ours (linked against /opt/rocm) are mapped first
        # and torch is imported later, the process ends up with two HIP runtimes and aborts in their exit handlers
        # ("double free or corruption").  Import torch first when it is installed so both bind to one runtime;
        # nothing from torch is used here.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.cdll = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        self.protos = {}
        for header in headers:
            self.protos.update(parse_header(os.path.join(INCLUDE, header)))
        for name, (res, args) in self.protos.items():
            fn = getattr(self.cdll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


_cache = {}


def lib():
    """libpegainfer_kernels_hip.so (include/pegainfer_kernels.h)."""
    if "k" not in _cache:
        _cache["k"] = _Lib("libpegainfer_kernels_hip.so", "pegainfer_kernels.h", "pegainfer_kernels_ext.h")
        try:
            import torch
            has_gpu = torch.cuda.is_available()
        except ImportError:   # symbol/ABI checks work without torch; compute entry points need a device anyway
            has_gpu = False
        if has_gpu:   # a reference rank thread calls this before its first op (executor.rs:438-458)
            _cache["k"].cublas_init()
    return _cache["k"]


def host_lib():
    """libpegainfer_qwen3.so (include/pegainfer_qwen3.h, pegainfer_qwen35.h, pegainfer_scheduler.h, pegainfer_comm.h)."""
    if "h" not in _cache:
        lib()
        _cache["h"] = _Lib("libpegainfer_qwen3.so", "pegainfer_qwen3.h", "pegainfer_qwen35.h", "pegainfer_scheduler.h",
                           "pegainfer_comm.h")
    return _cache["h"]
