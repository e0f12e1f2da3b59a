"""hipcc build driver for the native libraries (replaces the reference's
pegainfer-kernels/build.rs:1032-1353: nvcc + ar + cuBLAS/Triton-AOT with a plain
hipcc -> shared-object build; single target arch gfx950, no CUDA/HIP dual build).

  libpegainfer_kernels_hip.so  the drop-in C ABI of include/pegainfer_kernels.h   (csrc/*.hip)
  libpegainfer_qwen3.so        C++ host runtime: KV pool, decode buffers, hipGraph capture,
                               Qwen3 prefill/decode DAG                             (csrc/host/*.cpp)

Objects are cached by source mtime; `python -m pegainfer_amd.build` rebuilds what changed.
Also builds the oracle's C helper when oracle/Makefile exists (checker only).
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(CSRC, "host")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
ARCH = "gfx950"

KERNEL_LIB = os.path.join(LIBDIR, "libpegainfer_kernels_hip.so")
HOST_LIB = os.path.join(LIBDIR, "libpegainfer_qwen3.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _compile(src, extra, headers, objdir=None):
    obj = os.path.join(objdir or OBJDIR, os.path.basename(src) + ".o")
    if _newer([src] + headers, obj):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall",
              "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + extra +
             ["-c", src, "-o", obj])
    return obj


def build(verbose=False, force=False, variant=None, extra_flags=()):
    """variant: build the same sources with `extra_flags` into pegainfer_amd/lib_<variant> (objects in build_<variant>) for a
    same-box A/B against the default build (PEGAINFER_LIB_DIR selects it at load time); the default build takes no flags."""
    libdir = LIBDIR if not variant else os.path.join(PKG, "lib_" + variant)
    objdir = OBJDIR if not variant else os.path.join(PKG, "build_" + variant)
    kernel_lib, host_lib = os.path.join(libdir, os.path.basename(KERNEL_LIB)), os.path.join(libdir, os.path.basename(HOST_LIB))
    extra_flags = list(extra_flags) if variant else []
    os.makedirs(libdir, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    if force:
        for f in os.listdir(objdir):
            if os.path.isfile(os.path.join(objdir, f)):
                os.remove(os.path.join(objdir, f))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    ksrcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    hsrcs, hheaders = [], []
    if os.path.isdir(HOST):
        hsrcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".cpp"))
        hheaders = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        kfut = [ex.submit(_compile, s, extra_flags, headers, objdir) for s in ksrcs]
        hfut = [ex.submit(_compile, s, ["-x", "hip", "-I", HOST] + extra_flags, headers + hheaders, objdir) for s in hsrcs]
        kobjs = [f.result() for f in kfut]
        hobjs = [f.result() for f in hfut]
    if _newer(kobjs, kernel_lib):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", kernel_lib] + kobjs)
    if hobjs and _newer(hobjs + [kernel_lib], host_lib):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", host_lib] + hobjs +
             ["-L", libdir, "-lpegainfer_kernels_hip", "-L", "/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN",
              "-Wl,-rpath,/opt/rocm/lib"])
    # the oracle's C helper (checker only - nothing in pegainfer_amd/ loads it): oracle/Makefile, plain gcc; building the checker
    # is not using it, and a missing gcc only means the oracle rounds in numpy
    if not variant and os.path.exists(os.path.join(ROOT, "oracle", "Makefile")) and shutil.which("make") and shutil.which("gcc"):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 and verbose:
            print("oracle helper not built:", r.stdout[-300:])
    if verbose:
        print("built", kernel_lib, host_lib if hobjs else "")
    return kernel_lib, (host_lib if hobjs else None)


if __name__ == "__main__":
    # python -m pegainfer_amd.build [--force] [--variant NAME -- extra hipcc flags ...]
    argv = sys.argv[1:]
    var, flags = None, []
    if "--variant" in argv:
        i = argv.index("--variant")
        var = argv[i + 1]
        flags = argv[argv.index("--") + 1:] if "--" in argv else []
    build(verbose=True, force="--force" in argv, variant=var, extra_flags=flags)
