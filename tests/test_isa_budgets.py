"""Static register budgets of the hot gfx950 kernels, read from the compiler's kernel metadata (no GPU needed: hipcc
cross-compiles here).  Occupancy on CDNA4 is decided by the unified VGPR allocation (512 per SIMD lane: <= 128 -> 4
waves per SIMD, <= 168 -> 3, <= 256 -> 2) and a spill turns a register into scratch traffic inside the hot loop - both
have bitten this code base (a register-staged GEMV prologue: 184-214 VGPRs; a prefill-attention variant: AGPR copies), so
the budgets the measured configurations depend on are pinned here.  The assembly is cached under pegainfer_amd/build/isa."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pegainfer_amd", "csrc")
OUT = os.path.join(ROOT, "pegainfer_amd", "build", "isa")


def _hipcc():
    return os.environ.get("HIPCC") or shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def kernel_metadata(source):
    """{demangled-ish kernel name: {vgpr, agpr, vgpr_spill, sgpr_spill}} of one .hip file."""
    hipcc = _hipcc()
    if not hipcc:
        pytest.skip("hipcc not available")
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(CSRC, source)
    asm = os.path.join(OUT, source + ".s")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(asm) or any(os.path.getmtime(d) > os.path.getmtime(asm) for d in deps):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                        "-S", "--cuda-device-only", "-o", asm, src], check=True, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    text = open(asm).read()
    mangled = re.findall(r"\.name:\s+(\S+)", text)
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        pytest.skip("no C++ demangler (c++filt) available")
    names = subprocess.run([filt], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.split("\n")
    demangle = dict(zip(mangled, names))
    out, cur = {}, None
    for line in text.split("\n"):
        if re.match(r"^  - \.", line):
            cur = {}
        m = re.match(r"^\s+(?:- )?\.(agpr_count|name|sgpr_spill_count|vgpr_count|vgpr_spill_count):\s+(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "vgpr_spill_count" and "name" in cur:
                n = demangle.get(cur["name"], cur["name"])
                n = re.sub(r"^void\s+", "", n).split("(")[0].replace("pk::", "")
                out[n] = {"vgpr": int(cur.get("vgpr_count", 0)), "agpr": int(cur.get("agpr_count", 0)),
                          "vgpr_spill": int(cur["vgpr_spill_count"]), "sgpr_spill": int(cur.get("sgpr_spill_count", 0))}
    assert out, "no kernel metadata parsed from " + asm
    return out


def check(md, pattern, max_vgpr, at_least=1):
    hits = {k: v for k, v in md.items() if re.fullmatch(pattern, k)}
    assert len(hits) >= at_least, f"no kernel matches {pattern}: {sorted(md)[:8]}..."
    for k, v in hits.items():
        assert v["vgpr_spill"] == 0 and v["sgpr_spill"] == 0, f"{k} spills: {v}"
        assert v["vgpr"] <= max_vgpr, f"{k}: {v['vgpr']} VGPRs > {max_vgpr}"


def test_decode_gemv_and_gemm_kernels_keep_their_occupancy():
    md = kernel_metadata("linear.hip")
    # bs 1 decode: four 256-thread workgroups per CU (the grid sizing in gemv_launch_one counts on it)
    check(md, r"gemv_fused_kernel<1, \d, \d, \d, \d>", 128, at_least=6)      # incl. the whole-row-in-flight forms (U = 5)
    # bs 2: three per CU
    check(md, r"gemv_fused_kernel<2, \d, \d, \d, \d>", 168, at_least=6)
    # batched decode (3..16 columns): resident-x skinny MFMA kernels, two waves per SIMD, nothing in scratch
    check(md, r"skinny_resident_kernel<\d, \d>", 256, at_least=4)
    # prefill: 8-wave 256 x 256 tiles need two waves per SIMD; the 128-tile LDS-DMA kernels two workgroups per CU
    check(md, r"mfma_gemm256_kernel<(true|false)>", 256, at_least=2)
    # round 3: 128 x 256 tiles (plain / split-K and SwiGLU forms), 8 waves = two per SIMD, 64 accumulators
    check(md, r"mfma_gemm128x256_kernel<(true|false), 0, 4>", 256, at_least=2)
    check(md, r"mfma_gemm128x256_kernel<(true|false), 4, 4>", 168, at_least=2)    # 8 compute + 4 feeder waves: three per SIMD
    # round 6: 96-row plain tiles (qkv at 768 / 1024 tokens) and (32 + 32)-row SwiGLU tiles (the gate_up tail), feeder form
    check(md, r"mfma_gemm128x256_kernel<false, 4, 3>", 168)
    check(md, r"mfma_gemm128x256_kernel<true, 4, 2>", 168)
    check(md, r"mfma_gemm_glds_kernel<\d+, \d, (true|false)>", 256, at_least=6)
    check(md, r"splitk_reduce\w*kernel", 256, at_least=3)
    # round 4: the weight-streaming GEMM (one workgroup per CU: 4 compute + 3..6 feeder waves; 64- / 128-token tiles): spill-free
    check(md, r"stream_gemm_kernel<[1-6], 64, \d+, \d+, \d+, \d+, \d+>", 128, at_least=6)
    check(md, r"stream_gemm_kernel<[3-6], 128, \d+, \d+, \d+, \d+, \d+>", 168, at_least=4)


def test_attention_kernels_keep_their_occupancy():
    md = kernel_metadata("attn_decode.hip")
    # group sizes of the BASELINE models (Qwen3-4B / 8B: 4 query heads per kv head; Qwen3.5: 2 and 4 at head dim 256)
    check(md, r"fused_decode_attn_kernel<[124], (true|false), [48]>", 256, at_least=6)
    check(md, r"decode_attn_kernel<(128|256), [124], (true|false), [48]>", 256, at_least=24)
    md = kernel_metadata("attn_prefill.hip")
    # head dim 128: two workgroups per CU (the __launch_bounds__(256, 2) budget), no AGPR shuffling, no spills
    check(md, r"batch_prefill_paged_kernel<\d, 128, (true|false), \d, 4, (true|false)>", 256, at_least=6)
    # round 3: the LDS-DMA forms the launcher takes for power-of-two pages (128-row, paired 64-row and single 64-row tiles)
    check(md, r"batch_prefill_paged_kernel<[12], 128, true, [12], 4, true>", 256, at_least=3)
    for k, v in md.items():
        if re.fullmatch(r"batch_prefill_paged_kernel<\d, 128, (true|false), \d, 4, (true|false)>", k):
            assert v["agpr"] == 0, f"{k} keeps accumulators in AGPRs: {v}"
    # head dim 256 (Qwen3.5): one workgroup per CU is accepted, spills are not
    check(md, r"batch_prefill_paged_kernel<\d, 256, (true|false), \d, 4, false>", 512, at_least=2)
