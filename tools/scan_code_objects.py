#!/usr/bin/env python3
"""Disassembles the gfx950 code objects inside libunet_hip.so and counts instructions per kernel.

    python tools/scan_code_objects.py                       # packed fp32 arithmetic per kernel
    python tools/scan_code_objects.py 'v_mfma|ds_read_b128'  # any other pattern

The library's .hip_fatbin section is a sequence of clang offload bundles (one per source file); each is
unbundled for gfx950 and run through llvm-objdump.  Needs only the ROCm LLVM tools, no GPU.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
LIB = os.path.join(ROOT, "unet_amd", "lib", "libunet_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
PACKED_FP32 = r"\bv_pk_(add|mul|fma)_f32\b"


def code_objects(lib=LIB, workdir=None):
    """-> paths of the gfx950 code objects found in `lib` (files under `workdir`)."""
    workdir = workdir or tempfile.mkdtemp(prefix="unet_co_")
    fat = os.path.join(workdir, "fatbin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        piece = os.path.join(workdir, f"bundle{i}")
        with open(piece, "wb") as f:
            f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = piece + ".co"
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", f"--targets={TARGET}",
                            f"--input={piece}", "--unbundle", f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def count_per_kernel(pattern=PACKED_FP32, lib=LIB):
    """-> ({kernel symbol: matches}, number of kernels seen, number of code objects)."""
    rx, label = re.compile(pattern), re.compile(r"^[0-9a-f]+ <([^>]+)>:")
    hits, kernels = {}, set()
    cos = code_objects(lib)
    for co in cos:
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True,
                              text=True, check=True).stdout
        cur = None
        for line in text.splitlines():
            m = label.match(line)
            if m:
                cur = m.group(1)
                kernels.add(cur)
            elif cur and rx.search(line):
                hits[cur] = hits.get(cur, 0) + 1
    return hits, len(kernels), len(cos)


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else PACKED_FP32
    hits, nk, nco = count_per_kernel(pat)
    for k, v in sorted(hits.items(), key=lambda kv: -kv[1]):
        print(f"{v:8d}  {k}")
    print(f"{sum(hits.values())} matches of /{pat}/ in {len(hits)} of {nk} functions, {nco} code objects")
