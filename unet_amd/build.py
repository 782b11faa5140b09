"""In-tree build of the native parts (no JIT cache, nothing installed):

  unet_amd/lib/libunet_hip.so        hipcc --offload-arch=gfx950, the C ABI of include/unet_hip.h
  neural_net.cpython-*.so (repo root) g++ + pybind11, the Python module `neural_net`

hipcc cross-compiles for gfx950 without a GPU, so this runs anywhere the ROCm
toolchain is installed.  A target is rebuilt when any of its sources is newer.
"""
import glob
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unet_amd", "csrc")
HOST = os.path.join(ROOT, "unet_amd", "host")
LIBDIR = os.path.join(ROOT, "unet_amd", "lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
HIP_LIB = os.path.join(LIBDIR, "libunet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No packed fp32 arithmetic in device code.  On MI355X v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 returned wrong
# results in waves that shared a SIMD with waves of this library's MFMA kernels (csrc/bf16_util.h, keep_scalar);
# the scalar form never did.  Two switches:
#   -target-feature -packed-fp32-ops   the device code is compiled for a gfx950 without that feature, so the
#       backend has no such instruction to select, whatever forms <2 x float> operations (SLP vectoriser,
#       ext_vector arithmetic in the sources, the legaliser).  It is a cc1 option and so reaches the host pass
#       too, which answers "not a recognized feature for this target (ignoring feature)"; build_hip() drops
#       those lines.  (A per-function target attribute does the same for the instructions but stops the inliner
#       at every call into the HIP headers and the device library: thousands of real calls.)
#   -fno-slp-vectorize                 kept: without the pass the scalar code is not first paired and then split.
# tests/test_no_packed_fp32_cpu.py disassembles the built library and fails on any such instruction.  The
# training step takes the same time either way (these kernels are bound by HBM or by the matrix cores, not by
# VALU issue).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
             "-Wno-unused-variable", "-Wno-unused-but-set-variable"] + NO_PACKED_FP32
# UNET_LAB_HIPFLAGS="-DTCONV_LAB_STAMPS ...": extra compiler flags of a laboratory build (rebuild with --force, and again
# without the variable afterwards: the library in the tree is the one the tests and the benchmark load).
HIP_FLAGS += os.environ.get("UNET_LAB_HIPFLAGS", "").split()
HOST_PASS_NOISE = "is not a recognized feature for this target (ignoring feature)"
# Per-file extra flags.  wgrad_bf16: MFMA accumulators stay in ordinary vector registers (no copies
# to and from the accumulation half of the register file around the loop-carried accumulators).
EXTRA_FLAGS = {"wgrad_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def module_path():
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    return os.path.join(ROOT, "neural_net" + ext)


def build_hip(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    objs, procs = [], []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            cmd = [HIPCC] + HIP_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    failed = []
    for src, p in procs:
        _, err = p.communicate()
        err = "".join(l for l in err.splitlines(True) if HOST_PASS_NOISE not in l)
        if err.strip():
            sys.stderr.write(err)
        if p.returncode != 0:
            failed.append(src)
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or _newer(HIP_LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


def build_module(force=False, verbose=False):
    """The pybind11 module `neural_net` (host C++ only; calls the C ABI)."""
    import pybind11
    target = module_path()
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    if not srcs:
        return None
    deps = srcs + glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    # UNET_ASAN=1: the host module with AddressSanitizer (host code only).  Use it with
    #   LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0
    #   PYTHONPATH=build/asan:. python -m pytest tests/test_module_cpu.py
    # (libstdc++ is preloaded too: python does not link it, and the sanitizer's __cxa_throw hook needs the real one).  The sanitized module goes to
    # build/asan/ so that it never replaces the one the GPU runs load.
    asan = os.environ.get("UNET_ASAN", "0") not in ("", "0")
    if asan:
        os.makedirs(os.path.join(ROOT, "build", "asan"), exist_ok=True)
        target = os.path.join(ROOT, "build", "asan", os.path.basename(target))
    if force or _newer(target, deps) or _newer(target, [HIP_LIB]):
        cmd = ["g++", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer"] if asan else ["g++", "-O2"]
        cmd += ["-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
               "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
               "-I" + os.path.join(ROOT, "include"), "-I" + HOST] + srcs + \
              ["-L" + LIBDIR, "-lunet_hip", "-Wl,-rpath," + ("$ORIGIN/../../unet_amd/lib" if asan else "$ORIGIN/unet_amd/lib"),
               "-o", target]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_module(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
