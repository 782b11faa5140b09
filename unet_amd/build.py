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
# -fno-slp-vectorize: the SLP vectoriser pairs independent fp32 operations into packed instructions
# (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  On MI355X such arithmetic returned wrong results in waves that
# shared a SIMD with waves of this library's MFMA kernels (csrc/bf16_util.h, keep_scalar); the scalar form never
# did.  Without the pass the device code holds almost no packed fp32 arithmetic, and the training step takes the
# same time (7.27 ms either way: these kernels are bound by HBM or by the matrix cores, not by VALU issue).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
             "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
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
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
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
    if force or _newer(target, deps) or _newer(target, [HIP_LIB]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
               "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
               "-I" + os.path.join(ROOT, "include"), "-I" + HOST] + srcs + \
              ["-L" + LIBDIR, "-lunet_hip", "-Wl,-rpath,$ORIGIN/unet_amd/lib", "-o", target]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_module(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
