"""ctypes binding of libunet_hip.so, generated from include/unet_hip.h.

Every `int unet_*(...)` declared in the header becomes `capi.<name>(...)` (the
`unet_` prefix kept) which raises RuntimeError(unet_last_error()) on a
non-zero status.  Pointer arguments take integers (e.g. tensor.data_ptr()),
None, or ctypes.byref(...) objects for host out-parameters.

The library is required: importing succeeds without it, but the first call
raises (there is no CPU fallback).
"""
import ctypes
import os
import re

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(_ROOT, "include", "unet_hip.h")
LIB_PATH = os.path.join(_ROOT, "unet_amd", "lib", "libunet_hip.so")

_SCALARS = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "unet_stream_t": ctypes.c_void_p,
    "unet_event_t": ctypes.c_void_p,
    "unet_comm_t": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [(argname, ctype), ...])} for every prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"\b(int|const char \*)\s*(unet_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    parsed.append((a.split("*")[-1].strip(), ctypes.c_void_p))
                else:
                    ty, nm = a.replace("const ", "").rsplit(" ", 1)
                    parsed.append((nm, _SCALARS[ty.strip()]))
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, parsed)
    return protos


PROTOTYPES = parse_header()
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m unet_amd.build` "
                "(this package has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in PROTOTYPES.items():
            fn = getattr(_lib, name)  # AttributeError here = header/library mismatch
            fn.restype = ret
            fn.argtypes = [t for _, t in args]
    return _lib


def last_error():
    return lib().unet_last_error().decode()


def _wrap(name):
    def call(*args):
        rc = getattr(lib(), name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name}: {last_error()} (status {rc})")
        return rc
    call.__name__ = name
    return call


for _name, (_ret, _args) in PROTOTYPES.items():
    if _ret is ctypes.c_int and _name != "unet_version":
        globals()[_name] = _wrap(_name)


def unet_version():
    return lib().unet_version()
