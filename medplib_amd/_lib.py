"""ctypes loader for libmedplib_hip.so.

`include/medplib_hip.h` is the single source of truth for the C ABI: the prototypes are parsed from it and turned into
ctypes signatures, so the header, the library and the Python host side cannot drift apart silently.  There is NO
fallback: if the library is missing or a symbol is absent, import of the op layer fails loudly."""
import ctypes
import os
import re

import torch  # noqa: F401  — MUST precede loading libmedplib_hip.so: torch brings its own libamdhip64; loading ours first
#                              would start a second HIP runtime instance that cannot see torch's device allocations.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEDPLIB_HIP_LIB") or os.path.join(_HERE, "lib", "libmedplib_hip.so")     # (the override serves A/B builds of the library)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "medplib_hip.h")

_CT = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float,
    "size_t": ctypes.c_size_t,
    "hipStream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [(argtype, argname), ...])} for every `mp_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#[^\n]*", "", src, flags=re.M)          # preprocessor lines
    src = re.sub(r"typedef[^;]*;", "", src)
    src = src.replace('extern "C" {', "")
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mp_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                arglist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


def _ctype(t):
    t = t.replace("const ", "").strip()
    if t.endswith("*"):
        return ctypes.c_char_p if t == "char*" else ctypes.c_void_p
    return _CT[t]


class MedplibError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m medplib_amd.build` (hipcc, gfx950). "
                "medplib_amd has no CPU fallback.")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self._dll, name)  # AttributeError if the header declares a symbol the library lacks
            fn.argtypes = [_ctype(t) for t, _ in args]
            fn.restype = _ctype(ret) if ret.replace("const ", "") != "char*" else ctypes.c_char_p
            setattr(self, "_raw_" + name, fn)

    def last_error(self):
        return self._raw_mp_last_error_string().decode()

    def call(self, name, *args):
        """Call an `int mp_*` entry point; raise MedplibError on a non-zero return."""
        rc = getattr(self, "_raw_" + name)(*args)
        if rc != 0:
            raise MedplibError(f"{name} failed (rc={rc}): {self.last_error()}")

    def raw(self, name):
        return getattr(self, "_raw_" + name)


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
