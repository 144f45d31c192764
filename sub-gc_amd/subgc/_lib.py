"""ctypes binding of libsubgc_hip.so, generated from include/subgc_hip.h.

The header is the single source of truth: every `int subgc_*(...)` declaration in it is parsed
into a ctypes prototype, so a symbol that the header declares and the library does not export
fails at import time (tests/test_abi.py checks exactly that without a GPU).

There is NO CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
import re

# torch bundles its own libamdhip64; it must be the HIP runtime of the process BEFORE
# libsubgc_hip.so is dlopen'ed, so that streams and device pointers are shared with torch.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "subgc_hip.h")
LIB_PATH = os.path.join(_HERE, "libsubgc_hip.so")

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
}


class SubgcError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for m in re.finditer(r"\b(int|const\s+char\s*\*)\s+(subgc_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        protos = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    protos.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    protos.append((_SCALARS[ty.replace("const ", "").strip()], nm))
        out[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, protos)
    return out


def parse_struct(name, path=HEADER):
    """ctypes.Structure mirror of `typedef struct <name> { ... } <name>;` in the header (one field per declaration; pointers of any
    type become c_void_p): the header stays the single source of truth for the layout, like the prototypes."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    m = re.search(r"typedef\s+struct\s+" + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", src, flags=re.S)
    if m is None:
        raise SubgcError(f"{name} is not declared in subgc_hip.h")
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        if "*" in decl:
            fields.append((decl.split("*")[-1].strip(), ctypes.c_void_p))
        else:
            ty, nm = decl.rsplit(" ", 1)
            fields.append((nm, _SCALARS[ty.replace("const ", "").strip()]))
    return type(name, (ctypes.Structure,), {"_fields_": fields})


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SubgcError(f"{LIB_PATH} is missing: run `python sub-gc_amd/build.py` (hipcc, gfx950). "
                             "There is no CPU fallback for the Sub-GC hot path.")
        L = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (ret, args) in _protos.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise SubgcError(f"libsubgc_hip.so does not export {name} declared in subgc_hip.h") from e
            fn.restype = ret
            fn.argtypes = [a for a, _ in args]
        if L.subgc_version() != 1:
            raise SubgcError("libsubgc_hip.so ABI version mismatch")
        _lib = L
    return _lib


_FN = {}          # bound entry points (one attribute lookup on the CDLL per name instead of one per call)


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero code."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise SubgcError(f"{name} failed with code {rc}: {lib().subgc_last_error().decode()}")


FAM = {"gemm": 1, "attn": 2, "lstm": 3, "gcn": 4, "pool": 5, "softmax": 6, "mid": 7}


def prof_enable(family, on=True):
    call("subgc_prof_enable", FAM[family], int(on))


def prof_collect(family):
    n = ctypes.c_int64(0); ms = ctypes.c_double(0); work = ctypes.c_double(0)
    call("subgc_prof_collect", FAM[family], ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work))
    return n.value, ms.value, work.value


def prof_last_moved(family):
    """Bytes the family's launches at the last prof_collect actually moved (LSTM cells: split-K planes, gate terms, saved gates included)."""
    b = ctypes.c_double(0)
    call("subgc_prof_last_moved", FAM[family], ctypes.byref(b))
    return b.value
