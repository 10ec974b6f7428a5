"""ctypes binding of libtan_hip.so (include/tan_hip.h).  There is NO fallback: if the library is
missing or a symbol is absent, using the product path raises."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtan_hip.so")
HEADER = os.path.join(HERE, "..", "include", "tan_hip.h")

TAN_F32, TAN_BF16 = 0, 1
ACT_NONE, ACT_QUICKGELU, ACT_QUICKGELU_GRAD = 0, 1, 2


class TanHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("out_dtype", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_kc", C.c_int), ("b_kc", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_long),
        ("B", C.c_void_p), ("ldb", C.c_long),
        ("C", C.c_void_p), ("ldc", C.c_long),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_long),
        ("act", C.c_int),
        ("aux", C.c_void_p), ("ldaux", C.c_long),
        ("accumulate", C.c_int), ("split_k", C.c_int), ("alpha", C.c_float),
        ("batch", C.c_int), ("sA", C.c_long), ("sB", C.c_long), ("sC", C.c_long),
    ]


_lib = None


def declared_symbols() -> list[str]:
    """Every function name include/tan_hip.h declares."""
    with open(HEADER) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(tan_[a-z0-9_]+)\s*\(", src)))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TanHipError(
                f"{LIB_PATH} not found: build it with `python -m temporalalignnet_amd.build` "
                "(the HIP path has no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        for name in declared_symbols():
            getattr(_lib, name).restype = C.c_long if name.endswith("_floats") or name.endswith("_bytes") else C.c_int
    return _lib


def check(code: int, what: str):
    if code != 0:
        raise TanHipError(f"{what} failed with code {code}" + (" (bad argument)" if code == -1 else " (hipError_t)"))
