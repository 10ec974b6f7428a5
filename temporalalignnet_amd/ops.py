"""Thin torch-tensor -> C-ABI wrappers (device pointers, sizes, current HIP stream).

PyTorch is plumbing here: it owns the device memory and the stream; every arithmetic op is a
hand-written HIP kernel in libtan_hip.so.  All wrappers require CUDA(HIP) tensors and raise otherwise.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_QUICKGELU, ACT_QUICKGELU_GRAD, TAN_BF16, TAN_F32  # noqa: F401


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return TAN_F32
    if t.dtype == torch.bfloat16:
        return TAN_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.TanHipError("HIP path needs device tensors (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(A, B, C_out, *, M, N, K, a_kc=True, b_kc=True, lda=None, ldb=None, ldc=None, bias=None, residual=None,
         ldr=None, act=ACT_NONE, aux=None, ldaux=None, accumulate=False, split_k=1, alpha=1.0, batch=1,
         sA=0, sB=0, sC=0):
    """C[M,N] (=|+=) alpha * opA(A) @ opB(B) (+bias)(act)(+residual); see include/tan_hip.h:tan_gemm."""
    d = _lib.GemmDesc()
    d.dtype, d.out_dtype = _dt(A), _dt(C_out)
    assert _dt(B) == d.dtype
    d.M, d.N, d.K = M, N, K
    d.a_kc, d.b_kc = int(a_kc), int(b_kc)
    d.A, d.lda = _ptr(A), lda if lda is not None else (K if a_kc else M)
    d.B, d.ldb = _ptr(B), ldb if ldb is not None else (K if b_kc else N)
    d.C, d.ldc = _ptr(C_out), ldc if ldc is not None else N
    d.bias = _ptr(bias)
    d.residual, d.ldr = _ptr(residual), ldr if ldr is not None else N
    d.act = act
    d.aux, d.ldaux = _ptr(aux), ldaux if ldaux is not None else N
    d.accumulate, d.split_k, d.alpha = int(accumulate), split_k, alpha
    d.batch, d.sA, d.sB, d.sC = batch, sA, sB, sC
    _lib.check(_lib.lib().tan_gemm(C.byref(d), _stream()), "tan_gemm")
    return C_out
