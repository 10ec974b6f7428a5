"""Thin torch-tensor -> C-ABI wrappers (device pointers, sizes, current HIP stream).

PyTorch is plumbing here: it owns the device memory and the stream; every arithmetic op is a
hand-written HIP kernel in libtan_hip.so.  All wrappers require CUDA(HIP) tensors and raise otherwise.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_QUICKGELU, ACT_QUICKGELU_GRAD, ACT_RELU, TAN_BF16, TAN_F32  # noqa: F401


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
    # the raw handle of torch's CURRENT stream on the current device; the public torch.cuda.current_stream() builds a Stream
    # object through three Python layers (~3 us) and this is called once per launch (~120 times per training step)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def gemm_atb(As, Bs, Cs, *, lda, M, N, K, accumulate=False, split=1):
    """C_p [M_p, N_p] (=|+=) A_p^T B_p for up to eight problems (lists of tensors; A_p [K, lda_p], B_p [K, N_p] bf16 row-major, C_p f32
    or bf16): tan_gemm_atb, the 256 x 256-tile kernel.  lda / M / N: ints (the same for every problem) or lists."""
    if len(As) > 8:                                    # the kernel takes eight problems per launch
        sl = lambda v, a, b: v if isinstance(v, int) else v[a:b]                        # noqa: E731
        for a in range(0, len(As), 8):
            gemm_atb(As[a:a + 8], Bs[a:a + 8], Cs[a:a + 8], lda=sl(lda, a, a + 8), M=sl(M, a, a + 8), N=sl(N, a, a + 8), K=K,
                     accumulate=accumulate, split=split)
        return
    n = len(As)
    ints = lambda v: (C.c_int * n)(*([v] * n if isinstance(v, int) else v))          # noqa: E731
    ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])                    # noqa: E731
    _lib.check(_lib.lib().tan_gemm_atb(n, ptrs(As), ptrs(Bs), ptrs(Cs), ints(lda), ints(M), ints(N), K, _dt(Cs[0]), int(accumulate),
                                       split, _stream()), "tan_gemm_atb")


def gemm(A, B, C_out, *, M, N, K, a_kc=True, b_kc=True, lda=None, ldb=None, ldc=None, bias=None, residual=None,
         ldr=None, act=ACT_NONE, aux=None, ldaux=None, accumulate=False, split_k=1, alpha=1.0, batch=1,
         sA=0, sB=0, sC=0, colsum=None):
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
    d.colsum = _ptr(colsum)
    _lib.check(_lib.lib().tan_gemm(C.byref(d), _stream()), "tan_gemm")
    return C_out


def _f32(t):
    assert t is None or t.dtype == torch.float32
    return _ptr(t)


_ln_ws = {}


def _ws_f32(n, device):
    # one workspace per (device, stream): two LayerNorm backwards issued on different streams must not share their partial tables
    key = (device, "ln", torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)
    buf = _ln_ws.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, device=device, dtype=torch.float32)
        _ln_ws[key] = buf
    return buf


def layernorm_fwd(x, gamma, beta, y, mean=None, rstd=None, add=None, add_period=0, eps=1e-5):
    rows, Cc = x.numel() // x.shape[-1], x.shape[-1]
    _lib.check(_lib.lib().tan_layernorm_fwd(_ptr(x), _f32(gamma), _f32(beta), _ptr(y), _f32(mean), _f32(rstd), _ptr(add),
                                             C.c_int(add_period), C.c_long(rows), C.c_int(Cc), C.c_float(eps), _dt(x),
                                             _stream()), "tan_layernorm_fwd")
    return y


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma=None, dbeta=None, dres=None, dx_colsum=None):
    rows, Cc = x.numel() // x.shape[-1], x.shape[-1]
    L = _lib.lib()
    ws = _ws_f32(L.tan_layernorm_bwd_ws_floats(C.c_int(Cc)), x.device)
    _lib.check(L.tan_layernorm_bwd(_ptr(dy), _ptr(x), _f32(gamma), _f32(mean), _f32(rstd), _ptr(dres), _ptr(dx),
                                   _f32(dgamma), _f32(dbeta), _f32(dx_colsum), _ptr(ws), C.c_long(rows), C.c_int(Cc), _dt(x),
                                   _stream()),
               "tan_layernorm_bwd")
    return dx


def l2norm_fwd(x, y, inv_norm, rows, Cc, grp=None, src_grp_rows=None, src_off=0):
    grp = grp or rows
    _lib.check(_lib.lib().tan_l2norm_fwd(_ptr(x), _ptr(y), _f32(inv_norm), C.c_long(rows), C.c_int(Cc), C.c_int(grp),
                                          C.c_int(src_grp_rows if src_grp_rows is not None else grp), C.c_int(src_off),
                                          _dt(x), _stream()), "tan_l2norm_fwd")
    return y


def l2norm_bwd(dy, y, inv_norm, dx, rows, Cc, grp=None, dst_grp_rows=None, dst_off=0):
    grp = grp or rows
    _lib.check(_lib.lib().tan_l2norm_bwd(_ptr(dy), _ptr(y), _f32(inv_norm), _ptr(dx), C.c_long(rows), C.c_int(Cc),
                                          C.c_int(grp), C.c_int(dst_grp_rows if dst_grp_rows is not None else grp),
                                          C.c_int(dst_off), _dt(y), _stream()), "tan_l2norm_bwd")
    return dx


def _ptr8(tensors):
    if not 1 <= len(tensors) <= 8:
        raise ValueError(f"1..8 stage buffers, got {len(tensors)}")
    t = _lib.Ptr8()
    for i, x in enumerate(tensors):
        t.p[i] = x.data_ptr()
    return t


def l2norm_fwd_multi(xs, y, inv_norm, rows, Cc, grp=None, src_grp_rows=None, src_off=0):
    """all stages of one feature family in one launch: xs = list of per-stage buffers, y [S, rows, C], inv_norm [S * rows] or None"""
    grp = rows if grp is None else grp
    if len(xs) > 8:          # the C entry point takes up to 8 stage pointers: deeper stacks go in groups of 8
        for s0 in range(0, len(xs), 8):
            l2norm_fwd_multi(xs[s0:s0 + 8], y[s0:s0 + 8], None if inv_norm is None else inv_norm[s0 * rows:(s0 + 8) * rows],
                             rows, Cc, grp, src_grp_rows, src_off)
        return y
    t = _ptr8(xs)
    _lib.check(_lib.lib().tan_l2norm_fwd_multi(C.byref(t), _ptr(y), _f32(inv_norm), C.c_int(len(xs)), C.c_long(rows), C.c_int(Cc),
                                                C.c_int(grp), C.c_int(src_grp_rows if src_grp_rows is not None else grp),
                                                C.c_int(src_off), _dt(y), _stream()), "tan_l2norm_fwd_multi")
    return y


def l2norm_bwd_multi(dy, y, inv_norm, dxs, rows, Cc, grp=None, dst_grp_rows=None, dst_off=0):
    grp = rows if grp is None else grp
    if len(dxs) > 8:
        for s0 in range(0, len(dxs), 8):
            l2norm_bwd_multi(dy[s0:s0 + 8], y[s0:s0 + 8], inv_norm[s0 * rows:(s0 + 8) * rows], dxs[s0:s0 + 8], rows, Cc, grp,
                             dst_grp_rows, dst_off)
        return
    t = _ptr8(dxs)
    _lib.check(_lib.lib().tan_l2norm_bwd_multi(_ptr(dy), _ptr(y), _f32(inv_norm), C.byref(t), C.c_int(len(dxs)), C.c_long(rows),
                                                C.c_int(Cc), C.c_int(grp), C.c_int(dst_grp_rows if dst_grp_rows is not None else grp),
                                                C.c_int(dst_off), _dt(y), _stream()), "tan_l2norm_bwd_multi")



def colsum_acc(x, out, rows, Cc):
    _lib.check(_lib.lib().tan_colsum_acc(_ptr(x), _f32(out), C.c_long(rows), C.c_int(Cc), _dt(x), _stream()), "tan_colsum_acc")
    return out


def reduce_add(parts, out, nparts, n):
    """out[n] (f32) += sum_p parts[p][n] -- folds split-K partial tiles."""
    assert parts.dtype == torch.float32 and out.dtype == torch.float32 and parts.is_contiguous() and out.is_contiguous()
    _lib.check(_lib.lib().tan_reduce_add(_f32(parts), _f32(out), C.c_int(nparts), C.c_long(n), _stream()), "tan_reduce_add")
    return out


def rows_copy(src, dst, G, R, Cc, src_grp_rows, src_off, dst_grp_rows, dst_off, accumulate=False):
    assert _dt(src) == _dt(dst)
    _lib.check(_lib.lib().tan_rows_copy(_ptr(src), _ptr(dst), C.c_int(G), C.c_int(R), C.c_int(Cc), C.c_long(src_grp_rows),
                                         C.c_long(src_off), C.c_long(dst_grp_rows), C.c_long(dst_off), C.c_int(int(accumulate)),
                                         _dt(src), _stream()), "tan_rows_copy")
    return dst


def group_sum(x, out, G, R, Cc):
    _lib.check(_lib.lib().tan_group_sum(_ptr(x), _ptr(out), C.c_int(G), C.c_int(R), C.c_int(Cc), _dt(x), _stream()),
               "tan_group_sum")
    return out


def cast(src, dst):
    assert src.numel() == dst.numel()
    _lib.check(_lib.lib().tan_cast(_ptr(src), _dt(src), _ptr(dst), _dt(dst), C.c_long(src.numel()), _stream()), "tan_cast")
    return dst


def head_fwd(x, w, b, out, rows, Cc):
    _lib.check(_lib.lib().tan_head_fwd(_ptr(x), _f32(w), _f32(b), _f32(out), C.c_long(rows), C.c_int(Cc), _dt(x), _stream()),
               "tan_head_fwd")
    return out


def head_bwd(dout, x, w, dx, dw, db, rows, Cc, accumulate_dx=False):
    _lib.check(_lib.lib().tan_head_bwd(_f32(dout), _ptr(x), _f32(w), _ptr(dx), _f32(dw), _f32(db), C.c_long(rows), C.c_int(Cc),
                                        C.c_int(int(accumulate_dx)), _dt(x), _stream()), "tan_head_bwd")
    return dx


def interp_linear(src, dst, L_in, L_out, Cc):
    _lib.check(_lib.lib().tan_interp_linear(_f32(src), _f32(dst), C.c_int(L_in), C.c_int(L_out), C.c_int(Cc), _stream()),
               "tan_interp_linear")
    return dst


def interp_linear_bwd(ddst, dsrc, L_in, L_out, Cc):
    _lib.check(_lib.lib().tan_interp_linear_bwd(_f32(ddst), _f32(dsrc), C.c_int(L_in), C.c_int(L_out), C.c_int(Cc), _stream()),
               "tan_interp_linear_bwd")
    return dsrc


def attn_fwd(qkv, keypad_u8, o, lse, B, L, H):
    _lib.check(_lib.lib().tan_attn_fwd(_ptr(qkv), _ptr(keypad_u8), _ptr(o), _f32(lse), C.c_int(B), C.c_int(L), C.c_int(H),
                                        _dt(qkv), _stream()), "tan_attn_fwd")
    return o


def attn_bwd(qkv, keypad_u8, o, lse, d_o, dqkv, B, L, H, g_b_qkv=None):
    """g_b_qkv [3C] f32: += the column sums of dqkv (in_proj bias gradient), by the same call."""
    if g_b_qkv is not None:
        _lib.check(_lib.lib().tan_attn_bwd_bias(_ptr(qkv), _ptr(keypad_u8), _ptr(o), _f32(lse), _ptr(d_o), _ptr(dqkv),
                                                 _f32(g_b_qkv), C.c_int(B), C.c_int(L), C.c_int(H), _dt(qkv), _stream()),
                   "tan_attn_bwd_bias")
        return dqkv
    _lib.check(_lib.lib().tan_attn_bwd(_ptr(qkv), _ptr(keypad_u8), _ptr(o), _f32(lse), _ptr(d_o), _ptr(dqkv), C.c_int(B),
                                        C.c_int(L), C.c_int(H), _dt(qkv), _stream()), "tan_attn_bwd")
    return dqkv
