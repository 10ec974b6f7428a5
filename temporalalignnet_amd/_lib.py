"""ctypes binding of libtan_hip.so (include/tan_hip.h).  There is NO fallback: if the library is
missing or a symbol is absent, using the product path raises."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAN_HIP_LIB") or os.path.join(HERE, "libtan_hip.so")   # override: A/B of two builds on one box
HEADER = os.path.join(HERE, "..", "include", "tan_hip.h")

TAN_F32, TAN_BF16 = 0, 1
ACT_NONE, ACT_QUICKGELU, ACT_QUICKGELU_GRAD, ACT_RELU = 0, 1, 2, 3


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
        ("colsum", C.c_void_p),
    ]


_lib = None


def _header_source() -> str:
    with open(HEADER) as f:
        src = f.read()
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def declared_symbols() -> list[str]:
    """Every function name include/tan_hip.h declares."""
    return sorted(set(re.findall(r"\b(?:int|long)\s+(tan_[a-z0-9_]+)\s*\(", _header_source())))


_SCALARS = {"int": C.c_int, "long": C.c_long, "float": C.c_float, "double": C.c_double}


def declared_prototypes() -> dict:
    """name -> (restype, [argtypes]) parsed from include/tan_hip.h: every pointer (including `const tan_*_desc*`) is a
    c_void_p, scalars map one to one.  With argtypes set, a Python int passed for a `long` / `float` / `double` parameter is
    converted by ctypes instead of relying on the caller to wrap it (ADVICE / VERDICT r1)."""
    out = {}
    for ret, name, args in re.findall(r"\b(int|long)\s+(tan_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", _header_source(), flags=re.S):
        args = " ".join(args.split())
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(C.c_void_p)
                else:
                    base = [t for t in a.replace("const", " ").replace("unsigned", " ").split()][0]
                    types.append(_SCALARS[base])
        out[name] = (_SCALARS[ret], types)
    return out


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TanHipError(
                f"{LIB_PATH} not found: build it with `python -m temporalalignnet_amd.build` "
                "(the HIP path has no CPU fallback)")
        # torch bundles its own ROCm runtime (torch/lib/libamdhip64.so); it must be the one already loaded when our
        # library's DT_NEEDED libamdhip64.so.7 is resolved, otherwise two HIP runtimes end up in one process
        # (observed: every launch fails with hipErrorNoDevice).  PyTorch is the memory/stream plumbing anyway.
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        protos = declared_prototypes()
        for name in declared_symbols():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = protos[name]
    return _lib


def check(code: int, what: str):
    if code != 0:
        raise TanHipError(f"{what} failed with code {code}" + (" (bad argument)" if code == -1 else " (hipError_t)"))


class LayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "w_qkv", "w_out", "w_fc", "w_proj", "b_qkv", "b_out", "b_fc", "b_proj", "ln1_g", "ln1_b", "ln2_g", "ln2_b",
        "g_w_qkv", "g_w_out", "g_w_fc", "g_w_proj", "g_b_qkv", "g_b_out", "g_b_fc", "g_b_proj",
        "g_ln1_g", "g_ln1_b", "g_ln2_g", "g_ln2_b", "wt_qkv", "wt_out", "wt_fc", "wt_proj",
        "wp_qkv", "wp_out", "wp_fc", "wp_proj", "wtp_qkv", "wtp_out", "wtp_fc", "wtp_proj")]


class LayerBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "xn1", "qkv", "attn_o", "x_mid", "xn2", "h_pre", "h_act", "x_out", "mean1", "rstd1", "mean2", "rstd2", "lse")]


class PackEntry(C.Structure):
    _fields_ = [("src_off", C.c_long), ("dst_off", C.c_long), ("N", C.c_int), ("K", C.c_int), ("TN", C.c_int), ("TK", C.c_int)]


class ImageEntry(C.Structure):
    _fields_ = [("off", C.c_long), ("N", C.c_int), ("K", C.c_int), ("tn_w", C.c_int), ("tk_w", C.c_int), ("tn_t", C.c_int), ("tk_t", C.c_int)]


class AdamwImagesDesc(C.Structure):
    _fields_ = [
        ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("mode", C.c_void_p), ("n", C.c_long),
        ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
        ("step", C.c_int), ("grad_scale", C.c_float),
        ("p_bf16", C.c_void_p), ("ema", C.c_void_p), ("ema_m", C.c_float), ("ema_bf16", C.c_void_p),
        ("table", C.c_void_p), ("unit_prefix", C.c_void_p), ("n_entries", C.c_int), ("n_units", C.c_long),
        ("p_packed", C.c_void_p), ("p_t", C.c_void_p), ("p_tpacked", C.c_void_p), ("ema_packed", C.c_void_p),
        ("rest_idx", C.c_void_p), ("n_rest", C.c_long), ("unit_begin", C.c_long), ("unit_end", C.c_long),
    ]


class Ptr8(C.Structure):
    _fields_ = [("p", C.c_void_p * 8)]


class SimFamDesc(C.Structure):
    """tan_simfam_desc (include/tan_hip.h): one feature family of the logits-free NCE, stage outputs -> stage gradients"""
    _fields_ = [
        ("S", C.c_int), ("St", C.c_int), ("B", C.c_int), ("T", C.c_int), ("N", C.c_int), ("C", C.c_int), ("Mc", C.c_int), ("flags", C.c_int),
        ("x_video", Ptr8), ("v_grp_rows", C.c_long), ("v_off", C.c_long),
        ("x_text", Ptr8), ("t_grp_rows", C.c_long), ("t_off", C.c_long),
        ("idx", C.c_void_p), ("colmap", C.c_void_p), ("col_invalid", C.c_void_p),
        ("tgt", C.c_void_p), ("row_leak", C.c_void_p),
        ("vn", C.c_void_p), ("inv_v", C.c_void_p), ("tn", C.c_void_p), ("inv_t", C.c_void_p),
        ("rowsum", C.c_void_p), ("colsum", C.c_void_p), ("possum_v", C.c_void_p), ("possum_t", C.c_void_p),
        ("e_keep", C.c_void_p), ("ws", C.c_void_p),
        ("v_terms", C.c_void_p), ("t_terms", C.c_void_p),
        ("g_v", C.c_void_p), ("g_t", C.c_void_p),
        ("dl", C.c_void_p), ("d_tn_acc", C.c_void_p),
        ("d_video", Ptr8), ("d_text", Ptr8),
        ("dtn_split_k", C.c_int),
    ]


class MlpDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_long), ("C", C.c_int), ("FF", C.c_int),
        ("x_mid", C.c_void_p), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p),
        ("pw_fc", C.c_void_p), ("pw_proj", C.c_void_p), ("b_fc", C.c_void_p), ("b_proj", C.c_void_p),
        ("xn2", C.c_void_p), ("mean2", C.c_void_p), ("rstd2", C.c_void_p),
        ("h_pre", C.c_void_p), ("h_act", C.c_void_p), ("x_out", C.c_void_p),
        ("nln_g", C.c_void_p), ("nln_b", C.c_void_p), ("xn_next", C.c_void_p), ("nmean", C.c_void_p), ("nrstd", C.c_void_p),
        ("eps", C.c_float), ("variant", C.c_int),
        ("attn_o", C.c_void_p), ("pw_out", C.c_void_p), ("b_out", C.c_void_p), ("x_in", C.c_void_p),
    ]


class MlpBwdDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_long), ("C", C.c_int), ("FF", C.c_int),
        ("dx", C.c_void_p), ("h_pre", C.c_void_p), ("x_mid", C.c_void_p),
        ("mean2", C.c_void_p), ("rstd2", C.c_void_p), ("ln_g", C.c_void_p),
        ("pwt_proj", C.c_void_p), ("pwt_fc", C.c_void_p),
        ("dh", C.c_void_p), ("dx2", C.c_void_p),
        ("g_b_fc", C.c_void_p), ("g_ln_g", C.c_void_p), ("g_ln_b", C.c_void_p), ("g_b_out", C.c_void_p),
        ("ln1_dxn", C.c_void_p), ("ln1_x", C.c_void_p), ("ln1_res", C.c_void_p),
        ("ln1_mean", C.c_void_p), ("ln1_rstd", C.c_void_p), ("ln1_g", C.c_void_p),
        ("g_ln1_g", C.c_void_p), ("g_ln1_b", C.c_void_p), ("g_dx_colsum", C.c_void_p), ("dx_out", C.c_void_p),
        ("pwt_out", C.c_void_p), ("d_o", C.c_void_p),
        ("dqkv", C.c_void_p), ("pwt_in", C.c_void_p), ("dstage", C.c_void_p),
    ]


class EmbedDesc(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("a_dtype", C.c_int), ("rows", C.c_long), ("K", C.c_int), ("T", C.c_int), ("C", C.c_int),
        ("pw", C.c_void_p), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p),
        ("a_bf16", C.c_void_p), ("proj", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("out", C.c_void_p * 2), ("out_grp_rows", C.c_long * 2), ("out_off", C.c_long * 2), ("pos", C.c_void_p * 2),
        ("ln1_g", C.c_void_p * 2), ("ln1_b", C.c_void_p * 2), ("xn1", C.c_void_p * 2), ("mean1", C.c_void_p * 2), ("rstd1", C.c_void_p * 2),
        ("pad_src", C.c_void_p), ("pad_dst", C.c_void_p), ("pad_grp_rows", C.c_long), ("pad_off", C.c_long),
    ]


class EmbedBwdDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_long), ("T", C.c_int), ("C", C.c_int),
        ("d_out", C.c_void_p * 2), ("d_out_grp_rows", C.c_long * 2), ("d_out_off", C.c_long * 2),
        ("proj", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("ln_g", C.c_void_p),
        ("d_proj", C.c_void_p), ("g_ln_g", C.c_void_p), ("g_ln_b", C.c_void_p), ("d_pos", C.c_void_p * 2),
    ]


class PosLnBwdUse(C.Structure):
    _fields_ = [("d_pos", C.c_void_p), ("x", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("g_table", C.c_void_p), ("n", C.c_int), ("nparts", C.c_int)]


class AttnBlkDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("L", C.c_int), ("C", C.c_int), ("H", C.c_int),
        ("xn1", C.c_void_p), ("x_in", C.c_void_p), ("key_padding_mask", C.c_void_p),
        ("pw_qkv", C.c_void_p), ("pw_out", C.c_void_p), ("b_qkv", C.c_void_p), ("b_out", C.c_void_p),
        ("qkv", C.c_void_p), ("attn_o", C.c_void_p), ("lse", C.c_void_p), ("x_mid", C.c_void_p),
    ]


class EncoderDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("B", C.c_int), ("L", C.c_int), ("C", C.c_int), ("H", C.c_int), ("layers", C.c_int),
        ("key_padding_mask", C.c_void_p), ("x0", C.c_void_p),
        ("params", C.POINTER(LayerParams)), ("bufs", C.POINTER(LayerBufs)),
        ("post_g", C.c_void_p), ("post_b", C.c_void_p), ("g_post_g", C.c_void_p), ("g_post_b", C.c_void_p),
        ("post_out", C.c_void_p), ("post_mean", C.c_void_p), ("post_rstd", C.c_void_p),
        ("scr_dx", C.c_void_p), ("scr_dx2", C.c_void_p), ("scr_do", C.c_void_p), ("scr_dxn", C.c_void_p),
        ("scr_dh", C.c_void_p), ("scr_dqkv", C.c_void_p), ("ln_ws", C.c_void_p),
        ("dw_ws", C.c_void_p), ("dw_ws_floats", C.c_long),
        ("d_stage", C.POINTER(C.c_void_p)), ("d_x0", C.c_void_p),
        ("layer_done", C.POINTER(C.c_void_p)),
        ("no_save", C.c_int), ("xn1_ready", C.c_int), ("dw_stream", C.c_void_p), ("dw_tail", C.c_int),
        ("scr2_dx", C.c_void_p), ("scr2_dx2", C.c_void_p), ("scr2_dh", C.c_void_p), ("scr2_dqkv", C.c_void_p),
        ("split_part", C.c_void_p),
    ]


# ---- HIP streams by ROLE, one per device for the whole process.  Streams map onto a few hardware queues in creation order: when
# every model / trainer created its own, two roles that must overlap (the joint stack's stream and the main stream) could land on one
# queue depending on how many objects had been built before -- the same configuration ran 6.7 or 8.4 ms per step depending on what
# the process had run earlier (tools/lab/seq_cfg.py).
_ROLE_STREAMS = {}


def role_stream(dev, role):
    import torch
    dev = torch.device(dev)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), role)
    st = _ROLE_STREAMS.get(key)
    if st is None:
        for r in ("stack", "loss", "comm", "opt"):                                        # fixed creation order, whatever is asked for first
            k = (key[0], key[1], r)                                                       # (the streams of a step's tail -- "loss", "opt" -- at high
            if k not in _ROLE_STREAMS:                                                    #  priority: +0.02 ms per step, ABBA x2 of 80 steps, round 5)
                _ROLE_STREAMS[k] = torch.cuda.Stream(device=torch.device(key[0], key[1]))
        st = _ROLE_STREAMS.setdefault(key, _ROLE_STREAMS.get(key) or torch.cuda.Stream(device=torch.device(key[0], key[1])))
    return st
