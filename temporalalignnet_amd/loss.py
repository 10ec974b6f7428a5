"""MI355X-native `get_loss` with the reference's signature and returned dict (train/loss.py:55-422), plus
`get_mask_from_time`, `get_text_pos`, `circulant`.

Everything O(B*T*B*N) or O(B*T*N) runs in hand-written HIP kernels (include/tan_hip.h: tan_nce_fwd/bwd, tan_selflabel,
tan_agreement, tan_diag_max, tan_masked_quantile); what remains in torch are O(B*N) vector reductions (masked means,
z-scores, the BCE over <= B*N texts) on device tensors.  Unlike the reference (boolean-mask indexing, torch.quantile on
compacted tensors, np.array of all sentences -- ~40 hidden device->host syncs per call) this version never synchronises:
padded texts are handled by masks/weights instead of compaction, which is the same arithmetic over the same elements.

Known, documented differences from the reference:
  * only sim='cos' (the reference model only produces cosine logits, tan_model.py:116-119);
  * loss.py:296,301 index a [#texts-with-positives] tensor with a [#texts] mask and raise IndexError if some
    non-padded text has no positive frame; here such a text is simply averaged like the others.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pad_sequence

from . import _lib, ops

# A/B switch of the one-pass d-logits + d_vn kernel (tan_simnce_bwd_dl_dvn_kept); 0 = element-wise pass + GEMM
_FUSED_DVN = os.environ.get("TAN_FUSED_DVN", "1") != "0"

_KIND = {"i": 0, "u": 1, "keep": 2, "keep-joint": 3}


def circulant(tensor, dim):
    """All cyclic shifts along `dim`, new axis last: circulant([0,1,2]) -> [[0,1,2],[2,0,1],[1,2,0]] (loss.py:16-23).
    Kept for API parity; the HIP self-labelling kernel builds its windows by index arithmetic instead."""
    S = tensor.shape[dim]
    x = tensor.movedim(dim, -1)
    j = torch.arange(S, device=tensor.device)
    out = x[..., (j[None, :] - j[:, None]) % S]
    return out if dim in (-1, tensor.dim() - 1) else out.movedim(-2, dim)


def get_mask_from_time(start_list, end_list, num_timestamp, num_text, device="cuda"):
    """[B, N, T] bool mask `start <= t < end` from ragged lists; padded starts T+100 / ends -100 (loss.py:26-41)."""
    B = len(start_list)
    start = pad_sequence([torch.FloatTensor(i) for i in start_list], batch_first=True, padding_value=num_timestamp + 1e2)
    end = pad_sequence([torch.FloatTensor(i) for i in end_list], batch_first=True, padding_value=-1e2)
    start, end = start.to(device, non_blocking=True), end.to(device, non_blocking=True)
    steps = torch.arange(num_timestamp, device=device)[None, None, :].expand(B, num_text, -1)
    mask = (start[:, :, None] <= steps) & (steps < end[:, :, None])
    return mask, start, end


def get_text_pos(start_list, end_list, device="cuda"):
    """zero-padded [B, N, 2] (loss.py:44-52)."""
    start = pad_sequence([torch.FloatTensor(i) for i in start_list], batch_first=True, padding_value=0)
    end = pad_sequence([torch.FloatTensor(i) for i in end_list], batch_first=True, padding_value=0)
    return torch.stack((start.to(device, non_blocking=True), end.to(device, non_blocking=True)), dim=-1)


def _stage_major(logits):
    """[B,S,T,B,N] -> contiguous f32 [S, B*T, B*N]; zero-copy for tensors produced by our TemporalAligner."""
    B, S, T, B2, N = logits.shape
    x = logits.permute(1, 0, 2, 3, 4)
    if not x.is_contiguous() or x.dtype != torch.float32:
        x = x.float().contiguous()
    return x.view(S, B * T, B * N)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _NCEFn(torch.autograd.Function):
    """v_terms [S,R], t_terms [S,Mp] of the symmetric multi-positive NCE (loss.py:240-253) on HIP."""

    @staticmethod
    def forward(ctx, lg, tgt, col_invalid, row_leak, B, T, N):
        S, R, Mp = lg.shape
        dev = lg.device
        stats = torch.empty(2 * S * R + 2 * S * Mp, device=dev)
        rowsum, possum_v = stats[:S * R], stats[S * R:2 * S * R]
        colsum, possum_t = stats[2 * S * R:2 * S * R + S * Mp], stats[2 * S * R + S * Mp:]
        v_terms, t_terms = torch.empty(S, R, device=dev), torch.empty(S, Mp, device=dev)
        L = _lib.lib()
        ws = torch.empty(L.tan_nce_ws_floats(C.c_int(S), C.c_int(B), C.c_int(T), C.c_int(N)), device=dev)
        _lib.check(L.tan_nce_fwd(_p(lg), _p(tgt), _p(col_invalid), _p(row_leak), _p(rowsum), _p(colsum), _p(possum_v),
                                 _p(possum_t), _p(v_terms), _p(t_terms), _p(ws), C.c_int(S), C.c_int(B), C.c_int(T), C.c_int(N),
                                 C.c_int(Mp), ops._stream()), "tan_nce_fwd")
        ctx.saved = (lg, tgt, col_invalid, row_leak, rowsum, colsum, possum_v, possum_t)
        ctx.dims = (S, B, T, N)
        return v_terms, t_terms

    @staticmethod
    def backward(ctx, g_v, g_t):
        lg, tgt, col_invalid, row_leak, rowsum, colsum, possum_v, possum_t = ctx.saved
        S, B, T, N = ctx.dims
        g_v = torch.zeros_like(rowsum).view(S, -1) if g_v is None else g_v.contiguous()
        g_t = torch.zeros_like(colsum).view(S, -1) if g_t is None else g_t.contiguous()
        dl = torch.empty_like(lg)
        _lib.check(_lib.lib().tan_nce_bwd(_p(lg), _p(tgt), _p(col_invalid), _p(row_leak), _p(rowsum), _p(colsum), _p(possum_v),
                                          _p(possum_t), _p(g_v), _p(g_t), _p(dl), _lib.TAN_F32, C.c_int(S), C.c_int(B),
                                          C.c_int(T), C.c_int(N), ops._stream()), "tan_nce_bwd")
        return dl, None, None, None, None, None, None


class _Blocks:
    """Last-stage same-video logit blocks: element (b,t,n) at base[b*sb + t*st + n] (see include/tan_hip.h)."""

    def __init__(self, tensor, offset_elems, sb, st):
        self.tensor, self.ptr, self.sb, self.st = tensor, C.c_void_p(tensor.data_ptr() + 4 * offset_elems), sb, st

    @staticmethod
    def of_logits(lg, B, T, N):           # stage-major [S, R, Mp]
        S, R, Mp = lg.shape
        return _Blocks(lg, (S - 1) * R * Mp, T * Mp + N, Mp)

    @staticmethod
    def of_diag(diag):                    # compact [B, T, N]
        B, T, N = diag.shape
        return _Blocks(diag, 0, T * N, N)


def _selflabel(blk, vpad_u8, tpad_u8, dur, B, T, N):
    dev = blk.tensor.device
    max_pos = torch.empty(B, N, dtype=torch.int32, device=dev)
    max_prob, max_logit = torch.empty(B, N, device=dev), torch.empty(B, N, device=dev)
    self_tgt = torch.empty(B, N, T, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().tan_selflabel(blk.ptr, C.c_long(blk.sb), C.c_long(blk.st), _p(vpad_u8), _p(tpad_u8), _p(dur),
                                        _p(max_pos), _p(max_prob), _p(max_logit), _p(self_tgt), C.c_int(B), C.c_int(T),
                                        C.c_int(N), ops._stream()), "tan_selflabel")
    return {"max_pos": max_pos, "max_prob": max_prob, "max_logit": max_logit, "tgt": self_tgt}


def _quantile(x, invalid_u8, q):
    out = torch.empty(1, device=x.device)
    _lib.check(_lib.lib().tan_masked_quantile(_p(x), _p(invalid_u8), C.c_int(x.numel()), C.c_float(q), _p(out), ops._stream()),
               "tan_masked_quantile")
    return out


def _quantile_global(x, invalid_u8, q):
    """_quantile over the sentences of EVERY rank (global-negatives mode: the reference computes these statistics over the
    whole batch, loss.py:191-194,286,315-320): all-gather of the per-sentence values and pad flags, then the same kernel; beyond
    its 8192 values a sort with at::lerp's rounding, still without a host sync."""
    from . import dist as _dist
    xg, ig = _dist.all_gather_cat(x), _dist.all_gather_cat(invalid_u8)
    if xg.numel() <= 8192:
        return _quantile(xg, ig, q)
    inf = torch.full_like(xg, float("inf"))
    xs = torch.where(ig != 0, inf, xg).sort().values
    n = (ig == 0).sum()
    pos = q * (n - 1).to(torch.float32)
    lo = pos.floor().long().clamp(min=0)
    hi = torch.minimum(lo + 1, (n - 1).clamp(min=0))
    w = pos - lo.to(torch.float32)
    a, b = xs[lo], xs[hi]
    return torch.where(w < 0.5, a + w * (b - a), b - (b - a) * (1 - w)).view(1)


def _diag_max(blk, row_leak, B, T, N):
    out = torch.empty(B * N, device=blk.tensor.device)
    _lib.check(_lib.lib().tan_diag_max(blk.ptr, C.c_long(blk.sb), C.c_long(blk.st), _p(row_leak), _p(out), C.c_int(B), C.c_int(T),
                                       C.c_int(N), ops._stream()), "tan_diag_max")
    return out


class FusedSim:
    """What a fused forward hands to get_loss instead of materialised logits: unit features (autograd-connected) of the
    dual and joint paths, stage-major.  vn [S,R,C], tn [1 or S, Mp, C] in the model's compute dtype (bf16)."""

    def __init__(self, vn_d, tn_d, vn_j, tn_j, B, T, N):
        self.vn_d, self.tn_d, self.vn_j, self.tn_j, self.B, self.T, self.N = vn_d, tn_d, vn_j, tn_j, B, T, N
        self.n_text_valid = None      # host-side count of real (unpadded) sentences when the caller knows it: enables column compaction
        self.global_negatives = False # row f3: sentences of every data-parallel rank are negatives (dist_nce.py)
        self.join_event = None        # forward(fused="defer"): the joint features are complete after this event (side stream)

    def diag_blocks(self, which):
        """[B,T,N] f32 last-stage same-video cosine logits (all that self-labelling / thresholding read of the B^2 tensor)."""
        vn, tn = (self.vn_d, self.tn_d) if which == "dual" else (self.vn_j, self.tn_j)
        B, T, N, Cw = self.B, self.T, self.N, vn.shape[-1]
        out = torch.empty(B, T, N, device=vn.device)
        ops.gemm(vn.detach()[-1], tn.detach()[-1], out, M=T, N=N, K=Cw, batch=B, sA=T * Cw, sB=N * Cw, sC=T * N)
        return out


def compaction_prep(col_invalid, n_valid):
    """(gather index [Mc], padded column -> compacted column map [Mp] int32, pad flags of the compacted columns [Mc]) for
    _FusedNCEFn, or None when nothing would be dropped.  Mc = n_valid rounded up to 64 (the d-feature GEMM contracts over Mc in
    64-deep K-steps); a stable sort of the 0/1 pad flags puts the real sentences first, in order: static shapes, no host sync."""
    Mp = col_invalid.shape[0]
    if n_valid is None:
        return None
    Mc = min(Mp, (int(n_valid) + 63) // 64 * 64)
    if Mc >= Mp:
        return None
    idx = torch.sort(col_invalid, stable=True).indices[:Mc]
    colmap = (torch.cumsum(col_invalid == 0, 0, dtype=torch.int32) - 1).masked_fill_(col_invalid != 0, -1)
    return idx, colmap, col_invalid.index_select(0, idx)


class _NCETail(torch.autograd.Function):
    """loss.py:254-275 in two launches: ((mean(v_d|rows) + mean(t_d|cols))/2, (mean(v_j|rows) + mean(t_j|cols))/2)."""

    @staticmethod
    def forward(ctx, v_d, t_d, v_j, t_j, rows_mask, cols_mask, counts=None):
        v_d, t_d, v_j, t_j = (x.contiguous() for x in (v_d, t_d, v_j, t_j))
        (Sd, R), (Sj, M) = v_d.shape, t_j.shape
        assert t_d.shape == (Sd, M) and v_j.shape == (Sj, R)
        out = torch.empty(5, device=v_d.device)               # [loss_dual, loss_joint, their mean, n_rows, n_cols]
        _lib.check(_lib.lib().tan_nce_tail_fwd(_p(v_d), _p(t_d), _p(v_j), _p(t_j), _p(rows_mask), _p(cols_mask), C.c_int(Sd),
                                               C.c_int(Sj), C.c_long(R), C.c_long(M), _p(out), _p(out[3:]), _p(counts), ops._stream()),
                   "tan_nce_tail_fwd")
        ctx.saved = (rows_mask, cols_mask, out, Sd, Sj, R, M)
        ctx.set_materialize_grads(False)          # an unused output costs nothing (a zero fill per output otherwise)
        return out[0], out[1], out[2]             # three scalars: no select / add / div nodes between the tail and the loss

    @staticmethod
    def backward(ctx, g_d, g_j, g_m):
        rows_mask, cols_mask, out, Sd, Sj, R, M = ctx.saved
        dev = out.device
        if g_d is None and g_j is None and g_m is None:
            return (None,) * 7
        g_d, g_j, g_m = (None if g is None else g.contiguous().float() for g in (g_d, g_j, g_m))
        g_v_d, g_t_d = torch.empty(Sd, R, device=dev), torch.empty(Sd, M, device=dev)
        g_v_j, g_t_j = torch.empty(Sj, R, device=dev), torch.empty(Sj, M, device=dev)
        _lib.check(_lib.lib().tan_nce_tail_bwd(_p(g_d), _p(g_j), _p(g_m), _p(rows_mask), _p(cols_mask), _p(out[3:]), C.c_int(Sd),
                                               C.c_int(Sj), C.c_long(R), C.c_long(M), _p(g_v_d), _p(g_t_d), _p(g_v_j), _p(g_t_j),
                                               ops._stream()), "tan_nce_tail_bwd")
        return g_v_d, g_t_d, g_v_j, g_t_j, None, None, None


def _pos_masks(tgt, tpad_u8, B, T, N):
    rows_pos, cols_pos = torch.empty(B * T, device=tgt.device), torch.empty(B * N, device=tgt.device)
    _lib.check(_lib.lib().tan_pos_masks(_p(tgt), _p(tpad_u8), _p(rows_pos), _p(cols_pos), C.c_int(B), C.c_int(T), C.c_int(N),
                                        ops._stream()), "tan_pos_masks")
    return rows_pos, cols_pos


class _FusedNCEFn(torch.autograd.Function):
    """_NCEFn without the logits: tan_simnce_fwd / tan_simnce_bwd_dl + the two d-feature GEMMs.

    `n_valid` (host int, optional) = number of real sentences in the batch.  Padded text columns take part in nothing
    (loss.py:64-70 drops them before the log-sum-exps), so when it is known the sweep runs on the COMPACTED text matrix
    (Mc = n_valid rounded up to 64 columns instead of B*N; ~37 % fewer similarity FLOPs at N ~ U[4,16]); the returned text terms
    stay in the compacted column order [S, Mc] (the callers compact their column masks with the same `prep` index)."""

    @staticmethod
    def forward(ctx, vn, tn, tgt, col_invalid, row_leak, B, T, N, prep=None):
        S, R, Cw = vn.shape
        Mp, dev = B * N, vn.device
        shared = tn.shape[0] == 1
        compact = prep is not None
        if compact:
            idx, colmap, ci_run = prep                       # compaction_prep(): shared by the dual and the joint sweep
            Mc = idx.shape[0]
            tn_run = tn.index_select(1, idx)
        else:
            idx = colmap = None
            Mc, tn_run, ci_run = Mp, tn, col_invalid
        stats = torch.empty(2 * S * R + 2 * S * Mc, device=dev)
        rowsum, possum_v = stats[:S * R], stats[S * R:2 * S * R]
        colsum, possum_t = stats[2 * S * R:2 * S * R + S * Mc], stats[2 * S * R + S * Mc:]
        v_terms, t_run = torch.empty(S, R, device=dev), torch.empty(S, Mc, device=dev)
        L = _lib.lib()
        ws = torch.empty(L.tan_simnce_ws_floats(C.c_int(S), C.c_int(B), C.c_int(T), C.c_int(N)), device=dev)
        # the sweep can keep its exponentials (bf16 [S,R,Mc]) so that the backward is an element-wise pass instead of a second sweep
        want_bwd = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]           # (not the no-grad sweeps of the EMA target)
        ekeep = None
        if want_bwd and L.tan_simnce_keeps(C.c_int(Cw)):
            ekeep = torch.empty(L.tan_simnce_keep_elems(C.c_int(S), C.c_int(R), C.c_int(Mc)), dtype=torch.bfloat16, device=dev)
        fwd_args = (_p(vn), _p(tn_run), C.c_long(0 if shared else Mc * Cw), _p(tgt), _p(ci_run), _p(row_leak),
                    _p(rowsum), _p(colsum), _p(possum_v), _p(possum_t), _p(v_terms), _p(t_run), _p(ws),
                    C.c_int(S), C.c_int(B), C.c_int(T), C.c_int(N), C.c_int(Cw), _p(tn) if compact else None,
                    C.c_long(0 if shared else Mp * Cw), _p(colmap), C.c_int(Mc), C.c_int(0))
        if ekeep is not None:
            _lib.check(L.tan_simnce_fwd_keep(*fwd_args, _p(ekeep), ops._stream()), "tan_simnce_fwd_keep")
        else:
            _lib.check(L.tan_simnce_fwd(*fwd_args, ops._stream()), "tan_simnce_fwd")
        # compacted sweeps return the text terms in COMPACTED column order ([S, Mc]; get_loss compacts the column masks of the tail
        # instead, once and off the critical path): scattering them back to [S, B*N] was a fill + an index_copy per sweep here and
        # a gather per sweep in the backward, all on the serial chain between the stacks and the backward
        t_terms = t_run
        ctx.saved = (vn, tn, tn_run, tgt, ci_run, row_leak, rowsum, colsum, possum_v, possum_t, ws, idx, colmap, ekeep)
        ctx.dims = (S, B, T, N, Cw, shared, Mc)
        return v_terms, t_terms

    @staticmethod
    def backward(ctx, g_v, g_t):
        vn, tn, tn_run, tgt, ci_run, row_leak, rowsum, colsum, possum_v, possum_t, ws, idx, colmap, ekeep = ctx.saved
        S, B, T, N, Cw, shared, Mc = ctx.dims
        R, Mp, dev = B * T, B * N, vn.device
        compact = idx is not None
        g_v = torch.zeros(S, R, device=dev) if g_v is None else g_v.contiguous()
        g_t = torch.zeros(S, Mc, device=dev) if g_t is None else g_t.contiguous()           # [S, Mc]: compacted order, like t_terms
        fusable = ekeep is not None and vn.dtype == torch.bfloat16 and Mc % 8 == 0 and N <= 32 and Mc < 32768
        dl = torch.empty(S, R, Mc, dtype=torch.bfloat16, device=dev)
        bwd_args = (_p(vn), _p(tn_run), C.c_long(0 if shared else Mc * Cw), _p(tgt), _p(ci_run),
                    _p(row_leak), _p(rowsum), _p(colsum), _p(possum_v), _p(possum_t), _p(g_v),
                    _p(g_t), _p(dl), _p(ws), C.c_int(S), C.c_int(B), C.c_int(T), C.c_int(N),
                    C.c_int(Cw), _p(tn) if compact else None, C.c_long(0 if shared else Mp * Cw),
                    _p(colmap), C.c_int(Mc), C.c_int(1 | 2 | 16), ops._stream())      # SWEEP | DIAG | DIAG_KEEP: `ws` still holds the forward's same-video blocks
        d_vn = torch.empty_like(vn)
        if fusable and _FUSED_DVN:
            # d logits and d_vn = dl . tn_run in one pass over the kept exponentials (the tile is the MFMA operand while it is in the LDS)
            _lib.check(_lib.lib().tan_simnce_bwd_dl_dvn_kept(_p(ekeep), *bwd_args[:13], _p(d_vn), *bwd_args[13:]),
                       "tan_simnce_bwd_dl_dvn_kept")
        else:
            if ekeep is not None:
                _lib.check(_lib.lib().tan_simnce_bwd_dl_kept(_p(ekeep), *bwd_args), "tan_simnce_bwd_dl_kept")
            else:
                _lib.check(_lib.lib().tan_simnce_bwd_dl(*bwd_args), "tan_simnce_bwd_dl")
            ops.gemm(dl, tn_run, d_vn, M=R, N=Cw, K=Mc, a_kc=True, b_kc=False, lda=Mc, ldb=Cw, batch=S, sA=R * Mc,
                     sB=0 if shared else Mc * Cw, sC=R * Cw)
        if shared:       # one text feature for every stage: contract over (stage, row) in a single split-K GEMM
            acc = torch.zeros(Mc, Cw, device=dev)
            ops.gemm(dl, vn, acc, M=Mc, N=Cw, K=S * R, a_kc=False, b_kc=False, lda=Mc, ldb=Cw, accumulate=True,
                     split_k=max(1, min(8, S * R // 512)))
            d_run = ops.cast(acc, torch.empty(1, Mc, Cw, dtype=tn.dtype, device=dev))
        else:
            d_run = torch.empty(S, Mc, Cw, dtype=tn.dtype, device=dev)   # S x (Mc/128 x Cw/128) tiles under an R-long contraction
            ops.gemm(dl, vn, d_run, M=Mc, N=Cw, K=R, a_kc=False, b_kc=False, lda=Mc, ldb=Cw, batch=S, sA=R * Mc, sB=R * Cw,
                     sC=Mc * Cw)
        if compact:                                  # back to the padded row order, zeros at the dropped columns: one launch
            d_tn = torch.empty_like(tn)
            _lib.check(_lib.lib().tan_rows_gather(_p(d_run.contiguous()), _p(d_tn), _p(colmap), C.c_int(tn.shape[0]), C.c_long(Mc),
                                                  C.c_long(Mp), C.c_int(Cw), ops._dt(d_tn), ops._stream()), "tan_rows_gather")
        else:
            d_tn = d_run
        return d_vn, d_tn, None, None, None, None, None, None, None


class _ManualCtx:
    """Stand-in for the autograd context of _FusedNCEFn / _NCETail when their forward and backward are driven by hand (Trainer's
    two-chain step)."""
    needs_input_grad = (True, True)

    def set_materialize_grads(self, flag):
        pass


def nce_family(vn, tn, tgt, col_invalid, B, T, N, nv, g_v, g_t):
    """Similarity + multi-positive NCE of ONE family, forward and backward back to back on the current stream (loss.py:240-253 and its
    autograd): -> (v_terms, t_terms, d_vn, d_tn).  g_v [S, R] / g_t [S, Mc]: d loss / d terms (`nce_term_grads`)."""
    ctx = _ManualCtx()
    v_terms, t_terms = _FusedNCEFn.forward(ctx, vn, tn, tgt, col_invalid, None, B, T, N, nv)
    d_vn, d_tn = _FusedNCEFn.backward(ctx, g_v, g_t)[:2]
    return v_terms, t_terms, d_vn, d_tn


_SIMFAM = os.environ.get("TAN_SIMFAM", "1") != "0"
# the statistics sweep L2-normalises its frame panel itself (TAN_SIMFAM_NORM_IN_SWEEP); 0 = a separate l2n_fwd launch in front of it
_SIMFAM_NORM = os.environ.get("TAN_SIMFAM_NORM", "1") != "0"


def simfam_ok(S, N, Mc, dtype, T=64):
    """Shapes `tan_simfam_fwd / bwd` take (include/tan_hip.h): bf16, <= 8 stages, <= 32 sentences per video, sweep columns a multiple
    of 8 within the resident sweep's limit, and a [T, N] same-video block the finishing launch can hold twice in a CU's LDS (the bound
    of `tan_simfam_fwd`: anything larger takes the separate-launch path instead of failing with BAD_ARG -- ADVICE r5)."""
    fin_lds = 4 * (2 * T * N + T + 256 + 96) + 4 * 32 + 32 + T + 16
    return (_SIMFAM and dtype == torch.bfloat16 and S <= 8 and N <= 32 and Mc % 8 == 0 and Mc < 32768
            and fin_lds <= 160 * 1024 and Mc <= _lib.lib().tan_simnce_max_cols())


def _ptr8(tensors):
    p = _lib.Ptr8()
    for i, t in enumerate(tensors):
        p.p[i] = t.data_ptr()
    return p


class SimFam:
    """One feature family through `tan_simfam_fwd / tan_simfam_bwd` with its buffers (see `nce_family_stages` for the arguments).
    `run()` = forward and backward back to back (stage 1: the targets and the terms' upstream gradients depend on the batch alone);
    `sweep()` / `finish()` / `backward()` = the same launches in three calls for a loss whose targets (EMA self-labelling) and upstream
    gradients (thresholds over BOTH families' forward results) arrive in between (stage 2, `Trainer._forward_backward_chains2`):
    `tgt`, `g_v`, `g_t` are then buffers that other launches fill before the call that reads them."""

    def __init__(self, x_video, v_grp, x_text, t_grp, d_video, d_text, tgt, col_invalid, B, T, N, nv, g_v, g_t, split_k=0):
        S, St = len(x_video), len(x_text)
        R, Mp = B * T, B * N
        dev, Cw = x_video[0].device, x_video[0].shape[-1]
        Mc = nv[0].shape[0] if nv is not None else Mp
        L = _lib.lib()
        bf = torch.bfloat16
        d = self.d = _lib.SimFamDesc()
        d.S, d.St, d.B, d.T, d.N, d.C, d.Mc, d.flags = S, St, B, T, N, Cw, Mc, (1 if _SIMFAM_NORM else 0)
        d.x_video, d.v_grp_rows, d.v_off = _ptr8(x_video), v_grp[0], v_grp[1]
        d.x_text, d.t_grp_rows, d.t_off = _ptr8(x_text), t_grp[0], t_grp[1]
        if nv is not None:
            d.idx, d.colmap, d.col_invalid = nv[0].data_ptr(), nv[1].data_ptr(), nv[2].data_ptr()
        else:
            d.col_invalid = col_invalid.data_ptr()
        d.tgt = tgt.data_ptr()
        # one f32 block (saved sums, terms, norms, the text-gradient accumulator; pieces 16-byte aligned) + the bf16 tensors
        sizes = [S * R] * 4 + [S * Mc] * 3 + [St * Mc, St * Mc * Cw]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 3) // 4 * 4)
        f32 = torch.empty(offs[-1], device=dev)
        rowsum, possum_v, inv_v, v_terms, colsum, possum_t, t_terms, inv_t, acc = (f32[a:a + n] for a, n in zip(offs, sizes))
        vn = torch.empty(S, R, Cw, dtype=bf, device=dev)
        tn = torch.empty(St, Mc, Cw, dtype=bf, device=dev)
        n_keep, n_ws = L.tan_simnce_keep_elems(S, R, Mc), L.tan_simfam_ws_bytes(S, St, B, T, N, Mc)       # (`long`: restype c_long, _lib.lib())
        assert n_keep > 0 and n_ws > 0, (n_keep, n_ws)
        ekeep = torch.empty(n_keep, dtype=bf, device=dev)
        dl = torch.empty(S * R * Mc + 256, dtype=bf, device=dev)      # (+ slack: the 256-wide tiles of tan_gemm_atb read past a ragged Mc)
        ws = torch.empty(n_ws, dtype=torch.uint8, device=dev)
        d.vn, d.inv_v, d.tn, d.inv_t = vn.data_ptr(), inv_v.data_ptr(), tn.data_ptr(), inv_t.data_ptr()
        d.rowsum, d.colsum, d.possum_v, d.possum_t = rowsum.data_ptr(), colsum.data_ptr(), possum_v.data_ptr(), possum_t.data_ptr()
        d.e_keep, d.ws = ekeep.data_ptr(), ws.data_ptr()
        d.v_terms, d.t_terms = v_terms.data_ptr(), t_terms.data_ptr()
        d.dl, d.d_tn_acc = dl.data_ptr(), acc.data_ptr()
        d.d_video, d.d_text = _ptr8(d_video), _ptr8(d_text)
        d.dtn_split_k = split_k          # (0: the library's choice -- include/tan_hip.h)
        self.g_v, self.g_t = g_v, g_t
        self._acc = acc
        self.v_terms, self.t_terms = v_terms.view(S, R), t_terms.view(S, Mc)
        # the last stage's same-video cosines [B, T, N] f32 inside `ws`, written by the finishing launch (train/loss.py:280-283 reads them)
        off = L.tan_simfam_diag_offset(S, St, B, T, N, Mc, S - 1)
        self.diag_last = ws[off:off + 4 * B * T * N].view(torch.float32).view(B, T, N)
        self._keep = (f32, vn, tn, ekeep, dl, ws, tgt, col_invalid, nv, x_video, x_text, d_video, d_text)
        self.base_flags = d.flags

    def _fwd(self, flags, with_g):
        d = self.d
        d.flags = self.base_flags | flags
        d.g_v, d.g_t = (self.g_v.data_ptr(), self.g_t.data_ptr()) if with_g else (None, None)
        _lib.check(_lib.lib().tan_simfam_fwd(C.byref(d), ops._stream()), "tan_simfam_fwd")
        self.base_flags |= d.flags & 2           # TAN_SIMFAM_CORR_DONE

    def run(self):
        self._fwd(0, True)
        self.backward()

    def sweep(self):
        self._acc.zero_()                        # the text-gradient accumulator, cleared here instead of in the backward's first launch
        self.base_flags |= 16                    # TAN_SIMFAM_ACC_ZEROED
        self._fwd(4, False)                      # TAN_SIMFAM_SWEEP_ONLY

    def finish(self):
        self._fwd(8, False)                      # TAN_SIMFAM_FINISH_ONLY (the upstream gradients are not known yet: no corrections)

    def backward(self):
        d = self.d
        d.flags = self.base_flags
        d.g_v, d.g_t = self.g_v.data_ptr(), self.g_t.data_ptr()
        _lib.check(_lib.lib().tan_simfam_bwd(C.byref(d), ops._stream()), "tan_simfam_bwd")

    def record_stream(self, stream):
        for t in self._keep[:6]:
            t.record_stream(stream)


def simfam_stages_ok(x_video, x_text, N, nv, B, T):
    S, St = len(x_video), len(x_text)
    Mc = nv[0].shape[0] if nv is not None else B * N
    return simfam_ok(S, N, Mc, x_video[0].dtype, T) and x_video[0].shape[-1] == 512 and St in (1, S)


def nce_family_stages(x_video, v_grp, x_text, t_grp, d_video, d_text, tgt, col_invalid, B, T, N, nv, g_v, g_t, split_k=0):
    """Similarity + multi-positive NCE of ONE family from the stacks' stage OUTPUTS to their stage GRADIENTS, forward and backward
    back to back on the current stream (tan_model.py:116-119 / 136-139, loss.py:240-253 and their autograd) -> (v_terms, t_terms).
      x_video  list of S stage buffers; frame row r = b*T + t at row (r // T) * v_grp[0] + v_grp[1] + r % T
      x_text   list of 1 (dual: the text embedding) or S (joint: the stack's stages) buffers; sentence m = b*N + k at row
               (m // N) * t_grp[0] + t_grp[1] + m % N
      d_video / d_text  where the gradients go, addressed the same way (every row is written: dropped sentences get zeros)
      nv       column compaction (idx, colmap, pad flags of the sweep's columns) or None
      g_v [S, R] / g_t [S, Mc]  d loss / d terms (`nce_term_grads`)
    Six to seven launches through `tan_simfam_fwd / tan_simfam_bwd` (18 before); shapes those do not take run the same arithmetic as
    separate launches (`nce_family` between L2-normalisation launches)."""
    S, St = len(x_video), len(x_text)
    R, Mp = B * T, B * N
    dev, cd, Cw = x_video[0].device, x_video[0].dtype, x_video[0].shape[-1]
    if not simfam_stages_ok(x_video, x_text, N, nv, B, T):
        vn = torch.empty(S, R, Cw, dtype=cd, device=dev)
        tn = torch.empty(St, Mp, Cw, dtype=cd, device=dev)
        inv_v, inv_t = torch.empty(S * R, device=dev), torch.empty(St * Mp, device=dev)
        ops.l2norm_fwd_multi(x_video, vn, inv_v, R, Cw, T, v_grp[0], v_grp[1])
        ops.l2norm_fwd_multi(x_text, tn, inv_t, Mp, Cw, N, t_grp[0], t_grp[1])
        v_terms, t_terms, d_vn, d_tn = nce_family(vn, tn, tgt, col_invalid, B, T, N, nv, g_v, g_t)
        ops.l2norm_bwd_multi(d_vn, vn, inv_v, d_video, R, Cw, T, v_grp[0], v_grp[1])
        ops.l2norm_bwd_multi(d_tn.view(St, Mp, Cw), tn, inv_t, d_text, Mp, Cw, N, t_grp[0], t_grp[1])
        return v_terms, t_terms
    fam = SimFam(x_video, v_grp, x_text, t_grp, d_video, d_text, tgt, col_invalid, B, T, N, nv, g_v, g_t, split_k)
    fam.run()
    return fam.v_terms, fam.t_terms


def nce_term_grads(rows_mask, cols_mask, Sd, Sj):
    """d loss_mean / d (v_d, t_d, v_j, t_j) of _NCETail for loss = (loss_dual + loss_joint) / 2 (loss.py:254-275,359-373): a function of
    the two masks alone (mean weights 1 / (S count)), so it is known before any similarity is -- what lets a family's backward start
    as soon as its own forward is through.  -> (g_v_d, g_t_d, g_v_j, g_t_j, counts)."""
    dev = rows_mask.device
    R, M = rows_mask.shape[0], cols_mask.shape[0]
    counts = torch.stack([rows_mask.sum(), cols_mask.sum()])
    one = torch.ones(1, device=dev)
    g_v_d, g_t_d = torch.empty(Sd, R, device=dev), torch.empty(Sd, M, device=dev)
    g_v_j, g_t_j = torch.empty(Sj, R, device=dev), torch.empty(Sj, M, device=dev)
    _lib.check(_lib.lib().tan_nce_tail_bwd(None, None, _p(one), _p(rows_mask), _p(cols_mask), _p(counts), C.c_int(Sd), C.c_int(Sj),
                                           C.c_long(R), C.c_long(M), _p(g_v_d), _p(g_t_d), _p(g_v_j), _p(g_t_j), ops._stream()),
               "tan_nce_tail_bwd")
    return g_v_d, g_t_d, g_v_j, g_t_j, counts


_SIDE = {}


def _side_stream(dev):
    st = _SIDE.get(dev)
    if st is None:
        st = _SIDE[dev] = _lib.role_stream(dev, "loss")
    return st


def _stage2_fused(fused, Mp):
    """tan_stage2_masks computes rank-local batch statistics of up to 8192 sentences (global negatives: all-rank statistics, torch glue
    with collectives); TAN_STAGE2_FUSED=0 keeps the torch glue (A/B and the parity test)."""
    return (not getattr(fused, "global_negatives", False)) and Mp <= 8192 and os.environ.get("TAN_STAGE2_FUSED", "1") != "0"


class _BCESelFn(torch.autograd.Function):
    """BCE-with-logits of the alignability head over the selected sentences with pos_weight = 1/mean(label) - 1, and the head's top-1
    agreement (train/loss.py:341-350): returns [bce, top1]; one launch each way."""

    @staticmethod
    def forward(ctx, x, y, sel, scal):
        x = x.contiguous()
        out = torch.empty(2, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().tan_bce_sel_fwd(_p(x), _p(y), _p(sel), _p(scal), C.c_int(x.numel()), _p(out), ops._stream()),
                   "tan_bce_sel_fwd")
        ctx.save_for_backward(x, y, sel, scal)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, sel, scal = ctx.saved_tensors
        dx = torch.empty_like(x)
        g = g.contiguous()
        _lib.check(_lib.lib().tan_bce_sel_bwd(_p(x), _p(y), _p(sel), _p(scal), _p(g), C.c_int(x.numel()), _p(dx), ops._stream()),
                   "tan_bce_sel_bwd")
        return dx, None, None, None


def agreement_targets(src_j, src_d, prep, B, T, N, kind, quant=None, tgt=None):
    """Self-labelling of both families + their agreement (train/loss.py:88-229, no gradient): the window arg-max per sentence on the
    last-stage same-video logits `src_j` / `src_d` (`_Blocks`), the 30 % confidence quantiles, and the de-duplicated agreement target
    [B, T, N] f32 (written into `tgt` when given).  -> (J, D, tgt, iou [B, N], conf [B, N] u8).  Five launches."""
    dev = src_j.tensor.device
    tpad_u8, vpad_u8 = prep["tpad_u8"], prep["vpad_u8"]
    quant = quant or _quantile
    dur = prep["dur"]                                                                                 # loss.py:113-115
    J = _selflabel(src_j, vpad_u8, tpad_u8, dur, B, T, N)
    D = _selflabel(src_d, vpad_u8, tpad_u8, dur, B, T, N)
    q_j = quant(J["max_logit"].view(-1), tpad_u8.view(-1), 0.3)                                       # loss.py:191-194
    q_d = quant(D["max_logit"].view(-1), tpad_u8.view(-1), 0.3)
    tgt = torch.empty(B, T, N, device=dev) if tgt is None else tgt
    iou = torch.empty(B, N, device=dev)
    conf = torch.empty(B, N, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().tan_agreement(_p(J["tgt"]), _p(D["tgt"]), _p(prep["yt"]), _p(J["max_logit"]), _p(D["max_logit"]),
                                        _p(q_j), _p(q_d), C.c_int(_KIND[kind]), _p(tgt),
                                        _p(iou), _p(conf), C.c_int(B), C.c_int(T), C.c_int(N), ops._stream()), "tan_agreement")
    return J, D, tgt, iou, conf


def stage2_masks(md, mj, tpad_u8, tgt, abs_text_pos, conf, loss_threshold, want_a, B, T, N):
    """Rank-local batch statistics of train/loss.py:280-290,309-328,345 in ONE launch (tan_stage2_masks): z-scores of the per-sentence
    maxima md / mj, the threshold metric and kept-sentence mask, the rows that still own a positive, the alignability labels / selection
    / targets / pos_weight, and confidence-ratio (scal[3], when `conf` is given)."""
    dev = md.device
    R, Mp = B * T, B * N
    f32 = dict(device=dev, dtype=torch.float32)
    s2 = dict(metric=torch.empty(Mp, **f32), th_mask=torch.empty(Mp, dtype=torch.bool, device=dev), th_f=torch.empty(Mp, **f32),
              rows=torch.empty(R, **f32), scal=torch.empty(8, **f32))
    if want_a:
        s2.update(lab=torch.empty(Mp, **f32), sel=torch.empty(Mp, **f32), y=torch.empty(Mp, **f32))
    pos = None
    if want_a and abs_text_pos is not None:
        pos = abs_text_pos.to(dev, torch.float32).contiguous()
        if pos.numel() != 2 * Mp:
            raise ValueError("abs_text_pos must be [B, N, 2]")
    _lib.check(_lib.lib().tan_stage2_masks(_p(md), _p(mj), _p(tpad_u8), _p(tgt), _p(pos), _p(conf),
                                           C.c_float(float(loss_threshold)), C.c_int(int(want_a)), C.c_int(B),
                                           C.c_int(T), C.c_int(N), _p(s2["metric"]), _p(s2["th_mask"]), _p(s2["th_f"]),
                                           _p(s2["rows"]), _p(s2.get("lab")), _p(s2.get("sel")), _p(s2.get("y")),
                                           _p(s2["scal"]), ops._stream()), "tan_stage2_masks")
    return s2


def prepare_inputs(input_data, video_padding_mask, text_padding_mask, T, N, dev, args, n_text_valid=None, want_compaction=False):
    """Everything get_loss derives from the BATCH alone (train/loss.py:58-70,236-237): pad masks in the kernels' formats, the
    start/end target, and -- when no self-labelling rewrites the target -- the positive row / column masks and the text-column
    compaction.  ~20 tiny launches that do not depend on the model: the training driver issues them on a side stream next to the
    forward (`Trainer.forward_backward`), get_loss computes them itself otherwise."""
    B = text_padding_mask.shape[0]
    tgt_raw = input_data.get("_tgt_raw") if isinstance(input_data, dict) else None
    if tgt_raw is None:
        tgt_raw, _, _ = get_mask_from_time(input_data["start"], input_data["end"], T, N, device=dev)   # [B,N,T] bool
    tgt_raw = tgt_raw.contiguous()
    tp = text_padding_mask.to(dev)
    tp = tp.contiguous() if tp.dtype in (torch.float32, torch.bool, torch.uint8) else tp.float().contiguous()
    vp = video_padding_mask.to(dev)
    vp = vp.contiguous() if vp.dtype in (torch.bool, torch.uint8) else vp.bool().contiguous()
    Mp = B * N
    Mc = 0
    if want_compaction and n_text_valid is not None:
        Mc = min(Mp, (int(n_text_valid) + 63) // 64 * 64)
        if Mc >= Mp:
            Mc = 0                                   # nothing would be dropped
    # ONE launch (tan_loss_prep) instead of ~15 tiny ATen kernels: pad masks in the kernels' formats, the transposed f32 target, and
    # the column compaction (a stable partition of the pad flags; was sort + cumsum + compare + masked_fill + gathers)
    tpad_u8 = torch.empty(B, N, dtype=torch.uint8, device=dev)
    vpad_u8 = torch.empty(vp.shape, dtype=torch.uint8, device=dev)
    valid = torch.empty(Mp, dtype=torch.bool, device=dev)
    valid_f = torch.empty(Mp, device=dev)
    tgt = torch.empty(B, T, N, device=dev)
    idx = torch.empty(Mc, dtype=torch.int64, device=dev) if Mc else None
    colmap = torch.empty(Mp, dtype=torch.int32, device=dev) if Mc else None
    ci_run = torch.empty(Mc, dtype=torch.uint8, device=dev) if Mc else None
    is_f = tp.dtype == torch.float32
    _lib.check(_lib.lib().tan_loss_prep(_p(tp) if is_f else None, None if is_f else _p(tp), _p(vp), _p(tgt_raw), _p(tpad_u8), _p(vpad_u8),
                                        _p(valid), _p(valid_f), _p(tgt), _p(idx), _p(colmap), _p(ci_run), C.c_int(B), C.c_int(T),
                                        C.c_int(N), C.c_int(Mc), ops._stream()), "tan_loss_prep")
    prep = {"tpad": tpad_u8.view(torch.bool), "tpad_u8": tpad_u8, "vpad_u8": vpad_u8, "valid": valid, "valid_f": valid_f,
            "tgt_raw": tgt_raw}
    if args.learn_agreement:         # self-labelling inputs that depend on the batch alone (loss.py:113-115), off the critical path
        prep["dur"] = tgt_raw.sum(-1).float().clamp(min=1.0).masked_fill(prep["tpad"], 0.0).contiguous()
        prep["yt"] = tgt_raw.view(torch.uint8) if tgt_raw.dtype == torch.bool else tgt_raw.to(torch.uint8).contiguous()
    else:
        prep["tgt"] = tgt                                                                             # [B,T,N]
        prep["rows_pos"], prep["cols_pos"] = _pos_masks(tgt, tpad_u8, B, T, N)
    if want_compaction:
        prep["nv"], prep["nv_for"] = ((idx, colmap, ci_run) if Mc else None), n_text_valid
        if prep["nv"] is not None and "cols_pos" in prep:
            prep["cols_pos_c"] = prep["cols_pos"].index_select(0, idx)                 # the tail's column mask, compacted order
    return prep


def prepare_inputs_async(input_data, video_padding_mask, text_padding_mask, T, N, dev, args, n_text_valid=None, want_compaction=False):
    """prepare_inputs on the loss side stream; the result carries the event and the tensor list get_loss joins on."""
    main, side = torch.cuda.current_stream(), _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        prep = prepare_inputs(input_data, video_padding_mask, text_padding_mask, T, N, dev, args, n_text_valid, want_compaction)
        prep["_event"] = side.record_event()
    flat = []
    for v in prep.values():
        for x in (v if isinstance(v, (tuple, list)) else (v,)):
            if torch.is_tensor(x) and x.is_cuda:
                flat.append(x)
    prep["_tensors"] = flat
    return prep


def get_loss(input_data, video_seq, text_embed, video_padding_mask, text_padding_mask, logits, args, abs_text_pos=None,
             return_aux=False):
    """Reference signature (train/loss.py:55-57); returns the reference's loss_dict ('loss' carries the graph)."""
    if args.sim != "cos":
        raise NotImplementedError("HIP get_loss supports sim='cos' only (the model only emits cosine logits)")
    cotrain = args.model == "cotrain"
    fused = logits.get("_fused")                    # FusedSim from TemporalAligner.forward(..., fused=True)
    dev = fused.vn_d.device if fused is not None else logits["logits_dual"].device
    if dev.type != "cuda":
        raise _lib.TanHipError("get_loss needs device tensors: the HIP path has no CPU fallback")
    B, T, _ = video_seq.shape
    N = text_embed.shape[1]
    R, Mp = B * T, B * N
    join_ev = getattr(fused, "join_event", None) if fused is not None else None
    if join_ev is not None and (args.learn_agreement or getattr(fused, "global_negatives", False)):
        torch.cuda.current_stream().wait_event(join_ev)      # these paths read the joint features on this stream right away
        join_ev = None
    if fused is None:
        lg_d, lg_j = _stage_major(logits["logits_dual"]), _stage_major(logits["logits_joint"])
        blk_d, blk_j = _Blocks.of_logits(lg_d.detach(), B, T, N), _Blocks.of_logits(lg_j.detach(), B, T, N)
    else:
        blk_d = blk_j = None                        # built lazily from the features below
    prep = input_data.get("_loss_prep") if isinstance(input_data, dict) else None
    if prep is not None:             # computed ahead by the training driver on a side stream (prepare_inputs): join it here
        main = torch.cuda.current_stream()
        main.wait_event(prep["_event"])
        for v in prep["_tensors"]:
            v.record_stream(main)
    else:
        prep = prepare_inputs(input_data, video_padding_mask, text_padding_mask, T, N, dev, args)
    tpad, tpad_u8, vpad_u8, valid, valid_f, tgt_raw = (prep[k] for k in ("tpad", "tpad_u8", "vpad_u8", "valid", "valid_f", "tgt_raw"))
    out, aux = {}, {}

    row_leak = None
    if args.learn_agreement:
        with torch.no_grad():
            ema_fused = logits.get("ema-_fused") if cotrain else None
            if cotrain and ema_fused is not None:
                src_j, src_d = _Blocks.of_diag(ema_fused.diag_blocks("joint")), _Blocks.of_diag(ema_fused.diag_blocks("dual"))
            elif cotrain:
                src_j = _Blocks.of_logits(_stage_major(logits["ema-logits_joint"]), B, T, N)
                src_d = _Blocks.of_logits(_stage_major(logits["ema-logits_dual"]), B, T, N)
            elif fused is not None:
                blk_j, blk_d = _Blocks.of_diag(fused.diag_blocks("joint")), _Blocks.of_diag(fused.diag_blocks("dual"))
                src_j, src_d = blk_j, blk_d
            else:
                src_j, src_d = blk_j, blk_d
            quant = _quantile_global if getattr(fused, "global_negatives", False) else _quantile
            J, D, tgt, iou, conf = agreement_targets(src_j, src_d, prep, B, T, N, args.temporal_agreement_type, quant)
            conf_done = False        # folded into tan_stage2_masks below when that launch runs anyway
            if not ((args.loss_threshold > 0 or args.use_alignability_head) and _stage2_fused(fused, Mp)):
                out["confidence-ratio"] = (conf.view(Mp).float() * valid_f).sum() / valid_f.sum()
                conf_done = True
            out["iou-threshold"] = torch.full((), 0.5, device=dev)       # (torch.tensor(x, device=...) is a SYNCHRONOUS host-to-device copy)
            if not cotrain:      # reference in-place quirk: the -6e4 fills leak into the online logits (loss.py:96-101)
                row_leak = vpad_u8.view(R)
            aux.update(max_position_joint=J["max_pos"], max_position_dual=D["max_pos"], max_logits_joint=J["max_logit"],
                       max_logits_dual=D["max_logit"], joint_self_tgt=J["tgt"], dual_self_tgt=D["tgt"], iou=iou,
                       confidence_mask=conf, agreement_tgt=tgt)
        rows_pos, cols_pos = _pos_masks(tgt, tpad_u8, B, T, N)                                        # loss.py:236-237
    else:
        tgt, rows_pos, cols_pos = prep["tgt"], prep["rows_pos"], prep["cols_pos"]                     # [B,T,N]; loss.py:236-237

    nce_counts = None            # global (all-rank) mask sums in global-negatives mode, else the tail kernel counts locally
    ci = tpad_u8.view(Mp)
    if fused is None:
        v_d, t_d = _NCEFn.apply(lg_d, tgt, ci, row_leak, B, T, N)
        v_j, t_j = _NCEFn.apply(lg_j, tgt, ci, row_leak, B, T, N)
    elif getattr(fused, "global_negatives", False):
        # row f3: every rank's sentences are negatives (dist_nce.py); the collectives stay on the current stream, in order
        from .dist_nce import _GlobalNCEFn, global_counts
        v_d, t_d = _GlobalNCEFn.apply(fused.vn_d, fused.tn_d, tgt, ci, row_leak, B, T, N)
        v_j, t_j = _GlobalNCEFn.apply(fused.vn_j, fused.tn_j, tgt, ci, row_leak, B, T, N)
        nce_counts = global_counts(rows_pos, cols_pos)
    else:
        # host-side count of real sentences (no sync), or None: padded text columns are then skipped by both sweeps
        n_text_valid = getattr(fused, "n_text_valid", None)
        nv = prep["nv"] if ("nv" in prep and prep["nv_for"] == n_text_valid) else compaction_prep(ci, n_text_valid)
        # The dual and joint similarity sweeps are independent until the final mean: the joint one runs on a second HIP
        # stream (each sweep alone fills 75 % of the workgroup slots).  autograd replays a node's backward on the stream its
        # forward ran on and synchronises producer/consumer streams itself, so the two backward chains (d-logits + the two
        # feature-gradient GEMMs each) overlap as well.
        main = torch.cuda.current_stream()
        side = _side_stream(dev)
        if side is not None:
            side.wait_stream(main)
            if join_ev is not None:          # deferred join of the forward: only the joint sweep waits for the joint stack
                side.wait_event(join_ev)
            with torch.cuda.stream(side):
                v_j, t_j = _FusedNCEFn.apply(fused.vn_j, fused.tn_j, tgt, ci, row_leak, B, T, N, nv)
            v_d, t_d = _FusedNCEFn.apply(fused.vn_d, fused.tn_d, tgt, ci, row_leak, B, T, N, nv)
            main.wait_stream(side)
            v_j.record_stream(main); t_j.record_stream(main)
        else:
            v_d, t_d = _FusedNCEFn.apply(fused.vn_d, fused.tn_d, tgt, ci, row_leak, B, T, N, nv)
            v_j, t_j = _FusedNCEFn.apply(fused.vn_j, fused.tn_j, tgt, ci, row_leak, B, T, N, nv)
    cols_idx = None                 # the sweeps ran on compacted text columns: their text terms are [S, Mc] in that order
    if fused is not None and not getattr(fused, "global_negatives", False) and nv is not None:
        cols_idx = nv[0]
        # prep's compacted column mask belongs to prep's OWN index: when the compaction was redone above (another n_text_valid), the
        # mask is compacted with the index the sweeps actually used
        own = nv is prep.get("nv")
        cols_tail = prep["cols_pos_c"] if (own and not args.learn_agreement and "cols_pos_c" in prep) else cols_pos.index_select(0, cols_idx)
    else:
        cols_tail = cols_pos
    loss_dual, loss_joint, loss_mean = _NCETail.apply(v_d, t_d, v_j, t_j, rows_pos, cols_tail, nce_counts)
    out["loss-dual"], out["loss-joint"] = loss_dual.detach(), loss_joint.detach()

    if args.loss_threshold > 0 or args.use_alignability_head:
        with torch.no_grad():
            if blk_d is None:
                blk_j, blk_d = _Blocks.of_diag(fused.diag_blocks("joint")), _Blocks.of_diag(fused.diag_blocks("dual"))
            md = _diag_max(blk_d, row_leak, B, T, N)                                                  # loss.py:280
            mj = _diag_max(blk_j, row_leak, B, T, N)                                                  # loss.py:283
            glob_stats = nce_counts is not None      # global negatives: batch statistics over the sentences of every rank
            quant = _quantile_global if glob_stats else _quantile
            s2 = None
        if _stage2_fused(fused, Mp):
            # rank-local statistics: z-scores, threshold, kept mask, surviving rows, alignability labels / counts / pos_weight and
            # confidence-ratio in ONE launch (tan_stage2_masks; ~70 tiny ATen kernels before)
            with torch.no_grad():
                cf = conf if (args.learn_agreement and not conf_done) else None
                s2 = stage2_masks(md, mj, tpad_u8, tgt, abs_text_pos, cf, args.loss_threshold, bool(args.use_alignability_head), B, T, N)
                th_mask, th_f, rows_pos_th = s2["th_mask"], s2["th_f"], s2["rows"]
                if cf is not None:
                    out["confidence-ratio"] = s2["scal"][3]
                aux.update(t_th_mask=th_mask, max_logits_dual_per_text=md, max_logits_joint_per_text=mj)
        else:
          with torch.no_grad():

              def gsum(x):
                  if glob_stats:
                      from . import dist as _dist
                      _dist.allreduce_sum_(x)
                  return x
              n_valid = gsum(valid_f.sum())

              def zscore(x):
                  mean = gsum((x * valid_f).sum()) / n_valid
                  var = gsum((((x - mean) ** 2) * valid_f).sum()) / (n_valid - 1)
                  return (x - mean) / var.sqrt()

              metric = -(zscore(md) + zscore(mj))
              th = quant(metric, tpad_u8.view(-1), float(args.loss_threshold))                          # loss.py:286
              th_mask = (metric <= th) & valid
              th_f = th_mask.float()
              tgt_valid = tgt * (~tpad)[:, None, :].float()
              rows_pos_th = ((tgt_valid * th_f.view(B, 1, N)).sum(-1) > 0).view(R).float()              # loss.py:288-290
              aux.update(t_th_mask=th_mask, max_logits_dual_per_text=md, max_logits_joint_per_text=mj)
        glob = nce_counts is not None        # global negatives: the rank losses are SUMMED over ranks (gradients too), so every mean
        #                                      below divides its rank-local sum by the ALL-rank count (a rank-local mean
        #                                      would weigh these terms W times too much against the globally normalised NCE)
        if args.loss_threshold > 0:
            out["loss-dual-all"], out["loss-joint-all"] = loss_dual.detach(), loss_joint.detach()
            th_counts = None
            if glob:
                from .dist_nce import global_counts
                th_counts = global_counts(rows_pos_th, th_f)
            loss_dual_th, loss_joint_th, loss_mean_th = _NCETail.apply(
                v_d, t_d, v_j, t_j, rows_pos_th, th_f if cols_idx is None else th_f.index_select(0, cols_idx), th_counts)
            out["loss-dual"], out["loss-joint"] = loss_dual_th.detach(), loss_joint_th.detach()
        if args.use_alignability_head and s2 is not None:
            aux["t_align_th_mask"] = s2["lab"]
            a_joint = logits["joint_logits_alignability"][:, 2, :, 0].reshape(Mp)                     # stage index 2 (loss.py:341)
            bt = _BCESelFn.apply(a_joint, s2["y"], s2["sel"], s2["scal"])
            bce_joint = bt[0]
            out["loss-joint-bce"] = bce_joint.detach()
            out["alignability_top1"] = bt[1].detach()
        elif args.use_alignability_head:
            with torch.no_grad():
                med_d = quant(md, tpad_u8.view(-1), 0.5)                                              # loss.py:315-320
                med_j = quant(mj, tpad_u8.view(-1), 0.5)
                lab = torch.full_like(metric, 2.0)
                lab = lab.masked_fill((md > med_d) & (mj > med_j), 1.0)
                lab = lab.masked_fill((md < med_d) & (mj < med_j), 0.0)
                if abs_text_pos is not None:                                                          # loss.py:325-328
                    centre = abs_text_pos.to(dev).mean(-1).view(Mp)
                    lab = lab.masked_fill((centre < 0.2) | (centre > 0.8), 0.0)
                sel = ((lab != 2) & valid).float()
                y = lab * sel
                n_sel, n_pos = sel.sum(), y.sum()
                if glob:
                    from . import dist as _dist
                    cnt = torch.stack([n_sel, n_pos])
                    _dist.allreduce_sum_(cnt)
                    n_sel, n_pos = cnt[0], cnt[1]
                pos_weight = n_sel / n_pos - 1.0                                                      # 1/mean(y) - 1
                aux["t_align_th_mask"] = torch.where(valid, lab, torch.full_like(lab, float("nan")))
            a_joint = logits["joint_logits_alignability"][:, 2, :, 0].reshape(Mp)                     # stage index 2 (loss.py:341)
            bce = F.binary_cross_entropy_with_logits(a_joint, y, pos_weight=pos_weight.expand(Mp), reduction="none")
            bce_joint = (bce * sel).sum() / n_sel
            out["loss-joint-bce"] = bce_joint.detach()
            out["alignability_top1"] = ((((a_joint.detach() > 0).float() == y).float()) * sel).sum() / sel.sum()

    nce_w = 0 if args.optim_policy == "bce" else 1
    if args.loss_threshold > 0:
        out["loss-total"] = loss_mean.detach()
        loss = loss_mean_th
    else:
        loss = loss_mean
    if args.use_alignability_head:
        loss = loss * nce_w + bce_joint
    out["loss"] = loss
    return (out, aux) if return_aux else out
