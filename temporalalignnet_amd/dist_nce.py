"""Global-negative NCE across data-parallel ranks: the multi-positive NCE of
train/loss.py:240-275 over the GLOBAL batch of W ranks x B_local videos -- the semantics of the reference run at
B = W*B_local on one device (train/readme.md:10 trains with B=128; `einsum('astc,bkc->astbk')`, model/tan_model.py:118,138,
makes every other video's sentences negatives) -- without ever holding the [W*B*T, W*B*N] logits, or even one rank-pair
block of them, outside MFMA accumulators.

Scheme ("W blocks per rank"): every rank keeps its own video rows and sweeps them against the text features of every rank.
    forward   all-gather  tn, pad flags                            (text side only: (1|S)*B*N*512 bf16 per rank, ~2-15 MB)
              for q in ranks:  tan_simnce_fwd(SWEEP [| ACC_ROWS])  row sums += , column sums of block q (local rows only)
              own block:       tan_simnce_fwd(DIAG)                positives, padded-frame quirk
              all-reduce  column-sum table [W,S,B*N] f32           (tiny)
              tan_simnce_fwd(TERMS)                                v_terms for local rows, t_terms for local sentences
    backward  all-gather  g_t                                      (tiny)
              for q in ranks:  tan_simnce_bwd_dl(SWEEP [| DIAG])   d logits of block q, bf16
                               d_vn += dl_q tn_q ;  d_tn_part[q] = dl_q^T vn
              reduce-scatter d_tn_part                             (text side again)
The per-rank cost is W similarity sweeps instead of one: that is what global negatives are.  Loss normalisation is global:
each rank divides its local sums by the GLOBAL number of rows / sentences that own a positive, so the global loss is the SUM of
the rank losses and gradients are summed (not averaged) over ranks (`Trainer(global_negatives=True)` passes grad_scale=1).

`BlockNCE` holds one rank's state and exposes the phases; `_GlobalNCEFn` strings them together with torch.distributed
collectives (RCCL over xGMI when the tensors are on MI355X GPUs).  Every rank must present the same B_local and the same
padded sentence count N per step (the all-gathers exchange equal-shaped blocks): pad `text_embed` / `text_padding_mask` to the
loader's maximum when the per-batch maximum can differ between ranks.  Batch-global statistics of the stage-2 extras (quantiles of
loss.py:191-194,286,315-320) stay local to the rank: only the NCE core is made global here.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib, ops

SWEEP, DIAG, TERMS, ACC_ROWS = 1, 2, 4, 8          # TAN_SIM_* of include/tan_hip.h


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class BlockNCE:
    """One rank, one logits family.  vn [S,R,C] bf16 unit video features (local rows), tgt [B,T,N] f32 positives of the local
    videos, row_leak u8 [R] or None.  Text blocks are passed per call: tn_q [1|S, Mp, C] bf16, pad flags u8 [Mp]."""

    def __init__(self, vn, tgt, row_leak, B, T, N, shared_text: bool, keep: bool = False):
        self.vn, self.tgt, self.row_leak = vn, tgt, row_leak
        self.want_keep = keep          # a backward will follow: the sweeps store their exponentials
        self.S, self.R, self.C = vn.shape
        self.B, self.T, self.N, self.Mp, self.shared = B, T, N, B * N, shared_text
        dev = vn.device
        L = _lib.lib()
        self.ws = torch.empty(L.tan_simnce_ws_floats(C.c_int(self.S), C.c_int(B), C.c_int(T), C.c_int(N)), device=dev)
        self.rowsum = torch.empty(self.S, self.R, device=dev)
        self.possum_v = torch.empty(self.S, self.R, device=dev)
        self.possum_t = torch.empty(self.S, self.Mp, device=dev)
        self._dummy_v, self._dummy_t = torch.empty(self.S, self.R, device=dev), torch.empty(self.S, self.Mp, device=dev)

    def _stride(self):
        return 0 if self.shared else self.Mp * self.C

    def _fwd(self, tn, ci, colsum, v_terms, t_terms, phases, keep=None):
        args = (_p(self.vn), _p(tn), C.c_long(self._stride()), _p(self.tgt), _p(ci), _p(self.row_leak), _p(self.rowsum), _p(colsum),
                _p(self.possum_v), _p(self.possum_t), _p(v_terms), _p(t_terms), _p(self.ws), C.c_int(self.S), C.c_int(self.B),
                C.c_int(self.T), C.c_int(self.N), C.c_int(self.C), None, C.c_long(0), None, C.c_int(0), C.c_int(phases))
        if keep is not None:      # the sweep also stores its exponentials (bf16): the backward is an element-wise pass
            _lib.check(_lib.lib().tan_simnce_fwd_keep(*args, _p(keep), ops._stream()), "tan_simnce_fwd_keep")
        else:
            _lib.check(_lib.lib().tan_simnce_fwd(*args, ops._stream()), "tan_simnce_fwd")

    # ------------------------------------------------------------------ forward phases
    def sweep(self, tn_blocks, ci_blocks, own: int):
        """Row sums over every block; returns this rank's column-sum contributions [W,S,Mp] (own block already corrected for
        the padded-frame quirk).  Positives of the local rows / sentences are computed from the own block."""
        W = len(tn_blocks)
        colparts = torch.empty(W, self.S, self.Mp, device=self.vn.device)
        L = _lib.lib()
        self._keep = None
        if self.want_keep:
            if L.tan_simnce_keeps(C.c_int(self.C)):
                n = L.tan_simnce_keep_elems(C.c_int(self.S), C.c_int(self.R), C.c_int(self.Mp))
                self._keep = [torch.empty(n, dtype=torch.bfloat16, device=self.vn.device) for _ in range(W)]
        for q in range(W):
            self._fwd(tn_blocks[q], ci_blocks[q], colparts[q], self._dummy_v, self._dummy_t, SWEEP | (ACC_ROWS if q else 0),
                      keep=self._keep[q] if self._keep else None)
        self._fwd(tn_blocks[own], ci_blocks[own], colparts[own], self._dummy_v, self._dummy_t, DIAG)
        self._tn, self._ci, self._own = tn_blocks, ci_blocks, own
        return colparts

    def finish(self, colsum_all):
        """colsum_all [W,S,Mp]: column sums over the rows of ALL ranks -> (v_terms [S,R], t_terms [S,Mp]) of this rank."""
        self.colsum_all = colsum_all
        v_terms, t_terms = torch.empty(self.S, self.R, device=self.vn.device), torch.empty(self.S, self.Mp, device=self.vn.device)
        self._fwd(self._tn[self._own], self._ci[self._own], colsum_all[self._own], v_terms, t_terms, TERMS)
        return v_terms, t_terms

    # ------------------------------------------------------------------ backward phase
    def backward(self, g_v, g_t_all):
        """g_v [S,R] (local rows), g_t_all [W,S,Mp] (every rank's sentences) -> d_vn [S,R,C] bf16 and this rank's contributions
        d_tn_parts [W, 1|S, Mp, C] f32 to every rank's text-feature gradient."""
        S, R, Cw, Mp, dev = self.S, self.R, self.C, self.Mp, self.vn.device
        W = len(self._tn)
        St = 1 if self.shared else S
        d_vn = torch.empty_like(self.vn)
        d_tn_parts = torch.zeros(W, St, Mp, Cw, device=dev)
        dl = torch.empty(S, R, Mp, dtype=torch.bfloat16, device=dev)
        g_v = g_v.contiguous()
        for q in range(W):
            args = (_p(self.vn), _p(self._tn[q]), C.c_long(self._stride()), _p(self.tgt), _p(self._ci[q]), _p(self.row_leak),
                    _p(self.rowsum), _p(self.colsum_all[q]), _p(self.possum_v), _p(self.possum_t), _p(g_v), _p(g_t_all[q].contiguous()),
                    _p(dl), _p(self.ws), C.c_int(S), C.c_int(self.B), C.c_int(self.T), C.c_int(self.N), C.c_int(Cw), None, C.c_long(0),
                    None, C.c_int(0), C.c_int(SWEEP | (DIAG if q == self._own else 0)), ops._stream())
            if self._keep:
                _lib.check(_lib.lib().tan_simnce_bwd_dl_kept(_p(self._keep[q]), *args), "tan_simnce_bwd_dl_kept")
            else:
                _lib.check(_lib.lib().tan_simnce_bwd_dl(*args), "tan_simnce_bwd_dl")
            ops.gemm(dl, self._tn[q], d_vn, M=R, N=Cw, K=Mp, a_kc=True, b_kc=False, lda=Mp, ldb=Cw, batch=S, sA=R * Mp,
                     sB=self._stride(), sC=R * Cw, residual=d_vn if q else None)
            if self.shared:
                ops.gemm(dl, self.vn, d_tn_parts[q, 0], M=Mp, N=Cw, K=S * R, a_kc=False, b_kc=False, lda=Mp, ldb=Cw, accumulate=True,
                         split_k=max(1, min(8, S * R // 512)))
            else:
                ops.gemm(dl, self.vn, d_tn_parts[q], M=Mp, N=Cw, K=R, a_kc=False, b_kc=False, lda=Mp, ldb=Cw, batch=S, sA=R * Mp,
                         sB=R * Cw, sC=Mp * Cw, accumulate=True, split_k=max(1, min(4, R // 512)))
        return d_vn, d_tn_parts


# ---------------------------------------------------------------------------------------------------------------------
def _world():
    return (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)


def _all_gather(t):
    W, _ = _world()
    t = t.contiguous()
    if W == 1 and not dist.is_initialized():
        return [t]
    out = [torch.empty_like(t) for _ in range(W)]
    dist.all_gather(out, t)
    return out


def _all_reduce_(t):
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def _reduce_scatter(parts):
    """parts [W, ...] on every rank -> sum over ranks of parts[rank]."""
    W, r = _world()
    if not dist.is_initialized():
        return parts[0]
    if dist.get_backend() == "gloo":          # no reduce-scatter in gloo (CPU / single-GPU test transport): all-reduce + own slice
        full = parts.contiguous()
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
        return full[r].clone()
    out = torch.empty_like(parts[0])
    dist.reduce_scatter_tensor(out, parts.contiguous(), op=dist.ReduceOp.SUM)
    return out


class _GlobalNCEFn(torch.autograd.Function):
    """(vn, tn) -> (v_terms [S,R], t_terms [S,Mp]) of the local rows / sentences against the global batch."""

    @staticmethod
    def forward(ctx, vn, tn, tgt, col_invalid, row_leak, B, T, N):
        W, rank = _world()
        blk = BlockNCE(vn, tgt, row_leak, B, T, N, shared_text=tn.shape[0] == 1,
                       keep=bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        tn_all, ci_all = _all_gather(tn), _all_gather(col_invalid)
        colsum_all = _all_reduce_(blk.sweep(tn_all, ci_all, rank))
        v_terms, t_terms = blk.finish(colsum_all)
        ctx.blk, ctx.tn_dtype = blk, tn.dtype
        return v_terms, t_terms

    @staticmethod
    def backward(ctx, g_v, g_t):
        blk = ctx.blk
        dev = blk.vn.device
        g_v = torch.zeros(blk.S, blk.R, device=dev) if g_v is None else g_v
        g_t = torch.zeros(blk.S, blk.Mp, device=dev) if g_t is None else g_t
        g_t_all = torch.stack(_all_gather(g_t), 0)
        d_vn, parts = blk.backward(g_v, g_t_all)
        d_tn = _reduce_scatter(parts)
        return d_vn, d_tn.to(ctx.tn_dtype), None, None, None, None, None, None


def global_counts(rows_pos, cols_pos):
    """[n_rows_with_positive, n_sentences_with_positive] summed over ranks (f32 [2], device)."""
    c = torch.stack([rows_pos.sum(), cols_pos.sum()])
    return _all_reduce_(c)
