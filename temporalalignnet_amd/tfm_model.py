"""Mirror of the reference's model/tfm_model.py module surface (QuickGELU, ResidualAttentionBlock_Step,
TemporalEncoder, get_position_embedding_sine), MI355X-native underneath.

These classes hold parameters under the reference's attribute names, so that `state_dict()` keys
(`resblocks.{i}.attn.in_proj_weight`, `...attn.out_proj.weight`, `...mlp.c_fc.weight`, `...ln_1.weight`, ...) match
the reference checkpoint format.  No ATen arithmetic happens here.  Inside TemporalAligner a stack
is executed (forward AND backward) by libtan_hip.so through `tan_encoder_fwd` / `tan_encoder_bwd` as part of ONE autograd
node (temporalalignnet_amd/tan_model.py).  Called on their own, `QuickGELU.forward`, `ResidualAttentionBlock_Step.forward`
and `TemporalEncoder.forward` keep the reference's signatures and layouts (sequence-first `[L, B, C]`, bool key-padding
mask, model/tfm_model.py:11-13,34-38,48-55) and run the same kernels -- forward only: they return tensors without an
autograd graph (training goes through TemporalAligner).  fp32 inputs use the exact-f32 MFMA path, bf16 inputs the bf16 one.
"""
from __future__ import annotations

import math

import torch
from torch import nn


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) -- reference model/tfm_model.py:11-13.  In the HIP path this lives in the c_fc GEMM epilogue."""

    def forward(self, x: torch.Tensor):
        import ctypes as C
        from . import _lib, ops
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.lib().tan_quickgelu(ops._ptr(x), ops._ptr(y), C.c_long(x.numel()), ops._dt(x), ops._stream()), "tan_quickgelu")
        return y


class _LayerNormParams(nn.Module):
    """Parameter container with nn.LayerNorm's names (`weight`, `bias`)."""

    def __init__(self, width: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))
        self.eps = 1e-5


class _LinearParams(nn.Module):
    """Parameter container with nn.Linear's names and default init (kaiming_uniform(a=sqrt(5)) + uniform bias)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)


class _MHAParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names: packed `in_proj_weight [3C,C]` (q,k,v order),
    `in_proj_bias`, `out_proj.{weight,bias}`; same default init (xavier_uniform / zeros)."""

    def __init__(self, width: int, heads: int):
        super().__init__()
        self.embed_dim, self.num_heads = width, heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = _LinearParams(width, width)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _MLPParams(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.c_fc = _LinearParams(width, width * 4)
        self.gelu = QuickGELU()
        self.c_proj = _LinearParams(width * 4, width)


class ResidualAttentionBlock_Step(nn.Module):
    """Pre-LN residual block; returns (x, ln_1(x_in)) in the reference (model/tfm_model.py:17-38)."""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = _MHAParams(d_model, n_head)
        self.ln_1 = _LayerNormParams(d_model)
        self.mlp = _MLPParams(d_model)
        self.ln_2 = _LayerNormParams(d_model)

    PARAM_ORDER = ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                   "ln_1.weight", "ln_1.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
                   "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias")

    def forward(self, x: torch.Tensor, key_padding_mask: torch.Tensor = None):
        """(x_out, ln_1(x)) for x [L, B, C] and key_padding_mask [B, L] bool (True = ignore) -- model/tfm_model.py:34-38."""
        outs, xn = _run_blocks([self], x, key_padding_mask)
        return outs[-1], xn[0]


def _run_blocks(blocks, x, key_padding_mask):
    """Run pre-LN blocks through tan_encoder_fwd.  x [L, B, C] sequence-first; returns ([x_out of every block], [ln_1 output of
    every block]) as [L, B, C] tensors.  No autograd graph."""
    import ctypes as C
    from . import _lib, ops
    if not x.is_cuda:
        raise _lib.TanHipError("tfm_model blocks run on the HIP path only: move the module and its input to the GPU")
    L, B, Cw = x.shape
    dt = x.dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise TypeError(f"unsupported dtype {dt}")
    heads = blocks[0].attn.num_heads
    R = B * L
    with torch.no_grad():
        xb = x.detach().permute(1, 0, 2).contiguous().view(R, Cw)            # batch-first rows b*L + t
        n = len(blocks)
        params, bufs = (_lib.LayerParams * n)(), (_lib.LayerBufs * n)()
        keep = []                                                             # tensors the descriptors point into

        def w_(t):                                                            # weights in the activation dtype
            t = t.detach()
            if t.dtype != dt:
                c = torch.empty(t.shape, dtype=dt, device=t.device)
                ops.cast(t.contiguous(), c)
                t = c
            keep.append(t.contiguous())
            return keep[-1].data_ptr()

        def f_(t):
            keep.append(t.detach().float().contiguous())
            return keep[-1].data_ptr()

        acts = []
        for i, blk in enumerate(blocks):
            params[i].w_qkv, params[i].w_out = w_(blk.attn.in_proj_weight), w_(blk.attn.out_proj.weight)
            params[i].w_fc, params[i].w_proj = w_(blk.mlp.c_fc.weight), w_(blk.mlp.c_proj.weight)
            params[i].b_qkv, params[i].b_out = f_(blk.attn.in_proj_bias), f_(blk.attn.out_proj.bias)
            params[i].b_fc, params[i].b_proj = f_(blk.mlp.c_fc.bias), f_(blk.mlp.c_proj.bias)
            params[i].ln1_g, params[i].ln1_b = f_(blk.ln_1.weight), f_(blk.ln_1.bias)
            params[i].ln2_g, params[i].ln2_b = f_(blk.ln_2.weight), f_(blk.ln_2.bias)
            a = {k: torch.empty(R, m * Cw, dtype=dt, device=x.device)
                 for k, m in (("xn1", 1), ("qkv", 3), ("attn_o", 1), ("x_mid", 1), ("xn2", 1), ("h_pre", 4), ("h_act", 4), ("x_out", 1))}
            st = {k: torch.empty(R, dtype=torch.float32, device=x.device) for k in ("mean1", "rstd1", "mean2", "rstd2")}
            st["lse"] = torch.empty(B * heads * L, dtype=torch.float32, device=x.device)
            for k, v in {**a, **st}.items():
                setattr(bufs[i], k, v.data_ptr())
            acts.append(a)
            keep.append(st)
        d = _lib.EncoderDesc()
        d.dtype = ops._dt(xb)
        d.B, d.L, d.C, d.H, d.layers = B, L, Cw, heads, n
        mask = None
        if key_padding_mask is not None:
            mask = key_padding_mask.to(torch.uint8).contiguous()
            d.key_padding_mask = mask.data_ptr()
        d.x0 = xb.data_ptr()
        d.params, d.bufs = params, bufs
        _lib.check(_lib.lib().tan_encoder_fwd(C.byref(d), ops._stream()), "tan_encoder_fwd")
        seq = lambda t: t.view(B, L, Cw).permute(1, 0, 2)
        return [seq(a["x_out"]) for a in acts], [seq(a["xn1"]) for a in acts]


class TemporalEncoder(nn.Module):
    """S residual attention blocks with deep-supervision outputs (model/tfm_model.py:41-55)."""

    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([ResidualAttentionBlock_Step(width, heads) for _ in range(layers)])

    def forward(self, x: torch.Tensor, key_padding_mask: torch.Tensor = None):
        """Deep-supervision outputs of model/tfm_model.py:48-55: the ln_1 outputs of blocks 2..S followed by the last block's
        residual stream -- a list of S tensors [L, B, C] (the caller applies its post-LayerNorm to the last one)."""
        outs, xn = _run_blocks(list(self.resblocks), x, key_padding_mask)
        return xn[1:] + [outs[-1]]


def get_position_embedding_sine(feature_dim=512, num_features=1024, temperature=10000):
    """Fixed sine table for pos_enc='sine' (model/tfm_model.py:137-149); host-side constant, built once."""
    pos = torch.arange(num_features, dtype=torch.float32)
    pos = pos / (pos[-1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(feature_dim, dtype=torch.float32)
    div = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / feature_dim)
    ang = pos[:, None] / div
    return torch.stack((ang[:, 0::2].sin(), ang[:, 1::2].cos()), dim=2).flatten(1)
