"""Mirror of the reference's model/tfm_model.py module surface (QuickGELU, ResidualAttentionBlock_Step,
TemporalEncoder, get_position_embedding_sine), MI355X-native underneath.

These classes only *hold parameters* under the reference's attribute names, so that `state_dict()` keys
(`resblocks.{i}.attn.in_proj_weight`, `...attn.out_proj.weight`, `...mlp.c_fc.weight`, `...ln_1.weight`, ...) match
the reference checkpoint format.  No ATen arithmetic happens here: a stack is executed by
libtan_hip.so (`tan_encoder_fwd` / `tan_encoder_bwd`, see temporalalignnet_amd/tan_model.py), and calling one of these
modules' own `forward` on its own is routed through the same kernels.
"""
from __future__ import annotations

import math

import torch
from torch import nn


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) -- reference model/tfm_model.py:11-13.  In the HIP path this lives in the c_fc GEMM epilogue."""

    def forward(self, x: torch.Tensor):
        raise RuntimeError("QuickGELU is fused into the c_fc GEMM epilogue of libtan_hip.so; it is not called standalone")


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


class TemporalEncoder(nn.Module):
    """S residual attention blocks with deep-supervision outputs (model/tfm_model.py:41-55)."""

    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([ResidualAttentionBlock_Step(width, heads) for _ in range(layers)])


def get_position_embedding_sine(feature_dim=512, num_features=1024, temperature=10000):
    """Fixed sine table for pos_enc='sine' (model/tfm_model.py:137-149); host-side constant, built once."""
    pos = torch.arange(num_features, dtype=torch.float32)
    pos = pos / (pos[-1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(feature_dim, dtype=torch.float32)
    div = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / feature_dim)
    ang = pos[:, None] / div
    return torch.stack((ang[:, 0::2].sin(), ang[:, 1::2].cos()), dim=2).flatten(1)
