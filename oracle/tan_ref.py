"""CPU restatement of the TemporalAligner forward (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Functional style: every function takes a flat {state_dict key: tensor} parameter dict `p`
(same key names as the reference's `TemporalAligner.state_dict()`, tan_model.py:43-72) and
CPU tensors.  Layout is batch-first [B, L, C] throughout; the reference runs the encoder
seq-first [L, B, C] (tan_model.py:168,202) which is the same arithmetic per sequence.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

WIDTH = 512
HEADS = 8
LN_EPS = 1e-5


def quick_gelu(x):
    """x * sigmoid(1.702 x) -- model/tfm_model.py:11-13."""
    return x * torch.sigmoid(1.702 * x)


def layer_norm(x, p, name):
    """nn.LayerNorm(512), eps 1e-5, affine -- model/tfm_model.py:22,28; tan_model.py:50-54."""
    return F.layer_norm(x, (x.shape[-1],), p[f"{name}.weight"], p[f"{name}.bias"], LN_EPS)


def mha(xn, key_padding_mask, p, prefix, heads=HEADS):
    """Self-attention as nn.MultiheadAttention computes it (model/tfm_model.py:21,30-32):
    packed in-proj (q,k,v order), q scaled by dh^-0.5, masked keys -> -inf, softmax over keys,
    out-proj.  No attn mask, dropout 0.  xn: [B,L,C]; key_padding_mask: [B,L] bool (True = ignore)."""
    B, L, C = xn.shape
    dh = C // heads
    qkv = xn @ p[f"{prefix}.attn.in_proj_weight"].t() + p[f"{prefix}.attn.in_proj_bias"]
    q, k, v = qkv.split(C, dim=-1)
    q = q.view(B, L, heads, dh).transpose(1, 2) * (dh ** -0.5)
    k = k.view(B, L, heads, dh).transpose(1, 2)
    v = v.view(B, L, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                   # [B,H,L,L]
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1) @ v                              # [B,H,L,dh]
    a = a.transpose(1, 2).reshape(B, L, C)
    return a @ p[f"{prefix}.attn.out_proj.weight"].t() + p[f"{prefix}.attn.out_proj.bias"]


def block(x, key_padding_mask, p, prefix):
    """ResidualAttentionBlock_Step.forward -- model/tfm_model.py:34-38. Returns (x_out, ln_1(x_in))."""
    xn = layer_norm(x, p, f"{prefix}.ln_1")
    x = x + mha(xn, key_padding_mask, p, prefix)
    h = layer_norm(x, p, f"{prefix}.ln_2")
    h = quick_gelu(h @ p[f"{prefix}.mlp.c_fc.weight"].t() + p[f"{prefix}.mlp.c_fc.bias"])
    x = x + (h @ p[f"{prefix}.mlp.c_proj.weight"].t() + p[f"{prefix}.mlp.c_proj.bias"])
    return x, xn


def encoder(x, key_padding_mask, p, prefix, layers):
    """TemporalEncoder.forward -- model/tfm_model.py:48-55: collect every block's ln_1 output,
    drop the first, append the final residual stream => S deep-supervision features."""
    feats = []
    for i in range(layers):
        x, xn = block(x, key_padding_mask, p, f"{prefix}.resblocks.{i}")
        feats.append(xn)
    return feats[1:] + [x]


def interp_linear(src, size):
    """F.interpolate(mode='linear', align_corners=False) along the first axis of src [L_in, C]
    (tan_model.py:157-160,189-192)."""
    L_in = src.shape[0]
    scale = L_in / size
    pos = (torch.arange(size, dtype=torch.float32) + 0.5) * scale - 0.5
    pos = pos.clamp(min=0.0)
    i0 = pos.floor().long().clamp(max=L_in - 1)
    i1 = (i0 + 1).clamp(max=L_in - 1)
    w1 = (pos - i0.float())[:, None]
    return src[i0] * (1.0 - w1) + src[i1] * w1


def video_embedding(video, p, T, pos_start, interpolate_from=None):
    """ln_video_init(video_pre_proj(video)) + ln_position_init(pos) -- tan_model.py:155-167 / 187-199."""
    x = layer_norm(video @ p["video_pre_proj.weight"].t(), p, "ln_video_init")
    if interpolate_from:
        pos = interp_linear(p["temporal_pos_embed"][0:interpolate_from], T)
    else:
        pos = p["temporal_pos_embed"][pos_start:pos_start + T]
    return x + layer_norm(pos, p, "ln_position_init")[None]


def textual_feature(lang_embed, p):
    """ln_text_init(text_pre_proj(lang)) -- tan_model.py:231-234."""
    return layer_norm(lang_embed @ p["text_pre_proj.weight"].t(), p, "ln_text_init")


def textual_feature_with_time(lang_embed, p, pos_start, interpolate_from=None):
    """tan_model.py:212-228 (only used when use_text_pos_enc=1)."""
    N = lang_embed.shape[1]
    x = textual_feature(lang_embed, p)
    if interpolate_from:
        pos = interp_linear(p["text_temporal_pos_embed"][0:interpolate_from], N)
    else:
        pos = p["text_temporal_pos_embed"][pos_start:pos_start + N]
    return x + layer_norm(pos, p, "ln_position_init")[None]


def visual_feature(video, video_padding_mask, p, E, pos_start=0, interpolate_from=None):
    """get_visual_feature -- tan_model.py:152-179. Returns [B,S,T,C] (last stage post-LN'ed)."""
    B, T, _ = video.shape
    x = video_embedding(video, p, T, pos_start, interpolate_from)
    feats = encoder(x, video_padding_mask, p, "video_temporal_encoder", E)
    feats[-1] = layer_norm(feats[-1], p, "ln_video_post_enc")
    return torch.stack(feats, dim=1)


def joint_feature(video, video_padding_mask, lang_with_time, lang_padding_mask, p, D,
                  pos_start=0, interpolate_from=None):
    """get_joint_feature -- tan_model.py:182-209. Returns ([B,S,T,C], [B,S,N,C])."""
    B, T, _ = video.shape
    x = video_embedding(video, p, T, pos_start, interpolate_from)
    xj = torch.cat([x, lang_with_time], dim=1)
    mj = torch.cat([video_padding_mask, lang_padding_mask], dim=1)
    feats = encoder(xj, mj, p, "joint_temporal_encoder", D)
    feats[-1] = layer_norm(feats[-1], p, "ln_joint_post_enc")
    out = torch.stack(feats, dim=1)
    return out[:, :, :T], out[:, :, T:]


def sine_position_table(feature_dim=512, num_features=1024, temperature=10000.0):
    """pos_enc='sine' (tan_model.py:60-62): positions scaled to [0, 2*pi), channel pair i shares the wavelength
    temperature^(2i/feature_dim); even channels sin, odd channels cos -- model/tfm_model.py:137-149.  Pinned by golden G10."""
    pos = torch.arange(num_features, dtype=torch.float32)
    pos = pos / (pos[-1] + 1e-6) * (2.0 * np.pi)
    idx = torch.arange(feature_dim, dtype=torch.float32)
    wavelength = temperature ** (2.0 * torch.div(idx, 2, rounding_mode="floor") / feature_dim)
    ang = pos[:, None] / wavelength[None, :]
    out = torch.empty(num_features, feature_dim)
    out[:, 0::2] = ang[:, 0::2].sin()
    out[:, 1::2] = ang[:, 1::2].cos()
    return out


def _unit(x):
    """x / ||x||_2 over channels, no epsilon -- tan_model.py:116-117,136-137."""
    return x / x.norm(dim=-1, keepdim=True)


def forward(p, video, lang_embed, video_padding_mask, lang_padding_mask, *, E, D,
            use_alignability_head=False, use_text_pos_enc=False, return_dual_feature=True,
            random_pos_start=False, interpolate_from=None, rng=np.random):
    """TemporalAligner.forward -- tan_model.py:100-149.  When random_pos_start is set the
    position offsets are drawn from `rng` in the reference's order: visual, [text], joint
    (tan_model.py:163,224,195)."""
    B, T, _ = video.shape
    N = lang_embed.shape[1]
    draw = lambda n: int(rng.randint(0, int(n / 2))) if (random_pos_start and not interpolate_from) else 0
    video_out = visual_feature(video, video_padding_mask, p, E, draw(T), interpolate_from)
    lang_raw = textual_feature(lang_embed, p)
    vn, tn = _unit(video_out), _unit(lang_raw)
    logits_dual = torch.einsum("astc,bkc->astbk", vn, tn)
    if use_text_pos_enc:
        lang_t = textual_feature_with_time(lang_embed, p, draw(N), interpolate_from)
    else:
        lang_t = lang_raw
    jv, jt = joint_feature(video, video_padding_mask, lang_t, lang_padding_mask, p, D,
                           draw(T), interpolate_from)
    logits_joint = torch.einsum("astc,bskc->astbk", _unit(jv), _unit(jt))
    out = {"logits_dual": logits_dual, "logits_joint": logits_joint}
    if return_dual_feature:
        out["dual_feature_video"] = vn
        out["dual_feature_text"] = tn
    if use_alignability_head:
        w, b = p["binary_head.weight"], p["binary_head.bias"]
        out["dual_logits_alignability"] = lang_raw @ w.t() + b
        out["joint_logits_alignability"] = jt @ w.t() + b
    return out


def _split_interp(interpolate_from):
    if isinstance(interpolate_from, (list, tuple)):
        assert len(interpolate_from) == 2
        return interpolate_from[0], interpolate_from[1]
    return interpolate_from, None


def text_visual_sim_joint(p, video, lang_embed, *, D, use_text_pos_enc=False, interpolate_from=None,
                          random_pos_start=False, rng=np.random):
    """get_text_visual_sim_joint -- tan_model.py:237-264 (zero masks, within-sample einsum)."""
    vi, ti = _split_interp(interpolate_from)
    B, T, _ = video.shape
    N = lang_embed.shape[1]
    draw = lambda n, itp: int(rng.randint(0, int(n / 2))) if (random_pos_start and not itp) else 0
    lang_t = (textual_feature_with_time(lang_embed, p, draw(N, ti), ti) if use_text_pos_enc
              else textual_feature(lang_embed, p))
    zv = torch.zeros(B, T, dtype=torch.bool)
    zt = torch.zeros(B, N, dtype=torch.bool)
    jv, jt = joint_feature(video, zv, lang_t, zt, p, D, draw(T, vi), vi)
    return torch.einsum("bstc,bskc->bstk", _unit(jv), _unit(jt))


def text_visual_sim_dual(p, video, lang_embed, *, E, interpolate_from=None, random_pos_start=False,
                         rng=np.random):
    """get_text_visual_sim_dual -- tan_model.py:267-283."""
    B, T, _ = video.shape
    lang_raw = textual_feature(lang_embed, p)
    zv = torch.zeros(B, T, dtype=torch.bool)
    ps = int(rng.randint(0, int(T / 2))) if (random_pos_start and not interpolate_from) else 0
    vo = visual_feature(video, zv, p, E, ps, interpolate_from)
    return torch.einsum("bstc,bkc->bstk", _unit(vo), _unit(lang_raw))


def alignability(p, video, lang_embed, *, D, use_text_pos_enc=False, interpolate_from=None,
                 random_pos_start=False, rng=np.random):
    """get_alignability -- tan_model.py:286-312."""
    vi, ti = _split_interp(interpolate_from)
    B, T, _ = video.shape
    N = lang_embed.shape[1]
    draw = lambda n, itp: int(rng.randint(0, int(n / 2))) if (random_pos_start and not itp) else 0
    lang_t = (textual_feature_with_time(lang_embed, p, draw(N, ti), ti) if use_text_pos_enc
              else textual_feature(lang_embed, p))
    zv = torch.zeros(B, T, dtype=torch.bool)
    zt = torch.zeros(B, N, dtype=torch.bool)
    _, jt = joint_feature(video, zv, lang_t, zt, p, D, draw(T, vi), vi)
    w, b = p["binary_head.weight"], p["binary_head.bias"]
    return {"alignability-dual": textual_feature(lang_embed, p) @ w.t() + b,
            "alignability-joint": jt @ w.t() + b}


def ema_update(p_target, p_online, m):
    """TwinTemporalAligner._momentum_update -- tan_model.py:339-344."""
    for k in p_target:
        p_target[k] = p_target[k] * m + p_online[k] * (1.0 - m)
    return p_target
