"""The attention branch of a block in one launch (csrc/tan_attnblk.hip) through the C ABI: tan_pack_weights (qkv16 format) +
tan_attnblk_fwd against a PyTorch fp32 reference of model/tfm_model.py:30-36 (nn.MultiheadAttention(512, 8) with key_padding_mask,
out_proj, residual) and against the three launches it replaces (in_proj GEMM, tan_attn_fwd, out_proj GEMM).  bf16 throughput mode
only; tolerances are bf16 rounding (the f32 parity mode never takes this path)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _lib_ops():
    from temporalalignnet_amd import _lib, ops
    return _lib, ops


def pack(mats_fmt):
    """[(matrix [N, K] bf16, TN, TK)] -> packed images (tan_pack_weights)"""
    _lib, ops = _lib_ops()
    src = torch.cat([m.reshape(-1) for m, _, _ in mats_fmt])
    dst = torch.empty_like(src)
    ents, off, mx = [], 0, 0
    for m, TN, TK in mats_fmt:
        N, K = m.shape
        ents.append(_lib.PackEntry(off, off, N, K, TN, TK))
        mx = max(mx, (N // TN) * (K // TK))
        off += N * K
    arr = (_lib.PackEntry * len(ents))(*ents)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    _lib.check(_lib.lib().tan_pack_weights(ops._ptr(src), ops._ptr(dst), C.c_void_p(tab.data_ptr()), len(ents), mx, ops._stream()),
               "tan_pack_weights")
    outs, off = [], 0
    for m, _, _ in mats_fmt:
        outs.append(dst[off:off + m.numel()])
        off += m.numel()
    return outs


def test_qkv16_pack_format_is_the_documented_permutation():
    torch.manual_seed(2)
    w = torch.randn(1536, 512, device="cuda").to(bf)
    (p,) = pack([(w, 384, 32)])
    assert torch.equal(torch.sort(p.view(torch.int16).flatten())[0], torch.sort(w.view(torch.int16).flatten())[0])
    t = p.view(4, 16, 8, 3, 64, 8)            # [head pair][k step of 32][wave][feature block][lane][8]
    for hp, ks, wv, fb, lane in ((0, 0, 0, 0, 0), (3, 15, 7, 2, 63), (1, 7, 4, 1, 37), (2, 3, 3, 0, 18)):
        q = 3 * wv + fb
        which, j, fblk = (q // 4) % 3, q // 12, q % 4
        row = which * 512 + (2 * hp + j) * 64 + fblk * 16 + (lane & 15)
        k = ks * 32 + 8 * (lane >> 4)
        assert torch.equal(t[hp, ks, wv, fb, lane], w[row, k:k + 8]), (hp, ks, wv, fb, lane)


def reference(xn1, x_in, w_in, b_in, w_out, b_out, keypad, B, L):
    """fp32 math on the bf16 inputs, rounded where the kernels round (qkv and attention output to bf16)."""
    qkv = (xn1.float() @ w_in.float().T + b_in).to(bf).float().view(B, L, 3, 8, 64)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))          # [B, H, L, 64]
    s = (q @ k.transpose(-1, -2)) * 0.125
    if keypad is not None:
        s = s.masked_fill(keypad.bool()[:, None, None, :], float("-inf"))
    lse = torch.logsumexp(s, -1)
    o = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, 512)
    x_mid = x_in.float() + o.to(bf).float() @ w_out.float().T + b_out
    return qkv.view(B * L, 1536), o, lse, x_mid


@pytest.mark.parametrize("B,L,pad,save", [(3, 64, False, True), (5, 80, True, True), (2, 70, True, True), (4, 56, True, False),
                                          (128, 64, False, True)])
def test_attnblk_fwd_matches_reference_and_unfused_path(B, L, pad, save):
    _lib, ops = _lib_ops()
    torch.manual_seed(B * 100 + L)
    R = B * L
    xn1 = torch.randn(R, 512, device="cuda").to(bf)
    x_in = (torch.randn(R, 512, device="cuda") * 1.5).to(bf)
    w_in = (torch.randn(1536, 512, device="cuda") * 512 ** -0.5).to(bf)
    w_out = (torch.randn(512, 512, device="cuda") * 512 ** -0.5).to(bf)
    b_in, b_out = torch.randn(1536, device="cuda") * 0.1, torch.randn(512, device="cuda") * 0.1
    keypad = None
    if pad:
        keypad = torch.zeros(B, L, dtype=torch.uint8, device="cuda")
        for b in range(B):
            keypad[b, L - 1 - (b * 5) % 17:] = 1          # a padded tail of 1..17 keys
        keypad[0, 3] = 1                                   # and one hole
    pw_qkv, pw_out = pack([(w_in, 384, 32), (w_out, 512, 16)])
    out = {"qkv": torch.zeros(R, 1536, device="cuda", dtype=bf), "attn_o": torch.zeros(R, 512, device="cuda", dtype=bf),
           "lse": torch.zeros(B, 8, L, device="cuda"), "x_mid": torch.zeros(R, 512, device="cuda", dtype=bf)}
    d = _lib.AttnBlkDesc()
    d.B, d.L, d.C, d.H = B, L, 512, 8
    d.xn1, d.x_in = xn1.data_ptr(), x_in.data_ptr()
    d.key_padding_mask = keypad.data_ptr() if keypad is not None else None
    d.pw_qkv, d.pw_out, d.b_qkv, d.b_out = pw_qkv.data_ptr(), pw_out.data_ptr(), b_in.data_ptr(), b_out.data_ptr()
    if save:
        d.qkv, d.attn_o, d.lse = out["qkv"].data_ptr(), out["attn_o"].data_ptr(), out["lse"].data_ptr()
    d.x_mid = out["x_mid"].data_ptr()
    assert _lib.lib().tan_attnblk_supported(L, 512, 8, _lib.TAN_BF16) == 1
    _lib.check(_lib.lib().tan_attnblk_fwd(C.byref(d), ops._stream()), "tan_attnblk_fwd")
    torch.cuda.synchronize()
    r_qkv, r_o, r_lse, r_xmid = reference(xn1, x_in, w_in, b_in, w_out, b_out, keypad, B, L)
    checks = [("x_mid", r_xmid)] + ([("qkv", r_qkv), ("attn_o", r_o), ("lse", r_lse)] if save else [])
    for k, r in checks:
        got = out[k].float()
        tol = 2.0 ** -7 * r.abs().max().item() if out[k].dtype == bf else 5e-3      # lse: q, k are bf16-rounded
        err = (got - r).abs().max().item()
        assert err <= tol, (k, err, tol)
    if not save:
        assert not out["qkv"].any() and not out["attn_o"].any() and not out["lse"].any()
    # the three launches it replaces: same rounding points
    u_qkv, u_o, u_x = torch.empty_like(out["qkv"]), torch.empty_like(out["attn_o"]), torch.empty_like(out["x_mid"])
    u_lse = torch.empty(B, 8, L, device="cuda")
    ops.gemm(xn1, w_in, u_qkv, M=R, N=1536, K=512, bias=b_in)
    ops.attn_fwd(u_qkv, keypad, u_o, u_lse, B, L, 8)
    ops.gemm(u_o, w_out, u_x, M=R, N=512, K=512, bias=b_out, residual=x_in)
    torch.cuda.synchronize()
    pairs = [(u_x, "x_mid")] + ([(u_qkv, "qkv"), (u_o, "attn_o")] if save else [])
    for u, k in pairs:
        diff = (u.float() - out[k].float()).abs()
        assert diff.max().item() <= 2.0 ** -6 * u.float().abs().max().item(), (k, diff.max().item())
        assert (diff > 0).float().mean().item() < 0.05, (k, (diff > 0).float().mean().item())
    if save:
        assert (u_lse - out["lse"]).abs().max().item() < 2e-2
    # the small-batch form (tan_attnblk_fwd_split: one workgroup per (video, head pair), f32 planes added by a second launch): the side
    # outputs come from the same instructions -> equal; x_mid differs by the order of four f32 additions.  `part` is scratch (NaNs here).
    part = torch.full((4, R, 512), float("nan"), device="cuda")
    out2 = {k: torch.zeros_like(v) for k, v in out.items()}
    if save:
        d.qkv, d.attn_o, d.lse = out2["qkv"].data_ptr(), out2["attn_o"].data_ptr(), out2["lse"].data_ptr()
    d.x_mid = out2["x_mid"].data_ptr()
    _lib.check(_lib.lib().tan_attnblk_fwd_split(C.byref(d), C.c_void_p(part.data_ptr()), ops._stream()), "tan_attnblk_fwd_split")
    torch.cuda.synchronize()
    for k in ("qkv", "attn_o", "lse"):
        assert torch.equal(out2[k], out[k]), k
    diff = (out2["x_mid"].float() - out["x_mid"].float()).abs()
    assert diff.max().item() <= 2.0 ** -7 * out["x_mid"].float().abs().max().item() and (diff > 0).float().mean().item() < 0.02


def test_attnblk_with_every_key_of_a_video_padded_gives_the_residual_plus_bias():
    """all keys padded -> attention output 0 (not the reference's NaN; tests/test_kernels_gpu.py pins the stand-alone kernels the
    same way): x_mid = x_in + b_out for that video, lse = -inf, the other videos unaffected."""
    _lib, ops = _lib_ops()
    torch.manual_seed(5)
    B, L = 3, 64
    R = B * L
    xn1 = torch.randn(R, 512, device="cuda").to(bf)
    x_in = torch.randn(R, 512, device="cuda").to(bf)
    w_in = (torch.randn(1536, 512, device="cuda") * 512 ** -0.5).to(bf)
    w_out = (torch.randn(512, 512, device="cuda") * 512 ** -0.5).to(bf)
    b_in, b_out = torch.randn(1536, device="cuda") * 0.1, torch.randn(512, device="cuda") * 0.1
    keypad = torch.zeros(B, L, dtype=torch.uint8, device="cuda")
    keypad[1] = 1
    pw_qkv, pw_out = pack([(w_in, 384, 32), (w_out, 512, 16)])
    qkv, o = torch.zeros(R, 1536, device="cuda", dtype=bf), torch.full((R, 512), float("nan"), device="cuda", dtype=bf)
    lse, x_mid = torch.zeros(B, 8, L, device="cuda"), torch.zeros(R, 512, device="cuda", dtype=bf)
    d = _lib.AttnBlkDesc()
    d.B, d.L, d.C, d.H = B, L, 512, 8
    d.xn1, d.x_in, d.key_padding_mask = xn1.data_ptr(), x_in.data_ptr(), keypad.data_ptr()
    d.pw_qkv, d.pw_out, d.b_qkv, d.b_out = pw_qkv.data_ptr(), pw_out.data_ptr(), b_in.data_ptr(), b_out.data_ptr()
    d.qkv, d.attn_o, d.lse, d.x_mid = qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), x_mid.data_ptr()
    _lib.check(_lib.lib().tan_attnblk_fwd(C.byref(d), ops._stream()), "tan_attnblk_fwd")
    torch.cuda.synchronize()
    assert (o.view(B, L, 512)[1] == 0).all() and (lse[1] == float("-inf")).all()
    want = (x_in.float() + b_out).to(bf).view(B, L, 512)[1]
    assert torch.equal(x_mid.view(B, L, 512)[1], want)
    _, _, _, r_xmid = reference(xn1, x_in, w_in, b_in, w_out, b_out, keypad, B, L)
    keep = torch.tensor([0, 2], device="cuda")
    rk = r_xmid.view(B, L, 512)[keep]                                    # (the torch reference is NaN for video 1)
    assert (x_mid.float().view(B, L, 512)[keep] - rk).abs().max().item() <= 2.0 ** -7 * rk.abs().max().item()


def test_attnblk_rejects_what_it_cannot_do():
    _lib, ops = _lib_ops()
    d = _lib.AttnBlkDesc()
    d.B, d.L, d.C, d.H = 2, 256, 512, 8
    assert _lib.lib().tan_attnblk_fwd(C.byref(d), ops._stream()) == -1
    assert _lib.lib().tan_attnblk_supported(16, 512, 8, _lib.TAN_BF16) == 0
    assert _lib.lib().tan_attnblk_supported(64, 512, 8, _lib.TAN_F32) == 0
