"""Fused input embeddings (csrc/tan_embed.hip) through the C ABI: tan_embed_fwd against a PyTorch fp32 restatement of
model/tan_model.py:155-167,187-203,231-234 + the first block's ln_1 (model/tfm_model.py:35), rounded where the kernel rounds (bf16
operands, bf16 proj / out / xn1), and the model's fused front-end against the launches it replaces (TAN_EMBED_FUSED=0)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from test_panel_gpu import pack

pytestmark = pytest.mark.gpu


def _ln(x, g, b, eps=1e-5):
    m = x.mean(-1, keepdim=True)
    v = ((x - m) ** 2).mean(-1, keepdim=True)
    r = torch.rsqrt(v + eps)
    return (x - m) * r * g + b, m.squeeze(-1), r.squeeze(-1)


def _bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize("B,T,N,Dv,Dt,a_dtype,pads", [(3, 16, 5, 1024, 512, torch.float32, True), (2, 64, 16, 1024, 512, torch.float32, False),
                                                      (4, 20, 7, 256, 128, torch.bfloat16, True), (128, 64, 13, 1024, 512, torch.float32, True)])
def test_embed_fwd_matches_torch(B, T, N, Dv, Dt, a_dtype, pads):
    from temporalalignnet_amd import _lib, ops
    torch.manual_seed(B * 1000 + T)
    dev, Cw, L = "cuda", 512, T + N
    R, Mp = B * T, B * N
    video = (torch.randn(B, T, Dv, device=dev).abs() * 0.3).to(a_dtype)
    lang = torch.randn(B, N, Dt, device=dev).to(a_dtype)
    Wv, Wt = (torch.randn(Cw, Dv, device=dev) * Dv ** -0.5).bfloat16(), (torch.randn(Cw, Dt, device=dev) * Dt ** -0.5).bfloat16()
    pwv, pwt = pack([Wv, Wt])
    par = {k: (1 + 0.2 * torch.randn(Cw, device=dev)) if k.endswith("g") else 0.2 * torch.randn(Cw, device=dev)
           for k in ("v_g", "v_b", "t_g", "t_b", "l1v_g", "l1v_b", "l1j_g", "l1j_b")}
    pos_a, pos_b, pos_t = (torch.randn(n, Cw, device=dev) for n in (T, T, N))
    vpad = (torch.rand(B, T, device=dev) < 0.2).to(torch.uint8) if pads else None
    tpad = (torch.rand(B, N, device=dev) < 0.3).to(torch.uint8) if pads else None
    bf = dict(dtype=torch.bfloat16, device=dev)
    out = {"video_c": torch.zeros(R, Dv, **bf), "lang_c": torch.zeros(Mp, Dt, **bf), "proj_v": torch.zeros(R, Cw, **bf),
           "proj_t": torch.zeros(Mp, Cw, **bf), "x0": torch.zeros(R, Cw, **bf), "xj": torch.zeros(B * L, Cw, **bf),
           "lang_raw": torch.zeros(Mp, Cw, **bf), "xn1_v": torch.zeros(R, Cw, **bf), "xn1_j": torch.zeros(B * L, Cw, **bf)}
    st = {k: torch.zeros(n, device=dev) for k, n in (("mean_v", R), ("rstd_v", R), ("mean_t", Mp), ("rstd_t", Mp), ("m1v", R), ("r1v", R),
                                                     ("m1j", B * L), ("r1j", B * L))}
    keypad = torch.full((B, L), 7, dtype=torch.uint8, device=dev)
    D = (_lib.EmbedDesc * 2)()
    p = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    dt = _lib.TAN_F32 if a_dtype == torch.float32 else _lib.TAN_BF16
    d = D[0]
    d.a, d.a_dtype, d.rows, d.K, d.T, d.C, d.pw, d.ln_g, d.ln_b = p(video), dt, R, Dv, T, Cw, p(pwv), p(par["v_g"]), p(par["v_b"])
    d.a_bf16, d.proj, d.mean, d.rstd = (p(out["video_c"]) if a_dtype == torch.float32 else None), p(out["proj_v"]), p(st["mean_v"]), p(st["rstd_v"])
    d.out[0], d.out_grp_rows[0], d.out_off[0], d.pos[0] = p(out["x0"]), T, 0, p(pos_a)
    d.out[1], d.out_grp_rows[1], d.out_off[1], d.pos[1] = p(out["xj"]), L, 0, p(pos_b)
    d.ln1_g[0], d.ln1_b[0], d.xn1[0], d.mean1[0], d.rstd1[0] = p(par["l1v_g"]), p(par["l1v_b"]), p(out["xn1_v"]), p(st["m1v"]), p(st["r1v"])
    d.ln1_g[1], d.ln1_b[1], d.xn1[1], d.mean1[1], d.rstd1[1] = p(par["l1j_g"]), p(par["l1j_b"]), p(out["xn1_j"]), p(st["m1j"]), p(st["r1j"])
    d.pad_src, d.pad_dst, d.pad_grp_rows, d.pad_off = p(vpad), p(keypad), L, 0
    d = D[1]
    d.a, d.a_dtype, d.rows, d.K, d.T, d.C, d.pw, d.ln_g, d.ln_b = p(lang), dt, Mp, Dt, N, Cw, p(pwt), p(par["t_g"]), p(par["t_b"])
    d.a_bf16, d.proj, d.mean, d.rstd = (p(out["lang_c"]) if a_dtype == torch.float32 else None), p(out["proj_t"]), p(st["mean_t"]), p(st["rstd_t"])
    d.out[0], d.out_grp_rows[0], d.out_off[0] = p(out["lang_raw"]), N, 0
    d.out[1], d.out_grp_rows[1], d.out_off[1], d.pos[1] = p(out["xj"]), L, T, p(pos_t)
    d.ln1_g[1], d.ln1_b[1], d.xn1[1], d.mean1[1], d.rstd1[1] = p(par["l1j_g"]), p(par["l1j_b"]), p(out["xn1_j"]), p(st["m1j"]), p(st["r1j"])
    d.pad_src, d.pad_dst, d.pad_grp_rows, d.pad_off = p(tpad), p(keypad), L, T
    _lib.check(_lib.lib().tan_embed_fwd(D, 2, ops._stream()), "tan_embed_fwd")
    torch.cuda.synchronize()

    # ---- reference
    def modality(a, W, g, b):
        a16 = _bf(a.float()).double()
        proj = _bf((a16 @ W.double().t()).float())
        y, m, r = _ln(proj, g, b)
        return a16.float(), proj, y, m, r
    v16, pv, yv, mv, rv = modality(video.view(R, Dv), Wv, par["v_g"], par["v_b"])
    t16, ptx, yt, mt, rt = modality(lang.view(Mp, Dt), Wt, par["t_g"], par["t_b"])
    x0 = _bf(yv.view(B, T, Cw) + pos_a)
    xj = torch.cat([_bf(yv.view(B, T, Cw) + pos_b), _bf(yt.view(B, N, Cw) + pos_t)], 1)         # tan_model.py:201
    xn1_v, m1v, r1v = _ln(x0.view(R, Cw), par["l1v_g"], par["l1v_b"])
    xn1_j, m1j, r1j = _ln(xj.view(B * L, Cw), par["l1j_g"], par["l1j_b"])

    def close(got, want, tol, what):
        err = (got.float() - want.float()).abs().max().item()
        assert err <= tol, (what, err)
    if a_dtype == torch.float32:
        assert torch.equal(out["video_c"].float(), v16) and torch.equal(out["lang_c"].float(), t16)
    close(out["proj_v"], pv, 0.04, "proj_v")          # one bf16 ulp at |x| <= 4 (f32 accumulation order differs from the f64 reference)
    close(out["proj_t"], ptx, 0.04, "proj_t")
    # the LayerNorms are checked on the kernel's own bf16 proj (a 1-ulp flip of proj is amplified by rstd)
    yv2, mv2, rv2 = _ln(out["proj_v"].float(), par["v_g"], par["v_b"])
    yt2, mt2, rt2 = _ln(out["proj_t"].float(), par["t_g"], par["t_b"])
    close(st["mean_v"], mv2, 1e-5, "mean_v"); close(st["rstd_v"] / rv2, torch.ones_like(rv2), 1e-5, "rstd_v")
    close(st["mean_t"], mt2, 1e-5, "mean_t"); close(st["rstd_t"] / rt2, torch.ones_like(rt2), 1e-5, "rstd_t")
    x0b = _bf(yv2.view(B, T, Cw) + pos_a)
    xjb = torch.cat([_bf(yv2.view(B, T, Cw) + pos_b), _bf(yt2.view(B, N, Cw) + pos_t)], 1)
    close(out["x0"].view(B, T, Cw), x0b, 0.04, "x0")
    close(out["xj"].view(B, L, Cw), xjb, 0.04, "xj")
    close(out["lang_raw"].view(B, N, Cw), _bf(yt2.view(B, N, Cw)), 0.04, "lang_raw")
    a, _, _ = _ln(out["x0"].float(), par["l1v_g"], par["l1v_b"])
    close(out["xn1_v"], a, 0.04, "xn1_v")
    a, m1, r1 = _ln(out["xj"].float(), par["l1j_g"], par["l1j_b"])
    close(out["xn1_j"], a, 0.04, "xn1_j")
    close(st["m1j"], m1, 1e-5, "mean1_j"); close(st["r1j"] / r1, torch.ones_like(r1), 1e-5, "rstd1_j")
    # and against the f64-projection reference end to end, loosely (catches a wrong operand / position row / destination)
    close(out["x0"].view(B, T, Cw), x0, 0.15, "x0 (end to end)")
    close(out["xj"].view(B, L, Cw), xj, 0.15, "xj (end to end)")
    close(out["xn1_v"], xn1_v, 0.15, "xn1_v (end to end)")
    close(out["xn1_j"], xn1_j, 0.15, "xn1_j (end to end)")
    want_pad = torch.cat([vpad if vpad is not None else torch.zeros(B, T, dtype=torch.uint8, device=dev),
                          tpad if tpad is not None else torch.zeros(B, N, dtype=torch.uint8, device=dev)], 1)
    assert torch.equal(keypad, want_pad)


def _step_outputs(fused, seed=3, text_pos=0):
    """one bf16 forward + backward of the whole model with the fused front-end on / off: loss inputs and every parameter gradient"""
    os.environ["TAN_EMBED_FUSED"] = "1" if fused else "0"
    try:
        from temporalalignnet_amd import synth
        from temporalalignnet_amd.tan_model import TemporalAligner
        torch.manual_seed(0)
        np.random.seed(5)
        m = TemporalAligner(num_encoder_layers=2, num_decoder_layers=2, compute_dtype="bf16", random_pos_start=1, language_model=None,
                            use_text_pos_enc=text_pos).cuda()
        sd = m.state_dict()
        for k, v in synth.make_params(seed, 2, 2, False).items():
            if k in sd and sd[k].shape == torch.Size(v.shape):
                sd[k].copy_(torch.from_numpy(v))
        m.invalidate_shadow()
        b = synth.make_batch(11, B=6, T=32, n_min=3, n_max=9)
        video, lang = torch.from_numpy(b["video"]).cuda(), torch.from_numpy(b["text_embed"]).cuda().requires_grad_(True)
        vp = torch.from_numpy(b["padding_mask"]).bool().cuda()
        vp[1, -5:] = True
        tp = torch.from_numpy(b["text_padding_mask"]).bool().cuda()
        out = m(video, lang, video_padding_mask=vp, lang_padding_mask=tp, text_timestamp=None)
        loss = (out["logits_dual"].float() ** 2).mean() + (out["logits_joint"].float() * 0.5).sin().mean()
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
        return out["logits_dual"].detach().float().clone(), out["logits_joint"].detach().float().clone(), lang.grad.clone(), grads
    finally:
        os.environ.pop("TAN_EMBED_FUSED", None)


@pytest.mark.parametrize("text_pos", [0, 1])
def test_fused_front_end_matches_the_launches_it_replaces(text_pos):
    ld1, lj1, gl1, g1 = _step_outputs(True, text_pos=text_pos)
    ld0, lj0, gl0, g0 = _step_outputs(False, text_pos=text_pos)
    assert (ld1 - ld0).abs().max().item() < 0.02 and (lj1 - lj0).abs().max().item() < 0.02      # cosines, bf16 features
    assert set(g1) == set(g0)

    def cos(a, b):
        return torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    assert cos(gl1, gl0) > 0.995
    for n in g1:
        if g0[n].abs().max().item() == 0 and g1[n].abs().max().item() == 0:
            continue
        assert cos(g1[n], g0[n]) > 0.99, (n, cos(g1[n], g0[n]))
