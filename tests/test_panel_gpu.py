"""Row-panel fused kernels (csrc/tan_panel.hip) through the C ABI: tan_pack_weights + tan_mlp_fwd against a PyTorch fp32
reference of model/tfm_model.py:23-27,35-37 and against the four launches they replace (LayerNorm, c_fc GEMM, c_proj GEMM,
LayerNorm).  bf16 throughput mode only; tolerances are bf16 rounding (the f32 parity mode never takes this path)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib_ops():
    from temporalalignnet_amd import _lib, ops
    return _lib, ops


def pack(mats):
    _lib, ops = _lib_ops()
    mats = list(mats)
    ents, off, mx = [], 0, 0
    fmts = []
    for i, m in enumerate(mats):
        if isinstance(m, tuple):                       # (matrix, TN, TK): an explicit tile format
            mats[i], tn, tk = m
            fmts.append((tn, tk))
        else:
            fmts.append((512, 16) if m.shape[0] == 512 else (256, 32))
    src = torch.cat([m.reshape(-1) for m in mats])
    dst = torch.empty_like(src)
    for m, (TN, TK) in zip(mats, fmts):
        N, K = m.shape
        ents.append(_lib.PackEntry(off, off, N, K, TN, TK))
        mx = max(mx, (N // TN) * (K // TK))
        off += N * K
    arr = (_lib.PackEntry * len(ents))(*ents)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    _lib.check(_lib.lib().tan_pack_weights(ops._ptr(src), ops._ptr(dst), C.c_void_p(tab.data_ptr()), len(ents), mx, ops._stream()),
               "tan_pack_weights")
    outs, off = [], 0
    for m in mats:
        outs.append(dst[off:off + m.numel()])
        off += m.numel()
    return outs


def test_pack_weights_is_a_permutation_with_the_documented_fragment_layout():
    torch.manual_seed(1)
    for (N, K) in ((2048, 512), (512, 2048), (1536, 512), (512, 512)):
        w = torch.randn(N, K, device="cuda").bfloat16()
        (p,) = pack([w])
        assert torch.equal(torch.sort(p.view(torch.int16).flatten())[0], torch.sort(w.view(torch.int16).flatten())[0])
        TN, TK = (512, 16) if N == 512 else (256, 32)
        W = _lib_ops()[0].lib().tan_panel_waves()
        KS, RB = TK // 16, TN // (32 * W)
        tiles = p.view(N // TN, K // TK, W, RB, KS, 64, 8)          # [n-block][k-block][wave][row block][k step][lane][8]
        nb, kb, wv, rb, ks, lane = 0, (K // TK) - 1, W - 1, RB - 1, KS - 1, 45
        rho = lane & 31                                              # MFMA row -> feature of the 32-row block (see tan_panel.hip)
        row = nb * TN + wv * (TN // W) + rb * 32 + (rho & 3) + 4 * (rho >> 3) + 16 * ((rho >> 2) & 1)
        k = kb * TK + ks * 16 + 8 * (lane >> 5)
        assert torch.equal(tiles[nb, kb, wv, rb, ks, lane], w[row, k:k + 8])


@pytest.mark.parametrize("R,next_ln", [(64, True), (640, True), (1024, False)])
def test_mlp_fwd_matches_reference_and_unfused_path(R, next_ln):
    _lib, ops = _lib_ops()
    torch.manual_seed(R)
    bf = torch.bfloat16
    x_mid = (torch.randn(R, 512, device="cuda") * 1.5).to(bf)
    wfc = (torch.randn(2048, 512, device="cuda") * 1024 ** -0.5).to(bf)
    wpj = (torch.randn(512, 2048, device="cuda") * 0.03).to(bf)
    bfc, bpj = torch.randn(2048, device="cuda") * 0.1, torch.randn(512, device="cuda") * 0.1
    g2, b2 = 1 + 0.1 * torch.randn(512, device="cuda"), 0.1 * torch.randn(512, device="cuda")
    g1, b1 = 1 + 0.1 * torch.randn(512, device="cuda"), 0.1 * torch.randn(512, device="cuda")
    pw_fc, pw_pj = pack([wfc, wpj])
    out = {k: torch.zeros(R, 512, device="cuda", dtype=bf) for k in ("xn2", "xout", "xn1")}
    out |= {k: torch.zeros(R, 2048, device="cuda", dtype=bf) for k in ("hpre", "hact")}
    out |= {k: torch.zeros(R, device="cuda") for k in ("mean2", "rstd2", "mean1", "rstd1")}
    d = _lib.MlpDesc()
    d.rows, d.C, d.FF = R, 512, 2048
    d.x_mid, d.ln_g, d.ln_b = x_mid.data_ptr(), g2.data_ptr(), b2.data_ptr()
    d.pw_fc, d.pw_proj, d.b_fc, d.b_proj = pw_fc.data_ptr(), pw_pj.data_ptr(), bfc.data_ptr(), bpj.data_ptr()
    d.xn2, d.mean2, d.rstd2 = out["xn2"].data_ptr(), out["mean2"].data_ptr(), out["rstd2"].data_ptr()
    d.h_pre, d.h_act, d.x_out = out["hpre"].data_ptr(), out["hact"].data_ptr(), out["xout"].data_ptr()
    if next_ln:
        d.nln_g, d.nln_b, d.xn_next = g1.data_ptr(), b1.data_ptr(), out["xn1"].data_ptr()
        d.nmean, d.nrstd = out["mean1"].data_ptr(), out["rstd1"].data_ptr()
    d.eps, d.variant = 1e-5, 0
    _lib.check(_lib.lib().tan_mlp_fwd(C.byref(d), ops._stream()), "tan_mlp_fwd")
    torch.cuda.synchronize()
    # fp32 reference from the same bf16 inputs, rounded where the kernel rounds
    x = x_mid.float()
    xn2 = torch.nn.functional.layer_norm(x, (512,), g2, b2, 1e-5)
    pre = xn2.to(bf).float() @ wfc.float().T + bfc
    act = pre * torch.sigmoid(1.702 * pre)
    xo = x + act.to(bf).float() @ wpj.float().T + bpj
    xob = xo.to(bf).float()
    ref = {"xn2": xn2, "hpre": pre, "hact": act, "xout": xo, "mean2": x.mean(-1), "rstd2": (x.var(-1, unbiased=False) + 1e-5).rsqrt()}
    if next_ln:
        ref |= {"xn1": torch.nn.functional.layer_norm(xob, (512,), g1, b1, 1e-5), "mean1": xob.mean(-1),
                "rstd1": (xob.var(-1, unbiased=False) + 1e-5).rsqrt()}
    for k, r in ref.items():
        got = out[k].float()
        tol = 2.0 ** -7 * r.abs().max().item() if out[k].dtype == bf else 2e-4      # one bf16 ulp at the top of the range
        assert (got - r).abs().max().item() <= tol, (k, (got - r).abs().max().item(), tol)
    # the unfused launches it replaces: same roundings, so agreement to a bf16 ulp of each tensor
    u_xn2, u_hpre, u_hact, u_xout = (torch.empty_like(out[k]) for k in ("xn2", "hpre", "hact", "xout"))
    ops.layernorm_fwd(x_mid, g2, b2, u_xn2)
    ops.gemm(u_xn2, wfc, u_hact, M=R, N=2048, K=512, bias=bfc, act=ops.ACT_QUICKGELU, aux=u_hpre)
    ops.gemm(u_hact, wpj, u_xout, M=R, N=512, K=2048, bias=bpj, residual=x_mid)
    torch.cuda.synchronize()
    assert torch.equal(u_xn2, out["xn2"])
    for u, k in ((u_hpre, "hpre"), (u_hact, "hact"), (u_xout, "xout")):
        diff = (u.float() - out[k].float()).abs()
        assert diff.max().item() <= 2.0 ** -7 * u.float().abs().max().item(), k
        assert (diff > 0).float().mean().item() < 0.02, k        # different summation order flips a rounding now and then
    # the split-hidden launches for small batches (tan_mlp_fwd_split: eight workgroups per panel, f32 partial sums met by atomics, row
    # epilogue in a second launch): the side outputs of a chunk come from the same instructions -> equal; x_out / xn_next differ by the
    # order of eight f32 additions.  `part` is scratch: whatever is in it (NaNs here) must not matter, two calls in a row.
    assert _lib.lib().tan_mlp_split_chunks() == 8
    part = torch.full((8, R, 512), float("nan"), device="cuda")
    for rep in range(2):
        out2 = {k: torch.full_like(v, float("nan")) for k, v in out.items()}
        d.xn2, d.mean2, d.rstd2 = out2["xn2"].data_ptr(), out2["mean2"].data_ptr(), out2["rstd2"].data_ptr()
        d.h_pre, d.h_act, d.x_out = out2["hpre"].data_ptr(), out2["hact"].data_ptr(), out2["xout"].data_ptr()
        if next_ln:
            d.xn_next, d.nmean, d.nrstd = out2["xn1"].data_ptr(), out2["mean1"].data_ptr(), out2["rstd1"].data_ptr()
        _lib.check(_lib.lib().tan_mlp_fwd_split(C.byref(d), C.c_void_p(part.data_ptr()), ops._stream()), "tan_mlp_fwd_split")
        torch.cuda.synchronize()
        for k in ("xn2", "hpre", "hact", "mean2", "rstd2"):
            assert torch.equal(out2[k], out[k]), (rep, k)
        for k in ("xout",) + (("xn1",) if next_ln else ()):
            diff = (out2[k].float() - out[k].float()).abs()
            assert diff.max().item() <= 2.0 ** -7 * out[k].float().abs().max().item(), (rep, k)
            assert (diff > 0).float().mean().item() < 0.02, (rep, k)
        if next_ln:
            for k in ("mean1", "rstd1"):
                assert (out2[k] - out[k]).abs().max().item() <= 2e-3 * out[k].abs().max().item(), (rep, k)
    # ... and without side outputs (the no-grad forward)
    d.xn2 = d.mean2 = d.rstd2 = d.h_pre = d.h_act = None
    xo3 = torch.full_like(out["xout"], float("nan"))
    d.x_out = xo3.data_ptr()
    _lib.check(_lib.lib().tan_mlp_fwd_split(C.byref(d), C.c_void_p(part.data_ptr()), ops._stream()), "tan_mlp_fwd_split")
    torch.cuda.synchronize()
    diff = (xo3.float() - out["xout"].float()).abs()
    assert diff.max().item() <= 2.0 ** -7 * out["xout"].float().abs().max().item()


@pytest.mark.parametrize("save", [True, False])
def test_mlp_fwd_with_the_out_projection_as_head(save):
    """tan_mlp_fwd with pw_out set: x_mid = x_in + attn_o W_out^T + b_out is computed in the kernel (and written) in front of LN2 --
    against the GEMM launch it replaces followed by the plain kernel; x_mid itself against fp32."""
    _lib, ops = _lib_ops()
    R, bf = 448, torch.bfloat16
    torch.manual_seed(5)
    x_in = (torch.randn(R, 512, device="cuda") * 1.5).to(bf)
    attn_o = torch.randn(R, 512, device="cuda").to(bf)
    w_out = (torch.randn(512, 512, device="cuda") * 512 ** -0.5).to(bf)
    b_out = torch.randn(512, device="cuda") * 0.1
    wfc = (torch.randn(2048, 512, device="cuda") * 1024 ** -0.5).to(bf)
    wpj = (torch.randn(512, 2048, device="cuda") * 0.03).to(bf)
    bfc, bpj = torch.randn(2048, device="cuda") * 0.1, torch.randn(512, device="cuda") * 0.1
    g2, b2 = 1 + 0.1 * torch.randn(512, device="cuda"), 0.1 * torch.randn(512, device="cuda")
    g1, b1 = 1 + 0.1 * torch.randn(512, device="cuda"), 0.1 * torch.randn(512, device="cuda")
    pw_fc, pw_pj, pw_out = pack([wfc, wpj, w_out])

    def run(head):
        out = {k: torch.zeros(R, 512, device="cuda", dtype=bf) for k in ("xn2", "xout", "xn1", "xmid")}
        out |= {k: torch.zeros(R, 2048, device="cuda", dtype=bf) for k in ("hpre", "hact")}
        out |= {k: torch.zeros(R, device="cuda") for k in ("mean2", "rstd2", "mean1", "rstd1")}
        d = _lib.MlpDesc()
        d.rows, d.C, d.FF = R, 512, 2048
        d.x_mid, d.ln_g, d.ln_b = out["xmid"].data_ptr(), g2.data_ptr(), b2.data_ptr()
        d.pw_fc, d.pw_proj, d.b_fc, d.b_proj = pw_fc.data_ptr(), pw_pj.data_ptr(), bfc.data_ptr(), bpj.data_ptr()
        d.x_out = out["xout"].data_ptr()
        if save:
            d.xn2, d.mean2, d.rstd2 = out["xn2"].data_ptr(), out["mean2"].data_ptr(), out["rstd2"].data_ptr()
            d.h_pre, d.h_act = out["hpre"].data_ptr(), out["hact"].data_ptr()
        d.nln_g, d.nln_b, d.xn_next = g1.data_ptr(), b1.data_ptr(), out["xn1"].data_ptr()
        d.nmean, d.nrstd = out["mean1"].data_ptr(), out["rstd1"].data_ptr()
        d.eps, d.variant = 1e-5, 0
        if head:
            d.attn_o, d.pw_out, d.b_out, d.x_in = attn_o.data_ptr(), pw_out.data_ptr(), b_out.data_ptr(), x_in.data_ptr()
        else:
            ops.gemm(attn_o, w_out, out["xmid"], M=R, N=512, K=512, bias=b_out, residual=x_in)
        _lib.check(_lib.lib().tan_mlp_fwd(C.byref(d), ops._stream()), "tan_mlp_fwd")
        torch.cuda.synchronize()
        return out

    o0, o1 = run(False), run(True)
    ref = x_in.float() + attn_o.float() @ w_out.float().T + b_out
    assert (o1["xmid"].float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    for k in o0:
        a, b = o0[k].float(), o1[k].float()
        assert torch.isfinite(b).all(), k
        tol = 2.0 ** -6 * a.abs().max().item() if o0[k].dtype == bf else 2e-3
        assert (a - b).abs().max().item() <= tol + 1e-6, (k, (a - b).abs().max().item())
        if o0[k].dtype == bf:
            assert (a != b).float().mean().item() < 0.10, k


def test_mlp_fwd_rejects_what_it_cannot_do():
    _lib, ops = _lib_ops()
    d = _lib.MlpDesc()
    d.rows, d.C, d.FF = 100, 512, 2048       # not a multiple of 64 rows, and nothing bound
    assert _lib.lib().tan_mlp_fwd(C.byref(d), ops._stream()) == -1


@pytest.mark.parametrize("R", [64, 640, 1024])
def test_mlp_bwd_matches_the_fp32_backward_of_the_branch(R):
    """tan_mlp_bwd = autograd of x_out = x_mid + c_proj(quickgelu(c_fc(LN2(x_mid)))) w.r.t. x_mid, b_fc, ln_2 and (through dx2's
    column sums) the attention out-projection bias; dh is the operand of the c_fc weight gradient.  Reference: fp32 math on the same
    bf16 tensors, rounded where the kernel rounds (dh to bf16 before the second GEMM)."""
    _lib, ops = _lib_ops()
    torch.manual_seed(100 + R)
    bf = torch.bfloat16
    x_mid = (torch.randn(R, 512, device="cuda") * 1.5).to(bf)
    dx = (torch.randn(R, 512, device="cuda") * 0.02).to(bf)
    wfc = (torch.randn(2048, 512, device="cuda") * 1024 ** -0.5).to(bf)
    wpj = (torch.randn(512, 2048, device="cuda") * 0.03).to(bf)
    bfc = torch.randn(2048, device="cuda") * 0.1
    g2, b2 = 1 + 0.1 * torch.randn(512, device="cuda"), 0.1 * torch.randn(512, device="cuda")
    x = x_mid.float()
    mean2, rstd2 = x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    xhat = (x - mean2[:, None]) * rstd2[:, None]
    xn2 = (xhat * g2 + b2).to(bf)
    h_pre = (xn2.float() @ wfc.float().T + bfc).to(bf)
    # packed images of the TRANSPOSED weights: W_proj^T [2048][512] (c_fc-like tiles), W_fc^T [512][2048] (c_proj-like tiles)
    pwt_proj, pwt_fc = pack([wpj.T.contiguous(), wfc.T.contiguous()])
    dh = torch.zeros(R, 2048, device="cuda", dtype=bf)
    dx2 = torch.zeros(R, 512, device="cuda", dtype=bf)
    g_b_fc = torch.full((2048,), 0.5, device="cuda")            # accumulated INTO: start from something
    g_ln_g, g_ln_b, g_b_out = (torch.full((512,), v, device="cuda") for v in (0.25, -0.5, 1.0))
    d = _lib.MlpBwdDesc()
    d.rows, d.C, d.FF = R, 512, 2048
    d.dx, d.h_pre, d.x_mid = dx.data_ptr(), h_pre.data_ptr(), x_mid.data_ptr()
    d.mean2, d.rstd2, d.ln_g = mean2.data_ptr(), rstd2.data_ptr(), g2.data_ptr()
    d.pwt_proj, d.pwt_fc = pwt_proj.data_ptr(), pwt_fc.data_ptr()
    d.dh, d.dx2 = dh.data_ptr(), dx2.data_ptr()
    d.g_b_fc, d.g_ln_g, d.g_ln_b, d.g_b_out = g_b_fc.data_ptr(), g_ln_g.data_ptr(), g_ln_b.data_ptr(), g_b_out.data_ptr()
    _lib.check(_lib.lib().tan_mlp_bwd(C.byref(d), ops._stream()), "tan_mlp_bwd")
    torch.cuda.synchronize()
    hp = h_pre.float()
    sg = torch.sigmoid(1.702 * hp)
    r_dh = (dx.float() @ wpj.float()) * (sg + 1.702 * hp * sg * (1 - sg))
    r_dxn = r_dh.to(bf).float() @ wfc.float()
    g = r_dxn * g2
    r_dx2 = rstd2[:, None] * (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True)) + dx.float()

    def close(got, ref, what, rel):
        err = (got.float() - ref).abs().max().item()
        assert err <= rel * ref.abs().max().item() + 1e-7, (what, err, ref.abs().max().item())
    close(dh, r_dh, "dh", 2.0 ** -7)
    close(dx2, r_dx2, "dx2", 2.0 ** -7)
    close(g_b_fc - 0.5, r_dh.sum(0), "g_b_fc", 2e-3)
    close(g_ln_g - 0.25, (r_dxn * xhat).sum(0), "g_ln_g", 2e-3)
    close(g_ln_b + 0.5, r_dxn.sum(0), "g_ln_b", 2e-3)
    close(g_b_out - 1.0, r_dx2.sum(0), "g_b_out", 2e-3)
    # the split-hidden launches for small batches (tan_mlp_bwd_split): dh comes from the same instructions -> equal; dx2 and the
    # column-sum gradients differ by the order of f32 additions
    part = torch.full((8, R, 512), float("nan"), device="cuda")
    for rep in range(2):
        dh_s, dx2_s = torch.full_like(dh, float("nan")), torch.full_like(dx2, float("nan"))
        gs = {"g_b_fc": torch.full((2048,), 0.5, device="cuda"), "g_ln_g": torch.full((512,), 0.25, device="cuda"),
              "g_ln_b": torch.full((512,), -0.5, device="cuda"), "g_b_out": torch.full((512,), 1.0, device="cuda")}
        d.dh, d.dx2 = dh_s.data_ptr(), dx2_s.data_ptr()
        d.g_b_fc, d.g_ln_g, d.g_ln_b, d.g_b_out = (gs[k].data_ptr() for k in ("g_b_fc", "g_ln_g", "g_ln_b", "g_b_out"))
        _lib.check(_lib.lib().tan_mlp_bwd_split(C.byref(d), C.c_void_p(part.data_ptr()), ops._stream()), "tan_mlp_bwd_split")
        torch.cuda.synchronize()
        assert torch.equal(dh_s, dh), rep
        diff = (dx2_s.float() - dx2.float()).abs()
        assert diff.max().item() <= 2.0 ** -7 * dx2.float().abs().max().item() and (diff > 0).float().mean().item() < 0.02, rep
        close(gs["g_b_fc"] - 0.5, r_dh.sum(0), "g_b_fc split", 2e-3)
        close(gs["g_ln_g"] - 0.25, (r_dxn * xhat).sum(0), "g_ln_g split", 2e-3)
        close(gs["g_ln_b"] + 0.5, r_dxn.sum(0), "g_ln_b split", 2e-3)
        close(gs["g_b_out"] - 1.0, r_dx2.sum(0), "g_b_out split", 2e-3)
    d.g_b_fc, d.g_ln_g, d.g_ln_b, d.g_b_out = g_b_fc.data_ptr(), g_ln_g.data_ptr(), g_ln_b.data_ptr(), g_b_out.data_ptr()
    # optional tail: d_o = dx2 W_out (the attention out-projection's dX GEMM) from the resident dx2 panel; everything else unchanged
    w_out = (torch.randn(512, 512, device="cuda") * 512 ** -0.5).to(bf)
    (pwt_out,) = pack([w_out.T.contiguous()])
    d_o = torch.full((R, 512), float("nan"), device="cuda", dtype=bf)
    dh2, dx22 = torch.zeros_like(dh), torch.zeros_like(dx2)
    d.dh, d.dx2, d.pwt_out, d.d_o = dh2.data_ptr(), dx22.data_ptr(), pwt_out.data_ptr(), d_o.data_ptr()
    _lib.check(_lib.lib().tan_mlp_bwd(C.byref(d), ops._stream()), "tan_mlp_bwd")
    torch.cuda.synchronize()
    assert torch.equal(dx22, dx2) and torch.equal(dh2, dh)
    close(d_o, dx2.float() @ w_out.float(), "d_o", 2.0 ** -7)
    u_do = torch.empty_like(d_o)                                      # the launch it replaces (K-strided W form)
    ops.gemm(dx2, w_out, u_do, M=R, N=512, K=512, a_kc=True, b_kc=False, ldb=512)
    torch.cuda.synchronize()
    diff = (u_do.float() - d_o.float()).abs()
    assert diff.max().item() <= 2.0 ** -7 * u_do.float().abs().max().item() and (diff > 0).float().mean().item() < 0.02


def test_mlp_bwd_with_the_next_blocks_ln1_backward_as_prologue():
    """tan_mlp_bwd with ln1_dxn set: dx = ln1_res + LayerNorm-backward(ln1_dxn; ...) is produced in the kernel (and stored to dx_out
    for the weight-gradient launch) instead of being read -- against tan_layernorm_bwd followed by the plain tan_mlp_bwd.  ln1_res
    aliases dx2, as in tan_encoder_bwd."""
    _lib, ops = _lib_ops()
    R, bf = 256, torch.bfloat16
    torch.manual_seed(7)
    x_mid = (torch.randn(R, 512, device="cuda") * 1.5).to(bf)
    wfc = (torch.randn(2048, 512, device="cuda") * 1024 ** -0.5).to(bf)
    wpj = (torch.randn(512, 2048, device="cuda") * 0.03).to(bf)
    g2 = 1 + 0.1 * torch.randn(512, device="cuda")
    x = x_mid.float()
    mean2, rstd2 = x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    h_pre = (torch.randn(R, 2048, device="cuda")).to(bf)
    pwt_proj, pwt_fc = pack([wpj.T.contiguous(), wfc.T.contiguous()])
    # the next block's ln_1: input x_out (this block's output), upstream gradient dxn, residual-stream gradient res
    x_out = (torch.randn(R, 512, device="cuda") * 2).to(bf)
    dxn = (torch.randn(R, 512, device="cuda") * 0.02).to(bf)
    res = (torch.randn(R, 512, device="cuda") * 0.02).to(bf)
    g1 = 1 + 0.1 * torch.randn(512, device="cuda")
    xo = x_out.float()
    mean1, rstd1 = xo.mean(-1).contiguous(), (xo.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()

    def run(fused):
        dh = torch.zeros(R, 2048, device="cuda", dtype=bf)
        dx2 = res.clone()                                  # the residual gradient arrives in the buffer dx2 leaves in
        dx = torch.full((R, 512), float("nan"), device="cuda", dtype=bf)
        acc = {k: torch.full((n,), 0.5, device="cuda") for k, n in (("g_b_fc", 2048), ("g_ln_g", 512), ("g_ln_b", 512),
                                                                       ("g_b_out", 512), ("g_ln1_g", 512), ("g_ln1_b", 512),
                                                                       ("g_b_proj", 512))}
        d = _lib.MlpBwdDesc()
        d.rows, d.C, d.FF = R, 512, 2048
        d.h_pre, d.x_mid = h_pre.data_ptr(), x_mid.data_ptr()
        d.mean2, d.rstd2, d.ln_g = mean2.data_ptr(), rstd2.data_ptr(), g2.data_ptr()
        d.pwt_proj, d.pwt_fc = pwt_proj.data_ptr(), pwt_fc.data_ptr()
        d.dh, d.dx2 = dh.data_ptr(), dx2.data_ptr()
        d.g_b_fc, d.g_ln_g, d.g_ln_b, d.g_b_out = (acc[k].data_ptr() for k in ("g_b_fc", "g_ln_g", "g_ln_b", "g_b_out"))
        if fused:
            d.ln1_dxn, d.ln1_x, d.ln1_res = dxn.data_ptr(), x_out.data_ptr(), dx2.data_ptr()
            d.ln1_mean, d.ln1_rstd, d.ln1_g = mean1.data_ptr(), rstd1.data_ptr(), g1.data_ptr()
            d.g_ln1_g, d.g_ln1_b, d.g_dx_colsum = acc["g_ln1_g"].data_ptr(), acc["g_ln1_b"].data_ptr(), acc["g_b_proj"].data_ptr()
            d.dx_out = dx.data_ptr()
        else:
            ops.layernorm_bwd(dxn, x_out, g1, mean1, rstd1, dx, acc["g_ln1_g"], acc["g_ln1_b"], dres=dx2, dx_colsum=acc["g_b_proj"])
            d.dx = dx.data_ptr()
        _lib.check(_lib.lib().tan_mlp_bwd(C.byref(d), ops._stream()), "tan_mlp_bwd")
        torch.cuda.synchronize()
        return dx, dh, dx2, acc

    dx0, dh0, dx20, acc0 = run(False)
    dx1, dh1, dx21, acc1 = run(True)
    # the same arithmetic compiled twice (fma contraction may differ by an f32 ulp): equal up to a bf16 rounding step
    for got, ref, what in ((dx1, dx0, "dx"), (dh1, dh0, "dh"), (dx21, dx20, "dx2")):
        assert torch.isfinite(got.float()).all(), what
        err = (got.float() - ref.float()).abs().max().item()
        assert err <= 2.0 ** -7 * ref.float().abs().max().item(), (what, err)
        assert (got != ref).float().mean().item() < 0.02, what
    for k in acc0:          # f32 column sums meet in atomics in a different order
        assert torch.allclose(acc0[k], acc1[k], rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("with_stage", [False, True])
def test_mlp_bwd_with_the_next_blocks_in_proj_dx_gemm_as_head(with_stage):
    """tan_mlp_bwd with pwt_in set: ln1_dxn = dqkv W_in (+ dstage) is computed in the kernel (LDS only) in front of the ln_1 backward
    prologue -- against the GEMM launch it replaces followed by the ln1_dxn form of the same kernel; and the GEMM itself against fp32."""
    _lib, ops = _lib_ops()
    R, bf = 320, torch.bfloat16
    torch.manual_seed(11)
    x_mid = (torch.randn(R, 512, device="cuda") * 1.5).to(bf)
    wfc = (torch.randn(2048, 512, device="cuda") * 1024 ** -0.5).to(bf)
    wpj = (torch.randn(512, 2048, device="cuda") * 0.03).to(bf)
    w_in = (torch.randn(1536, 512, device="cuda") * 512 ** -0.5).to(bf)
    g2 = 1 + 0.1 * torch.randn(512, device="cuda")
    x = x_mid.float()
    mean2, rstd2 = x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    h_pre = (torch.randn(R, 2048, device="cuda")).to(bf)
    pwt_proj, pwt_fc, pwt_in = pack([wpj.T.contiguous(), wfc.T.contiguous(), w_in.T.contiguous()])
    x_out = (torch.randn(R, 512, device="cuda") * 2).to(bf)
    dqkv = (torch.randn(R, 1536, device="cuda") * 0.02).to(bf)
    dstage = (torch.randn(R, 512, device="cuda") * 0.02).to(bf) if with_stage else None
    res = (torch.randn(R, 512, device="cuda") * 0.02).to(bf)
    g1 = 1 + 0.1 * torch.randn(512, device="cuda")
    xo = x_out.float()
    mean1, rstd1 = xo.mean(-1).contiguous(), (xo.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
    r_dxn = dqkv.float() @ w_in.float() + (dstage.float() if with_stage else 0.0)
    dxn = r_dxn.to(bf)                                     # what the stand-alone GEMM stores (one rounding of the f32 sum)

    def run(head):
        dh = torch.zeros(R, 2048, device="cuda", dtype=bf)
        dx2 = res.clone()
        dx = torch.full((R, 512), float("nan"), device="cuda", dtype=bf)
        acc = {k: torch.full((n,), 0.5, device="cuda") for k, n in (("g_b_fc", 2048), ("g_ln_g", 512), ("g_ln_b", 512),
                                                                       ("g_b_out", 512), ("g_ln1_g", 512), ("g_ln1_b", 512),
                                                                       ("g_b_proj", 512))}
        d = _lib.MlpBwdDesc()
        d.rows, d.C, d.FF = R, 512, 2048
        d.h_pre, d.x_mid = h_pre.data_ptr(), x_mid.data_ptr()
        d.mean2, d.rstd2, d.ln_g = mean2.data_ptr(), rstd2.data_ptr(), g2.data_ptr()
        d.pwt_proj, d.pwt_fc = pwt_proj.data_ptr(), pwt_fc.data_ptr()
        d.dh, d.dx2 = dh.data_ptr(), dx2.data_ptr()
        d.g_b_fc, d.g_ln_g, d.g_ln_b, d.g_b_out = (acc[k].data_ptr() for k in ("g_b_fc", "g_ln_g", "g_ln_b", "g_b_out"))
        d.ln1_x, d.ln1_res = x_out.data_ptr(), dx2.data_ptr()
        d.ln1_mean, d.ln1_rstd, d.ln1_g = mean1.data_ptr(), rstd1.data_ptr(), g1.data_ptr()
        d.g_ln1_g, d.g_ln1_b, d.g_dx_colsum = acc["g_ln1_g"].data_ptr(), acc["g_ln1_b"].data_ptr(), acc["g_b_proj"].data_ptr()
        d.dx_out = dx.data_ptr()
        if head:
            d.dqkv, d.pwt_in = dqkv.data_ptr(), pwt_in.data_ptr()
            d.dstage = dstage.data_ptr() if with_stage else None
        else:
            d.ln1_dxn = dxn.data_ptr()
        _lib.check(_lib.lib().tan_mlp_bwd(C.byref(d), ops._stream()), "tan_mlp_bwd")
        torch.cuda.synchronize()
        return dx, dh, dx2, acc

    dx0, dh0, dx20, acc0 = run(False)
    dx1, dh1, dx21, acc1 = run(True)
    # the head accumulates K = 1536 in another order than the reference product: dxn1 differs by a bf16 rounding step on a few
    # elements, and everything downstream by as much
    for got, ref, what in ((dx1, dx0, "dx"), (dh1, dh0, "dh"), (dx21, dx20, "dx2")):
        assert torch.isfinite(got.float()).all(), what
        err = (got.float() - ref.float()).abs().max().item()
        assert err <= 2.0 ** -6 * ref.float().abs().max().item(), (what, err)
        assert (got != ref).float().mean().item() < 0.10, what
    for k in acc0:
        assert torch.allclose(acc0[k], acc1[k], rtol=2e-3, atol=2e-4), k
    # both at once is a contradiction; the head without the ln_1 fields too
    d = _lib.MlpBwdDesc()
    d.rows, d.C, d.FF = R, 512, 2048
    d.pwt_in = pwt_in.data_ptr()
    assert _lib.lib().tan_mlp_bwd(C.byref(d), ops._stream()) == -1


def test_mlp_bwd_rejects_what_it_cannot_do():
    _lib, ops = _lib_ops()
    d = _lib.MlpBwdDesc()
    d.rows, d.C, d.FF = 128, 512, 2048       # nothing bound
    assert _lib.lib().tan_mlp_bwd(C.byref(d), ops._stream()) == -1
    d.rows = 100
    assert _lib.lib().tan_mlp_bwd(C.byref(d), ops._stream()) == -1

