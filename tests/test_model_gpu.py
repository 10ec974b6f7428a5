"""TemporalAligner (HIP) vs the CPU oracle and the reference-generated goldens: forward outputs, gradients."""
import numpy as np
import pytest
import torch

from oracle import tan_ref, train_ref
from temporalalignnet_amd import synth

gpu = pytest.mark.gpu


def make_model(seed, E, D, head, dtype="fp32", **kw):
    from temporalalignnet_amd.tan_model import TemporalAligner
    m = TemporalAligner(num_encoder_layers=E, num_decoder_layers=D, use_alignability_head=int(head), language_model=None,
                        compute_dtype=dtype, **kw)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(seed, E, D, head).items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m.cuda()


def dev_batch(b):
    t = train_ref.to_torch_batch(b)
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in t.items()}


def hip_forward(m, b):
    d = dev_batch(b)
    return m(d["video"], d["text_embed"], d["padding_mask"], d["text_padding_mask"].bool(), None)


@gpu
def test_g1_forward_matches_reference_golden(golden):
    g = golden("g1_forward_e1d1")
    b = synth.make_batch(11, B=4, T=16, n_min=2, n_max=5, video_pad_tail=3)
    m = make_model(101, 1, 1, True)
    np.random.seed(123)
    with torch.no_grad():
        out = hip_forward(m, b)
    assert set(out) == set(g.files)
    for k in g.files:
        assert tuple(out[k].shape) == g[k].shape, k
        np.testing.assert_allclose(out[k].cpu().numpy(), g[k], rtol=1e-4, atol=1e-5, err_msg=k)


@gpu
def test_g13_bert_width_matches_reference_golden(golden):
    """language_model='bert' (tan_model.py:37-41,49): the aligner takes 768-d sentence embeddings (fp32 forward against the reference's
    outputs; a bf16 train step through the fused input embeddings and their backward runs and gives finite gradients)."""
    from temporalalignnet_amd.tan_model import TemporalAligner
    g = golden("g13_bert_width")
    b = synth.make_batch(23, B=3, T=16, n_min=2, n_max=6, d_text=768, video_pad_tail=2)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(113, 1, 2, True, d_text=768).items()}
    outs = {}
    for dtype in ("fp32", "bf16"):
        m = TemporalAligner(num_encoder_layers=1, num_decoder_layers=2, use_alignability_head=1, language_model="bert", compute_dtype=dtype)
        assert m.text_pre_proj.weight.shape == (512, 768)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith("bert.") for k in missing), (missing, unexpected)
        m.cuda()
        np.random.seed(77)
        if dtype == "fp32":
            with torch.no_grad():
                out = hip_forward(m, b)
            assert set(out) == set(g.files)
            for k in g.files:
                np.testing.assert_allclose(out[k].cpu().numpy(), g[k], rtol=1e-4, atol=1e-5, err_msg=k)
        else:
            out = hip_forward(m, b)
            (out["logits_dual"].float().square().mean() + out["logits_joint"].float().square().mean()).backward()
            gw = m.text_pre_proj.weight.grad
            assert gw is not None and gw.shape == (512, 768) and torch.isfinite(gw).all() and gw.abs().sum() > 0
            for k in ("logits_dual", "logits_joint"):
                assert np.abs(out[k].detach().float().cpu().numpy() - g[k]).max() < 3e-2, k


@gpu
def test_g2_forward_e6d6_matches_reference_golden(golden):
    g = golden("g2_forward_e6d6")
    b = synth.make_batch(12, B=2, T=64, n_min=8, n_max=12)
    m = make_model(102, 6, 6, True, random_pos_start=0)
    with torch.no_grad():
        out = hip_forward(m, b)
    for k in g.files:
        err = np.abs(out[k].cpu().numpy() - g[k]).max()
        assert err < 1e-3, (k, err)          # north-star tolerance: 1e-3 in fp32 mode
        assert err < 5e-5, (k, err)          # what the exact-f32 MFMA path actually delivers


@gpu
def test_g7_long_sequence_and_eval_entry_points(golden):
    g = golden("g7_long_interp")
    b = synth.make_batch(17, B=1, T=256, n_min=8, n_max=8)
    m = make_model(107, 2, 3, True, random_pos_start=0)
    d = dev_batch(b)
    with torch.no_grad():
        out = hip_forward(m, b)
        np.testing.assert_allclose(out["logits_dual"].cpu().numpy(), g["logits_dual"], atol=5e-5)
        np.testing.assert_allclose(out["logits_joint"].cpu().numpy(), g["logits_joint"], atol=5e-5)
        v100 = d["video"][:, :100]
        np.testing.assert_allclose(m.get_text_visual_sim_joint(v100, d["text_embed"], interpolate_from=64).cpu().numpy(),
                                   g["sim_joint_interp"], atol=5e-5)
        np.testing.assert_allclose(m.get_text_visual_sim_dual(v100, d["text_embed"], interpolate_from=64).cpu().numpy(),
                                   g["sim_dual_interp"], atol=5e-5)
        al = m.get_alignability(v100, d["text_embed"], interpolate_from=(64, 16))
        np.testing.assert_allclose(al["alignability-dual"].cpu().numpy(), g["align_dual_interp"], atol=5e-5)
        np.testing.assert_allclose(al["alignability-joint"].cpu().numpy(), g["align_joint_interp"], atol=5e-5)
        vf = m.get_visual_feature(d["video"][:, :40], torch.zeros(1, 40, dtype=torch.bool, device="cuda"))
        np.testing.assert_allclose(vf.cpu().numpy(), g["visual_feature_T40"], atol=5e-5)


@gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-2)])
def test_tfm_model_classes_run_standalone(dtype, tol):
    """QuickGELU / ResidualAttentionBlock_Step / TemporalEncoder called on their own keep the reference's signatures and
    layouts (model/tfm_model.py:11-13,34-38,48-55: [L, B, C] in, (x, ln_1(x)) / list of S tensors out) -- vs the CPU oracle."""
    from temporalalignnet_amd.tfm_model import QuickGELU, TemporalEncoder
    torch.manual_seed(3)
    L, B, C, S = 24, 3, 512, 3
    enc = TemporalEncoder(C, S, 8)
    for p_ in enc.parameters():
        if p_.dim() > 1:
            torch.nn.init.normal_(p_, std=C ** -0.5)
        else:
            torch.nn.init.normal_(p_, mean=1.0 if p_.shape[0] == C and p_.mean().item() > 0.5 else 0.0, std=0.1)
    p = {f"enc.{k}": v.detach().clone() for k, v in enc.state_dict().items()}
    x = torch.randn(L, B, C)
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[0, -5:] = True
    mask[2, -1:] = True
    enc.cuda()
    xd = x.cuda().to(dtype)
    feats = enc(xd, mask.cuda())
    want = tan_ref.encoder(x.permute(1, 0, 2), mask, p, "enc", S)
    assert isinstance(feats, list) and len(feats) == S and all(f.shape == (L, B, C) and f.dtype == dtype for f in feats)
    for got, w in zip(feats, want):
        assert (got.float().cpu().permute(1, 0, 2) - w).abs().max().item() <= tol * max(1.0, w.abs().max().item())
    x1, xn = enc.resblocks[0](xd, mask.cuda())
    w1, wn = tan_ref.block(x.permute(1, 0, 2), mask, p, "enc.resblocks.0")
    assert (x1.float().cpu().permute(1, 0, 2) - w1).abs().max().item() <= tol * max(1.0, w1.abs().max().item())
    assert (xn.float().cpu().permute(1, 0, 2) - wn).abs().max().item() <= tol * max(1.0, wn.abs().max().item())
    g = QuickGELU()(xd)
    assert (g.float().cpu() - tan_ref.quick_gelu(xd.float().cpu())).abs().max().item() <= (1e-6 if dtype == torch.float32 else 3e-2)


@gpu
def test_g11_text_pos_enc_and_sine_goldens(golden):
    """use_text_pos_enc=1 + random_pos_start=1 (three np.random draws, tan_model.py:163,224,195) and pos_enc='sine'
    (tan_model.py:60-62) against outputs of the reference itself."""
    from temporalalignnet_amd.tan_model import TemporalAligner
    b = synth.make_batch(21, B=3, T=16, n_min=2, n_max=6, video_pad_tail=2)
    g = golden("g11_text_pos_enc")
    m = make_model(111, 1, 3, True, use_text_pos_enc=1, random_pos_start=1)
    np.random.seed(321)
    with torch.no_grad():
        out = hip_forward(m, b)
    assert set(out) == set(g.files)
    for k in g.files:
        np.testing.assert_allclose(out[k].cpu().numpy(), g[k], atol=1e-4 if "alignability" in k else 5e-5, err_msg=k)
    g2 = golden("g11_sine_forward")
    m2 = TemporalAligner(num_encoder_layers=2, num_decoder_layers=1, use_alignability_head=0, language_model=None, pos_enc="sine",
                         random_pos_start=0)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(112, 2, 1, False).items() if k != "temporal_pos_embed"}
    missing, unexpected = m2.load_state_dict(sd, strict=False)
    assert missing == ["temporal_pos_embed"] and not unexpected           # the table is the model's own buffer, as in the reference
    m2.cuda()
    with torch.no_grad():
        out2 = hip_forward(m2, b)
    for k in g2.files:
        np.testing.assert_allclose(out2[k].cpu().numpy(), g2[k], atol=5e-5, err_msg=k)


@gpu
@pytest.mark.parametrize("E,D,B,T,vpad", [(1, 1, 4, 16, 3), (2, 3, 3, 32, 0)])
def test_backward_matches_oracle_autograd(E, D, B, T, vpad):
    """Random linear functional of ALL outputs -> every parameter gradient, vs torch autograd on the CPU oracle."""
    b = synth.make_batch(21, B=B, T=T, n_min=2, n_max=5, video_pad_tail=vpad)
    params = synth.make_params(301, E, D, True)
    m = make_model(301, E, D, True, random_pos_start=0)
    out = hip_forward(m, b)
    gen = torch.Generator().manual_seed(5)
    ws = {k: torch.randn(v.shape, generator=gen) for k, v in out.items()}
    loss = sum((out[k] * ws[k].cuda()).sum() for k in out)
    loss.backward()
    p = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    tb = train_ref.to_torch_batch(b)
    ref = tan_ref.forward(p, tb["video"], tb["text_embed"], tb["padding_mask"], tb["text_padding_mask"].bool(), E=E, D=D,
                          use_alignability_head=True)
    sum((ref[k] * ws[k]).sum() for k in ref).backward()
    assert abs(loss.item() - sum((ref[k] * ws[k]).sum() for k in ref).item()) < 1e-2
    for name, prm in m.named_parameters():
        want = p[name].grad
        if want is None:
            assert prm.grad is None or prm.grad.abs().max().item() == 0, name
            continue
        got = prm.grad.cpu()
        scale = want.abs().max().item() + 1e-6
        err = (got - want).abs().max().item() / scale
        assert err < 2e-3, (name, err, scale)


@gpu
def test_bf16_mode_tracks_fp32(golden):
    g = golden("g2_forward_e6d6")
    b = synth.make_batch(12, B=2, T=64, n_min=8, n_max=12)
    m = make_model(102, 6, 6, True, dtype="bf16", random_pos_start=0)
    with torch.no_grad():
        out = hip_forward(m, b)
    for k in ("logits_dual", "logits_joint"):
        err = np.abs(out[k].cpu().numpy() - g[k]).max()
        assert err < 6e-2, (k, err)     # cosine logits in [-1, 1]; bf16 activations through 6+6 layers


def test_state_dict_keys_match_reference_layout():
    from temporalalignnet_amd.tan_model import TemporalAligner, TwinTemporalAligner
    want = set(synth.param_shapes(1, 1, False))
    got = set(TemporalAligner(1, 1, language_model=None).state_dict())
    assert got == want
    tw = TwinTemporalAligner(0.999, num_encoder_layers=1, num_decoder_layers=3, use_alignability_head=1, language_model=None)
    keys = set(tw.state_dict())
    assert len(keys) == 132 and all(k.startswith(("online.", "target.")) for k in keys)


@gpu
@pytest.mark.parametrize("text_pos,rps,dual", [(1, 1, 1), (0, 1, 0), (1, 0, 1)])
def test_constructor_options_match_oracle(text_pos, rps, dual):
    """use_text_pos_enc (tan_model.py:212-228), random_pos_start (the np.random draws of :163,195,224 in the reference's
    order) and return_dual_feature=0: forward outputs and gradients vs the CPU oracle under the same numpy seed."""
    from temporalalignnet_amd.tan_model import TemporalAligner
    E, D = 2, 2
    params = synth.make_params(404, E, D, True)
    m = TemporalAligner(num_encoder_layers=E, num_decoder_layers=D, use_alignability_head=1, language_model=None,
                        use_text_pos_enc=text_pos, random_pos_start=rps, return_dual_feature=dual)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m.cuda()
    b = synth.make_batch(31, B=3, T=32, n_min=2, n_max=6, video_pad_tail=4)
    np.random.seed(77)
    out = hip_forward(m, b)
    assert ("dual_feature_video" in out) == bool(dual)
    gen = torch.Generator().manual_seed(9)
    ws = {k: torch.randn(v.shape, generator=gen) for k, v in out.items()}
    sum((out[k] * ws[k].cuda()).sum() for k in out).backward()
    p = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    tb = train_ref.to_torch_batch(b)
    np.random.seed(77)
    ref = tan_ref.forward(p, tb["video"], tb["text_embed"], tb["padding_mask"], tb["text_padding_mask"].bool(), E=E, D=D,
                          use_alignability_head=True, use_text_pos_enc=bool(text_pos), random_pos_start=bool(rps),
                          return_dual_feature=bool(dual))
    assert set(ref) == set(out)
    for k in ref:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), rtol=1e-4, atol=2e-5, err_msg=k)
    sum((ref[k] * ws[k]).sum() for k in ref).backward()
    for name in ("temporal_pos_embed", "text_temporal_pos_embed", "text_pre_proj.weight", "ln_position_init.weight"):
        want, got = p[name].grad, dict(m.named_parameters())[name].grad
        if want is None:
            assert got is None or got.abs().max().item() == 0, name
            continue
        scale = want.abs().max().item() + 1e-6
        assert (got.cpu() - want).abs().max().item() / scale < 2e-3, name


@gpu
def test_feature_entry_points_match_oracle():
    """get_visual_feature / get_textual_feature / get_textual_feature_with_time / get_joint_feature (the methods drivers and the
    retrieval evaluation call directly, tan_model.py:152-234) vs the CPU oracle; TwinTemporalAligner routes them to `online`."""
    from temporalalignnet_amd.tan_model import TwinTemporalAligner
    E, D = 2, 3
    params = synth.make_params(505, E, D, True)
    tw = TwinTemporalAligner(0.999, num_encoder_layers=E, num_decoder_layers=D, use_alignability_head=1, language_model=None,
                             random_pos_start=0)
    tw.online.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    tw._copy_param()
    tw.cuda()
    b = synth.make_batch(41, B=3, T=32, n_min=2, n_max=6, video_pad_tail=5)
    d = dev_batch(b)
    tb = train_ref.to_torch_batch(b)
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    tpad = tb["text_padding_mask"].bool()
    with torch.no_grad():
        vf = tw.get_visual_feature(d["video"], d["padding_mask"])
        tf = tw.get_textual_feature(d["text_embed"])
        tft = tw.get_textual_feature_with_time(d["text_embed"])
        jv, jt = tw.get_joint_feature(d["video"], d["padding_mask"], tf, d["text_padding_mask"].bool())
        ema = tw.forward_from_ema(d["video"], d["text_embed"], d["padding_mask"], d["text_padding_mask"].bool(), None)
        onl = tw(d["video"], d["text_embed"], d["padding_mask"], d["text_padding_mask"].bool(), None)
    rvf = tan_ref.visual_feature(tb["video"], tb["padding_mask"], p, E)
    rtf = tan_ref.textual_feature(tb["text_embed"], p)
    rtft = tan_ref.textual_feature_with_time(tb["text_embed"], p, 0)
    rjv, rjt = tan_ref.joint_feature(tb["video"], tb["padding_mask"], rtf, tpad, p, D)
    for got, want, name in ((vf, rvf, "visual"), (tf, rtf, "textual"), (tft, rtft, "textual+time"), (jv, rjv, "joint video"),
                            (jt, rjt, "joint text")):
        assert tuple(got.shape) == tuple(want.shape), name
        valid = slice(None)
        np.testing.assert_allclose(got.cpu().numpy()[valid], want.numpy()[valid], rtol=1e-4, atol=3e-5, err_msg=name)
    for k in onl:                                         # target == online right after _copy_param
        torch.testing.assert_close(ema[k], onl[k], rtol=0, atol=0, msg=k)


@gpu
def test_momentum_update_refreshes_every_weight_image_of_the_bf16_target():
    """TwinTemporalAligner._momentum_update() rewrites the target's bf16 shadow in the kernel; the packed / transposed
    images derived from it (row-panel kernels) must be rebuilt too.  After the update the target must compute exactly what a freshly
    constructed model loaded with the same parameters computes (model/tan_model.py:339-351)."""
    from temporalalignnet_amd.tan_model import TemporalAligner, TwinTemporalAligner
    torch.manual_seed(3)
    kw = dict(num_encoder_layers=2, num_decoder_layers=2, language_model=None, compute_dtype="bf16", random_pos_start=0)
    tw = TwinTemporalAligner(0.5, **kw).cuda()
    b = synth.make_batch(21, B=4, T=64, n_min=4, n_max=16)
    d = dev_batch(b)
    args = (d["video"], d["text_embed"], d["padding_mask"], d["text_padding_mask"].bool(), None)
    with torch.no_grad():
        tw.forward_from_ema(*args)                     # builds the target's shadow + packed images at the initial weights
        for p in tw.online.parameters():               # move the online weights far away, then average them in
            p.add_(torch.randn_like(p) * 0.05)
        tw.online.invalidate_shadow()
        tw._momentum_update()
        got = tw.forward_from_ema(*args)
        fresh = TemporalAligner(**kw).cuda()
        fresh.load_state_dict(tw.target.state_dict())
        want = fresh(*args)
    for k in want:
        assert torch.equal(got[k], want[k]), k


@gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_no_grad_forward_skips_the_saved_tensors_and_changes_nothing(dtype):
    """The no-grad forward (EMA target, evaluation entry points; tan_model.py:348-351) does not write the tensors that only a
    backward reads (tan_encoder_desc.no_save); every output is bit-identical to the training-mode forward's."""
    m = make_model(105, 2, 3, True, dtype=dtype, random_pos_start=0)
    b = synth.make_batch(23, B=4, T=64, n_min=4, n_max=16, video_pad_tail=5)
    with torch.no_grad():
        quiet = hip_forward(m, b)
    loud = hip_forward(m, b)
    assert any(v.requires_grad for v in loud.values())
    for k in loud:
        assert torch.equal(quiet[k], loud[k].detach()), k
