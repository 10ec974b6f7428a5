"""BASELINE configs[1] at FULL size (E6D6, T=64, B=128, N<=16), where the CPU oracle is too slow to run in a test: parity
through size-independent properties of the path -- video-permutation equivariance of the model and invariance of the loss,
the directional derivative of the whole step against finite differences (fp32 mode), bf16 vs fp32, and the fused
(logits-free, column-compacted, multi-stream) loss against the materialised reference-layout one."""
import numpy as np
import pytest
import torch

from temporalalignnet_amd import synth

pytestmark = pytest.mark.gpu
B, T, E, D = 128, 64, 6, 6


def _model(dtype, seed=7, head=False):
    from temporalalignnet_amd.tan_model import TemporalAligner
    m = TemporalAligner(num_encoder_layers=E, num_decoder_layers=D, use_alignability_head=int(head), language_model=None,
                        compute_dtype=dtype, random_pos_start=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(seed, E, D, head).items()})
    return m.cuda()


def _batch(seed=21, perm=None):
    from temporalalignnet_amd.train import to_device_batch
    b = synth.make_batch(seed, B=B, T=T, n_min=4, n_max=16)
    if perm is not None:
        for k in ("video", "padding_mask", "text_embed", "text_padding_mask", "abs_text_pos"):
            b[k] = b[k][perm]
        for k in ("start", "end"):
            b[k] = [b[k][i] for i in perm]
    return to_device_batch(b)


def _loss(m, b, fused, args=None):
    from temporalalignnet_amd.loss import get_loss
    from temporalalignnet_amd.train import default_args
    args = args or default_args(model="init")
    out = m(b["video"], b["text_embed"], b["padding_mask"], b["text_padding_mask"].bool(), b["_tgt_raw"], fused=fused)
    if fused:
        out["_fused"].n_text_valid = b["n_text"]
    return get_loss(b, b["video"], b["text_embed"], b["padding_mask"], b["text_padding_mask"], out, args, b["abs_text_pos"]), out


def test_permuting_the_videos_permutes_the_features_and_keeps_the_loss():
    """Each video only meets the others through the similarity matrix: features are per-video functions, the NCE loss is
    symmetric in the batch order."""
    m = _model("fp32")
    perm = np.random.RandomState(0).permutation(B)
    with torch.no_grad():
        l0, o0 = _loss(m, _batch(), fused=False)
        l1, o1 = _loss(m, _batch(perm=perm), fused=False)
    p = torch.from_numpy(perm).cuda()
    torch.testing.assert_close(o1["dual_feature_video"], o0["dual_feature_video"][p], rtol=0, atol=2e-6)
    torch.testing.assert_close(o1["dual_feature_text"], o0["dual_feature_text"][p], rtol=0, atol=2e-6)
    blk0 = o0["logits_joint"][p][:, :, :, p]                       # [B,S,T,B,N] -> permute both video axes
    torch.testing.assert_close(o1["logits_joint"], blk0, rtol=0, atol=5e-6)
    for k in ("loss", "loss-dual", "loss-joint"):
        assert abs(l0[k].item() - l1[k].item()) < 2e-5 * abs(l0[k].item()), (k, l0[k].item(), l1[k].item())


def test_directional_derivative_of_the_full_step_matches_finite_differences():
    """<grad, d> from the hand-written backward (12 layers, both stacks, similarity, NCE) vs (L(p + h d) - L(p - h d)) / 2h."""
    m = _model("fp32")
    b = _batch()
    flat = m.flat_parameters()
    m.zero_grad(set_to_none=True)
    l, _ = _loss(m, b, fused=False)
    l["loss"].backward()
    g = m.flat_grad().clone()
    gen = torch.Generator(device="cuda").manual_seed(3)
    for scale_by_grad in (False, True):
        d = torch.randn(flat.shape, device="cuda", generator=gen)
        if scale_by_grad:
            d = d.abs() * g.sign()                                 # an ascent direction: large, well-conditioned derivative
        d = d / d.norm()
        analytic = float((g.double() * d.double()).sum())
        h = 2e-2
        vals = []
        with torch.no_grad():
            for sgn in (1.0, -1.0):
                flat.add_(d, alpha=sgn * h)
                vals.append(float(_loss(m, b, fused=False)[0]["loss"].double()))
                flat.add_(d, alpha=-sgn * h)
        numeric = (vals[0] - vals[1]) / (2 * h)
        assert abs(numeric - analytic) <= 2e-2 * abs(analytic) + 2e-4, (scale_by_grad, numeric, analytic)


def test_bf16_step_tracks_fp32_and_fused_matches_materialised():
    mf, mb = _model("fp32"), _model("bf16")
    b = _batch()
    res = {}
    for name, m, fused in (("fp32", mf, False), ("bf16", mb, False), ("bf16-fused", mb, True)):
        m.zero_grad(set_to_none=True)
        l, _ = _loss(m, b, fused=fused)
        l["loss"].backward()
        res[name] = ({k: v.item() for k, v in l.items()}, m.flat_grad().clone())
    for k in ("loss", "loss-dual", "loss-joint"):
        ref = res["fp32"][0][k]
        assert abs(res["bf16"][0][k] - ref) < 1e-2 * abs(ref), (k, res["bf16"][0][k], ref)
        assert abs(res["bf16-fused"][0][k] - res["bf16"][0][k]) < 1e-3 * abs(ref), k
    g32, g16, g16f = res["fp32"][1], res["bf16"][1], res["bf16-fused"][1]
    cos = lambda a, c: float((a * c).sum() / (a.norm() * c.norm()))
    assert cos(g16, g32) > 0.995, cos(g16, g32)
    assert cos(g16f, g16) > 0.999, cos(g16f, g16)
    assert abs(float(g16f.norm() / g16.norm()) - 1.0) < 2e-2
