"""Logits-free similarity+NCE (tan_simnce_*) vs the materialised path on the same bf16 model, and vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import loss_ref, train_ref
from temporalalignnet_amd import synth

pytestmark = pytest.mark.gpu


def _setup(model_kind, E, D, B, T, seed=31, **akw):
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    args = default_args(model=model_kind, num_encoder_layers=E, num_decoder_layers=D, lr=1e-3, wd=1e-2, **akw)
    params = synth.make_params(seed, E, D, bool(args.use_alignability_head))
    trainers = []
    for fused in (False, True):
        m = build_model(args, compute_dtype="bf16", random_pos_start=0)
        tgt = m.online if model_kind == "cotrain" else m
        sd = tgt.state_dict()
        for k, v in params.items():
            sd[k].copy_(torch.from_numpy(v))
        if model_kind == "cotrain":
            m._copy_param()
        m.cuda()
        trainers.append(Trainer(m, args, fused_loss=fused))
    b_np = synth.make_batch(seed + 1, B=B, T=T, n_min=3, n_max=9)
    return args, params, trainers, b_np, to_device_batch(b_np)


@pytest.mark.parametrize("kind,E,D,B,T,akw", [("init", 2, 2, 12, 64, {}), ("init", 1, 1, 5, 16, {"learn_agreement": 1}),
                                              ("cotrain", 1, 3, 10, 32, {"loss_threshold": 0.5})])
def test_fused_matches_materialised(kind, E, D, B, T, akw):
    args, params, (t_mat, t_fus), b_np, b = _setup(kind, E, D, B, T, **akw)
    out = []
    for tr in (t_mat, t_fus):
        tr.zero_grad()
        ld = tr.forward_backward(b)
        out.append(({k: v.item() for k, v in ld.items()}, tr.online.flat_grad().clone()))
    (l0, g0), (l1, g1) = out
    assert set(l0) == set(l1)
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 2e-3 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    # same bf16 model, same kernels up to the similarity/NCE stage: gradients agree to bf16 rounding of dl
    denom = g0.norm().item()
    assert (g0 - g1).norm().item() / denom < 2e-2, (g0 - g1).norm().item() / denom
    f = t_mat.online._flat
    for name in ("video_pre_proj.weight", "text_pre_proj.weight", f"joint_temporal_encoder.resblocks.{D - 1}.mlp.c_fc.weight"):
        a, c = f.view(g0, name), f.view(g1, name)
        assert (a - c).norm().item() / (a.norm().item() + 1e-12) < 3e-2, name


def test_fused_loss_close_to_oracle():
    E, D, B, T = 2, 2, 12, 64
    args, params, (_, t_fus), b_np, b = _setup("init", E, D, B, T)
    t_fus.zero_grad()
    ld = t_fus.forward_backward(b)
    ref = train_ref.RefTrainer(params, E=E, D=D, args=loss_ref.default_args(), lr=1e-3, wd=1e-2, random_pos_start=False)
    lr_, _ = ref.step(train_ref.to_torch_batch(b_np))
    for k in ("loss", "loss-dual", "loss-joint"):
        assert abs(ld[k].item() - lr_[k].item()) < 2e-2 * max(1.0, abs(lr_[k].item())), (k, ld[k].item(), lr_[k].item())


def test_fused_forward_has_no_logits_and_three_steps_train():
    args, params, (_, t_fus), b_np, b = _setup("cotrain", 1, 3, 6, 16, loss_threshold=0.5)
    m = t_fus.model
    out = m(b["video"], b["text_embed"], b["padding_mask"], b["text_padding_mask"].bool(), None, fused=True)
    assert "_fused" in out and "logits_dual" not in out and "joint_logits_alignability" in out
    losses = [t_fus.step(b)["loss"].item() for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[2] < losses[0]


def test_column_compaction_is_a_noop_on_the_result():
    """The sweep over the compacted text matrix (host-known sentence count) == the sweep over all B*N padded columns."""
    args, params, (_, t_fus), b_np, b = _setup("init", 2, 2, 12, 64, learn_agreement=1)
    assert b["n_text"] == int((b_np["text_padding_mask"] == 0).sum()) < b_np["text_padding_mask"].size
    res = []
    for nt in (b["n_text"], None):
        bb = dict(b, n_text=nt)
        t_fus.zero_grad()
        ld = t_fus.forward_backward(bb)
        res.append(({k: v.item() for k, v in ld.items()}, t_fus.online.flat_grad().clone()))
    (l0, g0), (l1, g1) = res
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-5 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    assert (g0 - g1).norm().item() / g1.norm().item() < 2e-3        # bf16 d-logits, different summation order only


def test_more_text_columns_than_the_fused_sweep_accepts_falls_back_to_logits():
    """B*N beyond tan_simnce_max_cols(): the trainer silently uses the materialised-logits path for that batch."""
    from temporalalignnet_amd import _lib
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    lim = _lib.lib().tan_simnce_max_cols()
    B = 72
    N = lim // B + 12                     # (8192 columns for the resident sweep at C = 512: B*N = 9 000)
    b_np = synth.make_batch(3, B=B, T=16, n_min=N - 2, n_max=N)
    assert b_np["text_embed"].shape[1] == N and (b_np["text_padding_mask"] == 0).sum() > lim
    args = default_args(model="init", num_encoder_layers=1, num_decoder_layers=1)
    torch.manual_seed(0)
    tr = Trainer(build_model(args, compute_dtype="bf16").cuda(), args)
    assert tr.fused_loss
    losses = [tr.step(to_device_batch(b_np))["loss"].item() for _ in range(2)]
    assert all(np.isfinite(losses))


def test_compaction_lifts_the_padded_column_count_over_the_sweep_limit():
    """B*N padded columns exceed tan_simnce_max_cols() but the real sentences fit: the fused sweep runs on the compacted
    matrix and agrees with the materialised-logits loss."""
    from temporalalignnet_amd import _lib
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    lim = _lib.lib().tan_simnce_max_cols()
    b_np = synth.make_batch(5, B=72, T=16, n_min=2, n_max=lim // 72 + 12)
    B, N = b_np["text_embed"].shape[:2]
    n_text = int((b_np["text_padding_mask"] == 0).sum())
    assert B * N > lim and (n_text + 63) // 64 * 64 <= lim
    args = default_args(model="init", num_encoder_layers=1, num_decoder_layers=1)
    losses = []
    for fused in (True, False):
        torch.manual_seed(0)
        tr = Trainer(build_model(args, compute_dtype="bf16", random_pos_start=0).cuda(), args, fused_loss=fused)
        tr.zero_grad()
        losses.append(tr.forward_backward(to_device_batch(b_np))["loss"].item())
    assert abs(losses[0] - losses[1]) < 2e-3 * abs(losses[1]), losses


@pytest.mark.parametrize("S,B,T,N,shared,compact,leak", [(2, 3, 70, 5, False, False, False), (3, 4, 64, 9, True, True, True),
                                                         (1, 5, 40, 16, False, True, False), (2, 6, 64, 30, True, False, True)])
def test_kept_exponentials_match_the_recomputing_backward(S, B, T, N, shared, compact, leak):
    """_FusedNCEFn with the exponentials kept by the statistics sweep (tan_simnce_fwd_keep / tan_simnce_bwd_dl_kept: one element-wise
    pass) against the recomputing backward (tan_simnce_bwd_dl) on the same inputs: ragged R = B*T and column counts that are not
    multiples of the 128 x 128 tiles, shared and per-stage text features, column compaction, leaked (padded) frames.  Terms are
    the same sweep's; the feature gradients differ by one extra bf16 rounding of e."""
    from temporalalignnet_amd import _lib, loss as L
    if not _lib.lib().tan_simnce_keeps(512):
        pytest.skip("kept-exponentials path not available")
    g = torch.Generator(device="cpu").manual_seed(1234 + S + B)
    R, Mp, Cw = B * T, B * N, 512
    vn = torch.nn.functional.normalize(torch.randn(S, R, Cw, generator=g), dim=-1).cuda().bfloat16()
    tn = torch.nn.functional.normalize(torch.randn(1 if shared else S, Mp, Cw, generator=g), dim=-1).cuda().bfloat16()
    tgt = (torch.rand(B, T, N, generator=g) < 0.15).float().cuda()
    tpad = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        tpad[b, max(1, N - b):] = True                     # video b has N - b real sentences (at least one)
    col_invalid = tpad.view(-1).to(torch.uint8).cuda()
    row_leak = None
    if leak:
        row_leak = torch.zeros(R, dtype=torch.uint8)
        row_leak[T - 3:T] = 1; row_leak[R - 2:] = 1
        row_leak = row_leak.cuda()
    prep = None
    if compact:
        n_valid = int((~tpad).sum())
        prep = L.compaction_prep(col_invalid, n_valid)
    outs = []
    real = _lib.lib

    class NoKeep:
        def __init__(self, lib): self._lib = lib
        def __getattr__(self, k): return (lambda *a: 0) if k == "tan_simnce_keeps" else getattr(self._lib, k)

    for keep in (True, False):
        if not keep:
            proxy = NoKeep(real())
            _lib.lib = lambda: proxy
        try:
            v = vn.clone().requires_grad_(True); t = tn.clone().requires_grad_(True)
            v_terms, t_terms = L._FusedNCEFn.apply(v, t, tgt, col_invalid, row_leak, B, T, N, prep)
            gv = torch.randn(v_terms.shape, generator=g).cuda() if keep else gv
            gt = torch.randn(t_terms.shape, generator=g).cuda() if keep else gt
            (v_terms * gv).sum().add((t_terms * gt).sum()).backward()
            outs.append((v_terms.detach(), t_terms.detach(), v.grad.float(), t.grad.float()))
        finally:
            _lib.lib = real
    (v0, t0, dv0, dt0), (v1, t1, dv1, dt1) = outs
    # (the same sweep; sums of leaked entries meet in f32 atomics, so the last bit may differ between two runs)
    assert torch.allclose(v0, v1, rtol=1e-5, atol=1e-5) and torch.allclose(t0, t1, rtol=1e-5, atol=1e-5)
    for a, c in ((dv0, dv1), (dt0, dt1)):
        assert torch.isfinite(a).all()
        assert (a - c).norm().item() <= 6e-3 * c.norm().item() + 1e-6, (a - c).norm().item() / c.norm().item()


@pytest.mark.parametrize("S,B,T,N,shared,compact,leak", [(2, 3, 70, 5, False, False, False), (3, 4, 64, 9, True, True, True),
                                                         (1, 5, 40, 16, False, True, False), (2, 6, 64, 30, True, False, True),
                                                         (2, 40, 64, 10, False, True, False), (3, 24, 64, 16, True, False, False)])
def test_one_pass_dlogits_and_dvn_match_the_pass_plus_gemm(S, B, T, N, shared, compact, leak):
    """tan_simnce_bwd_dl_dvn_kept (d logits + d_vn = dl . tn in one pass: the tile is the MFMA operand while it is in the LDS; same-video
    corrections from the dense array of simnce_corr_kernel) against tan_simnce_bwd_dl_kept + the GEMM it replaces, same kept exponentials.  d_tn is computed from the d-logits either path wrote by the
    same GEMM (equal up to the run-to-run last-bit noise of the sweep's atomically summed row sums); d_vn differs by the f32 summation
    order only (one bf16 ulp)."""
    from temporalalignnet_amd import _lib, loss as L
    if not _lib.lib().tan_simnce_keeps(512):
        pytest.skip("kept-exponentials path not available")
    g = torch.Generator(device="cpu").manual_seed(4321 + S + B)
    R, Mp, Cw = B * T, B * N, 512
    vn = torch.nn.functional.normalize(torch.randn(S, R, Cw, generator=g), dim=-1).cuda().bfloat16()
    tn = torch.nn.functional.normalize(torch.randn(1 if shared else S, Mp, Cw, generator=g), dim=-1).cuda().bfloat16()
    tgt = (torch.rand(B, T, N, generator=g) < 0.15).float().cuda()
    tpad = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        tpad[b, max(1, N - (b % N)):] = True
    col_invalid = tpad.view(-1).to(torch.uint8).cuda()
    row_leak = None
    if leak:
        row_leak = torch.zeros(R, dtype=torch.uint8)
        row_leak[T - 3:T] = 1; row_leak[R - 2:] = 1
        row_leak = row_leak.cuda()
    prep = L.compaction_prep(col_invalid, int((~tpad).sum())) if compact else None
    outs, gv, gt = [], None, None
    keep_flag = L._FUSED_DVN
    try:
        for flag in (True, False):
            L._FUSED_DVN = flag
            v = vn.clone().requires_grad_(True); t = tn.clone().requires_grad_(True)
            v_terms, t_terms = L._FusedNCEFn.apply(v, t, tgt, col_invalid, row_leak, B, T, N, prep)
            if gv is None:
                gv = torch.randn(v_terms.shape, generator=g).cuda(); gt = torch.randn(t_terms.shape, generator=g).cuda()
            (v_terms * gv).sum().add((t_terms * gt).sum()).backward()
            outs.append((v.grad.float(), t.grad.float()))
    finally:
        L._FUSED_DVN = keep_flag
    dv1, dt1 = outs[1]
    for dv0, dt0 in outs[:1]:
        assert torch.isfinite(dv0).all() and torch.isfinite(dt0).all()
        # (two forward sweeps: their row sums meet in f32 atomics)
        assert (dt0 - dt1).norm().item() <= 3e-3 * dt1.norm().item() + 1e-7, (dt0 - dt1).norm().item() / dt1.norm().item()
        assert (dt0 - dt1).abs().max().item() <= 1e-2 * dt1.abs().max().item() + 1e-7
        assert (dv0 - dv1).norm().item() <= 3e-3 * dv1.norm().item() + 1e-7, (dv0 - dv1).norm().item() / dv1.norm().item()
        assert (dv0 - dv1).abs().max().item() <= 1e-2 * dv1.abs().max().item() + 1e-7
