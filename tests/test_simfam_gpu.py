"""One feature family of the logits-free NCE from the stacks' stage outputs to their stage gradients in 6-7 launches
(`tan_simfam_fwd / tan_simfam_bwd`: tan_model.py:116-119,136-139 + train/loss.py:240-253 and their autograd) against the 18 separate
launches it replaces (L2-normalisation launches around `nce_family`), which are themselves pinned to the oracle / the reference goldens
by tests/test_fused_gpu.py and tests/test_loss_gpu.py: same arithmetic on the same values, so the results agree to the last bits of
f32 summation order."""
import numpy as np
import pytest
import torch

from temporalalignnet_amd import synth

pytestmark = pytest.mark.gpu
C = 512


def _problem(B, T, N_max, S, joint, seed, compaction=True, pad_frames=False):
    from temporalalignnet_amd import loss as L
    from temporalalignnet_amd.train import default_args, to_device_batch
    b = to_device_batch(synth.make_batch(seed, B=B, T=T, n_min=max(1, N_max // 4), n_max=N_max))
    N = b["text_embed"].shape[1]
    Lr = T + N if joint else T
    g = torch.Generator(device="cuda").manual_seed(seed)
    stages = [(torch.randn(B * Lr, C, device="cuda", generator=g) * (1.0 + 0.3 * s)).to(torch.bfloat16) for s in range(S)]
    text = stages if joint else [torch.randn(B * N, C, device="cuda", generator=g).to(torch.bfloat16)]
    prep = L.prepare_inputs(b, b["padding_mask"], b["text_padding_mask"], T, N, "cuda", default_args(model="init"), b.get("n_text"),
                            want_compaction=compaction)
    nv = prep.get("nv") if compaction else None
    cols = prep["cols_pos_c"] if nv is not None else prep["cols_pos"]
    Sd = S
    g_v, g_t, _, _, _ = L.nce_term_grads(prep["rows_pos"], cols, Sd, Sd)
    # (not the 1/count weights only: every row / column gets its own upstream gradient)
    g_v = g_v * (0.5 + torch.rand(g_v.shape, device="cuda", generator=g))
    g_t = g_t * (0.5 + torch.rand(g_t.shape, device="cuda", generator=g))
    return dict(B=B, T=T, N=N, S=S, Lr=Lr, stages=stages, text=text, tgt=prep["tgt"], ci=prep["tpad_u8"].view(B * N), nv=nv,
                g_v=g_v.contiguous(), g_t=g_t.contiguous(), joint=joint)


def _run(p, fused, monkeypatch, norm_in_sweep=True):
    from temporalalignnet_amd import loss as L
    monkeypatch.setattr(L, "_SIMFAM", bool(fused))
    monkeypatch.setattr(L, "_SIMFAM_NORM", bool(norm_in_sweep))
    B, T, N, S, Lr = p["B"], p["T"], p["N"], p["S"], p["Lr"]
    d_video = [torch.full((B * Lr, C), 7.0, device="cuda").to(torch.bfloat16) for _ in range(S)]
    if p["joint"]:
        d_text, v_grp, t_grp = d_video, (Lr, 0), (Lr, T)
    else:
        d_text, v_grp, t_grp = [torch.full((B * N, C), 7.0, device="cuda").to(torch.bfloat16)], (T, 0), (N, 0)
    v, t = L.nce_family_stages(p["stages"], v_grp, p["text"], t_grp, d_video, d_text, p["tgt"], p["ci"], B, T, N, p["nv"],
                               p["g_v"], p["g_t"])
    torch.cuda.synchronize()
    return v.clone(), t.clone(), [x.float() for x in d_video], [x.float() for x in d_text]


@pytest.mark.parametrize("B,T,N_max,S,joint,compaction", [
    (8, 64, 16, 3, False, True),          # dual family, compacted columns
    (8, 64, 16, 3, True, True),           # joint family: frame and sentence rows of the same stage buffers
    (8, 24, 5, 1, True, False),           # no compaction (Mc = B*N), B*T not a multiple of the 128-row panels
    (16, 64, 9, 2, False, False),
    (8, 256, 24, 2, True, True),          # len=256, more than 16 sentences
    (128, 64, 16, 6, True, True),         # BASELINE configs[1] at full size
])
def test_family_launches_match_the_separate_launches(B, T, N_max, S, joint, compaction, monkeypatch):
    p = _problem(B, T, N_max, S, joint, seed=B + T + S, compaction=compaction)
    from temporalalignnet_amd import loss as L
    Mc = p["nv"][0].shape[0] if p["nv"] is not None else B * p["N"]
    assert L.simfam_ok(S, p["N"], Mc, torch.bfloat16), (p["N"], Mc)
    v0, t0, dv0, dt0 = _run(p, False, monkeypatch)
    v1, t1, dv1, dt1 = _run(p, True, monkeypatch)
    # the sweep normalising its own frame panel (default) against the separate L2-normalisation launch: the same values
    v2, t2, dv2, dt2 = _run(p, True, monkeypatch, norm_in_sweep=False)
    torch.testing.assert_close(v2, v1, rtol=2e-4, atol=2e-4)          # (the norm sums eight elements per lane instead of four: a unit feature may round the other way in bf16)
    for a, c in zip(dv2 + dt2, dv1 + dt1):
        assert (a - c).norm().item() <= 2e-3 * c.norm().item()
    # terms: f32 sums in another order; the in-sweep norm sums eight elements per lane instead of four, so a unit feature may round the
    # other way in bf16 (a handful of rows move by ~1e-4 relative, the rest by f32 rounding)
    assert torch.isfinite(v1).all()
    torch.testing.assert_close(v1, v0, rtol=3e-4, atol=3e-4)
    assert ((v1 - v0).abs() > 2e-5 * (1 + v0.abs())).float().mean().item() < 2e-3
    real = (p["nv"][2] == 0) if p["nv"] is not None else (p["ci"] == 0)
    torch.testing.assert_close(t1[:, real], t0[:, real], rtol=3e-4, atol=3e-4)
    assert torch.isfinite(t1).all()        # (the filler columns too: they are multiplied by a zero mask, NaN would poison the mean)
    for a, c in zip(dv1 + ([] if p["joint"] else dt1), dv0 + ([] if p["joint"] else dt0)):
        assert torch.isfinite(a).all()
        assert not (a == 7.0).any()            # every row written
        scale = c.abs().max().item()
        # bf16 outputs of the same f32 arithmetic up to summation order: a few entries may round the other way
        assert (a - c).abs().max().item() <= 2e-2 * scale, ((a - c).abs().max().item(), scale)
        assert (a - c).norm().item() <= 2e-3 * c.norm().item(), ((a - c).norm().item(), c.norm().item())


def test_family_path_is_what_the_two_chain_step_runs(monkeypatch):
    """The two-chain training step reaches `tan_simfam_fwd / bwd` (and not the separate launches) at the benchmarked shapes."""
    from temporalalignnet_amd import _lib, loss as L
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    calls = []
    lib = _lib.lib()
    real_fwd = lib.tan_simfam_fwd

    class Spy:
        def __getattr__(self, name):
            if name == "tan_simfam_fwd":
                def f(*a):
                    calls.append(name)
                    return real_fwd(*a)
                return f
            return getattr(lib, name)
    monkeypatch.setattr(_lib, "lib", lambda: Spy())
    args = default_args(model="init", num_encoder_layers=2, num_decoder_layers=2)
    m = build_model(args, compute_dtype="bf16", random_pos_start=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(3, 2, 2, False).items()})
    tr = Trainer(m.cuda(), args)
    b = to_device_batch(synth.make_batch(5, B=8, T=64, n_min=4, n_max=16))
    assert tr._chains_eligible(b, tr.fused_loss)
    ld = tr.step(b)
    torch.cuda.synchronize()
    assert calls == ["tan_simfam_fwd", "tan_simfam_fwd"] and np.isfinite(float(ld["loss"]))
