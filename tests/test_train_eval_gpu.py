"""Train-step driver (a16) and HTM-Align evaluation harness (a17) on the HIP path vs reference-generated goldens."""
import numpy as np
import pytest
import torch

from temporalalignnet_amd import synth

pytestmark = pytest.mark.gpu


def fingerprint(t):
    g = t.detach().double().flatten().cpu()
    idx = torch.linspace(0, g.numel() - 1, 16).long()
    return np.concatenate([[g.sum().item(), g.norm().item()], g[idx].numpy()])


def check_fp(got, want, name, rtol):
    scale = abs(want[1]) + 1e-30
    if name.endswith("in_proj_bias"):          # key-bias third: zero true gradient, Adam amplifies rounding noise
        idx = np.linspace(0, 1535, 16).astype(np.int64)
        ok = (idx < 512) | (idx >= 1024)
        np.testing.assert_allclose(got[2:][ok], want[2:][ok], rtol=rtol, atol=rtol * 0.05, err_msg=name)
        return
    assert abs(got[0] - want[0]) <= rtol * scale * 8 + 1e-6, (name, got[0], want[0])
    np.testing.assert_allclose(got[1:], want[1:], rtol=rtol, atol=rtol * scale * 0.05 + 1e-8, err_msg=name)


def load(model, params, prefix=""):
    sd = model.state_dict()
    for k, v in params.items():
        sd[prefix + k].copy_(torch.from_numpy(v))


def test_g5_three_train_steps_init(golden):
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    g = golden("g5_train_steps")
    args = default_args(model="init", num_encoder_layers=1, num_decoder_layers=1, lr=1e-3, wd=1e-2)
    model = build_model(args)                       # random_pos_start=1 (reference default for 'init')
    load(model, synth.make_params(105, 1, 1, False))
    model.cuda()
    tr = Trainer(model, args)
    b = to_device_batch(synth.make_batch(15, B=8, T=16, n_min=2, n_max=5))
    np.random.seed(7)
    losses = [tr.step(b)["loss"].item() for _ in range(3)]
    np.testing.assert_allclose(losses, g["init/losses"], rtol=2e-4)
    for k in g.files:
        if k.startswith("init/param/"):
            name = k[len("init/param/"):]
            check_fp(fingerprint(dict(model.named_parameters())[name]), g[k], name, 2e-3)


def test_g5_three_train_steps_cotrain(golden):
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    g = golden("g5_train_steps")
    args = default_args(model="cotrain", num_encoder_layers=1, num_decoder_layers=3, lr=1e-3, wd=1e-2, loss_threshold=0.5,
                        momentum_m=0.99)
    model = build_model(args)
    load(model.online, synth.make_params(106, 1, 3, True))
    model._copy_param()
    model.cuda()
    tr = Trainer(model, args)
    b = to_device_batch(synth.make_batch(16, B=6, T=16, n_min=2, n_max=5))
    losses = [tr.step(b)["loss"].item() for _ in range(3)]
    np.testing.assert_allclose(losses, g["cotrain/losses"], rtol=2e-4)
    named = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("cotrain/param/"):
            name = k[len("cotrain/param/"):]
            check_fp(fingerprint(named[name]), g[k], name, 2e-3)
    # parameters the reference never updates (grad is None there) must be untouched, decay included
    init = synth.make_params(106, 1, 3, True)
    assert torch.equal(named["online.mlp.weight"].cpu(), torch.from_numpy(init["mlp.weight"]))


def test_g6_eval_harness(golden):
    from temporalalignnet_amd.eval_align import make_sim_fn, test_alignment_htm
    from temporalalignnet_amd.tan_model import TemporalAligner
    g = golden("g6_eval_harness")
    m = TemporalAligner(1, 3, use_alignability_head=1, random_pos_start=0, language_model=None)
    load(m, synth.make_params(108, 1, 3, True))
    m.cuda().eval()
    videos = synth.align_videos()
    emb = {s: torch.from_numpy(e).cuda() for v in videos for s, e in zip(v["str"], v["emb"])}
    fn = make_sim_fn(m, lambda strs: torch.stack([emb[s] for s in strs]))
    metric, per_video = test_alignment_htm(fn, videos, return_per_video=True)
    assert metric["Recall"] == pytest.approx(float(g["Recall"]), abs=1e-12)
    assert metric["AUC"] == pytest.approx(float(g["AUC"]), abs=1e-9)
    for i, pv in enumerate(per_video):
        al = torch.from_numpy(np.asarray(videos[i]["aligned"]).astype(bool))
        assert (pv["argmax"].numpy() == g[f"v{i}/argmax"]).all()           # bit-exact alignment indices
        np.testing.assert_allclose(pv["sim"][al].numpy(), g[f"v{i}/sim_aligned"], rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(pv["score"].numpy(), g[f"v{i}/align_score"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("max_windows", [256, 3])
def test_g6_eval_harness_with_batched_windows(golden, max_windows):
    """The batched evaluation (all windows of a video through `eval_windows`: one pass of each stack, short windows and
    unequal sentence counts padded + masked) must reproduce the reference's window-by-window numbers -- the same golden."""
    from temporalalignnet_amd.eval_align import make_batched_sim_fn, make_sim_fn, test_alignment_htm
    from temporalalignnet_amd.tan_model import TemporalAligner
    g = golden("g6_eval_harness")
    m = TemporalAligner(1, 3, use_alignability_head=1, random_pos_start=0, language_model=None)
    load(m, synth.make_params(108, 1, 3, True))
    m.cuda().eval()
    videos = synth.align_videos()
    emb = {s: torch.from_numpy(e).cuda() for v in videos for s, e in zip(v["str"], v["emb"])}
    embed = lambda strs: torch.stack([emb[s] for s in strs])
    metric, per_video = test_alignment_htm(None, videos, return_per_video=True,
                                           batched_sim=make_batched_sim_fn(m, embed, max_windows=max_windows))
    assert metric["Recall"] == pytest.approx(float(g["Recall"]), abs=1e-12)
    assert metric["AUC"] == pytest.approx(float(g["AUC"]), abs=1e-9)
    ref_metric, ref_pv = test_alignment_htm(make_sim_fn(m, embed), videos, return_per_video=True)
    for i, pv in enumerate(per_video):
        al = torch.from_numpy(np.asarray(videos[i]["aligned"]).astype(bool))
        assert (pv["argmax"].numpy() == g[f"v{i}/argmax"]).all()
        np.testing.assert_allclose(pv["sim"][al].numpy(), g[f"v{i}/sim_aligned"], rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(pv["score"].numpy(), g[f"v{i}/align_score"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(pv["sim"].numpy(), ref_pv[i]["sim"].numpy(), rtol=1e-4, atol=2e-4)   # every sentence


def test_batched_eval_in_bf16_agrees_with_window_by_window_and_is_faster():
    import time
    from temporalalignnet_amd.eval_align import make_batched_sim_fn, make_sim_fn, test_alignment_htm
    from temporalalignnet_amd.train import build_model, default_args
    args = default_args(model="init", num_encoder_layers=3, num_decoder_layers=3, use_alignability_head=1)
    torch.manual_seed(3)
    m = build_model(args, compute_dtype="bf16", random_pos_start=0).cuda().eval()
    videos = synth.align_videos()
    emb = {s: torch.from_numpy(e).cuda() for v in videos for s, e in zip(v["str"], v["emb"])}
    embed = lambda strs: torch.stack([emb[s] for s in strs])
    out = {}
    for name, kw in (("window", dict(get_text_visual_sim=make_sim_fn(m, embed))),
                     ("batched", dict(get_text_visual_sim=None, batched_sim=make_batched_sim_fn(m, embed)))):
        test_alignment_htm(videos=videos, return_per_video=True, **kw)           # warm-up (workspaces)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        metric, pv = test_alignment_htm(videos=videos, return_per_video=True, **kw)
        torch.cuda.synchronize()
        out[name] = (metric, pv, time.perf_counter() - t0)
    (m0, pv0, t_w), (m1, pv1, t_b) = out["window"], out["batched"]
    print(f"eval harness: window-by-window {t_w * 1e3:.1f} ms, batched {t_b * 1e3:.1f} ms")
    for a, b in zip(pv0, pv1):
        # logits are cosines / 0.07 in bf16: one bf16 ulp of a feature moves them by ~1e-2
        assert (a["sim"] - b["sim"]).abs().max() <= 0.25
        assert (a["argmax"] == b["argmax"]).float().mean() >= 0.9
    assert abs(m0["AUC"] - m1["AUC"]) <= 0.05
    assert t_b < t_w


def test_lr_schedule_lag_matches_reference_lambda_lr():
    """Replays the reference's LambdaLR usage with torch on a dummy parameter: args.iteration starts at 1 (train/main.py:281),
    lr_scheduler.step(args.iteration) once before training (main.py:499) and after every batch with the pre-increment counter
    (main.py:138-140); a resume re-enters at main.py:444,499 with the saved counter."""
    import functools
    import warnings
    from temporalalignnet_amd.train import Trainer, default_args, lr_multiplier
    args = default_args(epochs=2)
    fn = functools.partial(lr_multiplier, iter_per_epoch=1500, epochs=2, warmup=1000)

    def reference_lrs(n_batches, start_iteration=1):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=args.lr)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, fn)
        it, used = start_iteration, []
        sched.step(it)                                       # main.py:499
        for _ in range(n_batches):
            used.append(opt.param_groups[0]["lr"])           # the optimizer step of this batch
            sched.step(it)                                   # main.py:138
            it += 1                                          # main.py:140
        return used, it

    tr = Trainer.__new__(Trainer)
    tr.args, tr.iter_per_epoch, tr.warmup, tr.iteration, tr.batches_seen, tr._lr_iter, tr._resume_bump = args, 1500, 1000, 0, 0, None, 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        used, it = reference_lrs(2990)
        for b in range(0, 2990, 7):
            tr.batches_seen = b
            assert tr.current_lr() == pytest.approx(used[b], rel=1e-12, abs=1e-18), b
        assert used[0] == used[1] == pytest.approx(args.lr * 1e-3)       # batches 0 and 1 both run at lambda(1)
        # resume after 40 batches: saved 'iteration' = 41; the first resumed batch runs one schedule position ahead
        used_r, _ = reference_lrs(3, start_iteration=41)
        tr.batches_seen, tr._resume_bump = 40, 1
        assert tr.current_lr() == pytest.approx(used_r[0], rel=1e-12)
        tr.batches_seen, tr._resume_bump = 41, 0
        assert tr.current_lr() == pytest.approx(used_r[1], rel=1e-12)
        tr.batches_seen = 42
        assert tr.current_lr() == pytest.approx(used_r[2], rel=1e-12)


@pytest.mark.gpu
def test_steps_do_not_accumulate_device_memory():
    """Regression: the autograd node of the forward once kept its own output tensors reachable from ctx (a cycle through C++
    that Python's gc cannot collect) and every step leaked its activation record."""
    import gc
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    for kind, dtype in (("init", "bf16"), ("cotrain", "bf16"), ("init", "fp32")):
        args = default_args(model=kind, num_encoder_layers=3, num_decoder_layers=3, loss_threshold=0.5 if kind == "cotrain" else 0.0)
        tr = Trainer(build_model(args, compute_dtype=dtype).cuda(), args)
        b = to_device_batch(synth.make_batch(4, B=8, T=32, n_min=3, n_max=7))
        for _ in range(3):
            tr.step(b)
        torch.cuda.synchronize(); gc.collect()
        before = torch.cuda.memory_allocated()
        for _ in range(6):
            tr.step(b)
        torch.cuda.synchronize(); gc.collect()
        assert torch.cuda.memory_allocated() - before < (1 << 20), (kind, dtype, torch.cuda.memory_allocated() - before)


def test_load_state_dict_into_a_used_model_refreshes_the_bf16_shadow():
    """Regression: parameters are views of a flat f32 buffer and the bf16 weights the kernels read are a shadow of it;
    load_state_dict writes through the parameter tensors (own version counters), so the shadow has to be invalidated
    explicitly -- otherwise a model that already ran keeps computing with the old weights."""
    from temporalalignnet_amd import synth
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    args = default_args(model="init", num_encoder_layers=2, num_decoder_layers=2)
    b = to_device_batch(synth.make_batch(5, B=4, T=32, n_min=3, n_max=6))
    torch.manual_seed(0)
    m = build_model(args, compute_dtype="bf16", random_pos_start=0).cuda()
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    tr = Trainer(m, args)
    tr.iteration = 2000                                   # past warm-up: a visible update
    l0 = tr.step(b)["loss"].item()
    l1 = Trainer(m, args).forward_backward(b)["loss"].item()
    assert abs(l1 - l0) > 1e-4                            # the step changed the weights
    m.load_state_dict(state0)
    l2 = Trainer(m, args).forward_backward(b)["loss"].item()
    assert abs(l2 - l0) <= 1e-5 * max(1.0, abs(l0)), (l0, l1, l2)
