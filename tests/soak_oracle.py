"""Randomised soak of the training step against the CPU oracle (test infrastructure, not collected by pytest: run by hand on a GPU box, `python tests/soak_oracle.py [n] [seed]`; it lives under tests/ because only tests may import `oracle/`).

For n random SMALL configurations (stage 1 and stage 2, every temporal-agreement type, tiny batches, one to 24 sentences, padded frames) the
fp32 HIP step -- parity mode: exact f32 fma chains, bit-exact index decisions -- runs ONE training step from `synth.make_params` on
`synth.make_batch`, and so does `oracle.train_ref.RefTrainer` (the restatement of train/main.py:81-122 + train/loss.py pinned by the goldens):
every entry of the loss dict must agree to 1e-3 relative, NaN where the oracle has NaN (the reference's own degenerate cases: no
selected sentence -> pos_weight = 1/0 - 1), and the updated parameters must agree like tests/test_train_eval_gpu.py's G5 check."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from oracle import train_ref          # (test infrastructure: this tool is a checker, not product)
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch


def load(model, params):
    sd = model.state_dict()
    for k, v in params.items():
        sd[k].copy_(torch.from_numpy(v))


BF16 = bool(os.environ.get("SOAK_BF16"))       # the throughput mode over random shapes that cross every kernel-selection boundary


def bf16_case(rng, it):
    """Stage 1 in bf16 against the fp32 oracle: loss to 2 %, the flat gradient NORM-relative to 2 % (measured: <= 0.8 %), every tensor with a gradient above
    the noise floor to 10 % -- window lengths and sentence counts chosen around the dispatch boundaries (rows per video 48 / 80 / 128 /
    288: fused attention branch | short | mid | streamed attention; row counts off the 64-row panels: tiled-GEMM fallback)."""
    T = int(rng.choice([8, 16, 31, 40, 47, 48, 56, 63, 64, 64, 65, 72, 79, 96, 120, 128, 129, 200, 256, 270]))
    nmax = int(rng.choice([1, 2, 5, 8, 15, 16, 17, 24, 33]))
    cfg = dict(B=int(rng.choice([1, 2, 3, 4, 8, 16])), T=T, nmin=int(rng.integers(1, nmax + 1)), nmax=nmax, vpad=int(rng.choice([0, 0, 3, T // 4])),
               E=int(rng.integers(1, 3)), D=int(rng.integers(1, 3)), seed=int(rng.integers(1, 10000)))
    args = default_args(model="init", num_encoder_layers=cfg["E"], num_decoder_layers=cfg["D"], lr=1e-3, wd=1e-2, seq_len=T)
    params = synth.make_params(cfg["seed"], cfg["E"], cfg["D"], False)
    b_np = synth.make_batch(cfg["seed"] + 1, B=cfg["B"], T=T, n_min=cfg["nmin"], n_max=cfg["nmax"], video_pad_tail=cfg["vpad"])
    ref = train_ref.RefTrainer(params, E=cfg["E"], D=cfg["D"], args=args, lr=1e-3, wd=1e-2, random_pos_start=0)
    r_out, _ = ref.step(train_ref.to_torch_batch(b_np))
    model = build_model(args, compute_dtype="bf16", random_pos_start=0)
    load(model, params)
    tr = Trainer(model.cuda(), args)
    tr.zero_grad()
    out = tr.forward_backward(to_device_batch(b_np))
    torch.cuda.synchronize()
    ok, notes = True, []
    for k in ("loss", "loss-dual", "loss-joint"):
        a, c = float(out[k]), float(r_out[k])
        if not np.isfinite(a) or abs(a - c) > 2e-2 * max(1.0, abs(c)):
            ok = False
            notes.append((k, a, c))
    named = dict(tr.online.named_parameters())
    num = den = 0.0
    gmax = max(float(v.grad.norm()) for v in ref.p.values() if v.grad is not None)
    for k, v in ref.p.items():
        if v.grad is None or named[k].grad is None:
            continue
        g, w = named[k].grad.detach().float().cpu(), v.grad.detach()
        num += float((g - w).norm()) ** 2
        den += float(w.norm()) ** 2
        if float(w.norm()) > 1e-3 * gmax and float((g - w).norm()) > 0.10 * float(w.norm()):
            ok = False
            notes.append((k, round(float((g - w).norm() / w.norm()), 4)))
    rel = (num / max(den, 1e-30)) ** 0.5
    if not rel <= 0.02:
        ok = False
        notes.append(("flat gradient", rel))
    print("ok  " if ok else "FAIL", cfg, "rows per video", T, "/", T + b_np["text_embed"].shape[1], "loss", round(float(r_out["loss"]), 4),
          "flat gradient rel", round(rel, 4), notes[:5], flush=True)
    return ok


EVAL = bool(os.environ.get("SOAK_EVAL"))       # the forward alone: no-grad (EMA / evaluation) against training mode, and against the oracle


def eval_case(rng, it):
    """`TemporalAligner.forward` on random shapes, alignability head on: every output of the no-grad forward (which skips the tensors only
    a backward reads: other kernel instantiations) is BIT-identical to the training-mode forward's; in fp32 every output equals the oracle's
    (`tan_ref.forward`) to 2e-4, in bf16 to 3e-2 (cosine logits live in [-1, 1])."""
    from oracle import tan_ref
    from temporalalignnet_amd.tan_model import TemporalAligner
    T = int(rng.choice([8, 16, 31, 47, 48, 56, 64, 64, 65, 79, 96, 128, 129, 200, 256, 270]))
    nmax = int(rng.choice([1, 2, 5, 8, 15, 16, 17, 24, 33]))
    cfg = dict(dtype=str(rng.choice(["fp32", "bf16", "bf16"])), B=int(rng.choice([1, 2, 3, 4, 8, 16])), T=T, nmin=int(rng.integers(1, nmax + 1)), nmax=nmax,
               vpad=int(rng.choice([0, 0, 3, T // 4])), E=int(rng.integers(1, 3)), D=int(rng.integers(3, 4)), seed=int(rng.integers(1, 10000)))
    params = synth.make_params(cfg["seed"], cfg["E"], cfg["D"], True)
    m = TemporalAligner(num_encoder_layers=cfg["E"], num_decoder_layers=cfg["D"], use_alignability_head=1, language_model=None,
                        compute_dtype=cfg["dtype"], random_pos_start=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda()
    b_np = synth.make_batch(cfg["seed"] + 1, B=cfg["B"], T=T, n_min=cfg["nmin"], n_max=cfg["nmax"], video_pad_tail=cfg["vpad"])
    t = train_ref.to_torch_batch(b_np)
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in t.items()}
    with torch.no_grad():
        quiet = m(d["video"], d["text_embed"], d["padding_mask"], d["text_padding_mask"].bool(), None)
    loud = m(d["video"], d["text_embed"], d["padding_mask"], d["text_padding_mask"].bool(), None)
    torch.cuda.synchronize()
    ref = tan_ref.forward({k: torch.from_numpy(v) for k, v in params.items()}, t["video"], t["text_embed"], t["padding_mask"], t["text_padding_mask"].bool(),
                          E=cfg["E"], D=cfg["D"], use_alignability_head=True, random_pos_start=False)
    ok, notes = True, []
    tol = 2e-4 if cfg["dtype"] == "fp32" else 3e-2
    valid_t = ~t["text_padding_mask"].bool()
    for k in loud:
        if not torch.is_tensor(loud[k]):
            continue
        if not torch.equal(quiet[k], loud[k].detach()):
            ok = False
            notes.append((k, "no-grad differs"))
        if k in ref and torch.is_tensor(ref[k]) and ref[k].shape == loud[k].shape:
            a, c = loud[k].detach().float().cpu(), ref[k].detach().float()
            fin = torch.isfinite(c)          # (padded frames of a fully padded ... the reference's own -inf / NaN entries are compared as a pattern)
            if not torch.equal(torch.isfinite(a), fin) and cfg["vpad"] == 0:
                ok = False
                notes.append((k, "finite pattern"))
            err = float((a - c)[fin & torch.isfinite(a)].abs().max()) if bool((fin & torch.isfinite(a)).any()) else 0.0
            if err > tol * max(1.0, float(c[fin].abs().max())):
                ok = False
                notes.append((k, err))
    print("ok  " if ok else "FAIL", cfg, "outputs", sorted(k for k in loud if torch.is_tensor(loud[k]))[:8], notes[:5], flush=True)
    return ok


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.set_num_threads(16)
    bad = nan_cases = stage2 = 0
    for it in range(n):
        if BF16 or EVAL:
            try:
                bad += 0 if (eval_case if EVAL else bf16_case)(rng, it) else 1
            except Exception as e:
                print("EXCEPTION", repr(e)[:300], flush=True)
                bad += 1
            continue
        cot = bool(rng.integers(0, 2))
        T = int(rng.choice([8, 16, 16, 32, 64]))
        nmax = int(rng.integers(1, 25))
        cfg = dict(stage=2 if cot else 1, B=int(rng.choice([1, 2, 2, 3, 4, 6, 8])), T=T, nmin=int(rng.integers(1, nmax + 1)), nmax=nmax,
                   vpad=int(rng.choice([0, 0, 2, T // 4])), E=int(rng.integers(1, 3)), D=int(rng.integers(3, 4) if cot else rng.integers(1, 3)),
                   seed=int(rng.integers(1, 10000)), agree=str(rng.choice(["i", "u", "keep", "keep-joint"])),
                   th=float(rng.choice([0.3, 0.5, 0.8])) if cot else 0.0)
        args = default_args(model="cotrain" if cot else "init", num_encoder_layers=cfg["E"], num_decoder_layers=cfg["D"], lr=1e-3, wd=1e-2,
                            seq_len=T, loss_threshold=cfg["th"], temporal_agreement_type=cfg["agree"], momentum_m=0.99)
        params = synth.make_params(cfg["seed"], cfg["E"], cfg["D"], cot)
        b_np = synth.make_batch(cfg["seed"] + 1, B=cfg["B"], T=T, n_min=cfg["nmin"], n_max=cfg["nmax"])
        if cfg["vpad"]:
            b_np["padding_mask"][::2, -cfg["vpad"]:] = True
        if cot and int((b_np["text_padding_mask"] == 0).sum()) < 3:
            # two real sentences in the whole batch: their z-scores are +-0.707 in both families, the threshold metric is 0 +- rounding
            # for both, and which of them `metric <= quantile` keeps is decided by the last bit (seed 5, configuration 6887: the oracle
            # keeps one, both HIP paths keep both) -- a tie, not a finding
            print("skip", cfg, "(fewer than three real sentences: the stage-2 threshold is a rounding tie)", flush=True)
            continue
        try:
            ref = train_ref.RefTrainer(params, E=cfg["E"], D=cfg["D"], args=args, lr=1e-3, wd=1e-2, m=0.99, random_pos_start=0)
            r_out, _ = ref.step(train_ref.to_torch_batch(b_np))
            model = build_model(args, compute_dtype="fp32", random_pos_start=0)
            load(model.online if cot else model, params)
            if cot:
                model._copy_param()
            tr = Trainer(model.cuda(), args)
            out = tr.step(to_device_batch(b_np))
            torch.cuda.synchronize()
        except Exception as e:
            print("EXCEPTION", cfg, repr(e)[:300], flush=True)
            bad += 1
            continue
        ok, notes = True, []
        stage2 += int(cot)
        nan_cases += int(any(np.isnan(float(v)) for v in r_out.values()))
        for k, v in r_out.items():
            a, c = float(out[k]), float(v)
            if np.isnan(c) != np.isnan(a) or (not np.isnan(c) and abs(a - c) > 1e-3 * max(1.0, abs(c))):
                ok = False
                notes.append((k, a, c))
        if not any(np.isnan(float(v)) for v in r_out.values()):       # (a NaN loss poisons every parameter on both sides alike)
            named = dict(tr.online.named_parameters())
            for k, v in ref.p.items():
                d = (named[k].detach().cpu() - v.detach()).abs()
                # Adam's first step moves every element by lr * sign(g): an element whose gradient is at the f32 noise floor may go the
                # other way (2 lr); anything beyond that, or many of those, is a finding
                if float(d.max()) > 2.05e-3 or float((d > 1e-4).float().mean()) > 0.02:
                    ok = False
                    notes.append((k, float(d.max()), float((d > 1e-4).float().mean())))
        print("ok  " if ok else "FAIL", cfg, {k: round(float(v), 4) for k, v in r_out.items() if k.startswith("loss")}, notes[:4], flush=True)
        bad += 0 if ok else 1
    print("configurations", n, "of them stage 2:", stage2, "with a NaN entry in the oracle's loss dict:", nan_cases, "failures", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
