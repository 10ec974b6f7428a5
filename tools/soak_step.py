"""Randomised soak of the stage-1 training step (not a test of the suite: run by hand on a GPU box, `python tools/soak_step.py [n] [seed]`).

For n random configurations -- batch size, window length, sentences per video, frame padding, random position offset, layer counts --
three PIPELINED two-chain steps (`Trainer.step`, the benchmarked schedule: chains, six-launch loss families, early optimizer launches,
padded sentence slots) are compared with three autograd steps (TAN_STEP_CHAINS=0: forward -> get_loss -> loss.backward() -> one AdamW
launch) from the same initial parameters on the same batches: losses of the three steps and the parameters after the third.  The bound is the
one of tests/test_train_options_gpu.py (the two schedules run the same kernels on the same values up to the order of f32 atomics and
bf16 rounding along another kernel path when a shape falls back).
SOAK_BIG=1 adds B = 48 ... 128; SOAK_STAGE2=1 soaks the co-training step instead: fused glue launches against the torch glue."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch


STAGE2 = bool(os.environ.get("SOAK_STAGE2"))      # stage 2 (co-training): the fused glue launches against the torch glue (TAN_STAGE2_FUSED)


def run(cfg, chains):
    if STAGE2:
        os.environ["TAN_STAGE2_FUSED"] = "1" if chains else "0"
        args = default_args(model="cotrain", num_encoder_layers=cfg["le"], num_decoder_layers=cfg["ld"], lr=1e-3, wd=1e-2, seq_len=cfg["T"],
                            loss_threshold=0.5, temporal_agreement_type=cfg["agree"])
    else:
        os.environ["TAN_STEP_CHAINS"] = "1" if chains else "0"
        args = default_args(model="init", num_encoder_layers=cfg["le"], num_decoder_layers=cfg["ld"], lr=1e-3, wd=1e-2, seq_len=cfg["T"])
    torch.manual_seed(cfg["seed"])
    m = build_model(args, compute_dtype="bf16", random_pos_start=cfg["rps"]).cuda()
    tr = Trainer(m, args, iter_per_epoch=50, warmup=2)
    tr.iteration = 5
    losses = []
    np.random.seed(cfg["seed"])
    for s in range(1 if STAGE2 else 3):       # (stage 2: one step -- its discrete decisions make the second step chaotic in the last bits of the first)
        b = to_device_batch(synth.make_batch(cfg["seed"] * 7 + s, B=cfg["B"], T=cfg["T"], n_min=cfg["nmin"], n_max=cfg["nmax"]))
        if cfg["vpad"]:
            for i in range(0, cfg["B"], 3):
                b["padding_mask"][i, -cfg["vpad"]:] = True
        if s == 0:
            eligible = STAGE2 or tr._chains_eligible(b, tr.fused_loss)
        losses.append(tr.step(b)["loss"])
        if STAGE2 and s == 0:
            eligible = bool(tr.__dict__.get("_last_step_chains"))       # (the two-chain co-training step, or its autograd fallback)
    torch.cuda.synchronize()
    tr.online._ensure_flat()
    return [float(x) for x in losses], tr.online.flat_parameters().clone(), tr.online._flat, eligible


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    bad = 0
    for it in range(n):
        T = int(rng.choice([16, 32, 64, 64, 64, 128]))
        nmax = int(rng.integers(2, 25))
        cfg = dict(B=int(rng.choice([2, 3, 5, 8, 8, 12, 16, 24, 32] + ([48, 64, 96, 128] if os.environ.get("SOAK_BIG") else []))), T=T, nmin=int(rng.integers(1, nmax + 1)), nmax=nmax,
                   vpad=int(rng.choice([0, 0, 3, T // 4])), rps=int(rng.integers(0, 2)), le=int(rng.integers(1, 4)),
                   ld=int(rng.integers(3, 4) if STAGE2 else rng.integers(1, 4)), seed=int(rng.integers(1, 10000)),
                   agree=str(rng.choice(["i", "u", "keep", "keep-joint"])) if STAGE2 else None)
        try:
            l1, p1, f, elig = run(cfg, True)
            l0, p0, _, _ = run(cfg, False)
        except Exception as e:      # a configuration the product rejects is a finding too
            print("EXCEPTION", cfg, repr(e)[:300], flush=True)
            bad += 1
            continue
        if any(np.isnan(l0)):       # the reference's own degenerate cases (stage 2, tiny batches: no selected sentence -> pos_weight = 1/0 - 1)
            same = [np.isnan(a) == np.isnan(c) for a, c in zip(l1, l0)]
            print("ok   (NaN on both paths)" if all(same) else "FAIL (NaN on one path)", cfg, l1, l0, flush=True)
            bad += 0 if all(same) else 1
            continue
        ok = all(np.isfinite(l1)) and all(abs(a - c) <= 2e-3 * max(1.0, abs(c)) for a, c in zip(l1, l0))
        # parameters after three steps (lr 1e-3): AdamW normalises the gradient, so an element whose gradient sits at the noise floor may
        # move by lr in either direction in each step -- the bound is the one of tests/test_fullsize_properties_gpu.py: at most
        # 2 x 3 x lr per element, and a small mean
        d = (p1 - p0).abs()
        dmax, dmean = float(d.max()), float(d.mean())
        if not (np.isfinite(dmax) and dmax <= (2.2e-3 if STAGE2 else 6.5e-3) and dmean <= 2e-4):
            ok = False
        worst = max(worst, dmean)
        print(("ok  " if ok else "FAIL"), cfg, "chains" if elig else "NOT-ELIGIBLE", "losses", [round(x, 4) for x in l1], "vs", [round(x, 4) for x in l0],
              "param diff max", round(dmax, 5), "mean", round(dmean, 7), flush=True)
        bad += 0 if ok else 1
    print("configurations", n, "failures", bad, "worst mean parameter difference", worst)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
