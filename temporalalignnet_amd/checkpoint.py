"""Checkpoint I/O in the reference's formats: train/main.py:407-484,511-523 and
utils/utils.py:38-57.

A checkpoint is `torch.save`d `{'epoch', 'state_dict', 'best_acc', 'optimizer', 'iteration'}`:
  * `state_dict` keys are the reference's; released checkpoints spell the language model
    `lang_model.*` while the module attribute is `bert` (tan_model.py:38-40 vs main.py:467-469) -- the aligner's
    `_load_from_state_dict` remaps, and so does `normalise_state_dict` here for dicts assembled by hand;
  * `optimizer` is a `torch.optim.AdamW.state_dict()`: two parameter groups in the order optim_policy builds them
    (no-decay first, main.py:354-355), parameters numbered in `named_parameters()` order inside each group, per-parameter
    `step / exp_avg / exp_avg_sq`.  Our optimizer state is two flat f32 buffers (+ a few language-model tensors); the
    functions below convert in both directions, so a run can resume from a reference checkpoint and vice versa.
No arithmetic happens here; nothing in this file needs the GPU.
"""
from __future__ import annotations

import glob
import os

import torch

from .tan_model import TwinTemporalAligner

NO_DECAY_TOKENS = (".ln_", ".bias", ".logit_scale", ".entropy_scale")      # train/main.py:332


def normalise_state_dict(state_dict: dict) -> dict:
    """`module.` (DataParallel) prefixes dropped; nothing else is renamed (lang_model./bert. is handled at load time)."""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def expand_for_cotrain(state_dict: dict) -> dict:
    """A stage-1 ('init') checkpoint loaded into the twin model: every tensor under `online.` and `target.`, language-model
    tensors also at top level (train/main.py:463-469)."""
    out = {f"target.{k}": v for k, v in state_dict.items()}
    out.update({f"online.{k}": v for k, v in state_dict.items()})
    out.update({k: v for k, v in state_dict.items() if "lang_model." in k})
    # checkpoints written by this build (and by the reference's own save) spell the module attribute, `bert.`: the twin's
    # top-level alias of the online language model expects those keys too (superset of main.py:467-469)
    out.update({k: v for k, v in state_dict.items() if k.startswith("bert.")})
    return out


def _is_frozen_word_table(name: str) -> bool:
    """The language model's word-embedding table.  The reference keeps it a plain nn.Embedding (requires_grad=True,
    model/s3d_milnce/s3dg.py:197) that is only ever read under no_grad (word2vec_model.py:84-85): optim_policy therefore lists it
    in the with-decay group as a member that never gets optimizer state.  This build freezes it (requires_grad=False), so it
    has to be put back when numbering parameters the reference's way."""
    return name.endswith(("bert.word_embd.weight", "lang_model.word_embd.weight")) and not name.startswith("target.")


def _trainable_named(model):
    """named_parameters() the reference's optimizer would see (main.py:336-338): requires_grad, plus the word table above."""
    return [(n, p) for n, p in model.named_parameters() if p.requires_grad or _is_frozen_word_table(n)]


def _groups(model, policy="default"):
    """(no_decay, with_decay) lists of (name, param) exactly as optim_policy assigns them (main.py:329-356)."""
    nd, wd = [], []
    for n, p in _trainable_named(model):
        if policy == "bce" and "binary_head" not in n:
            continue
        (nd if any(t in n for t in NO_DECAY_TOKENS) else wd).append((n, p))
    return nd, wd


def _moment_views(trainer):
    """name -> (exp_avg, exp_avg_sq) views for every trainable parameter that has optimizer state."""
    f, st = trainer._ensure_state()
    prefix = "online." if trainer.twin else ""
    out = {}
    for n in f.names:
        o, k, shp = f.off[n]
        if int(st["mode"][o]) == 2:              # never receives a gradient: torch.optim keeps no state for it
            continue
        out[prefix + n] = (st["m"][o:o + k].view(shp), st["v"][o:o + k].view(shp))
    lm_state = (trainer._state or {}).get("lm")
    if lm_state:
        for n, (m, v) in lm_state.items():
            out[prefix + "bert." + n] = (m, v)
    return out


def optimizer_state_dict(trainer) -> dict:
    """Our AdamW state as a `torch.optim.AdamW.state_dict()` (see module doc)."""
    a = trainer.args
    nd, wd = _groups(trainer.model, a.optim_policy)
    moments = _moment_views(trainer)
    state, groups, idx = {}, [], 0
    for members, decay in ((nd, 0.0), (wd, a.wd)):
        ids = []
        for n, _ in members:
            if n in moments and trainer.iteration > 0:
                m, v = moments[n]
                state[idx] = {"step": torch.tensor(float(trainer.iteration)), "exp_avg": m.detach().cpu().clone(),
                              "exp_avg_sq": v.detach().cpu().clone()}
            ids.append(idx)
            idx += 1
        groups.append({"lr": trainer.current_lr(), "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": decay,
                       "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                       "fused": None, "initial_lr": a.lr, "params": ids})
    return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(trainer, opt_state: dict):
    """Inverse of optimizer_state_dict; accepts a reference-written AdamW state (same grouping / numbering rule)."""
    a = trainer.args
    nd, wd = _groups(trainer.model, a.optim_policy)
    order = [n for n, _ in nd] + [n for n, _ in wd]
    sizes = [len(g["params"]) for g in opt_state["param_groups"]]
    if sizes != [len(nd), len(wd)]:
        raise ValueError(f"optimizer state has parameter groups of {sizes}, this model/policy needs {[len(nd), len(wd)]}")
    if trainer._lm_params() and (trainer._state is None or "lm" not in trainer._state):
        trainer._ensure_state()
        trainer._state["lm"] = {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in trainer._lm_params()}
    moments = _moment_views(trainer)
    flat_ids = [i for g in opt_state["param_groups"] for i in g["params"]]
    steps = set()
    for name, idx in zip(order, flat_ids):
        s = opt_state["state"].get(idx)
        if s is None:
            continue
        if name not in moments:
            raise ValueError(f"optimizer state for {name!r}, which never receives a gradient here")
        m, v = moments[name]
        if tuple(s["exp_avg"].shape) != tuple(m.shape):
            raise ValueError(f"{name}: optimizer moment shape {tuple(s['exp_avg'].shape)} != parameter shape {tuple(m.shape)}")
        m.copy_(s["exp_avg"].to(m.device, m.dtype))
        v.copy_(s["exp_avg_sq"].to(v.device, v.dtype))
        steps.add(int(float(s["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused AdamW keeps one bias-correction step")
    if steps:
        trainer.iteration = steps.pop()


def make_state(trainer, epoch: int, best_acc: float) -> dict:
    """The dict train/main.py:515-520 saves.  'iteration' is the reference's BATCH counter args.iteration (starts at 1,
    main.py:281, +1 per batch, main.py:140) -- not the optimizer-step count, which differs once backprop_freq > 1 and which the
    AdamW state carries itself (state[*]['step'])."""
    return {"epoch": epoch, "state_dict": {k: v.detach().cpu().clone() for k, v in trainer.model.state_dict().items()},
            "best_acc": best_acc, "optimizer": optimizer_state_dict(trainer), "iteration": trainer.batches_seen + 1,
            "iteration_kind": "batch"}


def save_checkpoint(state: dict, is_best=0, gap=1, filename="models/checkpoint.pth.tar", keep_all=False):
    """utils/utils.py:38-57: write `filename`, drop `epoch{epoch-gap}.pth.tar` unless keep_all, keep the 5 newest
    `model_best_epoch*.pth.tar`."""
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(state, filename)
    if not keep_all:
        try:
            os.remove(os.path.join(d, f"epoch{state['epoch'] - gap}.pth.tar"))
        except OSError:
            pass
    if is_best:
        past = sorted(glob.glob(os.path.join(d, "model_best_*.pth.tar")), key=lambda x: int("".join(filter(str.isdigit, x))))
        if len(past) >= 5:
            try:
                os.remove(past[0])
            except OSError:
                pass
        torch.save(state, os.path.join(d, f"model_best_epoch{state['epoch']}.pth.tar"))


def _load_state(model, state_dict):
    """strict load, falling back to a reported non-strict one (main.py:415-419,447-456)."""
    state_dict = normalise_state_dict(state_dict)
    try:
        model.load_state_dict(state_dict)
        return [], []
    except RuntimeError:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        return list(missing), list(unexpected)


def load_for_test(model, path):
    """--test (main.py:407-419) -> (epoch, missing_keys, unexpected_keys)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    missing, unexpected = _load_state(model, ckpt["state_dict"])
    return ckpt.get("epoch", -1), missing, unexpected


def load_for_resume(trainer, path):
    """--resume (main.py:437-456) -> dict(start_epoch, best_acc, missing, unexpected); restores iteration + AdamW moments."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    missing, unexpected = _load_state(trainer.model, ckpt["state_dict"])
    load_optimizer_state_dict(trainer, ckpt["optimizer"])            # restores the Adam bias-correction step from the state
    # args.iteration = checkpoint['iteration'] (main.py:444) drives the LR schedule; lr_scheduler.step(args.iteration) right
    # after (main.py:499) makes the FIRST batch after a resume run at lambda(iteration) -- one ahead of the uninterrupted run
    # 'iteration_kind' == 'batch' (written by make_state since round 3) or a reference file: the reference's batch counter,
    # 1 + batches seen.  Files of earlier revisions of this repo stored the optimizer-step count there, without a marker: they are
    # recognised by 'iteration' being equal to the AdamW step of the saved optimizer state (a reference file has 1 + step * freq).
    freq = int(getattr(trainer.args, "backprop_freq", 1) or 1)
    steps = [int(v["step"]) for v in ckpt["optimizer"].get("state", {}).values() if "step" in v]
    legacy_step_count = "iteration_kind" not in ckpt and bool(steps) and int(ckpt["iteration"]) == max(steps) and max(steps) > 0
    if legacy_step_count:
        import warnings
        warnings.warn(f"{path}: no 'iteration_kind' marker and 'iteration' == the AdamW step count ({max(steps)}): read as a legacy "
                      f"file that stored optimizer steps; batches seen = steps x backprop_freq ({freq}, the CURRENT setting -- "
                      "resume with the backprop_freq the file was written with, or the LR schedule position is off)")
        trainer.batches_seen = int(ckpt["iteration"]) * freq
    else:
        trainer.batches_seen = max(int(ckpt["iteration"]) - 1, 0)
    trainer._resume_bump = 1
    return {"start_epoch": ckpt["epoch"] + 1, "best_acc": ckpt["best_acc"], "missing": missing, "unexpected": unexpected}


def load_pretrain(model, path):
    """--pretrain (main.py:458-484): a stage-1 checkpoint into either model; for the twin model the tensors go to both
    streams (unless the file already is a cotrain checkpoint, `_cotrain_` in its name) and `_copy_param()` follows."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = normalise_state_dict(ckpt["state_dict"])
    twin = isinstance(model, TwinTemporalAligner)
    if twin and "_cotrain_" not in os.path.basename(str(path)):
        sd = expand_for_cotrain(sd)
    missing, unexpected = _load_state(model, sd)
    if twin:
        model._copy_param()
    return missing, unexpected
