"""CPU restatement of the train() step of train/main.py:33-162 (TEST INFRASTRUCTURE).

Step = zero_grad -> forward (+ EMA forward for cotrain) -> get_loss -> backward -> AdamW ->
EMA update, with the reference's parameter grouping (optim_policy, main.py:330-356) and
learning-rate schedule (main.py:486-499).  CPU autocast/GradScaler are no-ops in the reference
(torch.cuda.amp on CPU), so plain fp32 here.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import loss_ref, tan_ref

NO_DECAY_TOKENS = (".ln_", ".bias", ".logit_scale", ".entropy_scale")


def decay_flag(full_name: str) -> bool:
    """True if AdamW weight decay applies (main.py:332,340-343): substring match on the *full*
    parameter name, so top-level `ln_text_init.weight` decays under model='init' (no leading dot)
    but not under 'cotrain' where it is `online.ln_text_init.weight`."""
    return not any(tok in full_name for tok in NO_DECAY_TOKENS)


def lr_multiplier(iteration, iter_per_epoch, epochs, warmup=1000):
    """main.py:488-494."""
    if iteration < warmup:
        return iteration / warmup
    return 0.5 * (1.0 + math.cos(math.pi * (iteration - warmup) / (epochs * iter_per_epoch - warmup)))


class RefTrainer:
    def __init__(self, params: dict, *, E, D, args, lr=1e-4, wd=1e-5, m=0.999, random_pos_start=None):
        self.args = args
        self.E, self.D, self.m = E, D, m
        self.cotrain = args.model == "cotrain"
        self.p = {k: torch.tensor(v).clone().requires_grad_(True) for k, v in params.items()}
        if self.cotrain:
            self.pt = {k: v.detach().clone() for k, v in self.p.items()}
        prefix = "online." if self.cotrain else ""
        # reference default: random_pos_start=1 for 'init' (tan_model.py:22), 0 for cotrain (main.py:389)
        self.random_pos_start = (not self.cotrain) if random_pos_start is None else random_pos_start
        no_decay = [v for k, v in self.p.items() if not decay_flag(prefix + k)]
        decay = [v for k, v in self.p.items() if decay_flag(prefix + k)]
        self.opt = torch.optim.AdamW([{"params": no_decay, "lr": lr, "weight_decay": 0.0},
                                      {"params": decay, "lr": lr, "weight_decay": wd}], lr=lr, weight_decay=wd)

    def _fwd(self, p, batch, rps):
        return tan_ref.forward(p, batch["video"], batch["text_embed"], batch["padding_mask"],
                               batch["text_padding_mask"].bool(), E=self.E, D=self.D,
                               use_alignability_head=bool(self.args.use_alignability_head),
                               random_pos_start=rps)

    def step(self, batch):
        self.opt.zero_grad()
        logits = self._fwd(self.p, batch, self.random_pos_start)
        if self.cotrain:
            with torch.no_grad():
                ema = self._fwd(self.pt, batch, False)
            logits = {**logits, **{f"ema-{k}": v for k, v in ema.items()}}
        out, aux = loss_ref.get_loss(batch, batch["video"], batch["text_embed"], batch["padding_mask"],
                                     batch["text_padding_mask"], logits, self.args, batch.get("abs_text_pos"))
        out["loss"].backward()
        self.opt.step()
        if self.cotrain:
            with torch.no_grad():
                for k in self.pt:
                    self.pt[k] = self.pt[k] * self.m + self.p[k].detach() * (1.0 - self.m)
        return out, aux


def to_torch_batch(b: dict) -> dict:
    out = dict(b)
    for k in ("video", "text_embed", "text_padding_mask", "abs_text_pos"):
        out[k] = torch.as_tensor(b[k])
    out["padding_mask"] = torch.as_tensor(b["padding_mask"]).bool()
    return out
