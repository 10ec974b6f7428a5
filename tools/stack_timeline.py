"""Where a training step's GPU time is: events on the stream each piece is issued on (the two encoder stacks run on two HIP streams),
relative to the step's start, averaged over steps.  The steps are NOT synchronised (the host runs ahead as in bench.py's loop; events are
read after the last step): `step:begin` of step k fires when the main stream has finished step k-1's optimizer launch.
Tool only.  usage: python tools/stack_timeline.py [--stage 2] [--sync]   (--sync: host waits for every step -- the round-3 behaviour)"""
import os
import sys
import threading

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

from temporalalignnet_amd import synth
from temporalalignnet_amd import train as TR
from temporalalignnet_amd.tan_model import TemporalAligner
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch

stage = 2 if "--stage" in sys.argv else 1
args = default_args(model="init" if stage == 1 else "cotrain", loss_threshold=0.0 if stage == 1 else 0.5)
model = build_model(args, compute_dtype="bf16").cuda()
if stage == 2:
    model._copy_param()
tr = Trainer(model, args)
tr.batches_seen = 1000
b = to_device_batch(synth.make_batch(888, B=int(os.environ.get("B", 128)), T=64, n_min=4, n_max=16))
marks, lock = [], threading.Lock()


def ev(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    with lock:
        marks.append((tag, e))


def wrap(cls, name, tagfn):
    orig = getattr(cls, name)

    def f(self, *a, **k):
        t = tagfn(self, *a, **k)
        ev(t + ":begin")
        out = orig(self, *a, **k)
        ev(t + ":end")
        return out
    setattr(cls, name, f)


wrap(TemporalAligner, "_encoder_fwd", lambda self, er, *a, **k: ("ema " if self is not tr.online else "") + "fwd " + er.prefix.split("_")[0])
wrap(TemporalAligner, "_encoder_bwd", lambda self, er, *a, **k: "bwd " + er.prefix.split("_")[0])
if "--boundary" in sys.argv:      # the pieces between the end of the stacks' backward and the next step's stacks (stage 1, pipelined step)
    wrap(TemporalAligner, "_embed_bwd_fused", lambda self, *a, **k: "embed bwd")
    wrap(TemporalAligner, "_embed_fused", lambda self, *a, **k: "embed fwd")
    wrap(Trainer, "early_update", lambda self, which, *a, **k: "adamw " + which)
orig_gl = TR.get_loss


def gl(*a, **k):
    ev("get_loss:begin")
    out = orig_gl(*a, **k)
    ev("get_loss:end")
    return out


TR.get_loss = gl
orig_opt = Trainer.optimizer_step


def opt(self, *a, **k):
    ev("optimizer:begin")
    out = orig_opt(self, *a, **k)
    ev("optimizer:end")
    return out


Trainer.optimizer_step = opt
for _ in range(5):
    tr.step(b)
torch.cuda.synchronize()
N = 20
acc = {}
per_step = []
for _ in range(N):
    marks.clear()
    ev("step:begin")
    tr.step(b)
    ev("step:end")
    if "--sync" in sys.argv:
        torch.cuda.synchronize()
    per_step.append(list(marks))
torch.cuda.synchronize()
for ms in per_step[3:]:              # (the first steps after the warm-up sync are host-bound)
    t0 = ms[0][1]
    for tag, e in ms[1:]:
        acc.setdefault(tag, []).append(t0.elapsed_time(e))
N = len(per_step) - 3
print(f"stage {stage}: GPU time since the step's first event [ms], mean of {N} steps")
for tag, v in sorted(acc.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print(f"{sum(v) / len(v):7.3f}  {tag}")
