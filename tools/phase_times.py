"""GPU time of the phases of a training step, untraced: events on the main stream at the Python boundaries (model forward issued,
get_loss issued, backward issued, optimizer issued), averaged over steps.  The host runs ahead, so a phase's event interval is the
GPU's time for it (side streams join the main stream inside each phase).  Tool only."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import synth, loss as L
from temporalalignnet_amd import train as TR
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
args = default_args(model="init")
model = build_model(args, compute_dtype="bf16").cuda()
tr = Trainer(model, args); tr.batches_seen = 1000
b = to_device_batch(synth.make_batch(888, B=int(os.environ.get("B", 128)), T=64, n_min=4, n_max=16))
marks = []
def ev(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
orig_call = type(model).__call__
def model_call(self, *a, **k):
    ev("fwd_begin"); out = orig_call(self, *a, **k); ev("fwd_issued"); return out
type(model).__call__ = model_call
orig_gl = TR.get_loss
def gl(*a, **k):
    out = orig_gl(*a, **k); ev("loss_issued"); return out
TR.get_loss = gl
orig_fb = Trainer.forward_backward
def fb(self, batch):
    out = orig_fb(self, batch); ev("bwd_issued"); return out
Trainer.forward_backward = fb
for _ in range(5): tr.step(b)
torch.cuda.synchronize(); marks.clear()
N = 20
ev("start")
for _ in range(N):
    tr.step(b); ev("step_end")
torch.cuda.synchronize()
import collections
acc = collections.OrderedDict()
for (t0, e0), (t1, e1) in zip(marks, marks[1:]):
    acc[f"{t0}->{t1}"] = acc.get(f"{t0}->{t1}", 0.0) + e0.elapsed_time(e1)
tot = 0
for k, v in acc.items():
    print(f"{k:28s} {v / N * 1e3:8.1f} us/step"); tot += v
print("sum", tot / N * 1e3)
