"""Does the host run ahead of the GPU across steps?  Per-step host time of 40 un-synchronised steps (then one sync).
env: B (batch, default 128), KIND (init | cotrain)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
KIND = os.environ.get("KIND", "init")
args = default_args(model=KIND, **({"loss_threshold": 0.5} if KIND == "cotrain" else {}))
model = build_model(args, compute_dtype="bf16").cuda()
if KIND == "cotrain":
    model._copy_param()
    for p in model.target.parameters():
        p.requires_grad = False
tr = Trainer(model, args, iter_per_epoch=2890, warmup=1000); tr.batches_seen = 1000
b = to_device_batch(synth.make_batch(888, B=int(os.environ.get("B", 128)), T=64, n_min=4, n_max=16))
for _ in range(5): tr.step(b)
torch.cuda.synchronize()
ts = [time.perf_counter()]
for _ in range(40):
    tr.step(b); ts.append(time.perf_counter())
t_issue = ts[-1] - ts[0]
torch.cuda.synchronize(); t_all = time.perf_counter() - ts[0]
d = [(ts[i + 1] - ts[i]) * 1e3 for i in range(40)]
print("host ms per step:", " ".join(f"{x:.1f}" for x in d))
print(f"host issue total {t_issue*1e3:.1f} ms, with final sync {t_all*1e3:.1f} ms -> GPU {t_all*1e3/40:.2f} ms/step, host {t_issue*1e3/40:.2f} ms/step")
