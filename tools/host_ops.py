import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
KIND = os.environ.get("KIND", "cotrain")
args = default_args(model=KIND, **({"loss_threshold": 0.5} if KIND == "cotrain" else {}))
model = build_model(args, compute_dtype="bf16").cuda()
if KIND == "cotrain":
    model._copy_param()
    for p in model.target.parameters(): p.requires_grad = False
tr = Trainer(model, args); tr.batches_seen = 1000
b = to_device_batch(synth.make_batch(888, B=int(os.environ.get("B", 128)), T=64, n_min=4, n_max=16))
for _ in range(4): tr.step(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    for _ in range(3): tr.step(b)
torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=50))
