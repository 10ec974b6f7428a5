#!/usr/bin/env python
"""Which ATen / runtime launches does one training step still make, and from which Python line?  
usage: python tools/aten_ops.py [--stage 2]"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import ProfilerActivity, profile

from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch

stage = 2 if "--stage" in sys.argv and sys.argv[sys.argv.index("--stage") + 1] == "2" else 1
args = default_args(model="init" if stage == 1 else "cotrain", loss_threshold=0.0 if stage == 1 else 0.5)
torch.manual_seed(0)
model = build_model(args, compute_dtype="bf16").cuda()
if stage == 1:
    model.random_pos_start = 1
else:
    model._copy_param()
tr = Trainer(model, args, iter_per_epoch=2890, warmup=1000)
tr.batches_seen = tr.iteration = 1000
batch = to_device_batch(synth.make_batch(888, B=128, T=64, n_min=4, n_max=16))
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        tr.step(batch)
    torch.cuda.synchronize()
ev = prof.events()
kernels = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kernels[e.name[:90]] += 1
print(f"--- device launches per step (not tal::), {N} steps")
tot = 0
for k, c in kernels.most_common():
    if "tal::" in k or k.startswith("simnce") or "Memset" in k and False:
        continue
    print(f"{c / N:6.1f}  {k}")
    tot += c / N
print(f"{tot:6.1f}  total")
print("--- aten ops that launch something, with the innermost repo frame")
ops = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and len(e.kernels) > 0:
        frame = next((s for s in e.stack if "/temporalalignnet_amd/" in s or "bench.py" in s), "?")
        ops[(e.name, frame.strip()[-110:])] += 1
for (name, frame), c in ops.most_common(60):
    print(f"{c / N:5.1f} {name:28s} {frame}")
