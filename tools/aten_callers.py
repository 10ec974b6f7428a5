#!/usr/bin/env python
"""Which Python lines of the product issue ATen tensor ops during one training step?  Every torch function call made from
temporalalignnet_amd/ (or bench glue) is counted through a TorchFunctionMode, keyed by the innermost repo frame.  Tool only.
usage: python tools/aten_callers.py [--stage 2]"""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.overrides import TorchFunctionMode

from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch

stage = 2 if "--stage" in sys.argv else 1
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
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
SKIP = {"empty", "empty_like", "view", "reshape", "permute", "__getitem__", "size", "data_ptr", "is_contiguous", "detach", "numel",
        "stride", "dim", "record_stream", "__get__", "element_size", "storage_offset", "is_cuda", "expand", "transpose", "t", "_version"}
counts = Counter()


class Mode(TorchFunctionMode):
    def __torch_function__(self, func, types, a=(), kw=None):
        name = getattr(func, "__name__", str(func))
        if name not in SKIP:
            fr = next((f for f in reversed(traceback.extract_stack()[:-1]) if "/temporalalignnet_amd/" in f.filename), None)
            if fr is not None:
                counts[(name, f"{os.path.basename(fr.filename)}:{fr.lineno}", fr.line.strip()[:100])] += 1
        return func(*a, **(kw or {}))


with Mode():
    tr.step(batch)
torch.cuda.synchronize()
for (name, where, line), c in sorted(counts.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{c:3d} {name:22s} {where:28s} {line}")
print(sum(counts.values()), "torch calls from the product in one step (excluding views / metadata)")
