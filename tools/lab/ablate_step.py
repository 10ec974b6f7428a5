"""Which parts of a training step are on its critical path: un-synchronised steps with single pieces switched off (results are
WRONG in those variants: timing only).  Tool only."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch

args = default_args(model="init")
model = build_model(args, compute_dtype="bf16").cuda()
tr = Trainer(model, args, iter_per_epoch=2890, warmup=1000); tr.batches_seen = 1000
b = to_device_batch(synth.make_batch(888, B=int(os.environ.get("B", 128)), T=64, n_min=4, n_max=16))
orig_opt, orig_zero = Trainer.optimizer_step, Trainer.zero_grad


def timeit(tag, n=40):
    for _ in range(6): tr.step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(b)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{tag:32s} {t / n * 1e3:7.3f} ms/step   (host issue {th / n * 1e3:.3f})", flush=True)


def noop_opt(self, grad_scale=1.0, stepped=None):
    self.iteration += 1


for rnd in range(2):
    timeit("baseline")
    Trainer.optimizer_step = noop_opt
    timeit("no optimizer (no repack either)")
    Trainer.zero_grad = lambda self: None
    timeit("no optimizer, no zero_grad")
    Trainer.optimizer_step = orig_opt
    timeit("no zero_grad")
    Trainer.zero_grad = orig_zero
