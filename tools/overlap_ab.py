import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
args = default_args(model="init")
model = build_model(args, compute_dtype="bf16").cuda()
tr = Trainer(model, args, iter_per_epoch=2890, warmup=1000); tr.batches_seen = 1000
b = to_device_batch(synth.make_batch(888, B=128, T=64, n_min=4, n_max=16))
def run(n):
    for _ in range(5): tr.step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(b)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for ov in (True, False):
        tr.online.overlap_stacks = ov
        print("overlap_stacks", ov, f"{run(40):.3f} ms/step", flush=True)
