"""Lists the distinct tan_gemm calls one training step makes from Python (shape, layouts, batch, K slices).  Tool only."""
import sys, os, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import synth, ops
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
KIND = os.environ.get("KIND", "init")
args = default_args(model=KIND)
model = build_model(args, compute_dtype="bf16").cuda()
tr = Trainer(model, args); tr.batches_seen = 1000
b = to_device_batch(synth.make_batch(888, B=int(os.environ.get("B", 128)), T=64, n_min=4, n_max=16))
for _ in range(3): tr.step(b)
torch.cuda.synchronize()
seen = collections.Counter()
orig = ops.gemm
def traced(A, B, C, **kw):
    key = tuple(sorted((k, v) for k, v in kw.items() if isinstance(v, (int, bool)))) + (("out", str(C.dtype)),)
    seen[key] += 1
    return orig(A, B, C, **kw)
ops.gemm = traced
import temporalalignnet_amd.tan_model as tm, temporalalignnet_amd.loss as ls
tr.step(b); torch.cuda.synchronize()
for k, n in seen.items(): print(n, dict(k))
