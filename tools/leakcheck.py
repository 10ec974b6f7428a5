"""Finds tensors that accumulate from step to step (allocated-bytes growth, live CUDA tensors by shape)."""
import gc, sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
KIND = os.environ.get("KIND", "init")
args = default_args(model=KIND, **({"loss_threshold": 0.5} if KIND == "cotrain" else {}))
m = build_model(args, compute_dtype="bf16").cuda()
if KIND == "cotrain": m._copy_param()
else: m.random_pos_start = 1
tr = Trainer(m, args, iter_per_epoch=2890, warmup=1000); tr.iteration = 1000
b = to_device_batch(synth.make_batch(888, B=int(os.environ.get("B", 32)), T=64, n_min=4, n_max=16))
def live():
    c = collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda: c[(tuple(o.shape), str(o.dtype))] += 1
        except Exception: pass
    return c
for _ in range(5): tr.step(b)
torch.cuda.synchronize(); gc.collect(); a0 = torch.cuda.memory_allocated(); c0 = live()
for _ in range(10): tr.step(b)
torch.cuda.synchronize(); gc.collect(); a1 = torch.cuda.memory_allocated(); c1 = live()
print("allocated growth per step: %.1f MB" % ((a1 - a0) / 10 / 2**20))
for k, v in (c1 - c0).most_common(12): print(v, k)
torch.cuda.synchronize()
print(torch.cuda.memory_summary(abbreviated=True)[:1500])

def chain(o, depth=0, seen=None):
    seen = seen or set()
    if depth > 6 or id(o) in seen: return
    seen.add(id(o))
    for r in gc.get_referrers(o):
        if r is seen or isinstance(r, type(sys._getframe())) or r is globals(): continue
        t = type(r).__name__
        desc = ""
        if isinstance(r, dict): desc = "keys=" + ",".join(str(k) for k in list(r.keys())[:12])
        elif isinstance(r, (list, tuple)): desc = f"len={len(r)}"
        else: desc = repr(r)[:100]
        print("  " * depth + f"<- {t} {desc}")
        if not isinstance(r, (type, type(sys))):
            chain(r, depth + 1, seen)
cands = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda and tuple(o.shape) == (6, 2048, 512)]
print(len(cands), "candidates")
chain(cands[0])
