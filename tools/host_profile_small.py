import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
KIND = os.environ.get("KIND", "init")
args = default_args(model=KIND, **({"loss_threshold": 0.5} if KIND == "cotrain" else {}))
model = build_model(args, compute_dtype="bf16").cuda()
if KIND == "cotrain":
    model._copy_param()
tr = Trainer(model, args)
b = to_device_batch(synth.make_batch(888, B=16, T=64, n_min=4, n_max=16))
for _ in range(5):
    tr.step(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.step(b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
st.sort_stats("cumulative").print_stats(30)
