import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
t0=time.perf_counter()
for _ in range(5):
    x = torch.empty(600_000_000, dtype=torch.bfloat16, device="cuda"); del x
torch.cuda.synchronize()
print(f"1.2GB torch.empty+del: {(time.perf_counter()-t0)/5*1e3:.2f} ms each")
args = default_args(model="init")
model = build_model(args, compute_dtype="bf16").cuda()
tr = Trainer(model, args)
b = to_device_batch(synth.make_batch(888, B=128, T=64, n_min=4, n_max=16))
for _ in range(3):
    tr.step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host-issue time/step {1e3*(t1-t0)/5:.2f} ms; with final sync {1e3*(t2-t0)/5:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(b)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
