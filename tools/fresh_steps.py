"""Per-step times of the headline configuration in a FRESH process: step-boundary events on the main stream,
no host sync inside the loop; prints the first N step times so that what the first steps pay is visible."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
t_start = time.perf_counter()
dev = torch.device("cuda", 0)
args_ns = default_args(model="init", num_encoder_layers=6, num_decoder_layers=6, loss_threshold=0.0, seq_len=64)
torch.manual_seed(888)
model = build_model(args_ns, compute_dtype="bf16", language_model=None).to(dev)
model.random_pos_start = 1
tr = Trainer(model, args_ns, iter_per_epoch=2890, warmup=1000)
tr.batches_seen = 1000; tr.iteration = 1000
batch = to_device_batch(synth.make_batch(888, B=128, T=64, n_min=4, n_max=16), device=dev)
torch.cuda.synchronize()
t_setup = time.perf_counter() - t_start
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
for i in range(n):
    ev[i].record()
    h0 = time.perf_counter()
    tr.step(batch)
    host.append((time.perf_counter() - h0) * 1e3)
ev[n].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(json.dumps({"setup_s": round(t_setup, 2), "gpu_ms": [round(x, 3) for x in ms], "host_ms": [round(x, 3) for x in host]}))
