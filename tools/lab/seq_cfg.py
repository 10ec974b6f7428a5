"""Lab: bench.run_config for several configurations in ONE process, in a given order (does an earlier configuration slow a later one?).
usage: TAN_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/lab/seq_cfg.py s1 gneg s2 s2"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from temporalalignnet_amd import dist
sys.argv, order = [sys.argv[0], "--no-cpu-baseline"], sys.argv[1:]
a = bench.parse()
world, rank, local = dist.init_from_env()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
for tag in order:
    kw = {"s1": dict(stage=1, bs=128), "flat": dict(stage=1, bs=128, ddp_mode="flat"), "gneg": dict(stage=1, bs=128, gneg=True),
          "s2": dict(stage=2, bs=128), "s2b16": dict(stage=2, bs=16)}[tag]
    r, _ = bench.run_config(a, world, rank, dev, kw["stage"], kw["bs"], 64, 10, 5, 5, ddp_mode=kw.get("ddp_mode"), global_negatives=kw.get("gneg", False))
    print(tag, r["ms_per_step"], flush=True)
