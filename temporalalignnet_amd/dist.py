"""Data parallelism for the TAN hot path: one process per GPU, videos sharded across ranks, ONE all-reduce of the flat
gradient bucket per step (RCCL over xGMI; `backend="nccl"` IS RCCL on ROCm).

The reference's TAN training path is single-process (train/main.py:256 "not using DDP in our exp"); its only DDP code is
end2end/main_nce.py:142-158,283 (DistributedDataParallel over NCCL).  Semantics here follow SURVEY.md section 8(e): each
rank computes the reference loss on its local B_local videos (local negatives / local batch statistics) and gradients are
averaged -- identical to averaging the per-rank reference gradients.  Parameters that never receive a gradient (`mlp.*`,
unused pos-embeds) sit in the same flat bucket as zeros, so no find_unused_parameters machinery is needed.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# TAN_FORCE_DIST=1: create the process group and run every collective even at world size 1 -- the only way to drive the
# RCCL code path (init, async all-reduce on the side stream, barrier) on a 1-GPU box; results are unchanged.
_FORCE = os.environ.get("TAN_FORCE_DIST") == "1"


def _active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def active():
    """True when collectives must run: more than one rank, or TAN_FORCE_DIST=1."""
    return _active()


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world, rank, local = env_world()
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # TAN_DIST_BACKEND=gloo: the same launcher / rank / barrier path without RCCL (tests: two ranks sharing ONE GPU)
            backend = os.environ.get("TAN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if os.environ.get("TAN_DIST_SHARE_GPU") == "1" and torch.cuda.is_available():
                local = local % torch.cuda.device_count()
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_range(n_items: int, world: int, rank_: int):
    """Contiguous, balanced shard [lo, hi) of n_items videos for this rank (DistributedSampler-style coverage)."""
    base, rem = divmod(n_items, world)
    lo = rank_ * base + min(rank_, rem)
    return lo, lo + base + (1 if rank_ < rem else 0)


def allreduce_sum_(flat_grad: torch.Tensor, async_op: bool = False):
    """Sum the flat gradient bucket over ranks in place (the averaging 1/world is folded into the optimizer kernel)."""
    if not _active():
        return None
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)


def all_gather_cat(x: torch.Tensor) -> torch.Tensor:
    """The ranks' equally-shaped tensors concatenated along dim 0 in rank order (= the order of the videos in the global
    batch); the tensor itself when no collective has to run."""
    if not _active():
        return x
    x = x.contiguous()
    parts = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, x)
    return torch.cat(parts, 0)


def broadcast_(flat: torch.Tensor, src: int = 0):
    if _active():
        dist.broadcast(flat, src=src)


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if _active():
        dist.barrier()
