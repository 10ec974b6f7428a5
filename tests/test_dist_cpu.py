"""N>1 path on CPU: two gloo ranks shard the videos, compute the per-rank reference loss gradient (the CPU oracle stands in
for the HIP kernels, which need a GPU), all-reduce ONE flat gradient bucket and apply the 1/world scale -- and must agree
with the mean of the per-rank gradients computed in a single process (SURVEY.md section 8(e) semantics)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _flat_grad(params, batch, lo, hi, E, D):
    from oracle import loss_ref, tan_ref, train_ref
    sub = {k: (v[lo:hi] if isinstance(v, (np.ndarray, list)) else v) for k, v in batch.items()}
    N = int(max(sub["n_per"]))
    for k in ("text_embed", "text_padding_mask", "abs_text_pos"):
        sub[k] = sub[k][:, :N]
    t = train_ref.to_torch_batch(sub)
    p = {k: torch.tensor(v).requires_grad_(True) for k, v in params.items()}
    out = tan_ref.forward(p, t["video"], t["text_embed"], t["padding_mask"], t["text_padding_mask"].bool(), E=E, D=D)
    ld, _ = loss_ref.get_loss(sub, t["video"], t["text_embed"], t["padding_mask"], t["text_padding_mask"], out,
                              loss_ref.default_args(), t["abs_text_pos"])
    ld["loss"].backward()
    return torch.cat([(p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])).flatten() for k in sorted(p)])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from temporalalignnet_amd import dist, synth
    w, r, _ = dist.init_from_env(backend="gloo")
    assert (w, r) == (world, rank) and dist.world_size() == world
    params = synth.make_params(3, 1, 1, False)
    batch = synth.make_batch(4, B=6, T=16, n_min=2, n_max=4)
    lo, hi = dist.shard_range(6, world, rank)
    flat = _flat_grad(params, batch, lo, hi, 1, 1)
    dist.allreduce_sum_(flat)                       # the one collective of the step
    flat *= 1.0 / world                             # what tan_adamw_step's grad_scale applies
    t = dist.max_over_ranks(float(rank + 1), "cpu")
    dist.barrier()
    if rank == 0:
        q.put((flat.numpy(), t))
    tdist.destroy_process_group()


def test_two_rank_gloo_flat_allreduce_equals_mean_of_shard_gradients():
    sys.path.insert(0, ROOT)
    from temporalalignnet_amd import dist, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, tmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params = synth.make_params(3, 1, 1, False)
    batch = synth.make_batch(4, B=6, T=16, n_min=2, n_max=4)
    want = sum(_flat_grad(params, batch, *dist.shard_range(6, 2, r), 1, 1) for r in range(2)) / 2
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-4, atol=1e-6)   # thread-count dependent summation order
    assert tmax == 2.0
