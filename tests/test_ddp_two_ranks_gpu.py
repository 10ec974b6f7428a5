"""Two REAL ranks through `Trainer.step` on one MI355X: both processes use cuda:0 and the gloo backend (RCCL refuses two ranks
on one device; gloo all-reduces CUDA tensors through pinned host memory), so everything above the transport is the production
path -- per-rank batches, the per-layer events recorded inside tan_encoder_bwd, the bucketed asynchronous all-reduces issued
under the communication-order stream, the trailing remainder, the 1/world scale in the AdamW kernel.  Checked against ONE
process that runs the two batches one after the other into the same flat gradient (sum) and steps with grad_scale = 1/2
(averaged per-rank gradients of the local-batch loss)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
B, T = 8, 32


def _setup(seed_batch, kind, layers, bucket):
    from temporalalignnet_amd import synth
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    args = default_args(model=kind, num_encoder_layers=layers, num_decoder_layers=layers,
                        loss_threshold=0.5 if kind == "cotrain" else 0.0)
    torch.manual_seed(0)
    model = build_model(args, compute_dtype="bf16", random_pos_start=0).cuda()
    if kind == "cotrain":
        model._copy_param()
    tr = Trainer(model, args, ddp_bucket_layers=bucket)
    tr.iteration = 2000                                      # past warm-up
    batch = to_device_batch(synth.make_batch(seed_batch, B=B, T=T, n_min=3, n_max=7))
    return tr, batch


def _worker(rank, world, port, out_dir, kind, layers, bucket, backend="gloo", mode="buckets"):
    sys.path.insert(0, ROOT)
    local = rank if backend == "nccl" else 0                # RCCL: one GPU per rank; gloo: both ranks share cuda:0
    mode, _, wire = mode.partition("+")                      # "flat+bf16": TAN_DDP_GRAD_DTYPE
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local),
                      TAN_DDP_MODE=mode, TAN_DDP_GRAD_DTYPE=wire or "f32", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as tdist
    from temporalalignnet_amd import dist
    torch.cuda.set_device(local)
    w, r, _ = dist.init_from_env(backend=backend)
    assert (w, r) == (world, rank) and dist.active()
    torch.manual_seed(1234 + rank)                          # ranks start from DIFFERENT parameters: Trainer.step must broadcast rank 0's
    tr, batch = _setup(100 + rank, kind, layers, bucket)
    if rank == 1:
        with torch.no_grad():
            tr.online.flat_parameters().mul_(1.5)
    ld = tr.step(batch)
    torch.cuda.synchronize()
    torch.save({"loss": ld["loss"].item(), "grad": tr.online.flat_grad().cpu(), "param": tr.online._flat.flat.cpu()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    tdist.destroy_process_group()


def _two_gpus():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


@pytest.mark.parametrize("kind,layers,bucket,backend,mode", [
    # stage 1 ('init') runs the two-chain step in every mode (round 5): bucket hooks behind the chains' layer events, the flat gradient in
    # pieces with each stack stepped behind its piece, the single call, and the bf16 wire; stage 2 the autograd step
    ("init", 2, 1, "gloo", "buckets"), ("cotrain", 3, 2, "gloo", "buckets"), ("init", 2, 1, "gloo", "flat"), ("init", 2, 1, "gloo", "single"),
    ("init", 2, 2, "gloo", "flat+bf16"), ("cotrain", 3, 2, "gloo", "flat"),
    # the same through RCCL over xGMI, one GPU per rank -- runs wherever the box has two GPUs (self-skips on the 1-GPU pool)
    ("init", 2, 1, "nccl", "buckets"), ("cotrain", 3, 2, "nccl", "buckets"), ("cotrain", 3, 2, "nccl", "flat")])
def test_two_ranks_equal_one_process_with_both_batches(tmp_path, kind, layers, bucket, backend, mode):
    if backend == "nccl" and not _two_gpus():
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    ctx = mp.get_context("spawn")
    port = 29600 + (os.getpid() + layers + 7 * len(mode) + 13 * len(backend)) % 1000
    wire_bf16 = mode.endswith("+bf16")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), kind, layers, bucket, backend, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    # both ranks hold the same reduced gradient and the same parameters after the step
    assert torch.equal(r0["grad"], r1["grad"])
    assert torch.equal(r0["param"], r1["param"])
    # ---- one process, both batches
    tr, b0 = _setup(100, kind, layers, bucket)
    _, b1 = _setup(101, kind, layers, bucket)
    tr.zero_grad()
    l0 = tr.forward_backward(b0)["loss"].item()
    l1 = tr.forward_backward(b1)["loss"].item()            # gradients accumulate in the flat buffer
    g = tr.online.flat_grad().clone().cpu()
    tr.optimizer_step(grad_scale=0.5)
    torch.cuda.synchronize()
    assert abs(l0 - r0["loss"]) <= 1e-4 * max(1.0, abs(l0)) and abs(l1 - r1["loss"]) <= 1e-4 * max(1.0, abs(l1))
    f = tr.online._flat
    for n in f.names:                                        # every parameter tensor: summed over the two ranks, once
        o, k, _ = f.off[n]
        a, b = r0["grad"][o:o + k], g[o:o + k]
        assert (a - b).norm() <= (3e-2 if wire_bf16 else 2e-2) * b.norm() + 1e-7, (n, float((a - b).norm()), float(b.norm()))
    lr = tr.current_lr()
    dp = (r0["param"] - f.flat.cpu()).abs()
    # Adam with zero moments at step 2001: |update| = lr * 0.1 / sqrt(0.001 / (1 - 0.999**2001)) = 2.94 lr, so a noise-level gradient
    # whose sign differs between the two runs moves a parameter by up to 5.9 lr; nothing else may differ
    assert dp.max() <= 6.0 * lr + 1e-7
    assert (dp > 0.1 * lr).float().mean() < 0.01


# ---------------------------------------------------------------------------------------------------------------------
# Row f3: negatives from every rank.  The reference semantics is simply "the loss at B = W * B_local on one device"
# (tan_model.py:118,138), so two ranks holding the halves of a batch must reproduce ONE process on the whole batch:
# loss = sum of the rank losses, gradient = sum of the rank gradients (grad_scale 1).
GB = 12


def _slice(batch, lo, hi):
    import numpy as np
    return {k: (v[lo:hi] if isinstance(v, (np.ndarray, list)) else v) for k, v in batch.items()}


def _gn_setup(global_negatives, kind="init"):
    from temporalalignnet_amd import synth
    from temporalalignnet_amd.train import Trainer, build_model, default_args
    extra = dict(loss_threshold=0.5) if kind == "cotrain" else {}
    args = default_args(model=kind, num_encoder_layers=2, num_decoder_layers=3 if kind == "cotrain" else 2, **extra)
    torch.manual_seed(0)
    model = build_model(args, compute_dtype="bf16", random_pos_start=0).cuda()
    if kind == "cotrain":
        model._copy_param()
        for p_ in model.target.parameters():
            p_.requires_grad = False
    tr = Trainer(model, args, global_negatives=global_negatives, ddp_bucket_layers=1)
    tr.iteration = 2000
    return tr, synth.make_batch(300, B=GB, T=32, n_min=3, n_max=7)


def _gn_worker(rank, world, port, out_dir, kind="init"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as tdist
    from temporalalignnet_amd import dist
    from temporalalignnet_amd.train import to_device_batch
    torch.cuda.set_device(0)
    dist.init_from_env(backend="gloo")
    tr, full = _gn_setup(True, kind)
    lo, hi = dist.shard_range(GB, world, rank)
    ld = tr.step(to_device_batch(_slice(full, lo, hi)))
    torch.cuda.synchronize()
    torch.save({"loss": ld["loss"].item(), "grad": tr.online.flat_grad().cpu()}, os.path.join(out_dir, f"gn{rank}.pt"))
    dist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("kind", ["init", "cotrain"])
def test_two_ranks_with_global_negatives_equal_one_process_on_the_whole_batch(tmp_path, kind):
    """Stage 1: the NCE over every rank's sentences.  Stage 2 (cotrain, loss_threshold, alignability head) adds the BATCH statistics
    of loss.py:191-194,281-286,315-320 -- quantiles, z-scores, medians -- which in this mode run over the sentences of all ranks."""
    from temporalalignnet_amd.train import to_device_batch
    ctx = mp.get_context("spawn")
    port = 29700 + os.getpid() % 1000 + (50 if kind == "cotrain" else 0)
    procs = [ctx.Process(target=_gn_worker, args=(r, 2, port, str(tmp_path), kind)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    r0, r1 = (torch.load(tmp_path / f"gn{r}.pt") for r in range(2))
    assert torch.equal(r0["grad"], r1["grad"])
    tr, full = _gn_setup(False, kind)                         # local negatives on the WHOLE batch = the global semantics
    tr.zero_grad()
    loss = tr.forward_backward(to_device_batch(full))["loss"].item()
    g = tr.online.flat_grad().cpu()
    assert abs((r0["loss"] + r1["loss"]) - loss) <= 2e-3 * max(1.0, abs(loss)), (r0["loss"], r1["loss"], loss)
    f = tr.online._flat
    for n in f.names:
        o, k, _ = f.off[n]
        a, b = r0["grad"][o:o + k], g[o:o + k]
        assert (a - b).norm() <= 4e-2 * b.norm() + 1e-6, (n, float((a - b).norm()), float(b.norm()))

