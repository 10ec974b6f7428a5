"""Data-parallel gradient reduction overlapped with backward: per-layer events recorded by
tan_encoder_bwd, bucketed asynchronous all-reduces issued from `Trainer.step`.  The boxes have one GPU, so a second,
IDENTICAL rank is simulated: the real one-rank RCCL all-reduce runs, then the bucket is doubled on the same communication
stream (what summing with an identical peer does).  With grad_scale = 1/2 the step must then equal the single-process step, and
the flat gradient must be exactly twice the single-process one -- which fails if a bucket is missed, reduced twice, or reduced
before the layers it covers have finished writing their gradients (the events are the only ordering between the two)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def one_rank_rccl():
    import torch.distributed as tdist
    if tdist.is_initialized():
        yield
        return
    tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:29543", rank=0, world_size=1)
    try:
        yield
    finally:
        tdist.destroy_process_group()


def _setup(layers, seed=0):
    from temporalalignnet_amd import synth
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    args = default_args(model="init", num_encoder_layers=layers, num_decoder_layers=layers)
    batch = to_device_batch(synth.make_batch(11, B=16, T=64, n_min=4, n_max=16))
    torch.manual_seed(seed)
    model = build_model(args, compute_dtype="bf16", random_pos_start=0).cuda()
    return Trainer, model, args, batch


@pytest.mark.parametrize("bucket_layers", [1, 2, 4])
def test_bucket_ranges_tile_each_stack(bucket_layers):
    Trainer, model, args, batch = _setup(3)
    tr = Trainer(model, args, ddp_bucket_layers=bucket_layers)
    tr.zero_grad()
    for tag, prefix in (("video", "video_temporal_encoder."), ("joint", "joint_temporal_encoder.")):
        buckets = tr._ddp_buckets(tag, 3)
        lo_s, hi_s = tr.online.flat_range(prefix)
        assert [b[2] for b in buckets] == sorted((b[2] for b in buckets), reverse=True)      # last layers first
        assert buckets[-1][2] == 0
        spans = sorted((lo, hi) for lo, hi, _ in buckets)
        assert spans[0][0] == lo_s and spans[-1][1] == hi_s
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))                            # disjoint, no gap


@pytest.mark.parametrize("bucket_layers", [1, 2])
def test_overlapped_bucketed_allreduce_with_a_simulated_identical_peer(one_rank_rccl, monkeypatch, bucket_layers):
    from temporalalignnet_amd import dist
    Trainer, model, args, batch = _setup(3)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    # ---- single process
    tr = Trainer(model, args)
    ld0 = tr.step(batch)
    g0 = tr.online.flat_grad().clone()
    p0 = tr.online._flat.flat.clone()
    # ---- "two ranks"
    model.load_state_dict(state0)
    calls = []

    class _Work:
        def __init__(self, st):
            self.ev = torch.cuda.Event()
            self.ev.record(st)

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    real = dist.allreduce_sum_

    def fake_allreduce(t, async_op=False):
        st = torch.cuda.current_stream()
        w = real(t, async_op=True)                 # the real RCCL collective, one rank
        w.wait()                                   # `st` waits for it ...
        t.mul_(2.0)                                # ... and the identical peer's contribution is added on the same stream
        calls.append((t.data_ptr(), t.numel(), async_op, st.cuda_stream))
        return _Work(st) if async_op else None

    monkeypatch.setattr(dist, "_FORCE", True)
    monkeypatch.setattr(dist, "allreduce_sum_", fake_allreduce)
    monkeypatch.setattr(dist, "world_size", lambda: 2)
    tr2 = Trainer(model, args, ddp_bucket_layers=bucket_layers)
    tr2.ddp_mode = "buckets"                     # (the default is 'flat': the whole gradient in contiguous pieces, tested below)
    assert tr2._chains_eligible(batch, tr2.fused_loss)          # the bucket hook rides the two-chain step (round 5)
    ld1 = tr2.step(batch)
    g1 = tr2.online.flat_grad().clone()
    p1 = tr2.online._flat.flat.clone()
    torch.cuda.synchronize()
    n_buckets = 2 * -(-3 // bucket_layers)
    assert sum(1 for c in calls if c[2]) == n_buckets                  # asynchronous buckets: both stacks
    assert len(calls) > n_buckets                                      # + the trailing synchronous remainder
    comm = tr2._comm_order_stream(g1.device).cuda_stream
    assert all(c[3] == comm for c in calls if c[2])                    # issued under the communication-order stream
    assert sum(c[1] for c in calls) == g1.numel()                      # every element reduced exactly once
    assert abs(ld0["loss"].item() - ld1["loss"].item()) <= 1e-5 * max(1.0, abs(ld0["loss"].item()))
    # f32 atomics in a few reductions make two runs differ in the last bits; a missed / early bucket is a factor-2 error
    assert (g1 - 2 * g0).norm() <= 1e-3 * (2 * g0).norm()
    f = tr2.online._flat
    for n in f.names:                                                  # ... in EVERY parameter tensor, however small
        o, k, _ = f.off[n]
        a, b = g1[o:o + k], 2 * g0[o:o + k]
        assert (a - b).norm() <= 2e-2 * b.norm() + 1e-7, (n, float((a - b).norm()), float(b.norm()))
    # Adam's first step is lr * sign(g) (up to eps): noise-level gradients may flip sign between two runs, nothing else moves
    lr = tr2.current_lr()
    dp = (p1 - p0).abs()
    assert dp.max() <= 2.1 * lr + 1e-7
    assert (dp > 0.1 * lr).float().mean() < 0.01


def test_single_mode_is_one_allreduce_of_the_whole_gradient(one_rank_rccl, monkeypatch):
    """TAN_DDP_MODE=single: literally one RCCL all-reduce of the gradients per step, after backward."""
    from temporalalignnet_amd import dist
    Trainer, model, args, batch = _setup(2)
    calls = []
    real = dist.allreduce_sum_

    def counting(t, async_op=False):
        calls.append((t.numel(), async_op))
        return real(t, async_op=async_op)
    monkeypatch.setattr(dist, "_FORCE", True)
    monkeypatch.setattr(dist, "allreduce_sum_", counting)
    tr = Trainer(model, args)
    assert tr.ddp_mode == "flat"
    tr.ddp_mode = "single"
    ld = tr.step(batch)
    torch.cuda.synchronize()
    assert calls == [(tr.online.flat_grad().numel(), False)] and tr.last_collectives == 1
    assert ld["loss"].item() == ld["loss"].item()


@pytest.mark.parametrize("wire", ["f32", "bf16"])
def test_flat_mode_reduces_every_element_once_in_pieces_behind_the_chains(one_rank_rccl, monkeypatch, wire):
    """The default mode (BASELINE north_star: one all-reduce of the gradients per step) on the two-chain step: the video stack's slice
    is reduced when its chain ends, the joint stack's when that ends, what is left behind the embeddings' backward -- with a simulated
    identical peer (see above) the flat gradient must come out exactly twice the single-process one in every parameter tensor, and the
    parameters (each stack stepped right behind its piece) must equal the single-process step's.  wire = bf16: TAN_DDP_GRAD_DTYPE."""
    from temporalalignnet_amd import dist
    Trainer, model, args, batch = _setup(3)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    tr = Trainer(model, args)
    ld0 = tr.step(batch)
    g0, p0 = tr.online.flat_grad().clone(), tr.online._flat.flat.clone()
    model.load_state_dict(state0)
    calls = []
    real = dist.allreduce_sum_

    def fake_allreduce(t, async_op=False):
        assert not async_op
        real(t)                                    # the real RCCL collective, one rank (the current stream waits for it)
        t.mul_(2.0)                                # the identical peer's contribution
        calls.append((t.numel(), t.dtype, torch.cuda.current_stream().cuda_stream))
        return None
    monkeypatch.setattr(dist, "_FORCE", True)
    monkeypatch.setattr(dist, "allreduce_sum_", fake_allreduce)
    monkeypatch.setattr(dist, "world_size", lambda: 2)
    tr2 = Trainer(model, args)
    tr2.ddp_grad_dtype = wire
    assert tr2.ddp_mode == "flat" and tr2._chains_eligible(batch, tr2.fused_loss)
    ld1 = tr2.step(batch)
    g1, p1 = tr2.online.flat_grad().clone(), tr2.online._flat.flat.clone()
    torch.cuda.synchronize()
    f = tr2.online._flat
    lo_v, hi_v = tr2.online.flat_range("video_temporal_encoder.")
    lo_j, hi_j = tr2.online.flat_range("joint_temporal_encoder.")
    assert [c[0] for c in calls[:2]] == [hi_v - lo_v, hi_j - lo_j]      # the two stacks first, each on the stream of its last dW launches
    assert calls[0][2] != calls[1][2] and 3 <= len(calls) <= 4 == tr2.last_collectives + (4 - len(calls))
    assert sum(c[0] for c in calls) == g1.numel()                        # every element reduced exactly once
    assert all(c[1] == (torch.bfloat16 if wire == "bf16" else torch.float32) for c in calls)
    assert abs(ld0["loss"].item() - ld1["loss"].item()) <= 1e-5 * max(1.0, abs(ld0["loss"].item()))
    tol = 1e-3 if wire == "f32" else 6e-3                                # (bf16: 2^-9 relative rounding per element)
    assert (g1 - 2 * g0).norm() <= tol * (2 * g0).norm()
    for n in f.names:                                                    # ... in EVERY parameter tensor, however small
        o, k, _ = f.off[n]
        a, b = g1[o:o + k], 2 * g0[o:o + k]
        assert (a - b).norm() <= 2e-2 * b.norm() + 1e-7, (n, float((a - b).norm()), float(b.norm()))
    lr = tr2.current_lr()
    dp = (p1 - p0).abs()
    assert dp.max() <= 2.1 * lr + 1e-7
    assert (dp > 0.1 * lr).float().mean() < 0.01
