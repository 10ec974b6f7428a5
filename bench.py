#!/usr/bin/env python
"""Headline benchmark: video-seq/s of the TemporalAlignNet training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|fp32] [--stage 1|2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): E6D6, T=64, bf16, stage-1 ('init': NCE only), B=128 videos per GPU
(train/readme.md:10), N ~ U[4,16] sentences per video, synthetic HTM-370K-shaped features (random S3D-like 1024-d
clip features and 512-d sentence embeddings), random-init weights.  One step = zero_grad + forward + get_loss +
backward + (gradient all-reduce over RCCL when N>1) + fused AdamW, inputs already resident in HBM (features, sentence
embeddings, padding masks and the [B,N,T] timestamp mask derived from the batch's start/end lists: `to_device_batch`; everything
that depends on the model -- and all of get_loss, including its positive masks and the column compaction -- runs every step).  Multi-GPU shards
by video (weak scaling: every rank its own 128 videos), one all-reduce of the flat gradient per step.

Prints ONE JSON line (rank 0): value = whole-job video-seq/s; `roofline` = the dominant kernel (the MFMA GEMM family)
timed live with HIP events on its own stream inside the timed steps; `cpu_baseline` = the CPU oracle (oracle/, a PyTorch
CPU restatement of the reference pinned to reference-generated goldens) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
F32_MFMA_PEAK_TFLOPS = 157.3
GEMM_KIND_NAMES = {0: "gemm_bf16<A:K-contig,B:K-contig> (Linear fwd, dX through W^T copies, same-video similarity blocks)",
                   1: "gemm_bf16<A:K-contig,B:K-strided> (d video features = dlogits x text features)",
                   2: "gemm_bf16<A:K-strided,B:K-contig>", 3: "gemm_bf16<A:K-strided,B:K-strided> (dW)",
                   4: "gemm_f32<kc,kc>", 5: "gemm_f32<kc,ks>", 6: "gemm_f32<ks,kc>", 7: "gemm_f32<ks,ks>",
                   8: "attn_fwd", 9: "attn_bwd", 10: "simnce (logits-free similarity+NCE, fwd stats / bwd dlogits)",
                   11: "row-panel fused MLP, forward (LN2 + c_fc + QuickGELU + c_proj + residual + next LN) and backward (both dX GEMMs + quickgelu' + LN2 backward)",
                   12: "attention branch in one launch per video (in_proj GEMM + 8-head attention + out_proj + residual), forward"}
NKINDS = 13
FAMILY = list(range(8)) + [10, 11, 12]   # every MFMA GEMM pipeline launch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="videos per GPU per step")
    ap.add_argument("--seq-len", type=int, default=64)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2], help="1 = 'init' NCE only; 2 = 'cotrain'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=12, help="timed CPU-oracle steps (~1 s each at the default --cpu-batch: a 10-15 s sample)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--global-negatives", action="store_true", help="row f3: NCE negatives from every rank (W similarity sweeps)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--timer-every", type=int, default=10, help="HIP-event kernel timer samples one timed step in n (the middle one of each n: two of the default 20)")
    ap.add_argument("--timer-stride", type=int, default=4, help="the kernel timer brackets every n-th launch of the MFMA family in the sampled step")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra driver-timed configurations (stage 2, len=256) of the N=1 run")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--settle-s", type=float, default=2.0, help="bound [s] of the extra untimed warm-up that runs until the step time is "
                    "steady (two consecutive steps and two consecutive 4-step blocks within 1 %%); 0 = only --warmup")
    ap.add_argument("--with-lm", action="store_true", help="the step starts from token ids (Word2Vec embedder inside it, train/main.py:55-65)")
    return ap.parse_args()


def cpu_baseline(a, args_ns):
    """The CPU oracle's train step on a bounded sample (same E/D/T/N distribution, fewer videos), host cores of this box."""
    from oracle import train_ref
    from temporalalignnet_amd import synth
    # PyTorch CPU ops stop scaling (and then collapse) far below the 256 hardware threads of the GPU box's host: the
    # same step ran 208 s on 256 threads.  32 threads is what the baseline actually uses; `cores` reports that.
    torch.set_num_threads(min(os.cpu_count(), a.cpu_threads))
    E = D = a.layers
    head = bool(args_ns.use_alignability_head)
    tr = train_ref.RefTrainer(synth.make_params(1, E, D, head, randomize_affine=False), E=E, D=D, args=args_ns, lr=1e-4, wd=1e-5,
                              random_pos_start=False)
    b = train_ref.to_torch_batch(synth.make_batch(888, B=a.cpu_batch, T=a.seq_len, n_min=4, n_max=16))
    t0 = time.perf_counter()
    tr.step(b)                                   # warm-up
    warm = time.perf_counter() - t0
    steps = a.cpu_steps if warm < 20 else 1       # keep the whole leg bounded
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(b)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(a.cpu_batch / dt, 2), "unit": "video-seq/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"CPU oracle (PyTorch CPU fp32 restatement of the reference) E{E}D{D} T={a.seq_len} N~U[4,16] "
                      f"stage-{a.stage} train step on {a.cpu_batch} videos, {steps} timed steps after 1 warm-up "
                      f"({dt:.2f} s/step)"}


def settle_steps(trainer, batch, dev, max_s=None, tol=0.01, block=4):
    """Extra UNTIMED warm-up after the --warmup steps: blocks of `block` steps, each step bracketed by events, until the last two
    steps of a block agree within `tol` AND the block's mean agrees with the previous block's, bounded at --settle-s seconds
    (default 2).  A fresh process on a fresh lease pays allocator growth, lazy code-object loads and the clock ramp in its first
    ~10 steps (profiles/r04_fresh_lease.txt: 150-210 ms, 5.9, 5.1, 4.9, 4.8 ... 4.7 ms); 5 warm-up steps = 25 ms of GPU work end before
    that is over.  With N > 1 ranks every rank must run the same number of steps (they contain collectives): the stop decision is
    the max over ranks of 'not settled yet'."""
    from temporalalignnet_amd import dist
    max_s = SETTLE_S if max_s is None else max_s
    t0, n, prev = time.perf_counter(), 0, None
    while True:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(block + 1)]
        for i in range(block):
            ev[i].record()
            trainer.step(batch)
        ev[block].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(block)]
        n += block
        mean = sum(ms) / block
        ok = abs(ms[-1] - ms[-2]) <= tol * ms[-1] and prev is not None and abs(mean - prev) <= tol * mean
        prev = mean
        more = 0.0 if (ok or time.perf_counter() - t0 > max_s) else 1.0
        if dist.max_over_ranks(more, dev) == 0.0:
            return n


SETTLE_S = 2.0


def traffic_profile(stage, batch_size, seq_len):
    """The committed rocprofv3 PMC passes of this configuration (profiles/README.md), newest round first: `traffic` itself cannot
    be measured inside this process."""
    tag = {(1, 128, 64): "", (2, 128, 64): "stage2_b128_", (1, 32, 256): "cfg4_len256_b32_"}.get((stage, batch_size, seq_len))
    if tag is None:
        return None
    rounds = sorted({f[:3] for f in os.listdir(os.path.join(ROOT, "profiles")) if f[:1] == "r" and f[1:3].isdigit() and f[3:4] == "_"}, reverse=True)
    for rnd in rounds:
        rel = f"profiles/{rnd}_{tag}pmc_traffic.json"
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            fam = json.load(open(path)).get("mfma_gemm_family")
            if fam:
                return {"file": rel, "hbm_bytes_per_launch": round(fam["hbm_read_bytes_per_launch"] + fam["hbm_write_bytes_per_launch"]),
                        "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run"}
    return None


def run_config(a, world, rank, dev, stage, batch_size, seq_len, steps, warmup, timer_every, ddp_mode=None, global_negatives=None,
               with_lm=False, grad_dtype=None):
    """One configuration: W untimed + K timed steps between barriers; returns the result fields (no printing)."""
    from temporalalignnet_amd import _lib, dist, synth
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    args_ns = default_args(model="init" if stage == 1 else "cotrain", num_encoder_layers=a.layers, num_decoder_layers=a.layers,
                           loss_threshold=0.0 if stage == 1 else 0.5, seq_len=seq_len)
    torch.manual_seed(888)
    gneg = a.global_negatives if global_negatives is None else global_negatives
    model = build_model(args_ns, compute_dtype=a.dtype, language_model="word2vec" if with_lm else None).to(dev)
    if stage == 1:
        model.random_pos_start = 1
    trainer = Trainer(model, args_ns, iter_per_epoch=2890, warmup=1000,   # 370k videos / 128
                      global_negatives=gneg)
    if ddp_mode is not None:
        trainer.ddp_mode = ddp_mode
    if grad_dtype is not None:
        trainer.ddp_grad_dtype = grad_dtype
    trainer.time_comm = dist.active()
    trainer.batches_seen = 1000                                          # past warm-up: non-zero learning rate
    trainer.iteration = 1000
    if stage == 2:
        model._copy_param()
    np_batch = synth.make_batch(888 + rank, B=batch_size, T=seq_len, n_min=4, n_max=16)
    batch = to_device_batch(np_batch, device=dev)
    if with_lm:
        # the reference's step starts from TOKENS: Word2Vec embedder forward + backward inside the step (train/main.py:55-65, row
        # f1); [n_b, 32] ids per video, the sentence counts of the synthetic batch
        n_per = [int(n) for n in (1 - np.asarray(np_batch["text_padding_mask"])).sum(1)]
        ids, _ = synth.w2v_tokens(888 + rank, sum(n_per), 66250)
        ids = torch.from_numpy(ids).to(dev)
        batch["token"] = list(torch.split(ids, n_per))

    for _ in range(warmup):                                              # (the first step broadcasts rank 0's parameters)
        trainer.step(batch)
    settle = settle_steps(trainer, batch, dev) if a.settle_s > 0 else 0
    trainer.comm_events.clear()
    L = _lib.lib()
    use_timer = not a.no_kernel_timer
    if use_timer:
        _lib.check(L.tan_prof_enable(1, 1200 * max(steps, 1)), "tan_prof_enable")
        L.tan_prof_stride(a.timer_stride)        # inside the timed steps: every n-th launch of the family (an event pair costs 8-14 us)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampled = 0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]     # step boundaries on the main stream (no host sync)
    for i in range(steps):
        marks[i].record()
        if use_timer:                                  # ~200 timing events cost 1-1.7 ms in the step they bracket: one step in
            every = min(timer_every, steps)            # `timer_every` is sampled, the middle one (the host is ahead of the GPU there)
            on = (i % every == every // 2)
            L.tan_prof_enable(2 if on else 0, 0)
            sampled += on
        loss = trainer.step(batch)
    marks[steps].record()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = dist.max_over_ranks(elapsed, dev)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    srt = sorted(step_ms)
    final_loss = float(loss["loss"].item())
    comm = None
    if dist.active():
        import torch.distributed as tdist
        exposed = [e0.elapsed_time(e1) for e0, e1 in trainer.comm_events]
        alone = trainer.allreduce_alone_ms()
        comm = {"world_size_seen_by_backend": tdist.get_world_size(), "backend": tdist.get_backend(),
                "ddp_mode": "global-negatives (feature all-gather + text-gradient reduce-scatter) + " + trainer.ddp_mode
                if gneg else trainer.ddp_mode,
                "gradient_bytes_per_step": alone[2], "collectives_per_step": alone[1], "collectives_last_step": trainer.last_collectives,
                "gradient_wire_dtype": trainer.ddp_grad_dtype, "two_chain_step": bool(getattr(trainer, "_last_step_chains", False)),
                "allreduce_ms_per_step_alone": round(alone[0], 3),
                "exposed_comm_ms_per_step": round(sum(exposed) / max(len(exposed), 1), 3),
                "exposed_comm_ms_per_step_max_over_ranks": round(dist.max_over_ranks(sum(exposed) / max(len(exposed), 1), dev), 3),
                "note": "alone = the step's gradient collectives back to back on an idle GPU; exposed = compute-stream time between the end "
                        "of backward's enqueue and the last collective's completion (what the overlap did not hide), mean over the timed steps"}

    roof = None
    if use_timer:
        nk = NKINDS
        ms, work, cnt = (C.c_double * nk)(), (C.c_double * nk)(), (C.c_long * nk)()
        L.tan_prof_collect(ms, work, cnt, nk)
        work_all, cnt_all = (C.c_double * nk)(), (C.c_long * nk)()
        L.tan_prof_collect_all(work_all, cnt_all, nk)          # every launch of the sampled step(s), timed or not
        # The timed steps run the video and joint stacks on two concurrent HIP streams, so the per-launch durations above
        # include contention between co-running kernels.  Three extra (untimed) steps with the overlap switched off give the
        # same kernels' stand-alone durations as a second reading.
        iso = None
        online = trainer.online
        if getattr(online, "overlap_stacks", False):
            # the SAME schedule (the two-chain step where the timed steps ran it) with every stream of the step collapsed onto one:
            # `serialize_streams` (engine._run_chains); a step under autograd serialises by switching the stack overlap off
            chained = bool(getattr(trainer, "_last_step_chains", False))
            attr = "serialize_streams" if chained else "overlap_stacks"
            for mod in [online] + ([model.target] if stage == 2 else []):
                setattr(mod, attr, chained)
            trainer.step(batch)
            torch.cuda.synchronize()
            L.tan_prof_stride(1)                   # the untimed extra steps bracket every launch
            _lib.check(L.tan_prof_enable(1, 1200 * 3), "tan_prof_enable")
            for _ in range(3):
                trainer.step(batch)
            torch.cuda.synchronize()
            ms2, work2, cnt2 = (C.c_double * nk)(), (C.c_double * nk)(), (C.c_long * nk)()
            L.tan_prof_collect(ms2, work2, cnt2, nk)
            fam = [k for k in FAMILY if cnt2[k] > 0]
            t2, w2, c2 = sum(ms2[k] for k in fam), sum(work2[k] for k in fam), sum(cnt2[k] for k in fam)
            iso = {"achieved": round(w2 / (t2 * 1e-3) / 1e12, 1), "avg_launch_us": round(t2 * 1e3 / c2, 2),
                   "gemm_ms_per_step": round(t2 / 3, 3), "two_chain_step": bool(getattr(trainer, "_last_step_chains", False)),
                   "note": "the same step schedule and kernels with every stream of the step collapsed onto one (3 extra steps)"}
            for mod in [online] + ([model.target] if stage == 2 else []):
                setattr(mod, attr, not chained)
        L.tan_prof_enable(0, 0)
        stride = max(1, a.timer_stride)
        # every stride-th launch was bracketed: a kind's time per step = its sampled time scaled by (all its work / its sampled work)
        for k in range(nk):
            if cnt[k] > 0 and work[k] > 0:
                ms[k] *= work_all[k] / work[k]
                work[k], cnt[k] = work_all[k], cnt_all[k]
        kinds = [{"kernel": GEMM_KIND_NAMES[k], "ms_per_step": ms[k] / sampled, "launches_per_step": round(cnt[k] / sampled, 1),
                  "tflops": (work[k] / (ms[k] * 1e-3) / 1e12) if ms[k] > 0 else 0.0} for k in range(nk) if cnt[k] > 0]
        gemm = [k for k in FAMILY if cnt[k] > 0]      # every launch of the MFMA GEMM pipeline
        if gemm:
            peak = BF16_MFMA_PEAK_TFLOPS if a.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
            tms, twork, tcnt = sum(ms[k] for k in gemm), sum(work[k] for k in gemm), sum(cnt[k] for k in gemm)
            ach = twork / (tms * 1e-3) / 1e12
            # `traffic` (HBM bytes per launch) needs rocprofv3 PMC passes, which cannot run inside this process: null here; the
            # committed passes of the same command are referenced separately (profiles/README.md), never passed off as in-run data
            prof = traffic_profile(stage, batch_size, seq_len) if (a.dtype == "bf16" and not with_lm) else None
            roof = {"bound": "mfma", "kernel": "tal::gemm_glds_kernel + tal::gemm_dw256_kernel + tal::simnce_res_kernel (direct-to-LDS MFMA pipelines, all operand layouts) + tal::mlp_panel_kernel (row-panel fused MLP, forward and backward) + tal::attnblk_fwd_kernel (attention branch per video)",
                    "achieved": round(ach, 1),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None, "traffic_profile": prof,
                    "avg_launch_us": round(tms * 1e3 / tcnt, 2), "launches_per_step": round(tcnt / sampled, 1),
                    "gemm_ms_per_step": round(tms / sampled, 3), "algorithmic_gflop_per_step": round(twork / sampled / 1e9, 1),
                    "step_frac": round(twork / sampled / (elapsed / steps) / 1e12 / peak, 4),
                    "timer": f"HIP events on each launch's own stream, {sampled} of the {steps} timed steps (one in {min(timer_every, steps)}), "
                             f"every {stride}-th launch OF EACH KIND in it (a kind's time = its sampled time x all its work / its sampled work; `isolated` brackets every launch)",
                    "concurrency": "2 HIP streams (video || joint stack): durations include co-running kernels",
                    "isolated": iso,
                    "by_kernel": [{**x, "ms_per_step": round(x["ms_per_step"], 3), "tflops": round(x["tflops"], 1)} for x in kinds]}
    res = {"value": round(batch_size * world * steps / elapsed, 1), "unit": "video-seq/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "dtype": a.dtype,
           # GPU time between consecutive step boundaries (events on the main stream; the step that carries the ~200 kernel-timer events
           # is the max), and the extra untimed warm-up the steady-state rule below asked for
           "step_ms_min": round(srt[0], 3), "step_ms_p50": round(srt[len(srt) // 2], 3), "step_ms_max": round(srt[-1], 3),
           "first_timed_step_ms": round(step_ms[0], 3), "settle_steps": settle,
           "config": {"workload": f"E{a.layers}D{a.layers} len={seq_len} {a.dtype} stage-{stage} "
                                  f"({'init: multi-positive NCE only' if stage == 1 else 'cotrain: EMA + alignability + NCE'}) "
                                  f"train step (fwd+loss+bwd+AdamW), synthetic HTM-370K-shaped features, N~U[4,16] sentences/video",
                      "global_batch": batch_size * world, "per_gpu_batch": batch_size, "seq_len": seq_len,
                      "parallelism": f"dp{world}" + ("+global-negatives" if gneg else ""),
                      "final_loss": round(final_loss, 4)},
           "roofline": roof}
    if with_lm:
        res["config"]["workload"] += "; the step starts from token ids: Word2Vec sentence embedder forward + backward + AdamW inside it (train/main.py:55-65)"
    if comm is not None:
        res["comm"] = comm
    del trainer, model, batch
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res, args_ns


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (the driver's own launch line, same flags)
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"),
                                   os.path.abspath(__file__)] + sys.argv[1:])
    from temporalalignnet_amd import dist
    world, rank, local = dist.init_from_env()
    if a.gpus != world and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); reporting n_gpus={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    if os.environ.get("TAN_DIST_SHARE_GPU") == "1":       # tests: N ranks on fewer GPUs (gloo: RCCL refuses two ranks on one device)
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    global SETTLE_S
    SETTLE_S = a.settle_s
    head, args_ns = run_config(a, world, rank, dev, a.stage, a.batch, a.seq_len, a.steps, a.warmup, a.timer_every, with_lm=a.with_lm)
    extra = []
    headline = (a.stage, a.batch, a.seq_len) == (1, 128, 64) and not a.with_lm
    if world == 1 and not dist.active() and headline and not a.no_extra and a.dtype == "bf16":
        # BASELINE configs[2] (stage-2 co-training, here on one GPU at the per-GPU batch) and configs[3] (len=256) are
        # driver-timed too, each with its own roofline; shorter runs (the headline keeps the driver's K / W).  VERDICT r2: the
        # reference's step also runs the sentence embedder (row f1): a fourth entry starts from token ids.
        for st, bs, sl, lm, tag in ((2, 128, 64, False, "configs[2] on 1 GPU: E6D6 len=64 stage-2 co-training, B=128"),
                                    (1, 32, 256, False, "configs[3]: len=256 (joint L=272), B=32"),
                                    (1, 128, 64, True, "configs[1] with the language model in the step (tokens -> Word2Vec -> aligner)"),
                                    (1, 16, 64, False, "configs[1] at B_local=16 (SURVEY 8(d) config 3's small-batch point, stage 1)"),
                                    (2, 16, 64, False, "configs[2]: stage-2 co-training at B_local=16 (global 128 at 8 GPUs)")):
            r, _ = run_config(a, world, rank, dev, st, bs, sl, a.extra_steps, 3, 5, with_lm=lm)
            r["name"] = tag
            extra.append(r)
    elif dist.active() and headline and not a.no_extra and a.dtype == "bf16" and not a.global_negatives:
        # Multi-GPU (or TAN_FORCE_DIST=1): the variants the first hardware scaling run has to decide between, in the same run --
        # the other gradient-reduction mode, global negatives, and BASELINE configs[2] (stage-2 co-training) at B_local 128 / 16
        # (SURVEY 8(d) config 3).  The headline line keeps the default mode; n_gpus == 1 without TAN_FORCE_DIST prints none of this.
        cur = os.environ.get("TAN_DDP_MODE", "flat")
        others = [m for m in ("flat", "buckets", "single") if m != cur]
        variants = [(dict(stage=1, bs=128, ddp_mode=m), f"configs[1], gradient reduction mode '{m}' (TAN_DDP_MODE)") for m in others]
        variants += [(dict(stage=1, bs=128, grad_dtype="bf16"), "configs[1], gradient on the wire as bf16 (TAN_DDP_GRAD_DTYPE=bf16: 80 MB per step)"),
                     (dict(stage=1, bs=16), "configs[1] at B_local=16 (SURVEY 8(d) config 3's small-batch point, stage 1)"),
                     (dict(stage=1, bs=128, gneg=True), "configs[1] with global negatives (row f3)"),
                     (dict(stage=2, bs=128), "configs[2]: stage-2 co-training, B_local=128"),
                     (dict(stage=2, bs=16), "configs[2]: stage-2 co-training, B_local=16 (global 128 at 8 GPUs)")]
        for kw, tag in variants:
            r, _ = run_config(a, world, rank, dev, kw["stage"], kw["bs"], 64, a.extra_steps, 5, 5, ddp_mode=kw.get("ddp_mode"),
                              global_negatives=kw.get("gneg", False), grad_dtype=kw.get("grad_dtype"))
            r["name"] = tag
            extra.append(r)
    if rank == 0:
        out = {"metric": "video-seq/sec (len=64, E6D6) at 1/2/4/8 MI355X; HTM-Align ROC-AUC parity", "value": head["value"],
               "unit": head["unit"], "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"], "ms_per_step": head["ms_per_step"],
               "step_ms_min": head["step_ms_min"], "step_ms_p50": head["step_ms_p50"], "step_ms_max": head["step_ms_max"],
               "first_timed_step_ms": head["first_timed_step_ms"], "settle_steps": head["settle_steps"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": head["config"], "roofline": head["roofline"]}
        if "comm" in head:
            out["comm"] = head["comm"]
        if extra:
            out["extra"] = extra
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a, args_ns)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to the C stdout buffer, which a pipe only flushes at exit: flush it now so that the
        # JSON line is the LAST line on stdout
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
