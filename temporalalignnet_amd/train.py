"""Training-step driver: the MI355X counterpart of train/main.py:train() (reference lines 33-162).

One step = zero_grad -> forward (-> EMA forward for 'cotrain') -> get_loss -> backward -> [all-reduce of the flat
gradient] -> AdamW on both parameter groups -> EMA update, in that order (main.py:81-122).  The optimizer is ONE fused
HIP launch over the flat parameter buffer (tan_adamw_step) that also refreshes the bf16 shadow weights and the EMA
target; parameter grouping follows optim_policy (main.py:330-356) including its substring rule on full parameter names.

The reference's GradScaler is not reproduced: throughput mode is bf16 (f32 exponent range), so no loss scaling.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import types

import numpy as np
import torch

from . import _lib, dist, ops
from .loss import get_loss, get_mask_from_time
from .tan_model import TemporalAligner, TwinTemporalAligner, _vp


def default_args(**kw):
    """The flags of train/config.py:6-53 that the hot path reads, with the reference's defaults."""
    a = dict(model="init", sim="cos", learn_agreement=0, temporal_agreement_type="keep", loss_threshold=0.0,
             use_alignability_head=0, optim_policy="default", seq_len=64, lr=1e-4, wd=1e-5, clip_grad=0.0,
             momentum_m=0.999, num_encoder_layers=6, num_decoder_layers=6, epochs=10, backprop_freq=1)
    a.update(kw)
    if a["model"] == "cotrain":            # train/main.py:361-363
        a["learn_agreement"] = 1
        a["use_alignability_head"] = 1
    return types.SimpleNamespace(**a)


def build_model(args, compute_dtype="fp32", language_model=None, **kw):
    """Model construction of train/main.py:370-390."""
    common = dict(num_encoder_layers=args.num_encoder_layers, num_decoder_layers=args.num_decoder_layers, sim=args.sim,
                  language_model=language_model, use_alignability_head=args.use_alignability_head,
                  compute_dtype=compute_dtype, **kw)
    if args.model == "init":
        return TemporalAligner(**common)
    common.setdefault("random_pos_start", 0)       # train/main.py:389
    return TwinTemporalAligner(m=args.momentum_m, **common)


def lr_multiplier(iteration, iter_per_epoch, epochs, warmup=1000):
    """Linear warm-up then cosine (train/main.py:488-494)."""
    if iteration < warmup:
        return iteration / warmup
    return 0.5 * (1.0 + math.cos(math.pi * (iteration - warmup) / (epochs * iter_per_epoch - warmup)))


def to_device_batch(batch: dict, device="cuda") -> dict:
    """numpy batch (temporalalignnet_amd.synth.make_batch / loader collate schema) -> device tensors, plus the
    [B,N,T] timestamp mask of train/main.py:71 computed once."""
    out = dict(batch)
    for k in ("video", "text_embed", "text_padding_mask", "abs_text_pos"):
        if k in batch and batch[k] is not None:
            out[k] = torch.as_tensor(batch[k]).to(device, non_blocking=True)
    out["padding_mask"] = torch.as_tensor(batch["padding_mask"]).bool().to(device, non_blocking=True)
    if "text_padding_mask" in out and torch.is_tensor(out["text_padding_mask"]):
        out["_text_pad_bool"] = out["text_padding_mask"].bool()          # main.py:86 passes .bool() every step: converted once here
    T, N = out["video"].shape[1], out["text_embed"].shape[1]
    out["_tgt_raw"], _, _ = get_mask_from_time(batch["start"], batch["end"], T, N, device=device)
    if "text_padding_mask" in batch and not torch.is_tensor(batch["text_padding_mask"]):
        out["n_text"] = int(batch["text_padding_mask"].size - np.count_nonzero(batch["text_padding_mask"]))   # host count, no sync
    return out


def pad_sequence_by_last(sequences):
    """Stack ragged [n_b, C] tensors to [B, max n_b, C], padding each with copies of its last row
    (data/loader_htm.py:13-23, used at train/main.py:61)."""
    n_max = max(s.shape[0] for s in sequences)
    return torch.stack([torch.cat([s, s[-1:].expand(n_max - s.shape[0], -1)], 0) if s.shape[0] < n_max else s
                        for s in sequences], 0)


def embed_sentences(model, token_list):
    """train/main.py:55-65: concatenate every video's [n_b, 32] token ids, run the language model, split per video and
    pad by repeating the last sentence (pad_sequence_by_last) -> (text_embed [B, N, 512], text_padding_mask [B, N] float 0/1).
    The reference does the split / pad / mask with a Python loop over the videos (~7 tiny kernels per video forward and backward,
    ~900 launches at B=128: 6 ms of host time per step); the sentence counts are known on the host, so here it is ONE row gather
    (index_select; its backward one index_add) through a host-built index: row (b, k) <- sentence offset_b + min(k, n_b - 1)."""
    n_per = np.asarray([t.shape[0] for t in token_list], dtype=np.int64)
    flat = torch.cat(list(token_list), 0).long()
    dev = flat.device
    emb = model.lang_model(input_ids=flat, attention_mask=flat != 0)["pooler_output"]
    N = int(n_per.max())
    k = np.arange(N, dtype=np.int64)[None, :]
    base = np.concatenate([[0], np.cumsum(n_per)[:-1]])[:, None]
    idx = torch.from_numpy((base + np.minimum(k, n_per[:, None] - 1)).reshape(-1)).to(dev, non_blocking=True)
    pad = torch.from_numpy((k >= n_per[:, None]).astype(np.float32)).to(dev, non_blocking=True)
    text_embed = emb.index_select(0, idx).view(len(n_per), N, emb.shape[-1])
    return text_embed, pad


def uncovered_ranges(done, total):
    """[lo, hi) pieces of [0, total) that none of the `done` ranges covers; overlapping `done` ranges are an error (a slice of
    the gradient would be summed over ranks twice)."""
    out, pos = [], 0
    for lo, hi in sorted(done) + [(total, total)]:
        if lo < pos:
            raise ValueError(f"overlapping gradient buckets at {lo} < {pos}")
        if lo > pos:
            out.append((pos, lo))
        pos = max(pos, hi)
    return out


class _GradReducer:
    """The gradient collectives of ONE data-parallel step (the reference idiom end2end/main_nce.py:142-158,283): which
    ranges of the flat gradient have been summed over ranks, and the asynchronous ones still in flight per stack.  Every collective is
    issued from the main host thread in an order that depends on the configuration only (video pieces, joint pieces, remainder):
    identical on every rank."""

    def __init__(self, tr):
        self.tr, self.flat, self.mode = tr, tr.online.flat_grad(), tr.ddp_mode
        self.done, self.pending, self.n = [], {}, 0
        self.stack_events = {}            # two-chain step: per stack, the event behind its reduction on the stack's role stream
        self.wire = None
        if tr.ddp_grad_dtype == "bf16":
            w = tr.__dict__.get("_wire")
            if w is None or w.numel() != self.flat.numel() or w.device != self.flat.device:
                w = tr._wire = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
            self.wire = w

    def reduce(self, lo, hi, async_op=False, tag=None):
        """all-reduce(SUM) of flat[lo:hi] under the current stream context (asynchronous: completed by `wait(tag)`)"""
        piece, after = self.flat[lo:hi], None
        if self.wire is not None:
            w = self.wire[lo:hi]
            ops.cast(piece, w)
            work = dist.allreduce_sum_(w, async_op=async_op)
            after = lambda: ops.cast(w, piece)       # noqa: E731
        else:
            work = dist.allreduce_sum_(piece, async_op=async_op)
        self.done.append((lo, hi))
        self.n += 1
        if async_op:
            self.pending.setdefault(tag, []).append((work, after))
        elif after is not None:
            after()

    def wait(self, tag):
        """the current stream waits for the asynchronous collectives issued under `tag`"""
        for work, after in self.pending.pop(tag, []):
            if work is not None:
                work.wait()
            if after is not None:
                after()

    def hook(self, tag, layer_events):
        """'buckets': called once a stack's backward is enqueued -- one asynchronous all-reduce per bucket, each made to wait (on the
        GPU) only for the event of its lowest layer"""
        tr = self.tr
        comm = tr._comm_order_stream(self.flat.device)
        for lo, hi, last in tr._ddp_buckets(tag, len(layer_events)):
            # comm waits (on the GPU) for the event of the bucket's lowest layer; the process group's own stream
            # then waits for comm, i.e. for exactly the layers this bucket covers
            _lib.check(_lib.lib().tan_stream_wait_event(C.c_void_p(comm.cuda_stream), C.c_void_p(layer_events[last])),
                       "tan_stream_wait_event")
            with torch.cuda.stream(comm):
                self.reduce(lo, hi, async_op=True, tag=tag)

    def stack_done(self, which):
        """Two-chain step, on the stream that carries the stack's last weight gradients: the stack's slice of the gradient is summed
        over ranks once this returns (in stream order) -- its optimizer launch may follow."""
        if self.mode == "buckets":
            self.wait(which)
        elif self.mode == "flat":
            prefix = {"video": "video_temporal_encoder.", "joint": "joint_temporal_encoder."}[which]
            self.reduce(*self.tr.online.flat_range(prefix))
            return True
        return self.mode == "buckets"

    def finish(self):
        cur = torch.cuda.current_stream()
        for ev in self.stack_events.values():      # (the stacks' reductions ran on their role streams)
            cur.wait_event(ev)
        self.stack_events = {}
        for lo, hi in uncovered_ranges(self.done, self.flat.numel()):        # whatever has not been reduced yet
            self.reduce(lo, hi)
        for tag in list(self.pending):
            self.wait(tag)
        return self.n


class Trainer:
    def __init__(self, model, args, *, betas=(0.9, 0.999), eps=1e-8, iter_per_epoch=None, warmup=1000, fused_loss=None,
                 global_negatives=False, ddp_bucket_layers=None):
        self.model, self.args = model, args
        online = model.online if isinstance(model, TwinTemporalAligner) else model
        # logits-free similarity+NCE whenever the model runs in bf16 (the fused kernels are bf16-only)
        self.fused_loss = (online.compute_dtype == torch.bfloat16) if fused_loss is None else bool(fused_loss)
        self.twin = isinstance(model, TwinTemporalAligner)
        self.online = model.online if self.twin else model
        self.betas, self.eps = betas, eps
        self.iteration = 0
        self.iter_per_epoch, self.warmup = iter_per_epoch, warmup
        self._state = None
        # row f3: NCE negatives from every rank (fused bf16 path only).  The global loss is then the SUM of the rank losses,
        # so gradients are summed over ranks instead of averaged.
        self.global_negatives = bool(global_negatives)
        self._lr_iter = None              # (kept for callers that pin the schedule position explicitly)
        self.batches_seen = 0             # batches processed so far = the reference's args.iteration - 1 (main.py:281,140)
        self._resume_bump = 0             # 1 for the first batch after checkpoint.load_for_resume (see there)
        self._accum_open = False          # gradients of earlier batches are waiting in the flat buffer (backprop_freq > 1)
        # data parallelism: encoder layers per gradient all-reduce bucket (one layer = 12.6 MB of f32 gradient)
        self.ddp_bucket_layers = max(1, int(os.environ.get("TAN_DDP_BUCKET_LAYERS", "2") if ddp_bucket_layers is None
                                            else ddp_bucket_layers))
        self._comm_streams = {}
        self._zero_ev = None
        # TAN_DDP_MODE -- how the flat gradient (39.9 M f32) is summed over ranks each step:
        #   "flat" (default)  ONE logical all-reduce of the whole gradient, what BASELINE.json's north_star states, issued in contiguous
        #                     pieces as they become final: in the two-chain step the video stack's 45 % when its chain ends (under the
        #                     joint stack's last layers), the joint stack's 47 % when that ends, the remaining 8 % behind the embeddings'
        #                     backward; each stack's optimizer launch follows its piece.  Under autograd (stage 2): one call after backward.
        #   "single"          literally one call after backward (the A/B of the above)
        #   "buckets"         an all-reduce per `ddp_bucket_layers` layers behind per-layer events, overlapped with backward (the
        #                     reference idiom's DistributedDataParallel, end2end/main_nce.py:283); works in both step schedules
        # No >1-GPU node was available to pick by measurement: bench.py reports the other modes as `extra` entries of a multi-GPU run.
        self.ddp_mode = os.environ.get("TAN_DDP_MODE", "flat")
        # TAN_DDP_GRAD_DTYPE=bf16: the gradient crosses xGMI as bf16 (80 MB instead of 160), rounded once before and summed in f32 after
        self.ddp_grad_dtype = os.environ.get("TAN_DDP_GRAD_DTYPE", "f32")
        # TAN_STEP_PIPELINE (default 1): the two-chain step does not join its role streams when it returns -- the stacks' last weight
        # gradients and optimizer launches run under the NEXT step's embeddings and video stack (`_run_chains`, `_Flat.pending`)
        self.pipeline = os.environ.get("TAN_STEP_PIPELINE", "1") != "0"
        self._ddp = None                  # the step's _GradReducer while `step` runs with more than one rank
        self.last_collectives = 0
        self._params_synced = False
        # bench.py: with `time_comm` set, every step records two events on the compute stream around the part of the step that
        # WAITS for gradient collectives (the remainder all-reduce and the joins of the asynchronous buckets): their distance is
        # the communication time the step could not hide behind backward ("exposed"); comm_events collects the pairs
        self.time_comm = False
        self.comm_events = []
        # tests: keep what get_loss decided without gradient (agreement targets, threshold mask, alignability labels) in `last_aux`
        self.keep_aux = False
        self.last_aux = None

    # -------------------------------------------------------------- optimizer state
    def _ensure_state(self):
        f = self.online._ensure_flat()
        if self._state is None or self._state["m"].device != f.flat.device:
            mode = self.online.param_modes("online." if self.twin else "")
            if self.args.optim_policy == "bce":        # only binary_head trains (main.py:345-352)
                for n in f.names:
                    if "binary_head" not in n:
                        o, k, _ = f.off[n]
                        mode[o:o + k] = 2
            self._state = {"m": torch.zeros_like(f.flat), "v": torch.zeros_like(f.flat), "mode": mode.to(f.flat.device)}
        return f, self._state

    def current_lr(self):
        """Learning rate of the batch being processed.  The reference starts args.iteration at 1 (main.py:281), calls
        lr_scheduler.step(args.iteration) once before training (main.py:499) and again after every batch with the
        PRE-increment counter (main.py:138-140): batch b (0-based) therefore runs at lambda(max(b, 1)) -- replayed against
        torch's LambdaLR in tests/test_train_eval_gpu.py::test_lr_schedule_lag_matches_reference_lambda_lr -- and the first
        batch after a resume at lambda(b + 1)."""
        if self.iter_per_epoch is None:
            return self.args.lr
        lr_iter = getattr(self, "_lr_iter", None)
        b = self.batches_seen if lr_iter is None else lr_iter
        return self.args.lr * lr_multiplier(max(b + getattr(self, "_resume_bump", 0), 1), self.iter_per_epoch, self.args.epochs,
                                            self.warmup)

    def _comm_order_stream(self, device):
        """The stream the gradient collectives are issued under: it only ever waits for layer events, so a bucket's all-reduce
        is ordered after ITS layers and not after whatever else the compute streams have queued."""
        st = self._comm_streams.get(device)
        if st is None:
            st = self._comm_streams[device] = _lib.role_stream(device, "comm")
        return st

    def sync_parameters(self, src=0):
        """Every rank starts from rank `src`'s parameters (what DistributedDataParallel does at construction,
        end2end/main_nce.py:283): online flat buffer, EMA twin, language model.  Called once, from the first step."""
        self._params_synced = True
        if not dist.active():
            return
        mods = [self.online] + ([self.model.target] if self.twin else [])
        for m in mods:
            dist.broadcast_(m.flat_parameters(), src)
            m.invalidate_shadow()
            if m.bert is not None:
                # every language-model tensor, like DistributedDataParallel's constructor: the trainable ones in ONE broadcast, the
                # frozen word table (~80 MB, random-initialised by nn.Embedding unless the caller loaded it on every rank) in place,
                # once -- this function runs on the first step only
                frozen_table = {id(m.bert.word_embd.weight)} if hasattr(m.bert, "word_embd") else set()
                for p in m.bert.parameters():
                    if id(p) in frozen_table:
                        dist.broadcast_(p.data, src)
                for b in m.bert.buffers():
                    dist.broadcast_(b.data, src)
                ps = [p for p in m.bert.parameters() if id(p) not in frozen_table]
                if ps:
                    flat = torch.cat([p.data.reshape(-1) for p in ps])
                    dist.broadcast_(flat, src)
                    off = 0
                    for p in ps:
                        p.data.copy_(flat[off:off + p.numel()].view_as(p))
                        off += p.numel()

    def _lm_params(self):
        lm = self.online.bert
        return [] if lm is None else [(n, p) for n, p in lm.named_parameters() if p.requires_grad]

    def zero_grad(self, side=None):
        """side: a stream to fill on -- the fill then runs next to the forward and `forward_backward` joins it before backward."""
        f = self.online._ensure_flat()
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            for e in f.pending.values():          # (a pipelined step: the optimizer launches that still read the gradient)
                if e is not None:
                    side.wait_event(e)
            with torch.cuda.stream(side):
                f.grad.zero_()
                self._zero_ev = side.record_event()
        else:
            f.grad.zero_()
        self.online._bind_grads()
        for _, p in self._lm_params():
            p.grad = None

    def _lm_step(self, grad_scale):
        """AdamW (+EMA) for the language model's fc1/fc2, which live outside the aligner's flat buffer: one fused launch per
        tensor with the same decay rule (train/main.py:332: '.bias' -> no decay; names are `bert.fc1.weight`, ...)."""
        named = self._lm_params()
        if not named:
            return
        if "lm" not in self._state:
            self._state["lm"] = {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in named}
        tgt = dict(self.model.target.bert.named_parameters()) if self.twin and self.model.target.bert is not None else {}
        a = self.args
        for n, p in named:
            if p.grad is None:       # no gradient this step: AdamW skips the tensor, the EMA twin still moves (tan_model.py:339-344)
                if n in tgt:
                    _lib.check(_lib.lib().tan_ema_update(_vp(tgt[n].data), _vp(p.data), C.c_long(p.numel()), C.c_float(self.model.m),
                                                         None, ops._stream()), "tan_ema_update")
                continue
            m, v = self._state["lm"][n]
            wd = 0.0 if n.endswith(".bias") else a.wd
            ema = tgt.get(n)
            _lib.check(_lib.lib().tan_adamw_step(
                _vp(p.data), _vp(p.grad.contiguous()), _vp(m), _vp(v), None, C.c_long(p.numel()), C.c_double(self.current_lr()),
                C.c_double(self.betas[0]), C.c_double(self.betas[1]), C.c_double(self.eps), C.c_double(wd), C.c_int(self.iteration),
                C.c_float(grad_scale), None, _vp(ema.data) if ema is not None else None,
                C.c_float(self.model.m if self.twin else 0.0), None, ops._stream()), "tan_adamw_step")

    def _embed_tokens(self, batch):
        """sentence embeddings from the language model (train/main.py:55-65) when the batch carries token ids; idempotent"""
        if "token" in batch and self.online.bert is not None and not batch.get("_lm_embedded"):
            batch = dict(batch)
            batch["text_embed"], batch["text_padding_mask"] = embed_sentences(self.model, batch["token"])
            batch["_text_pad_bool"] = None
            batch["n_text"] = int(sum(t.shape[0] for t in batch["token"]))
            batch["_lm_embedded"] = True
        return batch

    def forward_backward(self, batch):
        a, m = self.args, self.model
        batch = self._embed_tokens(batch)
        fused = self.fused_loss
        if fused:       # the logits-free sweep keeps one LDS accumulator per text column: beyond its limit use materialised logits
            Bn, Nn = batch["text_embed"].shape[:2]
            cols = Bn * Nn
            if batch.get("n_text") is not None and not self.global_negatives:
                cols = min(cols, (batch["n_text"] + 63) // 64 * 64)
            if cols > _lib.lib().tan_simnce_max_cols():
                if self.global_negatives:
                    raise _lib.TanHipError(f"global_negatives: {Bn}x{Nn} text columns per rank exceed the fused sweep's limit")
                fused = False
        if self.online.compute_dtype == torch.bfloat16 and batch["video"].is_cuda:
            batch = self._pad_sentence_slots(batch)          # (whole 64-row panels for the joint stack: both step schedules)
        self._last_step_chains = self._chains_eligible(batch, fused)
        if self._last_step_chains:
            return self._forward_backward_chains2(batch) if self.twin else self._forward_backward_chains(batch)
        # (`step` decided on pipelining from `_will_chain` before this point; should the two ever disagree, the autograd schedule below must
        #  not start under the previous step's pending optimizer launches: wait for them here, on this stream)
        for fl_ in [self.online._flat] + ([self.model.target._flat] if self.twin else []):
            if fl_.in_step:
                fl_.in_step = False
                fl_.drain()
        if batch["video"].is_cuda:
            # what get_loss derives from the batch alone (masks, targets, column compaction: ~20 tiny launches) runs on the loss
            # side stream next to the forward instead of between the stacks and the similarity sweeps
            from .loss import prepare_inputs_async
            batch = dict(batch)
            Tn, Nn = batch["video"].shape[1], batch["text_embed"].shape[1]
            batch["_loss_prep"] = prepare_inputs_async(batch, batch["padding_mask"], batch["text_padding_mask"], Tn, Nn,
                                                       batch["video"].device, a, batch.get("n_text"),
                                                       want_compaction=bool(fused) and not self.global_negatives)
        tp_bool = batch["_text_pad_bool"] if batch.get("_text_pad_bool") is not None else batch["text_padding_mask"].bool()
        logits = m(batch["video"], batch["text_embed"], video_padding_mask=batch["padding_mask"],
                   lang_padding_mask=tp_bool, text_timestamp=batch.get("_tgt_raw"),
                   abs_text_pos=batch.get("abs_text_pos"),
                   fused="defer" if (fused and not a.learn_agreement and not self.global_negatives) else fused)
        if "_fused" in logits and batch.get("n_text") is not None:
            logits["_fused"].n_text_valid = batch["n_text"]        # padded text columns are skipped by the similarity sweep
        if self.global_negatives:
            if "_fused" not in logits:
                raise _lib.TanHipError("global_negatives needs the fused (bf16) similarity path")
            logits["_fused"].global_negatives = True
        if a.model == "cotrain":
            ema = m.forward_from_ema(batch["video"], batch["text_embed"], video_padding_mask=batch["padding_mask"],
                                     lang_padding_mask=tp_bool,
                                     text_timestamp=batch.get("_tgt_raw"), abs_text_pos=batch.get("abs_text_pos"),
                                     fused=fused)
            logits = {**logits, **{f"ema-{k}": v for k, v in ema.items()}}
        loss_dict = get_loss(batch, batch["video"], batch["text_embed"], batch["padding_mask"], batch["text_padding_mask"],
                             logits, a, batch.get("abs_text_pos"), return_aux=self.keep_aux)
        if self.keep_aux:
            loss_dict, self.last_aux = loss_dict
        if self._zero_ev is not None:                  # the gradient buffer's fill ran on a side stream next to the forward
            torch.cuda.current_stream().wait_event(self._zero_ev)
            self._zero_ev = None
        loss_dict["loss"].backward()
        return loss_dict

    def _chains_eligible(self, batch, fused):
        """Stage 1 ('init': multi-positive NCE only, train/readme.md:10) on the fused bf16 path with rank-local negatives: the loss is
        (loss_dual + loss_joint) / 2 with weights that depend on the batch's masks only, so the step runs as two chains that never wait
        for each other (`_AlignerEngine._run_chains`) instead of forward -> loss -> backward under autograd.  TAN_STEP_CHAINS=0: autograd."""
        a = self.args
        if not (bool(fused) and a.optim_policy != "bce" and not self.global_negatives and batch["video"].is_cuda
                and os.environ.get("TAN_STEP_CHAINS", "1") != "0"):
            return False
        if self.twin:
            return self._chains2_eligible(batch)
        return (a.model == "init" and not a.learn_agreement and a.loss_threshold <= 0 and not a.use_alignability_head
                and self.online._chains_ok(batch["video"], batch["text_embed"]))

    def _chains2_eligible(self, batch):
        """Stage 2 ('cotrain' as train/readme.md:13 runs it: EMA self-labelling, loss threshold, alignability head + BCE) on the fused bf16
        path with rank-local statistics: the step runs as two chains as well (`_forward_backward_chains2`), synchronised where the loss
        needs both families -- the agreement targets (both EMA stacks) and the thresholds / labels (both online families' forward).
        TAN_STAGE2_CHAINS=0: forward -> get_loss -> backward under autograd."""
        a, m, tg = self.args, self.online, self.model.target
        video, lang = batch["video"], batch["text_embed"]
        B, T, N = video.shape[0], video.shape[1], lang.shape[1]
        if not (a.model == "cotrain" and a.learn_agreement and a.loss_threshold > 0 and a.use_alignability_head
                and m.use_alignability_head and m.num_decoder_layers >= 3 and B * N <= 8192 and lang.shape[1] <= 32
                and os.environ.get("TAN_STAGE2_CHAINS", "1") != "0" and os.environ.get("TAN_STAGE2_FUSED", "1") != "0"
                and m._chains_ok(video, lang, allow_head=True) and tg._chains_ok(video, lang, allow_head=True)
                and tg.compute_dtype == m.compute_dtype and not m.use_text_pos_enc):
            return False
        from .loss import simfam_ok
        Mp = B * N
        nt = batch.get("n_text")
        Mc = Mp if nt is None else min(Mp, (int(nt) + 63) // 64 * 64)
        return all(simfam_ok(S, N, Mc, torch.bfloat16, T) for S in (m.num_encoder_layers, m.num_decoder_layers))

    @staticmethod
    def _pad_sentence_slots(batch):
        """Small batches: the row-panel kernels take stacks of B * (T + N) rows in whole 64-row panels (at B = 128 every N does; at
        B = 16 only N % 4 == 0, and the joint stack fell back to LayerNorm + tiled-GEMM launches for the other three quarters of the
        batches).  Up to three more PADDED sentence slots per video make the row count fit: a padded sentence is masked as an attention
        key and dropped from the loss (train/main.py:61-65 pads to the longest video of the batch the same way), so nothing a real row
        or the loss sees changes."""
        video, lang = batch["video"], batch["text_embed"]
        B, T, N = video.shape[0], video.shape[1], lang.shape[1]
        if (B * (T + N)) % 64 == 0:
            return batch
        extra = next((e for e in (1, 2, 3) if (B * (T + N + e)) % 64 == 0), 0)
        if not extra:
            return batch
        out = dict(batch)
        out["text_embed"] = torch.cat([lang, lang[:, -1:].expand(-1, extra, -1)], 1)
        tp = batch["text_padding_mask"]
        out["text_padding_mask"] = torch.cat([tp, torch.ones(B, extra, dtype=tp.dtype, device=tp.device)], 1)
        if batch.get("_text_pad_bool") is not None:
            pb = batch["_text_pad_bool"]
            out["_text_pad_bool"] = torch.cat([pb, torch.ones(B, extra, dtype=torch.bool, device=pb.device)], 1)
        tr_ = batch.get("_tgt_raw")
        if tr_ is None:
            tr_, _, _ = get_mask_from_time(batch["start"], batch["end"], T, N, device=video.device)
        out["_tgt_raw"] = torch.cat([tr_, torch.zeros(B, extra, tr_.shape[2], dtype=tr_.dtype, device=tr_.device)], 1)
        ap = batch.get("abs_text_pos")
        if torch.is_tensor(ap):
            out["abs_text_pos"] = torch.cat([ap, torch.zeros(B, extra, ap.shape[2], dtype=ap.dtype, device=ap.device)], 1)
        return out

    def _forward_backward_chains(self, batch):
        from .loss import _ManualCtx, _NCETail, nce_family_stages, nce_term_grads, prepare_inputs_async
        a, m = self.args, self.online
        video, lang = batch["video"], batch["text_embed"]
        B, T = video.shape[:2]
        N = lang.shape[1]
        dev = video.device
        Se, Sd = m.num_encoder_layers, m.num_decoder_layers
        main = torch.cuda.current_stream()
        # everything the loss derives from the batch alone, incl. the gradients of the NCE terms, on the loss side stream
        prep = prepare_inputs_async(batch, batch["padding_mask"], batch["text_padding_mask"], T, N, dev, a, batch.get("n_text"),
                                    want_compaction=True)
        from .loss import _side_stream
        ls = _side_stream(dev)
        with torch.cuda.stream(ls):
            nv = prep.get("nv")
            cols_tail = prep["cols_pos_c"] if nv is not None else prep["cols_pos"]
            g_v_d, g_t_d, g_v_j, g_t_j, _ = nce_term_grads(prep["rows_pos"], cols_tail, Se, Sd)
            ready = ls.record_event()
        tgt, ci = prep["tgt"], prep["tpad_u8"].view(B * N)

        def family(which, x_video, v_grp, x_text, t_grp, d_video, d_text):
            torch.cuda.current_stream().wait_event(ready)          # (a finished event costs nothing)
            gv, gt = (g_v_d, g_t_d) if which == "dual" else (g_v_j, g_t_j)
            return nce_family_stages(x_video, v_grp, x_text, t_grp, d_video, d_text, tgt, ci, B, T, N, nv, gv, gt)
        pipe = None
        if getattr(self, "_in_step", False) and self.online._flat.in_step:
            pipe = {"zero": self._zero_ev}               # (each chain waits for the fill in front of its first backward kernel)
        elif self._zero_ev is not None:                  # the gradient buffer's fill ran on a side stream
            main.wait_event(self._zero_ev)
        self._zero_ev = None
        tp_bool = batch["_text_pad_bool"] if batch.get("_text_pad_bool") is not None else batch["text_padding_mask"].bool()
        early_v = early_j = None
        if getattr(self, "_in_step", False):
            early = os.environ.get("TAN_OPT_EARLY", "1") != "0" and self._early_ok()
            ddp, gs = self._ddp, 1.0 / dist.world_size()

            def stack_done(which):
                # (main host thread, current stream = the one that carries the stack's last weight gradients) data parallel: the stack's
                # slice of the gradient is summed over ranks first; then its matrices may be stepped
                reduced = ddp.stack_done(which) if ddp is not None else True
                if ddp is not None:
                    # the remainder's optimizer launch (main stream) reads this stack's bias / LayerNorm gradients: behind the stack's
                    # collective AND the bf16 wire's cast back to f32, which run on this role stream
                    ddp.stack_events[which] = torch.cuda.current_stream().record_event()
                if early and reduced:
                    self.early_update(which, gs)
            if early or ddp is not None:
                early_v, early_j = (lambda: stack_done("video")), (lambda: stack_done("joint"))       # noqa: E731
        tail = {}

        def loss_tail():
            # the masked means of the four term tensors (loss.py:254-275), on the main stream right behind the video stack's backward:
            # the joint family's terms were final ~1.5 ms earlier on their stream; issued at the end of the step the launch sat between
            # the embeddings' backward and the optimizer launch (4.126 -> 4.078 ms per step, ABBA x2 of 60 steps)
            terms, ready = m._joint_terms
            if not ready.wait(timeout=60.0) or not terms:
                raise _lib.TanHipError("the joint chain did not reach its loss family")      # (`_run_chains` re-raises the chain's own error)
            v_j_, t_j_, ev_j = terms[0]
            main.wait_event(ev_j)
            v_d_, t_d_ = m._dual_terms
            for t in (v_j_, t_j_):
                t.record_stream(main)
            tail["out"] = _NCETail.forward(_ManualCtx(), v_d_, t_d_, v_j_, t_j_, prep["rows_pos"], cols_tail, None)
        v_d, t_d, v_j, t_j = m._run_chains(video, lang, m._mask_u8(batch["padding_mask"]), m._mask_u8(tp_bool), family, early_v, early_j,
                                           pipe=pipe, mid=loss_tail, need_d_lang=lang.requires_grad)
        m._joint_terms = m._dual_terms = None
        if lang.requires_grad:
            # the step started from token ids: the sentence embeddings' gradient (the embeddings' backward produced it) goes on through
            # the language model under autograd (index_select of `embed_sentences`, Word2VecModel's HIP nodes), on this stream
            lang.backward(m.__dict__.pop("_chain_d_lang").to(lang.dtype))
        for t in (g_v_d, g_t_d, g_v_j, g_t_j, cols_tail, prep["rows_pos"], v_j, t_j):
            t.record_stream(main)
        loss_dual, loss_joint, loss_mean = tail["out"]
        if pipe is not None:
            self._pipe_out = pipe["out"]
        return {"loss-dual": loss_dual, "loss-joint": loss_joint, "loss": loss_mean}

    def _forward_backward_chains2(self, batch):
        """Stage-2 co-training step (train/main.py:89-98,122; train/loss.py:88-229,277-373) as two chains without autograd:
            main:  EMA video stack -> cosines | online video stack -> sweep(dual)  .. finish(dual)  .. backward(dual)  -> stack backward
            side:  EMA joint stack -> cosines | online joint stack -> head -> sweep(joint) .. finish(joint) .. backward(joint) + head -> stack backward
        with two meeting points: (1) the agreement targets need both EMA stacks' same-video cosines (small launches on the loss stream,
        issued by the main chain's host thread: they are through long before the online sweeps are); (2) the terms' upstream gradients
        need both families' forward results -- per-sentence maxima -> z-scores / quantile threshold / kept rows, alignability labels,
        BCE (train/loss.py:277-357) -- issued on the main chain's stream between its family's finishing launch and backward.
        Every launch is the one `get_loss` issues on the autograd path (same kernels, `tan_simfam_*` for the families)."""
        import threading
        from .loss import (SimFam, _Blocks, _diag_max, _pos_masks, _side_stream, agreement_targets, prepare_inputs_async, stage2_masks)
        from .loss import _p
        a, m, tg = self.args, self.online, self.model.target
        video, lang = batch["video"], batch["text_embed"]
        B, T = video.shape[:2]
        N = lang.shape[1]
        R, Mp, L = B * T, B * N, T + N
        dev = video.device
        Se, Sd, Cw = m.num_encoder_layers, m.num_decoder_layers, 512
        lib = _lib.lib()
        main = torch.cuda.current_stream()
        ls = _side_stream(dev)
        prep = prepare_inputs_async(batch, batch["padding_mask"], batch["text_padding_mask"], T, N, dev, a, batch.get("n_text"),
                                    want_compaction=True)
        nv = prep.get("nv")
        Mc = nv[0].shape[0] if nv is not None else Mp
        tpad_u8, ci = prep["tpad_u8"], prep["tpad_u8"].view(Mp)
        tp_bool = batch["_text_pad_bool"] if batch.get("_text_pad_bool") is not None else batch["text_padding_mask"].bool()
        vmask, tmask = m._mask_u8(batch["padding_mask"]), m._mask_u8(tp_bool)
        # buffers that the meeting points fill (their addresses go into the families' descriptors up front)
        tgt = torch.empty(B, T, N, device=dev)
        g = {"dual": (torch.empty(Se, R, device=dev), torch.empty(Se, Mc, device=dev)),
             "joint": (torch.empty(Sd, R, device=dev), torch.empty(Sd, Mc, device=dev))}
        d_head = torch.empty(Mp, device=dev)                    # d loss / d alignability logits of joint stage 2 (train/loss.py:341)
        # host-side hand-overs between the two issuing threads (an event can only be waited for once it has been recorded) + their events
        flags = {k: threading.Event() for k in ("ema_joint", "tgt", "fin_joint", "g")}
        evs, fams, ema_diag, failed, aux, head = {}, {}, {}, [], {}, {}

        def hand_over(name):
            if not flags[name].wait(timeout=120.0) or failed:
                err = _lib.TanHipError(f"stage-2 step: the other chain did not reach '{name}'" + (f" ({failed[0]} chain failed)" if failed else ""))
                err.tan_consequence = True          # (`_run_chains` reports the other thread's error instead)
                raise err
            return evs[name]

        def guarded(fn):
            def run(*args_, **kw):
                try:
                    return fn(*args_, **kw)
                except BaseException:
                    failed.append(args_[0] if args_ else "?")
                    for f_ in flags.values():       # the other thread must not wait out its timeout
                        f_.set()
                    raise
            return run

        # ---- the EMA target (tan_model.py:346-351 `forward_from_ema`, no gradient): its input embeddings here, its stacks on the chains
        with torch.no_grad():
            tg._ensure_flat()
            fe_t = tg._embed_fused(video, lang, vmask, tmask, 0, 0, 0, False)

        @guarded
        def pre(which):
            cur = torch.cuda.current_stream()
            with torch.no_grad():
                ema_diag[which] = tg._ema_stack_diag(which, fe_t, vmask, tmask, B, T, N)
            evs["ema_" + which] = cur.record_event()
            if which == "joint":
                flags["ema_joint"].set()

        def targets():
            # meeting point 1, on the loss stream: train/loss.py:88-229 (self-labelling of both EMA families, agreement 'keep' / ...,
            # de-duplication) + the positive masks of loss.py:236-237
            hand_over("ema_joint")
            ls.wait_event(evs["ema_video"])
            ls.wait_event(evs["ema_joint"])
            ls.wait_event(prep["_event"])
            with torch.cuda.stream(ls):
                for t_ in ema_diag.values():
                    t_.record_stream(ls)
                J, D, _, iou, conf = agreement_targets(_Blocks.of_diag(ema_diag["joint"]), _Blocks.of_diag(ema_diag["video"]), prep, B, T, N,
                                                       a.temporal_agreement_type, tgt=tgt)
                rows_pos, cols_pos = _pos_masks(tgt, tpad_u8, B, T, N)
                cols_tail = cols_pos.index_select(0, nv[0]) if nv is not None else cols_pos
                evs["tgt"] = ls.record_event()
            aux.update(max_position_joint=J["max_pos"], max_position_dual=D["max_pos"], max_logits_joint=J["max_logit"],
                       max_logits_dual=D["max_logit"], joint_self_tgt=J["tgt"], dual_self_tgt=D["tgt"], iou=iou, confidence_mask=conf,
                       agreement_tgt=tgt, rows_pos=rows_pos, cols_tail=cols_tail)
            flags["tgt"].set()

        def upstream():
            # meeting point 2, on the main chain's stream: train/loss.py:277-357 and the tail's backward
            cur = torch.cuda.current_stream()
            fd, fj = fams["dual"], fams["joint"]
            fj.record_stream(cur)
            md, mj = head["md"], head["mj"]          # per-sentence maxima (loss.py:280,283): each chain's own launch behind its finish
            mj.record_stream(cur)
            s2 = stage2_masks(md, mj, tpad_u8, tgt, batch.get("abs_text_pos"), aux["confidence_mask"], a.loss_threshold, True, B, T, N)
            cols_th = s2["th_f"].index_select(0, nv[0]) if nv is not None else s2["th_f"]
            out_th = torch.empty(5, device=dev)              # [loss_dual_th, loss_joint_th, their mean, n_rows, n_cols]
            _lib.check(lib.tan_nce_tail_fwd(_p(fd.v_terms), _p(fd.t_terms), _p(fj.v_terms), _p(fj.t_terms), _p(s2["rows"]), _p(cols_th),
                                            C.c_int(Se), C.c_int(Sd), C.c_long(R), C.c_long(Mc), _p(out_th), _p(out_th[3:]), None,
                                            ops._stream()), "tan_nce_tail_fwd")
            one, gb = self._stage2_consts(dev)
            _lib.check(lib.tan_nce_tail_bwd(None, None, _p(one), _p(s2["rows"]), _p(cols_th), _p(out_th[3:]), C.c_int(Se), C.c_int(Sd),
                                            C.c_long(R), C.c_long(Mc), _p(g["dual"][0]), _p(g["dual"][1]), _p(g["joint"][0]),
                                            _p(g["joint"][1]), ops._stream()), "tan_nce_tail_bwd")
            # BCE of the alignability head on joint stage 2 (loss.py:341-349), forward and backward
            a_joint = head["a_j2"]
            a_joint.record_stream(cur)
            bce = torch.empty(2, device=dev)
            _lib.check(lib.tan_bce_sel_fwd(_p(a_joint), _p(s2["y"]), _p(s2["sel"]), _p(s2["scal"]), C.c_int(Mp), _p(bce), ops._stream()),
                       "tan_bce_sel_fwd")
            _lib.check(lib.tan_bce_sel_bwd(_p(a_joint), _p(s2["y"]), _p(s2["sel"]), _p(s2["scal"]), _p(gb), C.c_int(Mp), _p(d_head),
                                           ops._stream()), "tan_bce_sel_bwd")
            evs["g"] = cur.record_event()
            flags["g"].set()
            aux.update(t_th_mask=s2["th_mask"], max_logits_dual_per_text=md, max_logits_joint_per_text=mj, t_align_th_mask=s2["lab"],
                       out_th=out_th, bce=bce, s2=s2)

        @guarded
        def family(which, x_video, v_grp, x_text, t_grp, d_video, d_text):
            cur = torch.cuda.current_stream()
            cur.wait_event(prep["_event"])
            fam = fams[which] = SimFam(x_video, v_grp, x_text, t_grp, d_video, d_text, tgt, ci, B, T, N, nv, *g[which])
            fam.sweep()
            if which == "joint":
                # alignability head on the text rows of joint stage 2 (tan_model.py:147-148; the only stage the loss reads, loss.py:341)
                jt = torch.empty(Mp, Cw, dtype=x_text[2].dtype, device=dev)
                ops.rows_copy(x_text[2], jt, B, N, Cw, L, T, N, 0)
                a_j2 = torch.empty(Mp, device=dev)
                ops.head_fwd(jt, m._f("binary_head.weight").view(-1), m._f("binary_head.bias"), a_j2, Mp, Cw)
                head.update(jt=jt, a_j2=a_j2)
                cur.wait_event(hand_over("tgt"))
                fam.finish()
                head["mj"] = _diag_max(_Blocks.of_diag(fam.diag_last), None, B, T, N)
                evs["fin_joint"] = cur.record_event()
                flags["fin_joint"].set()
                cur.wait_event(hand_over("g"))
            else:
                targets()
                cur.wait_event(evs["tgt"])
                fam.finish()
                head["md"] = _diag_max(_Blocks.of_diag(fam.diag_last), None, B, T, N)
                cur.wait_event(hand_over("fin_joint"))
                upstream()
            fam.backward()
            if which == "joint":
                d_jt = torch.empty(Mp, Cw, dtype=head["jt"].dtype, device=dev)
                ops.head_bwd(d_head, head["jt"], m._f("binary_head.weight").view(-1), d_jt, m._g("binary_head.weight").view(-1),
                             m._g("binary_head.bias"), Mp, Cw)
                ops.rows_copy(d_jt, d_text[2], B, N, Cw, N, 0, L, T, accumulate=True)
                d_head.record_stream(cur)
            return fam.v_terms, fam.t_terms

        pipe = None
        if getattr(self, "_in_step", False) and m._flat.in_step:
            pipe = {"zero": self._zero_ev}
        elif self._zero_ev is not None:
            main.wait_event(self._zero_ev)
        self._zero_ev = None
        early_v = early_j = None
        if getattr(self, "_in_step", False):
            early = os.environ.get("TAN_OPT_EARLY", "1") != "0" and self._early_ok()
            ddp, gs = self._ddp, 1.0 / dist.world_size()

            def stack_done(which):
                reduced = ddp.stack_done(which) if ddp is not None else True
                if ddp is not None:
                    ddp.stack_events[which] = torch.cuda.current_stream().record_event()
                if early and reduced:
                    self.early_update(which, gs)
            if early or ddp is not None:
                early_v, early_j = (lambda: stack_done("video")), (lambda: stack_done("joint"))       # noqa: E731
        m._run_chains(video, lang, vmask, tmask, family, early_v, early_j, pipe=pipe, need_d_lang=lang.requires_grad,
                      pre={"video": lambda: pre("video"), "joint": lambda: pre("joint")})
        m._joint_terms = m._dual_terms = None
        tg._release_ws(fe_t["em"])
        if lang.requires_grad:
            lang.backward(m.__dict__.pop("_chain_d_lang").to(lang.dtype))
        # ---- the loss dictionary (monitoring entries on the loss stream; train/loss.py:296-304,359-373)
        fd, fj = fams["dual"], fams["joint"]
        out_th, bce = aux.pop("out_th"), aux.pop("bce")
        s2 = aux.pop("s2")
        with torch.cuda.stream(ls):
            ls.wait_event(evs["g"])
            for f_ in (fd, fj):
                f_.record_stream(ls)
            out_all = torch.empty(5, device=dev)
            _lib.check(lib.tan_nce_tail_fwd(_p(fd.v_terms), _p(fd.t_terms), _p(fj.v_terms), _p(fj.t_terms), _p(aux["rows_pos"]),
                                            _p(aux["cols_tail"]), C.c_int(Se), C.c_int(Sd), C.c_long(R), C.c_long(Mc), _p(out_all),
                                            _p(out_all[3:]), None, ops._stream()), "tan_nce_tail_fwd")
            loss = out_th[2] + bce[0]
        main.wait_stream(ls)
        # (allocator bookkeeping for what crossed streams: main joined `ls` just above and joined the side stream inside `_run_chains`,
        #  so a block freed after this point is safe for main's allocator; what was allocated under `ls` / the side stream and read on
        #  main is recorded here -- a minimal set: ~130 record_stream calls cost 0.13 ms of host time at the end of every step)
        for t_ in (out_all, loss, aux["confidence_mask"], head["mj"], head["a_j2"]):
            t_.record_stream(main)
        for t_ in prep["_tensors"]:
            t_.record_stream(main)
        fj.record_stream(main)
        if pipe is not None:
            self._pipe_out = pipe["out"]
        if self.keep_aux:
            self.last_aux = {k: v for k, v in aux.items() if k not in ("rows_pos", "cols_tail")}
        return {"loss-dual": out_th[0], "loss-joint": out_th[1], "loss-dual-all": out_all[0], "loss-joint-all": out_all[1],
                "loss-total": out_all[2], "loss-joint-bce": bce[0], "alignability_top1": bce[1], "confidence-ratio": s2["scal"][3],
                "iou-threshold": torch.full((), 0.5, device=dev), "loss": loss}

    def _stage2_consts(self, dev):
        c = self.__dict__.setdefault("_s2_consts", {})
        if dev not in c:
            c[dev] = (torch.ones(1, device=dev), torch.tensor([1.0, 0.0], device=dev))      # d loss / d (mean NCE), d loss / d (bce, top-1)
        return c[dev]

    def _lm_allreduce(self):
        """Sum the language model's gradients over ranks in ONE bucket (they live outside the flat buffer).  Must run before
        the per-parameter clip: the clip coefficient is a function of the AVERAGED gradient (utils/train_utils.py:3-13)."""
        if not dist.active():
            return
        # the bucket covers EVERY trainable language-model tensor (zeros where this rank has no gradient): its size must not depend
        # on which ranks happened to get one, or the collective hangs / mixes tensors
        params = [p for _, p in self._lm_params()]
        if not params:
            return
        # ... followed by one has-gradient flag per tensor: a tensor NO rank produced a gradient for (batches with precomputed
        # text_embed) must stay `grad is None`, so that AdamW skips it like the single-GPU run and the reference do
        flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], device=params[0].device)
        bucket = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] + [flags])
        dist.allreduce_sum_(bucket)
        # a rank that has the gradient itself knows the sum is >= 1 without looking; only a rank WITHOUT one reads the flags back
        seen = bucket[-len(params):].tolist() if any(p.grad is None for p in params) else None
        off = 0
        for i, p in enumerate(params):
            g = bucket[off:off + p.numel()].view_as(p)
            if p.grad is None:
                if seen[i] > 0:
                    p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()

    def optimizer_step(self, grad_scale=1.0):
        f, st = self._ensure_state()
        a = self.args
        self._lm_allreduce()
        if a.clip_grad > 0:                            # per-parameter L2 clip, utils/train_utils.py:3-13 (language model included)
            for p in list(f.params) + [q for _, q in self._lm_params()]:
                if p.grad is not None:
                    coef = a.clip_grad / (p.grad.norm(2) * grad_scale + 1e-6)
                    p.grad.mul_(torch.clamp(coef, max=1.0))
        ema = self.model.target._ensure_flat() if self.twin else None
        self.iteration += 1
        if self._images_in_optimizer(f, ema):
            early = self.__dict__.pop("_early", ())          # stacks `early_update` already stepped (same step count, lr, grad_scale)
            u0 = 0
            if early:
                assert "video" in early, early              # (units: video stack, joint stack, pre-projections)
                u0 = f.mats_units if "joint" in early else f.video_units
            self._adamw_images(f, st, ema, grad_scale, units=(u0, 0) if u0 else None)
            self._lm_step(grad_scale)
            return
        _lib.check(_lib.lib().tan_adamw_step(
            _vp(f.flat), _vp(f.grad), _vp(st["m"]), _vp(st["v"]), _vp(st["mode"]), C.c_long(f.total),
            C.c_double(self.current_lr()), C.c_double(self.betas[0]), C.c_double(self.betas[1]), C.c_double(self.eps),
            C.c_double(a.wd), C.c_int(self.iteration), C.c_float(grad_scale), _vp(f.shadow),
            _vp(ema.flat) if ema is not None else None, C.c_float(self.model.m if self.twin else 0.0),
            _vp(ema.shadow) if ema is not None else None, ops._stream()), "tan_adamw_step")
        f.shadow_epoch += 1                             # the kernel rewrote the bf16 shadow: transposed copies are stale
        if ema is not None:
            ema.shadow_epoch += 1                       # (shadow_version is left alone: the flat buffers' version counters did
            #                                             not move, the kernel writes through raw pointers)
        self._lm_step(grad_scale)

    def _images_in_optimizer(self, f, ema):
        """The optimizer launch writes the weight images itself (tan_adamw_step_images) when the model computes from them: bf16
        with the row-panel kernels and the W^T copies (TAN_OPT_IMAGES=0: AdamW, then transpose + 2 x pack on the side stream)."""
        on = self.online
        return (f.shadow is not None and on.panel_kernels and on.transposed_dx and os.environ.get("TAN_OPT_IMAGES", "1") != "0"
                and (ema is None or (ema.shadow is not None and self.model.target.panel_kernels)))

    def _adamw_tables(self, f, st):
        tabs = f.image_table()
        if "rest_idx" not in st:                 # every element outside the matrices: the plain kernel's work list
            own = torch.zeros(f.total, dtype=torch.bool, device=f.flat.device)
            for lo, hi in tabs[4]:
                own[lo:hi] = True
            st["rest_idx"] = (~own).nonzero().flatten().to(torch.int32)
        return tabs

    def _adamw_images(self, f, st, ema, grad_scale, units=None, rest=True, step=None):
        """units = (u0, u1): only those units of the image table (see `early_update`); rest: also everything outside the matrices."""
        tab, prefix, n_ent, n_units, ranges = self._adamw_tables(f, st)
        if ema is not None:
            ema.sync_shadow_p()                  # (allocates the twin's packed image on first use)
        d = _lib.AdamwImagesDesc()
        d.p, d.g, d.m, d.v, d.mode = (t.data_ptr() for t in (f.flat, f.grad, st["m"], st["v"], st["mode"]))
        d.n = f.total
        d.lr, d.beta1, d.beta2, d.eps, d.weight_decay = self.current_lr(), self.betas[0], self.betas[1], self.eps, self.args.wd
        d.step, d.grad_scale = (self.iteration if step is None else step), grad_scale
        d.p_bf16 = f.shadow.data_ptr()
        d.table, d.unit_prefix, d.n_entries, d.n_units = tab.data_ptr(), prefix.data_ptr(), n_ent, n_units
        d.p_packed, d.p_t, d.p_tpacked = f.shadow_p.data_ptr(), f.shadow_t.data_ptr(), f.shadow_tp.data_ptr()
        d.rest_idx, d.n_rest = st["rest_idx"].data_ptr(), (st["rest_idx"].numel() if rest else 0)
        if units is not None:
            d.unit_begin, d.unit_end = units
        if ema is not None:
            d.ema, d.ema_m, d.ema_bf16, d.ema_packed = ema.flat.data_ptr(), self.model.m, ema.shadow.data_ptr(), ema.shadow_p.data_ptr()
        _lib.check(_lib.lib().tan_adamw_step_images(C.byref(d), ops._stream()), "tan_adamw_step_images")
        if rest:                                 # (the call that completes the step)
            f.images_rewritten()
            if ema is not None:
                ema.images_rewritten(transposes=False)

    def _early_ok(self):
        f, st = self._ensure_state()
        # (data parallel: a stack is stepped behind the all-reduce of its slice -- `_GradReducer.stack_done`; not in 'single' mode)
        ok = (self._images_in_optimizer(f, self.model.target._ensure_flat_nosync() if self.twin else None)
              and not self.args.clip_grad > 0 and not self._accum_open
              and (not dist.active() or (self._ddp is not None and self.ddp_mode in ("flat", "buckets"))))
        if ok:
            self._adamw_tables(f, st)        # (built here, by ONE thread: the two early launches are issued from two host threads)
            self._early = set()
        return ok

    def early_update(self, which, grad_scale):
        """Two-chain step: AdamW + weight images of ONE stack's matrices as soon as that stack's backward is enqueued, on the stream
        that carries the stack's last weight-gradient launches (`_run_chains`): the video stack's under the joint stack's last layers
        (its chain ends ~0.25 ms earlier), the joint stack's next to the embeddings' backward -- HBM-bound launches next to MFMA-bound /
        small ones; `optimizer_step` then updates what is left (the pre-projections and everything outside the matrices).  The units of
        the image table are ordered video stack, joint stack, pre-projections.  (Called from the two host threads that issue the chains.)"""
        f, st = self._ensure_state()
        lo, hi = (0, f.video_units) if which == "video" else (f.video_units, f.mats_units)
        ema = self.model.target._ensure_flat_nosync() if self.twin else None       # (stage 2: the EMA twin's stack moves in the same launch)
        self._adamw_images(f, st, ema, grad_scale, units=(lo, hi), rest=False, step=self.iteration + 1)
        self._early.add(which)

    def train_iteration(self, batch, idx):
        """One iteration of the reference loop INCLUDING its gradient accumulation (train/main.py:112-139): backward every
        batch; clip / optimizer step / zero_grad / EMA only when `idx % backprop_freq == 0` (idx = batch index inside the epoch,
        so the first batch of an epoch always steps; accumulated gradients are summed, not averaged); the LR schedule advances
        with every batch.  With backprop_freq == 1 this is `step`."""
        freq = int(getattr(self.args, "backprop_freq", 1))
        if freq <= 1:
            return self.step(batch)
        if not self._params_synced:
            self.sync_parameters()
        if not self._accum_open:
            self.zero_grad()
            self._accum_open = True
        loss_dict = self.forward_backward(batch)
        if idx % freq == 0:
            world = dist.world_size()
            if dist.active():
                dist.allreduce_sum_(self.online.flat_grad())
            self._lr_iter = self.batches_seen
            try:
                self.optimizer_step(grad_scale=1.0 if self.global_negatives else 1.0 / world)
            finally:
                self._lr_iter = None
            self._accum_open = False
        self.batches_seen += 1
        self._resume_bump = 0
        return loss_dict

    def _ddp_buckets(self, tag, layers):
        """[(lo, hi, last_layer)] in the order backward finishes them (last layers first): `bucket_layers` consecutive layers of
        one stack per bucket; the bucket is final once layer `last_layer` (its lowest) is.  Cached per stack."""
        cache = self.__dict__.setdefault("_bucket_cache", {})
        key = (tag, layers, self.ddp_bucket_layers)
        if key not in cache:
            prefix = {"video": "video_temporal_encoder.", "joint": "joint_temporal_encoder."}[tag]
            out, top = [], layers
            while top > 0:
                bot = max(0, top - self.ddp_bucket_layers)
                spans = [self.online.flat_range(f"{prefix}resblocks.{i}.") for i in range(bot, top)]
                lo, hi = min(sp[0] for sp in spans), max(sp[1] for sp in spans)
                assert sum(sp[1] - sp[0] for sp in spans) == hi - lo, "layers of a stack are not contiguous in the flat buffer"
                out.append((lo, hi, bot))
                top = bot
            slo, shi = self.online.flat_range(prefix)
            assert all(slo <= lo and hi <= shi for lo, hi, _ in out)
            cache[key] = out
        return cache[key]

    def _will_chain(self, batch):
        """Whether `forward_backward` will take the two-chain path for this batch (decided before the step touches anything)."""
        if not batch["video"].is_cuda or "text_embed" not in batch:
            return False
        fused = self.fused_loss
        if fused:
            Bn, Nn = batch["text_embed"].shape[:2]
            cols = Bn * Nn
            if batch.get("n_text") is not None and not self.global_negatives:
                cols = min(cols, (batch["n_text"] + 63) // 64 * 64)
            fused = cols <= _lib.lib().tan_simnce_max_cols()
        return self._chains_eligible(batch, fused)

    def step(self, batch):
        """One optimizer step on an already device-resident batch (see to_device_batch): zero_grad -> forward -> loss -> backward ->
        [gradient all-reduce] -> AdamW, train/main.py:81-122.
        With N>1 ranks the flat gradient is summed over RCCL (`_GradReducer`, TAN_DDP_MODE): in pieces behind the two chains, in
        buckets behind per-layer events recorded inside `tan_encoder_bwd`, or in one call; the order of issue is fixed and identical
        on every rank.
        Stage 1 (the two-chain step) is PIPELINED across the step boundary (TAN_STEP_PIPELINE): this call returns with the stacks' last
        weight-gradient launches and the optimizer launches of their matrices still running on their role streams; their events wait
        in `_Flat.pending`, the next step waits for each where it needs it, anything else that touches the parameters goes through
        `_ensure_flat()` / `state_dict()`, which wait for all of them."""
        if getattr(self, "_poisoned", None):
            raise _lib.TanHipError(self._poisoned)
        if not self._params_synced:
            self.sync_parameters()
        fl = self.online._flat
        tfl = self.model.target._flat if self.twin else None       # (stage 2: the optimizer launches write the EMA twin's buffers too)
        batch = self._embed_tokens(batch)      # (the language model's forward, when the step starts from token ids: before anything is decided)
        if self.online.compute_dtype == torch.bfloat16 and batch["video"].is_cuda and "text_embed" in batch:
            batch = self._pad_sentence_slots(batch)      # (before the schedule is decided: stage 2's eligibility depends on the padded N)
        piped = self.pipeline and fl.bound() and (tfl is None or tfl.bound()) and self._will_chain(batch)
        fl.in_step = piped                     # (a pipelined step waits for what the previous one left running itself, piece by piece)
        if tfl is not None:
            tfl.in_step = piped
        self._pipe_out = None
        try:
            dev0 = self.online._ensure_flat().flat.device        # (not pipelined: waits for everything pending)
            aside = None
            if dev0.type == "cuda":      # the gradient fill and the image refresh run on a side stream, next to the forward's first kernels
                from .loss import _side_stream
                aside = _side_stream(dev0)
            self.zero_grad(side=aside)
            world = dist.world_size()
            gscale = 1.0 if self.global_negatives else 1.0 / world
            self._ddp = None
            if dist.active():
                self._ddp = _GradReducer(self)
                if self.ddp_mode == "buckets":
                    self.online._grad_ready_hook = self._ddp.hook
            self._in_step = True               # (forward_backward may hand finished gradients to the optimizer early: only inside step)
            try:
                loss_dict = self.forward_backward(batch)
            except BaseException:
                self._ddp = None
                self._join_role_streams(dev0)          # nothing of the failed step is recorded in `pending`: join what it enqueued
                fl.pending = {}
                if tfl is not None:
                    tfl.pending = {}
                if self.__dict__.pop("_early", None):
                    # `early_update` has already stepped a stack's matrices with this step's count: the parameters are half way between
                    # two steps and a retried step would apply AdamW to them twice -- say so instead of going on
                    self._poisoned = ("a training step failed after the early optimizer launch of a stack: parameters are inconsistent; "
                                      "reload a checkpoint")
                raise
            finally:
                self.online._grad_ready_hook = None
                self._in_step = False
            out_ev = self._pipe_out
            if out_ev is not None:
                # a pipelined two-chain step: what is left for this stream is the remainder of the gradient reduction and the small
                # optimizer launch (pre-projections + everything outside the matrices) -- behind the stacks' last weight-gradient
                # launches, not behind the optimizer launches that follow those on their streams
                # (the last weight-gradient launches only write matrix gradients, which the early launches step; without those this launch
                #  steps the matrices too and waits for them.  Single-call data parallelism reduces everything here: same)
                if not self.__dict__.get("_early"):
                    cur = torch.cuda.current_stream()
                    for k in ("video", "joint"):
                        cur.wait_event(out_ev[k])
                fl.pending = {}                # (every event of the previous step has been waited for inside this one)
                if tfl is not None:
                    tfl.pending = {}
            if self._ddp is not None:
                ev = None
                if self.time_comm:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                self.last_collectives = self._ddp.finish()
                self._ddp = None
                if ev is not None:
                    ev[1].record()
                    self.comm_events.append(ev)
            self.optimizer_step(grad_scale=gscale)
            if out_ev is not None:
                fl.run_image_hooks()           # the LayerNorm'ed position tables: one small launch, on this stream
                fl.pending = dict(out_ev)
                if tfl is not None:            # (the same launches wrote the twin's parameters and images)
                    tfl.run_image_hooks()
                    tfl.pending = dict(out_ev)
            elif aside is not None:
                self.online._ensure_flat_nosync().refresh_images_async(aside, backward=True)
                if self.twin:
                    self.model.target._ensure_flat_nosync().refresh_images_async(aside, backward=False)
        finally:
            fl.in_step = False
            if tfl is not None:
                tfl.in_step = False
            self._pipe_out = None
        self.batches_seen += 1
        self._resume_bump = 0
        return loss_dict

    def _join_role_streams(self, dev):
        """The current stream waits for everything enqueued on the step's role streams (the failure path of `step`)."""
        if dev.type != "cuda":
            return
        from .loss import _side_stream
        cur = torch.cuda.current_stream()
        streams = [_lib.role_stream(dev, r) for r in ("loss", "opt")] + [_side_stream(dev), self.online._side_stream(dev)]
        for st in streams:
            if st is not None and st.cuda_stream != cur.cuda_stream:
                cur.wait_stream(st)

    def allreduce_alone_ms(self, reps=5):
        """bench.py: the gradient collectives of one step issued back to back on an otherwise idle GPU (the same ranges as `step`
        reduces: the buckets of both stacks + the remainder, or the whole flat gradient in 'flat' mode) -- the communication time a
        step has to hide, measured without anything to hide it behind.  The gradient buffer is scaled back afterwards."""
        if not dist.active():
            return None
        flat = self.online.flat_grad()
        if self.ddp_mode == "single" or (self.ddp_mode == "flat" and not self.__dict__.get("_last_step_chains")):
            ranges = [(0, flat.numel())]
        elif self.ddp_mode == "flat":
            ranges = [self.online.flat_range("video_temporal_encoder."), self.online.flat_range("joint_temporal_encoder.")]
            ranges += uncovered_ranges(sorted(ranges), flat.numel())
        else:
            ranges = []
            for tag, layers in (("video", self.online.num_encoder_layers), ("joint", self.online.num_decoder_layers)):
                ranges += [(lo, hi) for lo, hi, _ in self._ddp_buckets(tag, layers)]
            ranges += uncovered_ranges(sorted(ranges), flat.numel())
        keep = flat.clone()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for lo, hi in ranges:
                dist.allreduce_sum_(flat[lo:hi])
        e1.record()
        torch.cuda.synchronize()
        flat.copy_(keep)
        return e0.elapsed_time(e1) / reps, len(ranges), sum(hi - lo for lo, hi in ranges) * 4
