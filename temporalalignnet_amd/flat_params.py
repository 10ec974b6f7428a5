"""Flat parameter storage of one TemporalAligner: every parameter is a view into ONE f32 buffer (+ one flat f32 gradient buffer
that `p.grad` aliases: one RCCL all-reduce bucket, one fused AdamW launch), plus the bf16 images the throughput mode computes
with -- the shadow copy, the transposes of the encoder Linear weights (K-contiguous operand of the dX GEMMs) and the
tan_pack_weights images of both (row-panel kernels).  Each derived image carries the epoch of the shadow it was built from."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib, ops

_ALIGN = 8  # flat offsets are multiples of 8 elements (16-byte aligned bf16 / 32-byte f32 views)


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _Flat:
    """Flat storage behind the aligner's own parameters (language model excluded)."""

    def __init__(self, owner: nn.Module, named):
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.off, total = {}, 0
        for n, p in named:
            self.off[n] = (total, p.numel(), tuple(p.shape))
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = total
        self.flat = self.grad = self.shadow = None
        self._views = {}
        self.shadow_version = -1
        # transposed bf16 copies of the encoder Linear weights (K-contiguous operand of the dX GEMMs), refreshed lazily:
        # shadow_epoch counts refreshes of `shadow`, shadow_t_epoch the epoch `shadow_t` was built from
        self.shadow_t, self.shadow_t_table, self.shadow_epoch, self.shadow_t_epoch = None, None, 0, -1
        # tan_pack_weights images (row-panel kernels) of the same weights / of their transposes, same element offsets
        self.shadow_p, self.shadow_tp, self.pack_table, self.shadow_p_epoch, self.shadow_tp_epoch = None, None, None, -1, -1
        self.device = None
        # more per-step images owned by the model (callables, run inside refresh_images_async on the side stream): the LayerNorm'ed
        # position tables of the fused input embeddings
        self.image_hooks = []
        # Events of work that `Trainer.step` left running on its role streams when it returned (the stacks' last weight-gradient launches
        # and the optimizer launches behind them: the NEXT step waits for each exactly where it needs it).  Anything else that touches the
        # buffers goes through `drain()` first -- `_ensure_flat()` calls it -- which makes the current stream wait for all of them.
        self.pending = {}
        self.in_step = False

    def drain(self):
        if self.pending and not self.in_step:
            cur = torch.cuda.current_stream()
            for e in self.pending.values():
                if e is not None:
                    cur.wait_event(e)
            self.pending = {}

    def bound(self):
        p0, pl = self.params[0], self.params[-1]
        return (self.flat is not None and p0.device == self.flat.device
                and p0.data_ptr() == self.flat.data_ptr()
                and pl.data_ptr() == self.flat.data_ptr() + 4 * self.off[self.names[-1]][0])

    def bind(self, want_shadow: bool):
        """(Re)build the flat buffers from the current parameter values and alias every parameter to its slice."""
        dev = self.params[0].device
        flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for n, p in zip(self.names, self.params):
            o, k, shp = self.off[n]
            flat[o:o + k].view(shp).copy_(p.data)
        old_grads = [p.grad for p in self.params]
        self.flat, self.grad, self.device = flat, torch.zeros_like(flat), dev
        for n, p, g in zip(self.names, self.params, old_grads):
            o, k, shp = self.off[n]
            p.data = flat[o:o + k].view(shp)
            if g is not None:
                self.grad[o:o + k].view(shp).copy_(g)
                p.grad = self.grad[o:o + k].view(shp)
        self.shadow = torch.empty(self.total, dtype=torch.bfloat16, device=dev) if want_shadow else None
        self.shadow_version = -1
        self._views = {}
        self.shadow_t, self.shadow_t_table, self.shadow_t_epoch = None, None, -1
        self.shadow_p, self.shadow_tp, self.pack_table, self.shadow_p_epoch, self.shadow_tp_epoch = None, None, None, -1, -1
        self._image_table = None

    def _pack_tables(self):
        """device tables of tan_pack_entry for the MLP weights of every block ([out, in]) and for their transposes ([in, out])"""
        if self.pack_table is None:
            names = [n for n in self.names if ".resblocks." in n and len(self.off[n][2]) == 2]
            pre = self._packable_pre()                                                                # tan_embed_fwd's operands

            def table(transposed):
                ents, mx = [], 0
                for n in names + ([] if transposed else pre):
                    o, _, (N, K) = self.off[n]
                    if transposed:
                        # (all four have a row-panel consumer in the backward: the MLP's two dX GEMMs, the out_proj dX tail, the
                        #  in_proj dX head)
                        N, K = K, N
                    TN, TK = (512, 16) if N == 512 else (256, 32)
                    if not transposed and n.endswith("attn.in_proj_weight"):
                        TN, TK = 384, 32                                # the "qkv16" format of tan_attnblk_fwd (include/tan_hip.h)
                    ents.append(_lib.PackEntry(o, o, N, K, TN, TK))
                    mx = max(mx, (N // TN) * (K // TK))
                arr = (_lib.PackEntry * len(ents))(*ents)
                dev_t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.shadow.device)
                return dev_t, len(ents), mx
            self.pack_table = (table(False), table(True))
        return self.pack_table

    def _packable_pre(self):
        """The pre-projection matrices that get a packed image: feature widths in whole 64-blocks (what `image_table` and the packer
        tile by).  Other widths (a constructor argument: SURVEY 8(d) config 5) train through the tiled-GEMM front-end -- the fused
        embedding launch asks for multiples of 128 (`_embed_fused_ok`) -- and are stepped with the rest of the parameters."""
        return [n for n in ("video_pre_proj.weight", "text_pre_proj.weight")
                if n in self.off and self.off[n][2][0] % 64 == 0 and self.off[n][2][1] % 64 == 0]

    def image_table(self):
        """For tan_adamw_step_images: (device table of tan_image_entry, device unit prefix, n entries, n units, [(lo, hi)] flat ranges
        of the matrices) over every matrix that has a packed image -- the optimizer launch then writes the shadow, the W^T copies and
        both packed images itself.  Allocates / builds the images once (the kernels that rebuild them lazily stay the fallback)."""
        if getattr(self, "_image_table", None) is None:
            self.sync_shadow_p(); self.sync_shadow_t(); self.sync_shadow_tp()
            import numpy as np
            ents, prefix, ranges = [], [0], []
            mats = [n for n in self.names if ".resblocks." in n and len(self.off[n][2]) == 2]
            pre = self._packable_pre()
            for n in mats + pre:
                o, k, (N, K) = self.off[n]
                assert N % 64 == 0 and K % 64 == 0, (n, N, K)
                tn_w, tk_w = (384, 32) if n.endswith("attn.in_proj_weight") else ((512, 16) if N == 512 else (256, 32))
                tn_t, tk_t = (0, 0) if n in pre else ((512, 16) if K == 512 else (256, 32))      # W^T is [K][N]: its "N" is K
                ents.append(_lib.ImageEntry(o, N, K, tn_w, tk_w, tn_t, tk_t))
                prefix.append(prefix[-1] + (N // 64) * (K // 64))
                ranges.append((o, o + k))
            arr = (_lib.ImageEntry * len(ents))(*ents)
            dev = self.shadow.device
            tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            pre_t = torch.from_numpy(np.asarray(prefix, dtype=np.int64)).to(dev)
            # (units of the video stack's matrices come first: `Trainer` may update them before the joint stack's backward is through)
            n_video = sum(1 for n in mats if n.startswith("video_temporal_encoder."))
            assert all(n.startswith("video_temporal_encoder.") for n in mats[:n_video])
            self.video_units = prefix[n_video]
            self.mats_units = prefix[len(mats)]            # ... then the joint stack's; the pre-projections are the last units
            self._image_table = (tab, pre_t, len(ents), prefix[-1], ranges)
        return self._image_table

    def images_rewritten(self, transposes=True):
        """tan_adamw_step_images has just rewritten the shadow AND the images built from it."""
        self.shadow_version = self.flat._version
        self.shadow_epoch += 1
        self.shadow_p_epoch = self.shadow_epoch
        if transposes:
            self.shadow_t_epoch = self.shadow_epoch
            self.shadow_tp_epoch = self.shadow_epoch

    def sync_shadow_p(self):
        """(Re)build the packed images of every 2-D `...resblocks.*` weight from the bf16 shadow: one launch."""
        if self.shadow is None:
            return None
        if self.shadow_p is None:
            self.shadow_p = torch.zeros(self.total, dtype=torch.bfloat16, device=self.shadow.device)
        if self.shadow_p_epoch != self.shadow_epoch:
            tab, n, mx = self._pack_tables()[0]
            _lib.check(_lib.lib().tan_pack_weights(_vp(self.shadow), _vp(self.shadow_p), _vp(tab), C.c_int(n), C.c_int(mx),
                                                   ops._stream()), "tan_pack_weights")
            self.shadow_p_epoch = self.shadow_epoch
        return self.shadow_p

    def refresh_images_async(self, side, backward=True):
        """After an optimizer step: rebuild every image derived from the bf16 shadow (packed tiles; W^T copies and their packed tiles
        when a backward will need them) on `side`, next to whatever the caller's stream does before its first encoder kernel (the
        input embeddings of the next step).  Consumers call `join_images` on the stream they launch on."""
        if self.shadow is None:
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.sync_shadow_p()
            if backward and self.shadow_t is not None:
                self.sync_shadow_t()
                if self.shadow_tp is not None:
                    self.sync_shadow_tp()
            for hook in self.image_hooks:
                hook()
            self.images_event = side.record_event()

    def run_image_hooks(self):
        """the model's per-step images on the CURRENT stream (a pipelined step: the side stream still carries an optimizer launch)"""
        for hook in self.image_hooks:
            hook()
        self.images_event = None

    def join_images(self):
        ev = getattr(self, "images_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)      # a finished event costs nothing on the GPU

    def sync_shadow(self):
        if self.shadow is not None and self.shadow_version != self.flat._version:
            ops.cast(self.flat, self.shadow)
            self.shadow_rewritten()

    def shadow_rewritten(self):
        """A kernel (the cast above, tan_adamw_step, tan_ema_update) has just rewritten the bf16 shadow from the f32 masters:
        the shadow is current, and every image derived from it (W^T copies, packed tiles) is stale."""
        if self.shadow is not None:
            self.shadow_version = self.flat._version
            self.shadow_epoch += 1

    def sync_shadow_tp(self):
        """Packed images of the TRANSPOSED MLP weights (row-panel backward): tan_pack_weights over `shadow_t`, same offsets."""
        if self.shadow is None or self.shadow_t is None:
            return None
        if self.shadow_tp is None:
            self.shadow_tp = torch.zeros(self.total, dtype=torch.bfloat16, device=self.shadow.device)
        if self.shadow_tp_epoch != self.shadow_t_epoch:
            tab, n, mx = self._pack_tables()[1]
            _lib.check(_lib.lib().tan_pack_weights(_vp(self.shadow_t), _vp(self.shadow_tp), _vp(tab), C.c_int(n), C.c_int(mx),
                                                   ops._stream()), "tan_pack_weights")
            self.shadow_tp_epoch = self.shadow_t_epoch
        return self.shadow_tp

    def sync_shadow_t(self):
        """(Re)build the transposed copies of every 2-D `...resblocks.*` weight from the bf16 shadow: one batched launch."""
        if self.shadow is None:
            return None
        if self.shadow_t is None:
            names = [n for n in self.names if ".resblocks." in n and len(self.off[n][2]) == 2]
            rows = [[self.off[n][0], self.off[n][2][0], self.off[n][2][1]] for n in names]
            self.shadow_t = torch.zeros(self.total, dtype=torch.bfloat16, device=self.shadow.device)
            self.shadow_t_table = (torch.tensor(rows, dtype=torch.int64).to(self.shadow.device), len(rows),
                                   max(r[1] for r in rows), max(r[2] for r in rows))
        if self.shadow_t_epoch != self.shadow_epoch:
            table, n, mr, mc = self.shadow_t_table
            _lib.check(_lib.lib().tan_transpose_batch(_vp(self.shadow), _vp(self.shadow_t), _vp(table), C.c_int(n), C.c_long(mr),
                                                      C.c_long(mc), C.c_int(_lib.TAN_BF16), ops._stream()), "tan_transpose_batch")
            self.shadow_t_epoch = self.shadow_epoch
        return self.shadow_t

    def view(self, buf, name):
        """Slice `name` of a flat buffer.  Cached: a train step asks for ~500 of these, and building each narrow+view pair
        was a fifth of the host time of the step."""
        o, k, shp = self.off[name]
        if buf is not self.flat and buf is not self.grad and buf is not self.shadow:
            return buf[o:o + k].view(shp)            # a caller's own buffer (e.g. a clone of the gradient): never cached
        key = (buf.data_ptr(), name)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = buf[o:o + k].view(shp)
        return v

    def ptr(self, buf, name):
        """device address of slice `name` of a flat buffer (no tensor view is built)"""
        return buf.data_ptr() + self.off[name][0] * buf.element_size()
