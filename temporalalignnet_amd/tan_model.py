"""MI355X-native TemporalAligner / TwinTemporalAligner with the reference's Python surface.

Drop-in boundary: same constructor arguments, `forward` signature and returned dict keys,
`get_visual_feature / get_joint_feature / get_textual_feature(_with_time) / get_text_visual_sim_{joint,dual} /
get_alignability`, `TwinTemporalAligner.{forward, forward_from_ema, _momentum_update, _copy_param}` and `state_dict`
keys as reference model/tan_model.py.  The interface drift of the released reference is resolved as a superset:
`lang_model` is an alias of `bert`, `abs_text_pos=` is accepted (and ignored, as nothing in the model reads it),
`get_text_visual_sim` aliases `get_text_visual_sim_joint`.

Underneath there is no ATen arithmetic: parameters live in one flat f32 buffer (+ flat f32 gradient, + bf16 shadow in
throughput mode) and every op is a hand-written HIP kernel of libtan_hip.so (include/tan_hip.h).  The whole forward is
ONE autograd node; its backward runs the HIP backward and accumulates parameter gradients straight into the flat
gradient buffer that `p.grad` views alias (one RCCL all-reduce bucket, one fused AdamW launch).

compute_dtype: 'fp32' = parity mode (exact-f32 MFMA, matches the reference CPU path to ~1e-6);
               'bf16' = throughput mode (bf16 operands / activations, f32 accumulation, statistics and logits).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .tfm_model import TemporalEncoder, _LayerNormParams, _LinearParams, get_position_embedding_sine

WIDTH, HEADS = 512, 8
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

    def _pack_tables(self):
        """device tables of tan_pack_entry for the MLP weights of every block ([out, in]) and for their transposes ([in, out])"""
        if self.pack_table is None:
            names = [n for n in self.names if ".resblocks." in n and len(self.off[n][2]) == 2]

            def table(transposed):
                ents, mx = [], 0
                for n in names:
                    o, _, (N, K) = self.off[n]
                    if transposed:
                        if ".mlp.c_" not in n:                          # (no row-panel consumer of the attention weights' transposes yet)
                            continue
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


class _Blocks:
    """Contiguous activation blocks carved out of one allocation."""

    def __init__(self, dtype, device, sizes: dict):
        self.off, total = {}, 0
        for k, n in sizes.items():
            self.off[k] = (total, n)
            total += (n + 63) // 64 * 64
        self.buf = torch.empty(total, dtype=dtype, device=device)

    def __getitem__(self, k):
        o, n = self.off[k]
        return self.buf[o:o + n]


class _EncRun:
    """Saved activations of one encoder stack + its ctypes descriptor."""

    def __init__(self, model, prefix, layers, B, L, cd, dev):
        R, Cw = B * L, WIDTH
        self.prefix, self.layers, self.B, self.L, self.R = prefix, layers, B, L, R
        per = {"xn1": R * Cw, "qkv": R * 3 * Cw, "attn_o": R * Cw, "x_mid": R * Cw, "xn2": R * Cw, "h_pre": R * 4 * Cw,
               "h_act": R * 4 * Cw, "x_out": R * Cw}
        self.act = _Blocks(cd, dev, {f"{i}.{k}": n for i in range(layers) for k, n in per.items()} | {"post": R * Cw})
        st = {"mean1": R, "rstd1": R, "mean2": R, "rstd2": R, "lse": B * HEADS * L}
        self.stat = _Blocks(torch.float32, dev, {f"{i}.{k}": n for i in range(layers) for k, n in st.items()}
                            | {"post_mean": R, "post_rstd": R})
        self.bufs = (_lib.LayerBufs * layers)()
        for i in range(layers):
            for k in per:
                setattr(self.bufs[i], k, self.act[f"{i}.{k}"].data_ptr())
            for k in st:
                setattr(self.bufs[i], k, self.stat[f"{i}.{k}"].data_ptr())

    def stage(self, s):
        """[R, C] deep-supervision output s (tfm_model.py:48-55)."""
        if s < self.layers - 1:
            return self.act[f"{s + 1}.xn1"].view(self.R, WIDTH)
        return self.act["post"].view(self.R, WIDTH)


class _AlignerFn(torch.autograd.Function):
    """The whole TemporalAligner.forward as one autograd node (HIP forward, HIP backward)."""

    @staticmethod
    def forward(ctx, model, video, lang, vmask_u8, tmask_u8, opts, *anchor):
        # `anchor`: ONE trainable parameter, only there to make this node require grad.  The parameter gradients are not returned
        # through autograd (backward adds them into the flat gradient buffer every p.grad aliases): with all ~160 parameters as
        # inputs the engine evaluated 160 AccumulateGrad nodes with undefined gradients per step (~0.25 ms of host time).
        ctx.set_materialize_grads(False)      # unused outputs must not materialise 100s of MB of zero gradients
        run = model._run_forward(video, lang, vmask_u8, tmask_u8, opts)
        # the returned tensor OBJECTS must not stay reachable from ctx: tensor -> grad_fn (this node, held from C++) -> ctx -> run
        # -> tensor is a cycle Python's gc cannot see -- it kept every step's whole activation record alive (1.3 MB per video)
        outputs = tuple(run.pop("outputs"))
        ctx.model, ctx.run = model, run
        ctx.lang_requires_grad = lang.requires_grad
        ctx.n_anchor = len(anchor)
        return outputs

    @staticmethod
    def backward(ctx, *grads):
        model = ctx.model
        d_lang = model._run_backward(ctx.run, grads, ctx.lang_requires_grad)
        return (None, None, d_lang, None, None, None) + (None,) * ctx.n_anchor


class TemporalAligner(nn.Module):
    def __init__(self, num_encoder_layers=2, num_decoder_layers=2, sim="cos", language_model="word2vec",
                 pos_enc="learned", use_text_pos_enc=0, return_dual_feature=1, random_pos_start=1,
                 use_alignability_head=0, *, compute_dtype="fp32", d_video=1024):
        super().__init__()
        self.num_encoder_layers = num_encoder_layers
        self.num_decoder_layers = num_decoder_layers
        self.sim = sim
        self.pos_enc = pos_enc
        self.language_model = language_model
        self.use_text_pos_enc = use_text_pos_enc
        self.return_dual_feature = return_dual_feature
        self.random_pos_start = random_pos_start
        self.use_alignability_head = use_alignability_head
        self.compute_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, torch.float32: torch.float32,
                              torch.bfloat16: torch.bfloat16}[compute_dtype]
        if num_encoder_layers < 1 or num_decoder_layers < 1:
            raise NotImplementedError("the HIP path needs >= 1 layer per stack (reference edge case tan_model.py:177-179)")

        if language_model == "word2vec":
            from .word2vec_model import Word2VecModel
            self.bert = Word2VecModel(compute_dtype=compute_dtype)
        elif language_model in (None, "none"):
            self.bert = None
        else:
            raise NotImplementedError(f"language_model={language_model!r}: only 'word2vec' (tan_model.py:39-40) or None")
        text_embed_dim = 512

        self.video_temporal_encoder = TemporalEncoder(width=WIDTH, layers=num_encoder_layers, heads=HEADS)
        self.joint_temporal_encoder = TemporalEncoder(width=WIDTH, layers=num_decoder_layers, heads=HEADS)
        self.video_pre_proj = _LinearParams(d_video, WIDTH, bias=False)
        self.text_pre_proj = _LinearParams(text_embed_dim, WIDTH, bias=False)
        self.ln_text_init = _LayerNormParams(WIDTH)
        self.ln_video_init = _LayerNormParams(WIDTH)
        self.ln_position_init = _LayerNormParams(WIDTH)
        self.ln_video_post_enc = _LayerNormParams(WIDTH)
        self.ln_joint_post_enc = _LayerNormParams(WIDTH)
        if pos_enc == "learned":
            self.temporal_pos_embed = nn.Parameter(torch.empty(1024, WIDTH))
            nn.init.normal_(self.temporal_pos_embed, std=0.01)
        elif pos_enc == "sine":
            self.register_buffer("temporal_pos_embed", get_position_embedding_sine(WIDTH, 1024))
        else:
            raise ValueError(pos_enc)
        self.text_temporal_pos_embed = nn.Parameter(torch.empty(1024, WIDTH))
        nn.init.normal_(self.text_temporal_pos_embed, std=0.01)
        self.mlp = _LinearParams(WIDTH, WIDTH)      # never used in any forward (tan_model.py:68)
        if use_alignability_head:
            self.binary_head = _LinearParams(WIDTH, 1)
            nn.init.normal_(self.binary_head.weight, std=0.01)
            nn.init.zeros_(self.binary_head.bias)
        self.initialize_parameters()
        # reference registration order differs from construction order only for the two pos-embeds, which the
        # reference registers before the encoders' parameters appear in state_dict(); key *names* are what matter.
        self._flat = _Flat(self, [(n, p) for n, p in self.named_parameters() if not n.startswith("bert.")])
        self._ws_pool, self._ws_lru, self._ws_tick = {}, {}, 0
        import threading
        self._ws_lock = threading.Lock()
        self.overlap_stacks = True        # run the video and joint stacks on two HIP streams
        self._side = None
        self._issuer = None               # helper thread issuing the side-stream stack (see _on_side)
        self._lp_cache = {}
        self.transposed_dx = os.environ.get("TAN_TRANSPOSED_DX", "1") != "0"   # dX GEMMs read W^T copies (K-contiguous)
        self.panel_kernels = os.environ.get("TAN_PANEL", "1") != "0"           # row-panel fused kernels (packed weight images)
        self._grad_ready_hook = None      # callable(tag, layer_events) fired inside backward once a stack's backward is enqueued
        # load_state_dict copies into the parameter tensors, whose version counters are not the flat buffer's: the bf16
        # shadow (and the W^T copies built from it) must be rebuilt from the f32 masters on the next forward
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_shadow())

    def invalidate_shadow(self):
        """Call after writing parameters in place other than through the optimizer kernel / load_state_dict (e.g.
        `p.data.copy_(...)`): the next forward re-casts the bf16 shadow weights from the f32 masters."""
        f = self.__dict__.get("_flat")
        if f is not None:
            f.shadow_version = -1

    # ------------------------------------------------------------------ init (tan_model.py:76-97)
    def initialize_parameters(self):
        nn.init.normal_(self.video_pre_proj.weight, std=0.01)
        nn.init.normal_(self.text_pre_proj.weight, std=0.01)
        nn.init.normal_(self.mlp.weight, std=0.01)
        nn.init.zeros_(self.mlp.bias)
        w, layers = self.joint_temporal_encoder.width, self.joint_temporal_encoder.layers
        proj_std = (w ** -0.5) * ((2 * layers) ** -0.5)
        attn_std = w ** -0.5
        fc_std = (2 * w) ** -0.5
        for enc in (self.video_temporal_encoder, self.joint_temporal_encoder):
            for blk in enc.resblocks:
                nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
                nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
                nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
                nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)

    @property
    def lang_model(self):          # train/main.py:58,174,202 spelling
        return self.bert

    # ------------------------------------------------------------------ flat-buffer plumbing
    def _autograd_anchor(self):
        f = self._flat
        for p in f.params:
            if p.requires_grad:
                return (p,)
        return ()

    def _ensure_flat(self):
        f = self._flat
        if not f.bound():
            if not f.params[0].is_cuda:
                raise _lib.TanHipError("TemporalAligner parameters are on the CPU: the HIP path has no CPU fallback; "
                                       "call .cuda() first")
            f.bind(want_shadow=self.compute_dtype == torch.bfloat16)
        f.sync_shadow()
        if self.panel_kernels:
            f.sync_shadow_p()
        return f

    def flat_parameters(self):
        return self._ensure_flat().flat

    def flat_grad(self):
        return self._ensure_flat().grad

    def flat_range(self, prefix):
        """[lo, hi) of the flat buffers covered by the parameters whose names start with `prefix` (contiguous by
        construction: registration order groups each encoder stack)."""
        f = self._flat
        spans = [(f.off[n][0], f.off[n][0] + f.off[n][1]) for n in f.names if n.startswith(prefix)]
        lo, hi = min(s[0] for s in spans), max(s[1] for s in spans)
        hi = (hi + _ALIGN - 1) // _ALIGN * _ALIGN
        inside = [n for n in f.names if lo <= f.off[n][0] < hi]
        assert all(n.startswith(prefix) for n in inside), "parameters of this prefix are not contiguous in the flat buffer"
        return lo, min(hi, f.total)

    def _w(self, name):
        """weight in compute dtype (bf16 shadow or the f32 master)"""
        f = self._flat
        return f.view(f.shadow if self.compute_dtype == torch.bfloat16 else f.flat, name)

    def _f(self, name):
        return self._flat.view(self._flat.flat, name)

    def _g(self, name):
        return self._flat.view(self._flat.grad, name)

    def _bind_grads(self):
        """Make every p.grad alias its slice of the flat gradient (zeroing the buffer if grads were None)."""
        f = self._flat
        # fast path (every training step after the first): three probes (first / middle / last trainable parameter) still alias
        # the flat gradient this method bound them to
        ends = self.__dict__.get("_grad_ends")
        if ends is None:
            tr = [(n, p) for n, p in zip(f.names, f.params) if p.requires_grad]
            ends = self.__dict__["_grad_ends"] = (tr[0], tr[len(tr) // 2], tr[-1]) if tr else ()
        if ends and all(p.grad is not None and p.grad.data_ptr() == f.grad.data_ptr() + 4 * f.off[n][0] for n, p in ends) \
                and self.__dict__.get("_grads_bound_to") == f.grad.data_ptr():
            return
        fresh = all(p.grad is None for p in f.params if p.requires_grad)
        if fresh:
            f.grad.zero_()
        for n, p in zip(f.names, f.params):
            if not p.requires_grad:
                continue
            gv = f.view(f.grad, n)
            if p.grad is None:
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv
        self.__dict__["_grads_bound_to"] = f.grad.data_ptr()

    def param_modes(self, name_prefix=""):
        """u8 per-element optimizer mode for tan_adamw_step: 1 decay / 0 no decay by the reference's substring rule on the
        FULL parameter name (train/main.py:332,340-343) / 2 never receives a gradient (torch skips .grad-is-None params)."""
        f = self._flat
        mode = torch.zeros(f.total, dtype=torch.uint8)
        unused = {"mlp.weight", "mlp.bias"}
        if not self.use_text_pos_enc:
            unused.add("text_temporal_pos_embed")
        for n in f.names:
            o, k, _ = f.off[n]
            full = name_prefix + n
            if n in unused:
                mode[o:o + k] = 2
            elif any(tok in full for tok in (".ln_", ".bias", ".logit_scale", ".entropy_scale")):
                mode[o:o + k] = 0
            else:
                mode[o:o + k] = 1
        return mode

    # ------------------------------------------------------------------ encoder descriptors
    def _layer_params(self, prefix, layers):
        """tan_layer_params[layers] of one stack: addresses into the flat parameter / gradient / bf16-shadow buffers.  Built by
        pointer arithmetic and cached until one of the three buffers is re-allocated."""
        f = self._flat
        wbuf = f.shadow if self.compute_dtype == torch.bfloat16 else f.flat
        wt = f.shadow_t if (self.compute_dtype == torch.bfloat16 and self.transposed_dx) else None
        wp = f.shadow_p if (self.compute_dtype == torch.bfloat16 and self.panel_kernels) else None
        wtp = f.shadow_tp if (wt is not None and wp is not None) else None
        sig = (f.flat.data_ptr(), f.grad.data_ptr(), wbuf.data_ptr(), wt.data_ptr() if wt is not None else 0,
               wp.data_ptr() if wp is not None else 0, wtp.data_ptr() if wtp is not None else 0)
        hit = self._lp_cache.get((prefix, layers))
        if hit is not None and hit[0] == sig:
            return hit[1]
        arr = (_lib.LayerParams * layers)()
        m = {"w_qkv": "attn.in_proj_weight", "w_out": "attn.out_proj.weight", "w_fc": "mlp.c_fc.weight",
             "w_proj": "mlp.c_proj.weight"}
        fm = {"b_qkv": "attn.in_proj_bias", "b_out": "attn.out_proj.bias", "b_fc": "mlp.c_fc.bias",
              "b_proj": "mlp.c_proj.bias", "ln1_g": "ln_1.weight", "ln1_b": "ln_1.bias", "ln2_g": "ln_2.weight",
              "ln2_b": "ln_2.bias"}
        for i in range(layers):
            base = f"{prefix}.resblocks.{i}."
            for k, v in m.items():
                setattr(arr[i], k, f.ptr(wbuf, base + v))
                setattr(arr[i], "g_" + k, f.ptr(f.grad, base + v))
                setattr(arr[i], "wt_" + k[2:], f.ptr(wt, base + v) if wt is not None else None)
                setattr(arr[i], "wp_" + k[2:], f.ptr(wp, base + v) if wp is not None else None)
                if k in ("w_fc", "w_proj"):
                    setattr(arr[i], "wtp_" + k[2:], f.ptr(wtp, base + v) if wtp is not None else None)
            for k, v in fm.items():
                setattr(arr[i], k, f.ptr(f.flat, base + v))
                setattr(arr[i], "g_" + k, f.ptr(f.grad, base + v))
        self._lp_cache[(prefix, layers)] = (sig, arr)
        return arr

    def _enc_desc(self, er: _EncRun, x0, keypad, post_name):
        d = _lib.EncoderDesc()
        d.dtype = ops._dt(x0)
        d.B, d.L, d.C, d.H, d.layers = er.B, er.L, WIDTH, HEADS, er.layers
        d.key_padding_mask = _vp(keypad)
        d.x0 = _vp(x0)
        er.params = self._layer_params(er.prefix, er.layers)
        d.params = er.params
        d.bufs = er.bufs
        d.post_g, d.post_b = _vp(self._f(post_name + ".weight")), _vp(self._f(post_name + ".bias"))
        d.g_post_g, d.g_post_b = _vp(self._g(post_name + ".weight")), _vp(self._g(post_name + ".bias"))
        d.post_out = _vp(er.act["post"])
        d.post_mean, d.post_rstd = _vp(er.stat["post_mean"]), _vp(er.stat["post_rstd"])
        return d

    def _encoder_fwd(self, er, x0, keypad, post_name, save=False):
        d = self._enc_desc(er, x0, keypad, post_name)
        d.no_save = 0 if save else 1         # no backward will follow: the tensors kept only for it are not written
        _lib.check(_lib.lib().tan_encoder_fwd(C.byref(d), ops._stream()), "tan_encoder_fwd")

    def _layer_events(self, prefix, layers):
        """tan_event handles (one per layer of a stack, created once) for tan_encoder_desc.layer_done; only used while a
        `_grad_ready_hook` is installed (data-parallel training)."""
        cache = self.__dict__.setdefault("_layer_event_cache", {})
        evs = cache.get(prefix)
        if evs is None or len(evs) != layers:
            evs = []
            for _ in range(layers):
                h = C.c_void_p()
                _lib.check(_lib.lib().tan_event_create(C.byref(h)), "tan_event_create")
                evs.append(h.value)
            cache[prefix] = evs
        return evs

    def _encoder_bwd(self, er, x0, keypad, post_name, d_stage, d_x0):
        cd, dev, R = x0.dtype, x0.device, er.R
        d = self._enc_desc(er, x0, keypad, post_name)
        if self._grad_ready_hook is not None:
            evs = self._layer_events(er.prefix, er.layers)
            ev_arr = (C.c_void_p * er.layers)(*evs)
            d.layer_done = ev_arr
        scr = self._take_scratch(R, cd, dev)       # stream-ordered reuse: one backward at a time per model
        d.scr_dx, d.scr_dx2, d.scr_do, d.scr_dxn = (_vp(scr[k]) for k in ("dx", "dx2", "do", "dxn"))
        d.scr_dh, d.scr_dqkv = _vp(scr["dh"]), _vp(scr["dqkv"])
        d.ln_ws = _vp(scr.ln_ws)
        d.dw_ws, d.dw_ws_floats = _vp(scr.dw_ws), scr.dw_ws.numel()
        arr = (C.c_void_p * er.layers)(*[(t.data_ptr() if t is not None else None) for t in d_stage])
        d.d_stage = arr
        d.d_x0 = _vp(d_x0)
        _lib.check(_lib.lib().tan_encoder_bwd(C.byref(d), ops._stream()), "tan_encoder_bwd")

    # ------------------------------------------------------------------ embedding front-ends
    def _pos_table(self, which):
        return self._f(which) if (which != "temporal_pos_embed" or self.pos_enc == "learned") else self.temporal_pos_embed

    def _pos_ln(self, which, n, start, interpolate_from, cd, keep):
        """ln_position_init(pos[start:start+n]) (or the linearly interpolated table) as [n, C] in compute dtype; f32 inside."""
        dev = self._flat.flat.device
        table = self._pos_table(which)
        if interpolate_from:
            src = table[0:interpolate_from].contiguous()
            pos = torch.empty(n, WIDTH, device=dev)
            ops.interp_linear(src, pos, interpolate_from, n, WIDTH)
        else:
            pos = table[start:start + n]
        out = torch.empty(n, WIDTH, device=dev)
        mean, rstd = torch.empty(n, device=dev), torch.empty(n, device=dev)
        ops.layernorm_fwd(pos, self._f("ln_position_init.weight"), self._f("ln_position_init.bias"), out, mean, rstd)
        out_c = out if cd == torch.float32 else ops.cast(out, torch.empty(n, WIDTH, device=dev, dtype=cd))
        saved = {"which": which, "n": n, "start": start, "interp": interpolate_from, "pos": pos, "mean": mean, "rstd": rstd}
        return out_c, saved

    def _pos_ln_bwd(self, saved, d_out_c):
        """backward of _pos_ln: d_out_c [n, C] compute dtype -> accumulate into the table / ln_position_init grads."""
        n, dev = saved["n"], d_out_c.device
        d_out = d_out_c if d_out_c.dtype == torch.float32 else ops.cast(d_out_c, torch.empty(n, WIDTH, device=dev))
        d_pos = torch.empty(n, WIDTH, device=dev)
        ops.layernorm_bwd(d_out, saved["pos"], self._f("ln_position_init.weight"), saved["mean"], saved["rstd"], d_pos,
                          self._g("ln_position_init.weight"), self._g("ln_position_init.bias"))
        which = saved["which"]
        if which == "temporal_pos_embed" and self.pos_enc != "learned":
            return
        g = self._g(which)
        if saved["interp"]:
            ops.interp_linear_bwd(d_pos, g, saved["interp"], n, WIDTH)
        else:
            ops.rows_copy(d_pos, g[saved["start"]:saved["start"] + n], 1, n, WIDTH, n, 0, n, 0, accumulate=True)

    def _draw(self, n, interpolate_from):
        """np.random draw of the position offset, in the reference's order and from its global RNG (tan_model.py:163,195,224)."""
        if interpolate_from or not self.random_pos_start:
            return 0
        return int(np.random.randint(0, int(n / 2)))

    # ------------------------------------------------------------------ the HIP forward
    def _prep(self, x):
        """input features -> contiguous compute-dtype device tensor"""
        cd = self.compute_dtype
        if not x.is_cuda:
            raise _lib.TanHipError("TemporalAligner needs device tensors: the HIP path has no CPU fallback")
        x = x.detach().float().contiguous()
        if cd != torch.float32:
            x = ops.cast(x, torch.empty(x.shape, dtype=cd, device=x.device))
        return x

    def _prep_inputs(self, video, lang):
        return self._prep(video), (self._prep(lang) if lang is not None else None)

    def _video_embed(self, video_c, pos_start, interpolate_from, keep):
        """x0 = ln_video_init(video_pre_proj(video)) + ln_position_init(pos)  (tan_model.py:155-167); computed ONCE and shared
        by the dual and joint paths (the reference evaluates it twice, bit-identically)."""
        B, T, Dv = video_c.shape
        cd, dev = video_c.dtype, video_c.device
        R = B * T
        proj = torch.empty(R, WIDTH, dtype=cd, device=dev)
        ops.gemm(video_c, self._w("video_pre_proj.weight"), proj, M=R, N=WIDTH, K=Dv)
        pos_c, pos_saved = self._pos_ln("temporal_pos_embed", T, pos_start, interpolate_from, cd, keep)
        x0 = torch.empty(R, WIDTH, dtype=cd, device=dev)
        mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
        ops.layernorm_fwd(proj, self._f("ln_video_init.weight"), self._f("ln_video_init.bias"), x0, mean, rstd, pos_c, T)
        return x0, {"proj": proj, "mean": mean, "rstd": rstd, "pos": pos_saved, "video_c": video_c}

    def _video_embed_repos(self, sv, pos_start, interpolate_from, keep):
        """The video embedding of `sv` again with another position offset: ln_video_init(proj) + ln_position_init(pos')."""
        proj = sv["proj"]
        R, (B, T, _) = proj.shape[0], sv["video_c"].shape
        cd, dev = proj.dtype, proj.device
        pos_c, pos_saved = self._pos_ln("temporal_pos_embed", T, pos_start, interpolate_from, cd, keep)
        x0 = torch.empty(R, WIDTH, dtype=cd, device=dev)
        ops.layernorm_fwd(proj, self._f("ln_video_init.weight"), self._f("ln_video_init.bias"), x0,
                          torch.empty(R, device=dev), torch.empty(R, device=dev), pos_c, T)
        return x0, {"pos": pos_saved, "repos_of": sv}

    def _video_embed_bwd_pair(self, sv, d_x0, have_x0, sv_j, d_x0j):
        """Backward of a video embedding used twice with two position offsets (dual path `d_x0`, joint path `d_x0j`): the position
        tables get their own row sums; the LayerNorm backward is linear in its upstream gradient and both uses share input and
        statistics, so ONE LayerNorm backward and ONE weight-gradient GEMM run on the sum of the two gradients."""
        B, T, _ = sv["video_c"].shape
        cd, dev = d_x0j.dtype, d_x0j.device
        for saved, d in ((sv["pos"], d_x0 if have_x0 else None), (sv_j["pos"], d_x0j)):
            if d is None:
                continue
            d_pos = torch.empty(T, WIDTH, dtype=cd, device=dev)
            ops.group_sum(d, d_pos, B, T, WIDTH)
            self._pos_ln_bwd(saved, d_pos)
        if have_x0:
            ops.rows_copy(d_x0j, d_x0, B, T, WIDTH, T, 0, T, 0, accumulate=True)
            d_sum = d_x0
        else:
            d_sum = d_x0j
        self._video_embed_bwd(sv, d_sum, pos_too=False)

    def _video_embed_bwd(self, sv, d_x0, pos_too=True):
        video_c = sv["video_c"]
        B, T, Dv = video_c.shape
        R, cd, dev = B * T, d_x0.dtype, d_x0.device
        d_proj = torch.empty(R, WIDTH, dtype=cd, device=dev)
        ops.layernorm_bwd(d_x0, sv["proj"], self._f("ln_video_init.weight"), sv["mean"], sv["rstd"], d_proj,
                          self._g("ln_video_init.weight"), self._g("ln_video_init.bias"))
        ops.gemm(d_proj, video_c, self._g("video_pre_proj.weight"), M=WIDTH, N=Dv, K=R, a_kc=False, b_kc=False,
                 lda=WIDTH, ldb=Dv, accumulate=True, split_k=max(1, min(32, R // 1024)))      # K-slices >= 1024 rows: the 4-stage K-strided kernel
        if pos_too:
            d_pos = torch.empty(T, WIDTH, dtype=cd, device=dev)
            ops.group_sum(d_x0, d_pos, B, T, WIDTH)
            self._pos_ln_bwd(sv["pos"], d_pos)

    def _text_embed(self, lang_c, with_time, pos_start, interpolate_from, keep):
        """ln_text_init(text_pre_proj(lang)) (+ ln_position_init(text_pos))  (tan_model.py:231-234 / 212-228)."""
        B, N, Dt = lang_c.shape
        cd, dev = lang_c.dtype, lang_c.device
        R = B * N
        proj = torch.empty(R, WIDTH, dtype=cd, device=dev)
        ops.gemm(lang_c, self._w("text_pre_proj.weight"), proj, M=R, N=WIDTH, K=Dt)
        pos_c = pos_saved = None
        if with_time:
            pos_c, pos_saved = self._pos_ln("text_temporal_pos_embed", N, pos_start, interpolate_from, cd, keep)
        out = torch.empty(R, WIDTH, dtype=cd, device=dev)
        mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
        ops.layernorm_fwd(proj, self._f("ln_text_init.weight"), self._f("ln_text_init.bias"), out, mean, rstd, pos_c, N if with_time else 0)
        return out, {"proj": proj, "mean": mean, "rstd": rstd, "pos": pos_saved, "lang_c": lang_c}

    def _text_embed_bwd(self, sv, d_out, need_d_lang):
        lang_c = sv["lang_c"]
        B, N, Dt = lang_c.shape
        R, cd, dev = B * N, d_out.dtype, d_out.device
        d_proj = torch.empty(R, WIDTH, dtype=cd, device=dev)
        ops.layernorm_bwd(d_out, sv["proj"], self._f("ln_text_init.weight"), sv["mean"], sv["rstd"], d_proj,
                          self._g("ln_text_init.weight"), self._g("ln_text_init.bias"))
        ops.gemm(d_proj, lang_c, self._g("text_pre_proj.weight"), M=WIDTH, N=Dt, K=R, a_kc=False, b_kc=False,
                 lda=WIDTH, ldb=Dt, accumulate=True, split_k=max(1, min(16, R // 256)))
        if sv["pos"] is not None:
            d_pos = torch.empty(N, WIDTH, dtype=cd, device=dev)
            ops.group_sum(d_out, d_pos, B, N, WIDTH)
            self._pos_ln_bwd(sv["pos"], d_pos)
        if need_d_lang:
            d_lang = torch.empty(R, Dt, dtype=cd, device=dev)
            ops.gemm(d_proj, self._w("text_pre_proj.weight"), d_lang, M=R, N=Dt, K=WIDTH, a_kc=True, b_kc=False, ldb=Dt)
            return d_lang.float().view(B, N, Dt)
        return None

    # Activation workspaces (~1 GB per stack at B=128) are pooled per shape: allocating them afresh every step costs
    # tens of ms of hipMalloc/hipFree on the host.  A workspace is taken at forward and handed back after backward
    # (or right after a no-grad forward, whose outputs never alias it).
    # Workspaces (saved activations of a stack, backward scratch) are pooled per shape: allocating ~1 GB afresh each step costs
    # 13 ms/GB of host time.  Real batches vary in N (hence in the joint length L = T + N), so the pool is an LRU over shapes
    # bounded to _WS_POOL_KEYS entries -- at most ~1.3 GB each at B=128.
    _WS_POOL_KEYS = 10

    def _pool_touch(self, key):
        with self._ws_lock:
            self._ws_tick += 1
            self._ws_lru[key] = self._ws_tick
            if len(self._ws_lru) > self._WS_POOL_KEYS:
                for old in sorted(self._ws_lru, key=self._ws_lru.get)[:len(self._ws_lru) - self._WS_POOL_KEYS]:
                    self._ws_lru.pop(old)
                    self._ws_pool.pop(old, None)          # tensors return to the caching allocator (stream-ordered reuse)

    def _take_ws(self, prefix, layers, B, L, cd, dev):
        key = (prefix, layers, B, L, cd, dev)
        self._pool_touch(key)
        with self._ws_lock:
            pool = self._ws_pool.setdefault(key, [])
            er = pool.pop() if pool else None
        if er is None:
            er = _EncRun(self, prefix, layers, B, L, cd, dev)
        er.pool_key = key
        return er

    def _release_ws(self, er):
        if er is not None and getattr(er, "pool_key", None) is not None:
            with self._ws_lock:
                if er.pool_key in self._ws_lru:           # its shape may have been evicted meanwhile: then just drop it
                    pool = self._ws_pool.setdefault(er.pool_key, [])
                    if len(pool) < 2:
                        pool.append(er)
            er.pool_key = None

    def _side_stream(self, dev):
        if not self.overlap_stacks:
            return None
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _on_side(self, side, fn):
        """Issue `fn` (one stack's launches, ~150 per call) on the side stream from a helper thread: the C entry points
        release the GIL, so the two stacks' host-side launch work overlaps too -- at ~5 us of host time per launch the step
        was bound by ONE thread issuing ~530 launches (7.1 of 8.3 ms).  Returns a future; .result() re-raises."""
        if self._issuer is None:
            from concurrent.futures import ThreadPoolExecutor
            self._issuer = ThreadPoolExecutor(max_workers=1, thread_name_prefix="tan-side")
        dev = side.device

        def run():
            torch.cuda.set_device(dev)                       # device and current stream are thread-local
            with torch.no_grad(), torch.cuda.stream(side):
                return fn()
        return self._issuer.submit(run)

    def _take_scratch(self, R, cd, dev):
        key = ("scr", R, cd, dev)
        self._pool_touch(key)
        scr = self._ws_pool.get(key)
        if scr is None:
            scr = _Blocks(cd, dev, {"dx": R * WIDTH, "dx2": R * WIDTH, "do": R * WIDTH, "dxn": R * WIDTH,
                                    "dh": R * 4 * WIDTH, "dqkv": R * 3 * WIDTH})
            n_ws = _lib.lib().tan_layernorm_bwd_ws_floats(C.c_int(WIDTH))
            scr.ln_ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
            scr.dw_ws = torch.empty(32 * 4 * WIDTH * WIDTH, dtype=torch.float32, device=dev)     # split-K partial tiles
            self._ws_pool[key] = scr
        return scr

    def _run_video_stack(self, x0, vmask_u8, B, T, save=False):
        er = self._take_ws("video_temporal_encoder", self.num_encoder_layers, B, T, x0.dtype, x0.device)
        self._encoder_fwd(er, x0, vmask_u8, "ln_video_post_enc", save)
        return er

    def _run_joint_stack(self, x0, text_t, vmask_u8, tmask_u8, B, T, N, save=False):
        cd, dev = x0.dtype, x0.device
        L = T + N
        xj = torch.empty(B * L, WIDTH, dtype=cd, device=dev)
        ops.rows_copy(x0, xj, B, T, WIDTH, T, 0, L, 0)
        ops.rows_copy(text_t, xj, B, N, WIDTH, N, 0, L, T)
        if vmask_u8 is None and tmask_u8 is None:
            keypad = None
        else:
            vm = vmask_u8 if vmask_u8 is not None else torch.zeros(B, T, dtype=torch.uint8, device=dev)
            tm = tmask_u8 if tmask_u8 is not None else torch.zeros(B, N, dtype=torch.uint8, device=dev)
            keypad = torch.cat([vm, tm], dim=1).contiguous()
        er = self._take_ws("joint_temporal_encoder", self.num_decoder_layers, B, L, cd, dev)
        self._encoder_fwd(er, xj, keypad, "ln_joint_post_enc", save)
        er.xj, er.keypad = xj, keypad
        return er

    def _run_forward(self, video, lang, vmask_u8, tmask_u8, opts, keep=True):  # noqa: C901
        """HIP forward of TemporalAligner.forward (tan_model.py:100-149).  Returns the run record used by backward."""
        self._ensure_flat()
        B, T, _ = video.shape
        N = lang.shape[1]
        cd, dev = self.compute_dtype, video.device
        Se, Sd, Cw = self.num_encoder_layers, self.num_decoder_layers, WIDTH
        itp = opts.get("interpolate_from")
        video_c, lang_c = self._prep_inputs(video, lang)
        # reference RNG order: visual, [text-with-time], joint
        p_v = self._draw(T, itp)
        p_t = self._draw(N, itp) if self.use_text_pos_enc else 0
        p_j = self._draw(T, itp)
        x0, sv_video = self._video_embed(video_c, p_v, itp, keep)
        if p_j != p_v:      # random_pos_start=1 draws independent offsets for the dual and joint paths: same projection and
            #                 LayerNorm input, another slice of the position table (one GEMM, not two; see _video_embed_bwd_pair)
            x0j, sv_video_j = self._video_embed_repos(sv_video, p_j, itp, keep)
        else:
            x0j, sv_video_j = x0, None
        lang_raw, sv_text = self._text_embed(lang_c, False, 0, None, keep)
        if self.use_text_pos_enc:
            lang_t, sv_text_t = self._text_embed(lang_c, True, p_t, itp, keep)
        else:
            lang_t, sv_text_t = lang_raw, None
        R, Mp, L = B * T, B * N, T + N
        # L2-normalised features (tan_model.py:116-117,136-137): all stages of a family in one launch, each family right behind the
        # stack that feeds it and on that stack's stream -- 19 per-stage launches after the join sat on the critical path before
        vn_d = torch.empty(Se, R, Cw, dtype=cd, device=dev)
        vn_j = torch.empty(Sd, R, Cw, dtype=cd, device=dev)
        tn_d = torch.empty(Mp, Cw, dtype=cd, device=dev)
        tn_j = torch.empty(Sd, Mp, Cw, dtype=cd, device=dev)
        inv = _Blocks(torch.float32, dev, {"vd": Se * R, "vj": Sd * R, "td": Mp, "tj": Sd * Mp})
        save = bool(opts.get("needs_grad", True))      # False: torch.no_grad() (EMA target, evaluation) -- nothing kept for backward

        def video_side():
            ev_ = self._run_video_stack(x0, vmask_u8, B, T, save)
            ops.l2norm_fwd_multi([ev_.stage(s) for s in range(Se)], vn_d, inv["vd"], R, Cw)
            ops.l2norm_fwd(lang_raw, tn_d, inv["td"], Mp, Cw)
            return ev_

        def joint_side():
            ej_ = self._run_joint_stack(x0j, lang_t, vmask_u8, tmask_u8, B, T, N, save)
            stages = [ej_.stage(s) for s in range(Sd)]
            ops.l2norm_fwd_multi(stages, vn_j, inv["vj"], R, Cw, T, L, 0)
            ops.l2norm_fwd_multi(stages, tn_j, inv["tj"], Mp, Cw, N, L, T)
            return ej_
        # the two stacks are independent (tan_model.py:108-134): the joint stack runs on a side HIP stream next to the video
        # stack, which fills the CUs left idle by each other's small kernels (attention, LayerNorm) and launch gaps
        main, side = torch.cuda.current_stream(), self._side_stream(dev)
        if side is not None:
            side.wait_stream(main)
            fut = self._on_side(side, joint_side)
            ev = video_side()
            ej = fut.result()
            if opts.get("defer_join") and not self.use_alignability_head:
                # the caller (get_loss) joins: the dual similarity sweep only needs the video stack and starts under the joint
                # stack's tail; whoever touches the joint features first waits for this event
                self._join_event = side.record_event()
            else:
                main.wait_stream(side)
        else:
            ev = video_side()
            ej = joint_side()
        if opts.get("fused"):
            # logits-free mode: hand the unit features to get_loss (tan_simnce_* never materialises [S,R,Mp])
            # (fresh view objects: the returned tensors must not be the objects kept in `run`, see _AlignerFn.forward)
            outputs = [vn_d.view(Se, B, T, Cw).permute(1, 0, 2, 3), tn_d.view(B, N, Cw), vn_j.view(Sd, R, Cw), tn_j.view(Sd, Mp, Cw)]
            names = ["vn_d", "tn_d", "vn_j", "tn_j"]
        else:
            # cosine logits, stage-major [S, R, Mp] f32; the reference layout [B,S,T,B,N] is a permuted view (tan_model.py:118,138)
            lg_d = torch.empty(Se, R, Mp, device=dev)
            lg_j = torch.empty(Sd, R, Mp, device=dev)
            ops.gemm(vn_d, tn_d, lg_d, M=R, N=Mp, K=Cw, batch=Se, sA=R * Cw, sB=0, sC=R * Mp)
            ops.gemm(vn_j, tn_j, lg_j, M=R, N=Mp, K=Cw, batch=Sd, sA=R * Cw, sB=Mp * Cw, sC=R * Mp)
            outputs = [lg_d.view(Se, B, T, B, N).permute(1, 0, 2, 3, 4), lg_j.view(Sd, B, T, B, N).permute(1, 0, 2, 3, 4),
                       vn_d.view(Se, B, T, Cw).permute(1, 0, 2, 3), tn_d.view(B, N, Cw)]
            names = ["lg_d", "lg_j", "vn_d", "tn_d"]
        run = {"B": B, "T": T, "N": N, "ev": ev, "ej": ej, "x0": x0, "x0j": x0j, "sv_video": sv_video,
               "sv_video_j": sv_video_j, "sv_text": sv_text, "sv_text_t": sv_text_t, "lang_raw": lang_raw, "lang_t": lang_t,
               "vn_d": vn_d, "vn_j": vn_j, "tn_d": tn_d, "tn_j": tn_j, "inv": inv, "vmask": vmask_u8, "tmask": tmask_u8}
        if self.use_alignability_head:
            w, b = self._f("binary_head.weight").view(-1), self._f("binary_head.bias")
            a_d = torch.empty(Mp, device=dev)
            ops.head_fwd(lang_raw, w, b, a_d, Mp, Cw)
            jt_raw = torch.empty(Sd, Mp, Cw, dtype=cd, device=dev)
            a_j = torch.empty(Sd, Mp, device=dev)
            for s in range(Sd):
                ops.rows_copy(ej.stage(s), jt_raw[s], B, N, Cw, L, T, N, 0)
            ops.head_fwd(jt_raw, w, b, a_j, Sd * Mp, Cw)
            run["jt_raw"] = jt_raw
            outputs += [a_d.view(B, N, 1), a_j.view(Sd, B, N, 1).permute(1, 0, 2, 3)]
            names += ["a_d", "a_j"]
        run["outputs"], run["names"] = outputs, names
        if not opts.get("needs_grad", True):       # nothing will call backward: the stacks' workspaces are free again
            self._release_ws(ev)
            self._release_ws(ej)
        return run

    # ------------------------------------------------------------------ the HIP backward
    def _run_backward(self, run, grads, need_d_lang):
        self._bind_grads()
        B, T, N = run["B"], run["T"], run["N"]
        ev, ej = run["ev"], run["ej"]
        cd, dev = self.compute_dtype, run["x0"].device
        Se, Sd, Cw = self.num_encoder_layers, self.num_decoder_layers, WIDTH
        R, Mp, L = B * T, B * N, T + N
        gd = dict(zip(run["names"], grads))
        g_ld, g_lj, g_vn, g_tn = gd.get("lg_d"), gd.get("lg_j"), gd.get("vn_d"), gd.get("tn_d")
        g_vnj, g_tnj = gd.get("vn_j"), gd.get("tn_j")
        g_ad, g_aj = gd.get("a_d"), gd.get("a_j")

        def stage_major(g, S):
            """[B,S,T,B,N] grad -> contiguous [S,R,Mp] in compute dtype (zero-copy when it is our own permuted buffer)."""
            g = g.permute(1, 0, 2, 3, 4)
            g = g if g.is_contiguous() else g.contiguous()
            g = g.view(S, R, Mp)
            if g.dtype != cd:
                g = ops.cast(g.float().contiguous() if g.dtype != torch.float32 else g, torch.empty(S, R, Mp, dtype=cd, device=dev))
            return g

        inv = run["inv"]
        dst_v = [None] * Se           # d stage outputs of the video stack
        dst_j = [None] * Sd
        d_lang_raw = torch.zeros(Mp, Cw, dtype=cd, device=dev)
        have_lang_raw = False
        # ---- dual similarity: logits_d[s] = vn_d[s] tn_d^T
        d_vn_d = None
        if g_ld is not None:
            dl = stage_major(g_ld, Se)
            d_vn_d = torch.empty(Se, R, Cw, dtype=cd, device=dev)
            ops.gemm(dl, run["tn_d"], d_vn_d, M=R, N=Cw, K=Mp, a_kc=True, b_kc=False, lda=Mp, ldb=Cw, batch=Se,
                     sA=R * Mp, sB=0, sC=R * Cw)
            acc = torch.zeros(Mp, Cw, device=dev)
            ops.gemm(dl, run["vn_d"], acc, M=Mp, N=Cw, K=Se * R, a_kc=False, b_kc=False, lda=Mp, ldb=Cw, accumulate=True,
                     split_k=max(1, min(32, Se * R // 512)))
            d_tn_d = acc if cd == torch.float32 else ops.cast(acc, torch.empty(Mp, Cw, dtype=cd, device=dev))
        else:
            d_tn_d = None
        if g_vn is not None:          # dual_feature_video is an output too
            gv = g_vn.permute(1, 0, 2, 3).contiguous().view(Se, R, Cw).to(cd)
            d_vn_d = gv if d_vn_d is None else d_vn_d + gv
        if g_tn is not None:
            gt = g_tn.contiguous().view(Mp, Cw).to(cd)
            d_tn_d = gt if d_tn_d is None else d_tn_d + gt
        if d_vn_d is not None:
            dst_all = torch.empty(Se, R, Cw, dtype=cd, device=dev)
            for s in range(Se):
                dst_v[s] = dst_all[s]
            ops.l2norm_bwd_multi(d_vn_d, run["vn_d"], inv["vd"], dst_v, R, Cw)
        if d_tn_d is not None:
            ops.l2norm_bwd(d_tn_d, run["tn_d"], inv["td"], d_lang_raw, Mp, Cw)
            have_lang_raw = True
        # ---- joint similarity: logits_j[s] = vn_j[s] tn_j[s]^T
        d_vn_j = d_tn_j = None
        if g_lj is not None:
            dl = stage_major(g_lj, Sd)
            d_vn_j = torch.empty(Sd, R, Cw, dtype=cd, device=dev)
            d_tn_j = torch.empty(Sd, Mp, Cw, dtype=cd, device=dev)
            ops.gemm(dl, run["tn_j"], d_vn_j, M=R, N=Cw, K=Mp, a_kc=True, b_kc=False, lda=Mp, ldb=Cw, batch=Sd,
                     sA=R * Mp, sB=Mp * Cw, sC=R * Cw)
            ops.gemm(dl, run["vn_j"], d_tn_j, M=Mp, N=Cw, K=R, a_kc=False, b_kc=False, lda=Mp, ldb=Cw, batch=Sd,
                     sA=R * Mp, sB=R * Cw, sC=Mp * Cw)
        if g_vnj is not None or g_tnj is not None:       # fused mode: feature gradients arrive directly from _FusedNCEFn
            d_vn_j = g_vnj.contiguous().to(cd) if g_vnj is not None else torch.zeros(Sd, R, Cw, dtype=cd, device=dev)
            d_tn_j = g_tnj.contiguous().to(cd) if g_tnj is not None else torch.zeros(Sd, Mp, Cw, dtype=cd, device=dev)
        if d_vn_j is not None:
            dst_all = torch.empty(Sd, B * L, Cw, dtype=cd, device=dev)
            for s in range(Sd):
                dst_j[s] = dst_all[s]
            ops.l2norm_bwd_multi(d_vn_j, run["vn_j"], inv["vj"], dst_j, R, Cw, T, L, 0)
            ops.l2norm_bwd_multi(d_tn_j, run["tn_j"], inv["tj"], dst_j, Mp, Cw, N, L, T)
        # ---- alignability heads (tan_model.py:147-148)
        if self.use_alignability_head and (g_ad is not None or g_aj is not None):
            w = self._f("binary_head.weight").view(-1)
            gw, gb = self._g("binary_head.weight").view(-1), self._g("binary_head.bias")
            if g_ad is not None:
                ops.head_bwd(g_ad.contiguous().view(Mp).float(), run["lang_raw"], w, d_lang_raw, gw, gb, Mp, Cw, accumulate_dx=True)
                have_lang_raw = True
            if g_aj is not None:
                gaj = g_aj.permute(1, 0, 2, 3).contiguous().view(Sd * Mp).float()
                d_jt = torch.empty(Sd, Mp, Cw, dtype=cd, device=dev)
                ops.head_bwd(gaj, run["jt_raw"], w, d_jt, gw, gb, Sd * Mp, Cw)
                for s in range(Sd):
                    if dst_j[s] is None:
                        dst_j[s] = torch.zeros(B * L, Cw, dtype=cd, device=dev)
                    ops.rows_copy(d_jt[s], dst_j[s], B, N, Cw, N, 0, L, T, accumulate=True)
        # ---- encoder stacks
        d_x0 = torch.zeros(R, Cw, dtype=cd, device=dev)
        d_x0j = d_x0
        any_v = any(t is not None for t in dst_v)
        any_j = any(t is not None for t in dst_j)
        d_lang_t = None
        d_xj = torch.empty(B * L, Cw, dtype=cd, device=dev) if any_j else None
        if cd == torch.bfloat16 and self.transposed_dx:
            self._flat.sync_shadow_t()         # W^T copies for the dX GEMMs, rebuilt once per optimizer step (main stream)
            if self.panel_kernels:
                self._flat.sync_shadow_tp()    # their packed images (MLP weights) for the row-panel backward
        main, side = torch.cuda.current_stream(), self._side_stream(dev)
        if any_j and any_v and side is not None:
            # joint stack backward on the side stream (issued by the helper thread), video stack backward on the main stream
            side.wait_stream(main)
            fut = self._on_side(side, lambda: self._encoder_bwd(ej, ej.xj, ej.keypad, "ln_joint_post_enc", dst_j, d_xj))
            self._encoder_bwd(ev, run["x0"], run["vmask"], "ln_video_post_enc", dst_v, d_x0)
            # DDP: each stack's slice of the flat gradient is final once its backward is enqueued.  Both collectives are issued
            # from THIS thread, video first (every rank must issue them in the same order), each in the stream context whose
            # work it has to wait for; they overlap whatever backward work is still running.
            if self._grad_ready_hook is not None:
                self._grad_ready_hook("video", self._layer_events(ev.prefix, ev.layers))
            fut.result()
            if self._grad_ready_hook is not None:
                with torch.cuda.stream(side):
                    self._grad_ready_hook("joint", self._layer_events(ej.prefix, ej.layers))
            main.wait_stream(side)
        else:
            if any_j:
                self._encoder_bwd(ej, ej.xj, ej.keypad, "ln_joint_post_enc", dst_j, d_xj)
                if self._grad_ready_hook is not None:    # joint-stack gradients are final: DDP starts reducing them now
                    self._grad_ready_hook("joint", self._layer_events(ej.prefix, ej.layers))
            if any_v:
                self._encoder_bwd(ev, run["x0"], run["vmask"], "ln_video_post_enc", dst_v, d_x0)
                if self._grad_ready_hook is not None:
                    self._grad_ready_hook("video", self._layer_events(ev.prefix, ev.layers))
        if any_j:
            if run["sv_video_j"] is not None:
                d_x0j = torch.empty(R, Cw, dtype=cd, device=dev)
                ops.rows_copy(d_xj, d_x0j, B, T, Cw, L, 0, T, 0)
            else:
                ops.rows_copy(d_xj, d_x0, B, T, Cw, L, 0, T, 0, accumulate=any_v)
            if run["sv_text_t"] is None:
                ops.rows_copy(d_xj, d_lang_raw, B, N, Cw, L, T, N, 0, accumulate=have_lang_raw)
                have_lang_raw = True
            else:
                d_lang_t = torch.empty(Mp, Cw, dtype=cd, device=dev)
                ops.rows_copy(d_xj, d_lang_t, B, N, Cw, L, T, N, 0)
        # ---- embeddings: two chains of ~10 small launches each that share nothing (video / text parameters): the text one on the
        # side stream (idle now) next to the video one (TAN_TAIL_STREAMS=0: one after the other)
        d_lang = None
        cur = torch.cuda.current_stream()
        aux = self._side_stream(dev) if os.environ.get("TAN_TAIL_STREAMS", "1") != "0" else None
        if aux is not None and aux.cuda_stream == cur.cuda_stream:
            aux = None

        def text_side():
            d = None
            if have_lang_raw:
                d = self._text_embed_bwd(run["sv_text"], d_lang_raw, need_d_lang)
            if d_lang_t is not None:
                d2 = self._text_embed_bwd(run["sv_text_t"], d_lang_t, need_d_lang)
                d = d2 if d is None else (d + d2 if d2 is not None else d)
            return d

        if aux is not None:
            aux.wait_stream(cur)
            # caching-allocator bookkeeping across the two streams: tensors allocated on `cur` and read on `aux` must not be
            # handed out again on `cur` before aux is done with them, and vice versa for the result
            d_lang_raw.record_stream(aux)
            if d_lang_t is not None:
                d_lang_t.record_stream(aux)
            with torch.cuda.stream(aux):
                d_lang = text_side()
            if d_lang is not None:
                d_lang.record_stream(cur)
        if any_j and run["sv_video_j"] is not None:
            self._video_embed_bwd_pair(run["sv_video"], d_x0, any_v, run["sv_video_j"], d_x0j)
        elif any_v or any_j:
            self._video_embed_bwd(run["sv_video"], d_x0)
        if aux is not None:
            cur.wait_stream(aux)
        else:
            d_lang = text_side()
        self._release_ws(ev)
        self._release_ws(ej)
        return d_lang

    # ------------------------------------------------------------------ public surface (tan_model.py:100-312)
    @staticmethod
    def _mask_u8(m):
        if m is None:
            return None
        return m.to(torch.uint8).contiguous()

    def forward(self, video_embed, lang_embed, video_padding_mask, lang_padding_mask, text_timestamp=None,
                interpolate_from=None, abs_text_pos=None, fused=False):
        """Reference signature (tan_model.py:100-103).  `fused=True` (bf16 mode only, not in the reference) skips the
        [B,S,T,B,N] logits: the dict then carries '_fused' (unit features) for temporalalignnet_amd.loss.get_loss, which
        runs the logits-free similarity+NCE kernels; 'logits_dual' / 'logits_joint' are absent."""
        self._ensure_flat()
        f = self._flat
        defer = fused == "defer"        # (training driver only) leave the join of the two stack streams to get_loss
        fused = bool(fused) and self.compute_dtype == torch.bfloat16
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in f.params) or lang_embed.requires_grad)
        self._join_event = None
        outs = _AlignerFn.apply(self, video_embed, lang_embed, self._mask_u8(video_padding_mask),
                                self._mask_u8(lang_padding_mask),
                                {"interpolate_from": interpolate_from, "needs_grad": needs_grad, "fused": fused,
                                 "defer_join": defer and fused}, *self._autograd_anchor())
        B, T, N = video_embed.shape[0], video_embed.shape[1], lang_embed.shape[1]
        if fused:
            from .loss import FusedSim
            Se = self.num_encoder_layers
            out = {"_fused": FusedSim(outs[0].permute(1, 0, 2, 3).reshape(Se, B * T, WIDTH), outs[1].reshape(1, B * N, WIDTH),
                                      outs[2], outs[3], B, T, N)}
            out["_fused"].join_event, self._join_event = self._join_event, None
            nxt = 4
            if self.return_dual_feature:
                out["dual_feature_video"], out["dual_feature_text"] = outs[0], outs[1]
        else:
            out = {"logits_dual": outs[0], "logits_joint": outs[1]}
            nxt = 4
            if self.return_dual_feature:
                out["dual_feature_video"], out["dual_feature_text"] = outs[2], outs[3]
        if self.use_alignability_head:
            out["dual_logits_alignability"], out["joint_logits_alignability"] = outs[nxt], outs[nxt + 1]
        return out

    @torch.no_grad()
    def get_visual_feature(self, video_embed, video_padding_mask, interpolate_from=None):
        """[B,S,T,C] deep-supervision video features of the dual encoder (tan_model.py:152-179).  Inference entry point
        (retrieval, eval_zeroshot_retrieval.py:180-184): gradients flow through forward(), not through this method."""
        self._ensure_flat()
        B, T, _ = video_embed.shape
        video_c, _ = self._prep_inputs(video_embed, None)
        x0, _ = self._video_embed(video_c, self._draw(T, interpolate_from), interpolate_from, False)
        ev = self._run_video_stack(x0, self._mask_u8(video_padding_mask), B, T)
        S = self.num_encoder_layers
        out = torch.stack([ev.stage(s).view(B, T, WIDTH) for s in range(S)], dim=1).float()
        self._release_ws(ev)
        return out

    @torch.no_grad()
    def get_textual_feature(self, lang_embed):
        """ln_text_init(text_pre_proj(lang)) (tan_model.py:231-234): [B,N,C], or any [..., 512] like the reference's Linear +
        LayerNorm (the retrieval evaluation passes the language model's [1, 512] pooler_output, eval_zeroshot_retrieval.py:190-193)."""
        self._ensure_flat()
        lead = lang_embed.shape[:-1]
        lang_c = self._prep(lang_embed.reshape(1, -1, lang_embed.shape[-1]))
        out, _ = self._text_embed(lang_c, False, 0, None, False)
        return out.view(*lead, WIDTH).float()

    @torch.no_grad()
    def get_textual_feature_with_time(self, lang_embed, text_timestamp=None, interpolate_from=None):
        """tan_model.py:212-228."""
        self._ensure_flat()
        lang_c = self._prep(lang_embed)
        N = lang_embed.shape[1]
        out, _ = self._text_embed(lang_c, True, self._draw(N, interpolate_from), interpolate_from, False)
        return out.view(*lang_embed.shape[:2], WIDTH).float()

    @torch.no_grad()
    def get_joint_feature(self, video_embed, video_padding_mask, lang_embed_with_time, lang_padding_mask,
                          interpolate_from=None):
        """([B,S,T,C], [B,S,N,C]) from the joint encoder (tan_model.py:182-209); `lang_embed_with_time` is the
        already-projected text feature, as in the reference."""
        self._ensure_flat()
        B, T, _ = video_embed.shape
        N = lang_embed_with_time.shape[1]
        video_c, _ = self._prep_inputs(video_embed, None)
        x0, _ = self._video_embed(video_c, self._draw(T, interpolate_from), interpolate_from, False)
        lt = lang_embed_with_time.detach().to(self.compute_dtype).contiguous().view(B * N, WIDTH)
        ej = self._run_joint_stack(x0, lt, self._mask_u8(video_padding_mask), self._mask_u8(lang_padding_mask), B, T, N)
        S, L = self.num_decoder_layers, T + N
        out = torch.stack([ej.stage(s).view(B, L, WIDTH) for s in range(S)], dim=1).float()
        self._release_ws(ej)
        return out[:, :, :T], out[:, :, T:]

    @staticmethod
    def _split_interp(interpolate_from):
        if isinstance(interpolate_from, (list, tuple)):
            assert len(interpolate_from) == 2
            return interpolate_from[0], interpolate_from[1]
        return interpolate_from, None

    def _eval_joint(self, video_embed, lang_embed, interpolate_from, video_padding_mask=None, lang_padding_mask=None):
        """shared by get_text_visual_sim_joint / get_alignability: joint stack on (video, text); the reference passes zero
        masks (tan_model.py:246-247,296-297), the batched evaluation passes the padding of its stacked windows."""
        vi, ti = self._split_interp(interpolate_from)
        self._ensure_flat()
        B, T, _ = video_embed.shape
        N = lang_embed.shape[1]
        video_c, lang_c = self._prep_inputs(video_embed, lang_embed)
        if self.use_text_pos_enc:   # reference order: text offset first, then video (tan_model.py:245-259)
            lang_t, _ = self._text_embed(lang_c, True, self._draw(N, ti), ti, False)
        else:
            lang_t, _ = self._text_embed(lang_c, False, 0, None, False)
        x0, _ = self._video_embed(video_c, self._draw(T, vi), vi, False)
        ej = self._run_joint_stack(x0, lang_t, self._mask_u8(video_padding_mask), self._mask_u8(lang_padding_mask), B, T, N)
        return ej, lang_c, B, T, N

    def _within_sample_sim(self, vn, tn, S, B, T, N, t_stage_stride):
        """einsum 'bstc,b(s)kc->bstk' as a batched GEMM over (s, b)."""
        out = torch.empty(S, B, T, N, device=vn.device)
        for s in range(S):
            ops.gemm(vn[s], tn[s] if t_stage_stride else tn, out[s], M=T, N=N, K=WIDTH, batch=B, sA=T * WIDTH,
                     sB=N * WIDTH, sC=T * N)
        return out.permute(1, 0, 2, 3)

    @torch.no_grad()
    def get_text_visual_sim_joint(self, video_embed, lang_embed, interpolate_from=None):
        """[B,S,T,K] within-sample cosine similarities from the joint encoder (tan_model.py:237-264)."""
        ej, _, B, T, N = self._eval_joint(video_embed, lang_embed, interpolate_from)
        S, L, cd, dev = self.num_decoder_layers, T + N, self.compute_dtype, video_embed.device
        vn = torch.empty(S, B * T, WIDTH, dtype=cd, device=dev)
        tn = torch.empty(S, B * N, WIDTH, dtype=cd, device=dev)
        for s in range(S):
            ops.l2norm_fwd(ej.stage(s), vn[s], None, B * T, WIDTH, T, L, 0)
            ops.l2norm_fwd(ej.stage(s), tn[s], None, B * N, WIDTH, N, L, T)
        self._release_ws(ej)
        return self._within_sample_sim(vn, tn, S, B, T, N, True)

    get_text_visual_sim = get_text_visual_sim_joint   # alias needed by the released Twin constructor (tan_model.py:328)

    @torch.no_grad()
    def get_text_visual_sim_dual(self, video_embed, lang_embed, interpolate_from=None):
        """[B,S,T,K] within-sample cosine similarities from the dual encoder (tan_model.py:267-283)."""
        self._ensure_flat()
        B, T, _ = video_embed.shape
        N = lang_embed.shape[1]
        cd, dev, S = self.compute_dtype, video_embed.device, self.num_encoder_layers
        video_c, lang_c = self._prep_inputs(video_embed, lang_embed)
        lang_raw, _ = self._text_embed(lang_c, False, 0, None, False)
        x0, _ = self._video_embed(video_c, self._draw(T, interpolate_from), interpolate_from, False)
        ev = self._run_video_stack(x0, None, B, T)
        vn = torch.empty(S, B * T, WIDTH, dtype=cd, device=dev)
        tn = torch.empty(B * N, WIDTH, dtype=cd, device=dev)
        for s in range(S):
            ops.l2norm_fwd(ev.stage(s), vn[s], None, B * T, WIDTH)
        ops.l2norm_fwd(lang_raw, tn, None, B * N, WIDTH)
        self._release_ws(ev)
        return self._within_sample_sim(vn, tn, S, B, T, N, False)

    @torch.no_grad()
    def get_alignability(self, video_embed, lang_embed, interpolate_from=None, abs_text_pos=None):
        """{'alignability-dual' [B,K,1], 'alignability-joint' [B,S,K,1]} (tan_model.py:286-312)."""
        ej, lang_c, B, T, N = self._eval_joint(video_embed, lang_embed, interpolate_from)
        S, L, cd, dev = self.num_decoder_layers, T + N, self.compute_dtype, video_embed.device
        w, b = self._f("binary_head.weight").view(-1), self._f("binary_head.bias")
        lang_raw, _ = self._text_embed(lang_c, False, 0, None, False)
        a_d = torch.empty(B * N, device=dev)
        ops.head_fwd(lang_raw, w, b, a_d, B * N, WIDTH)
        jt = torch.empty(S, B * N, WIDTH, dtype=cd, device=dev)
        for s in range(S):
            ops.rows_copy(ej.stage(s), jt[s], B, N, WIDTH, L, T, N, 0)
        self._release_ws(ej)
        a_j = torch.empty(S, B * N, device=dev)
        ops.head_fwd(jt, w, b, a_j, S * B * N, WIDTH)
        return {"alignability-dual": a_d.view(B, N, 1), "alignability-joint": a_j.view(S, B, N, 1).permute(1, 0, 2, 3)}

    @torch.no_grad()
    def eval_windows(self, video_embed, lang_embed, video_padding_mask=None, lang_padding_mask=None, interpolate_from=None):
        """Everything the evaluation closure of train/main.py:171-189 asks of the model, for a BATCH of windows in one pass of each
        stack: {'sim' [B,S_d,T,K], 'dual-sim' [B,S_e,T,K], 'alignability-dual' [B,K,1], 'alignability-joint' [B,S_d,K,1]} (raw
        cosines / logits; the closure transposes and divides by 0.07).  The reference evaluates window by window at B=1 and runs
        the joint stack twice per window (get_text_visual_sim_joint + get_alignability) -- ~600 launches of a few microseconds each,
        latency-bound on any GPU; here the windows of a video are stacked, short last windows / unequal sentence counts are
        padded and masked as attention keys (a masked key has probability exactly 0, so real rows are unaffected)."""
        ej, lang_c, B, T, N = self._eval_joint(video_embed, lang_embed, interpolate_from, video_padding_mask, lang_padding_mask)
        vi, _ = self._split_interp(interpolate_from)
        Sd, Se, L, cd, dev = self.num_decoder_layers, self.num_encoder_layers, T + N, self.compute_dtype, video_embed.device
        vn = torch.empty(Sd, B * T, WIDTH, dtype=cd, device=dev)
        tn = torch.empty(Sd, B * N, WIDTH, dtype=cd, device=dev)
        for s in range(Sd):
            ops.l2norm_fwd(ej.stage(s), vn[s], None, B * T, WIDTH, T, L, 0)
            ops.l2norm_fwd(ej.stage(s), tn[s], None, B * N, WIDTH, N, L, T)
        out = {"sim": self._within_sample_sim(vn, tn, Sd, B, T, N, True)}
        lang_raw, _ = self._text_embed(lang_c, False, 0, None, False)
        if self.use_alignability_head:
            w, b = self._f("binary_head.weight").view(-1), self._f("binary_head.bias")
            a_d = torch.empty(B * N, device=dev)
            ops.head_fwd(lang_raw, w, b, a_d, B * N, WIDTH)
            jt = torch.empty(Sd, B * N, WIDTH, dtype=cd, device=dev)
            for s in range(Sd):
                ops.rows_copy(ej.stage(s), jt[s], B, N, WIDTH, L, T, N, 0)
            a_j = torch.empty(Sd, B * N, device=dev)
            ops.head_fwd(jt, w, b, a_j, Sd * B * N, WIDTH)
            out["alignability-dual"] = a_d.view(B, N, 1)
            out["alignability-joint"] = a_j.view(Sd, B, N, 1).permute(1, 0, 2, 3)
        self._release_ws(ej)
        # dual encoder on the same windows
        video_c, _ = self._prep_inputs(video_embed, None)
        x0, _ = self._video_embed(video_c, self._draw(T, vi), vi, False)
        ev = self._run_video_stack(x0, self._mask_u8(video_padding_mask), B, T)
        vd = torch.empty(Se, B * T, WIDTH, dtype=cd, device=dev)
        td = torch.empty(B * N, WIDTH, dtype=cd, device=dev)
        for s in range(Se):
            ops.l2norm_fwd(ev.stage(s), vd[s], None, B * T, WIDTH)
        ops.l2norm_fwd(lang_raw, td, None, B * N, WIDTH)
        self._release_ws(ev)
        out["dual-sim"] = self._within_sample_sim(vd, td, Se, B, T, N, False)
        return out

    # checkpoint compatibility: the released checkpoint spells the language model `lang_model.` (train/main.py:467-469)
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in list(state_dict.keys()):
            if k.startswith(prefix + "lang_model."):
                state_dict[prefix + "bert." + k[len(prefix + "lang_model."):]] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class TwinTemporalAligner(nn.Module):
    """Online + EMA target copy (model/tan_model.py:315-351)."""

    def __init__(self, m=0.999, *args, **kwargs):
        super().__init__()
        self.m = m
        self.online = TemporalAligner(*args, **kwargs)
        self.target = TemporalAligner(*args, **kwargs)
        self._copy_param()
        self.bert = self.online.bert
        self.get_visual_feature = self.online.get_visual_feature
        self.get_joint_feature = self.online.get_joint_feature
        self.get_textual_feature_with_time = self.online.get_textual_feature_with_time
        self.get_textual_feature = self.online.get_textual_feature
        self.get_text_visual_sim = self.online.get_text_visual_sim
        self.get_text_visual_sim_joint = self.online.get_text_visual_sim_joint
        self.get_text_visual_sim_dual = self.online.get_text_visual_sim_dual
        self.get_alignability = self.online.get_alignability
        self.eval_windows = self.online.eval_windows
        self.target.random_pos_start = 0

    @property
    def lang_model(self):
        return self.bert

    # train/main.py:466-469 adds the stage-1 language-model tensors at top level under `lang_model.`; the attribute is `bert`
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in list(state_dict.keys()):
            if k.startswith(prefix + "lang_model."):
                state_dict[prefix + "bert." + k[len(prefix + "lang_model."):]] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _copy_param(self):
        for po, pt in zip(self.online.parameters(), self.target.parameters()):
            pt.data.copy_(po.data)
            pt.requires_grad = False
        self.target.invalidate_shadow()         # the copies went through the parameter tensors, not the flat buffer

    @torch.no_grad()
    def _momentum_update(self):
        """target = m * target + (1 - m) * online (tan_model.py:339-344): one HIP launch over the flat buffers."""
        fo, ft = self.online._ensure_flat(), self.target._ensure_flat()
        _lib.check(_lib.lib().tan_ema_update(_vp(ft.flat), _vp(fo.flat), C.c_long(fo.total), C.c_float(self.m),
                                             _vp(ft.shadow), ops._stream()), "tan_ema_update")
        ft.shadow_rewritten()                # also the packed / transposed images built from the shadow
        if self.online.bert is not None:     # language-model parameters live outside the flat buffers
            for po, pt in zip(self.online.bert.parameters(), self.target.bert.parameters()):
                pt.data.mul_(self.m).add_(po.data, alpha=1.0 - self.m)

    def forward(self, *args, **kwargs):
        return self.online(*args, **kwargs)

    @torch.no_grad()
    def forward_from_ema(self, *args, **kwargs):
        return self.target(*args, **kwargs)
