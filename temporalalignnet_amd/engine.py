"""The HIP forward / backward of TemporalAligner.forward (model/tan_model.py:100-149) behind ONE autograd node: embedding
front-ends, the two encoder stacks on two HIP streams (one C call per stack and direction), L2-normalised features, cosine logits
(or the unit features for the logits-free loss) and the alignability heads -- and the same in reverse, adding parameter gradients
straight into the flat gradient buffer.  `_AlignerEngine` is mixed into TemporalAligner (tan_model.py keeps the reference's
public surface); nothing here is ATen arithmetic."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib, ops
from .flat_params import _vp
from .workspace import HEADS, WIDTH, _Blocks, _EncRun, _WorkspaceMixin


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


class _AlignerEngine(_WorkspaceMixin):
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
        d.split_part = _vp(getattr(er, "split_part", None))
        return d

    def _encoder_fwd(self, er, x0, keypad, post_name, save=False, xn1_ready=False):
        d = self._enc_desc(er, x0, keypad, post_name)
        self._flat.join_images()
        d.no_save = 0 if save else 1         # no backward will follow: the tensors kept only for it are not written
        d.xn1_ready = 1 if xn1_ready else 0  # tan_embed_fwd already wrote the first block's ln_1 output
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

    def _encoder_bwd(self, er, x0, keypad, post_name, d_stage, d_x0, dw_stream=None, dw_tail=1):
        cd, dev, R = x0.dtype, x0.device, er.R
        d = self._enc_desc(er, x0, keypad, post_name)
        self._flat.join_images()
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
        if dw_stream is not None:          # the last blocks' weight gradients off the chain (the caller joins that stream)
            d.dw_stream, d.dw_tail = C.c_void_p(dw_stream.cuda_stream), dw_tail
            d.scr2_dx, d.scr2_dx2, d.scr2_dh, d.scr2_dqkv = (_vp(scr[k]) for k in ("dx_b", "dx2_b", "dh_b", "dqkv_b"))
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

    def _embed_bwd_fused(self, run, d_x0, d_xj, d_lang_dual, need_d_lang):
        """Backward of the fused front-end: tan_embed_bwd (both modalities: LayerNorm backward on the sum of the dual- and joint-path
        gradients read in place, position-row sums, LayerNorm parameter gradients), tan_pos_ln_bwd (all used table slices), and the
        two pre-projection weight-gradient GEMMs (video on this stream, text -- and d lang when the language model trains -- on the
        side stream)."""
        em, B, T, N = run["em"], run["B"], run["T"], run["N"]
        L, R, Mp = T + N, B * T, B * N
        cd, dev = self.compute_dtype, em["x0"].device
        sv_v, sv_vj, sv_t, sv_tt = run["sv_video"], run["sv_video_j"], run["sv_text"], run["sv_text_t"]
        D = (_lib.EmbedBwdDesc * 2)()
        dpos = em.dpos
        d = D[0]
        d.rows, d.T, d.C = R, T, WIDTH
        if d_x0 is not None:
            d.d_out[0], d.d_out_grp_rows[0], d.d_out_off[0], d.d_pos[0] = d_x0.data_ptr(), T, 0, dpos[0].data_ptr()
        if d_xj is not None:
            d.d_out[1], d.d_out_grp_rows[1], d.d_out_off[1], d.d_pos[1] = d_xj.data_ptr(), L, 0, dpos[1].data_ptr()
        d.proj, d.mean, d.rstd = sv_v["proj"].data_ptr(), sv_v["mean"].data_ptr(), sv_v["rstd"].data_ptr()
        d.ln_g, d.d_proj = self._f("ln_video_init.weight").data_ptr(), em["dproj_v"].data_ptr()
        d.g_ln_g, d.g_ln_b = self._g("ln_video_init.weight").data_ptr(), self._g("ln_video_init.bias").data_ptr()
        nprob = 1
        if d_lang_dual is not None or d_xj is not None:
            d = D[1]
            nprob = 2
            d.rows, d.T, d.C = Mp, N, WIDTH
            if d_lang_dual is not None:
                d.d_out[0], d.d_out_grp_rows[0], d.d_out_off[0] = d_lang_dual.data_ptr(), N, 0
            if d_xj is not None:
                d.d_out[1], d.d_out_grp_rows[1], d.d_out_off[1] = d_xj.data_ptr(), L, T
                if sv_tt is not None:
                    d.d_pos[1] = dpos[2].data_ptr()
            d.proj, d.mean, d.rstd = sv_t["proj"].data_ptr(), sv_t["mean"].data_ptr(), sv_t["rstd"].data_ptr()
            d.ln_g, d.d_proj = self._f("ln_text_init.weight").data_ptr(), em["dproj_t"].data_ptr()
            d.g_ln_g, d.g_ln_b = self._g("ln_text_init.weight").data_ptr(), self._g("ln_text_init.bias").data_ptr()
        _lib.check(_lib.lib().tan_embed_bwd(D, nprob, ops._stream()), "tan_embed_bwd")
        # ---- ln_position_init backward on the used slices (the dual offset's slice only if the video stack had a gradient)
        uses = []

        def use(saved, buf):
            which = saved["which"]
            learned = not (which == "temporal_pos_embed" and self.pos_enc != "learned")
            g_tab = self._g(which)[saved["start"]:saved["start"] + saved["n"]] if learned else None
            uses.append(_lib.PosLnBwdUse(buf.data_ptr(), saved["pos"].data_ptr(), saved["mean"].data_ptr(), saved["rstd"].data_ptr(),
                                         g_tab.data_ptr() if g_tab is not None else None, saved["n"], em.nparts))
        if d_x0 is not None:
            use(sv_v["pos"], dpos[0])
        if d_xj is not None:
            use((sv_vj or sv_v)["pos"], dpos[1])
            if sv_tt is not None:
                use(sv_tt["pos"], dpos[2])
        if uses:
            U = (_lib.PosLnBwdUse * len(uses))(*uses)
            _lib.check(_lib.lib().tan_pos_ln_bwd(U, len(uses), self._f("ln_position_init.weight").data_ptr(),
                                                 self._g("ln_position_init.weight").data_ptr(), self._g("ln_position_init.bias").data_ptr(),
                                                 WIDTH, ops._stream()), "tan_pos_ln_bwd")
        # ---- weight gradients of the two pre-projections (train/main.py: autograd of tan_model.py:48-49)
        cur = torch.cuda.current_stream()
        aux = self._side_stream(dev)
        if aux is not None and aux.cuda_stream == cur.cuda_stream:
            aux = None
        d_lang = None

        def text_side():
            lang_c = sv_t["lang_c"]
            Dt = lang_c.shape[-1]
            ops.gemm(em["dproj_t"], lang_c, self._g("text_pre_proj.weight"), M=WIDTH, N=Dt, K=Mp, a_kc=False, b_kc=False,
                     lda=WIDTH, ldb=Dt, accumulate=True, split_k=max(1, min(16, Mp // 256)))
            if need_d_lang:
                dl = torch.empty(Mp, Dt, dtype=cd, device=dev)
                ops.gemm(em["dproj_t"], self._w("text_pre_proj.weight"), dl, M=Mp, N=Dt, K=WIDTH, a_kc=True, b_kc=False, ldb=Dt)
                return dl.float().view(B, N, Dt)
            return None
        if nprob == 2 and aux is not None:
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                d_lang = text_side()
            if d_lang is not None:
                d_lang.record_stream(cur)
        video_c = sv_v["video_c"]
        Dv = video_c.shape[-1]
        ops.gemm(em["dproj_v"], video_c, self._g("video_pre_proj.weight"), M=WIDTH, N=Dv, K=R, a_kc=False, b_kc=False,
                 lda=WIDTH, ldb=Dv, accumulate=True, split_k=max(1, min(32, R // 1024)))      # (K slices of 256 rows: +0.03 ms per step, ABBA x2)
        if nprob == 2:
            if aux is not None:
                cur.wait_stream(aux)
            else:
                d_lang = text_side()
        return d_lang

    def _side_stream(self, dev):
        if not self.overlap_stacks:
            return None
        if self._side is None or self._side.device != dev:
            self._side = _lib.role_stream(dev, "stack")          # one per device and process (see _lib.role_stream)
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

    def _run_video_stack(self, x0, vmask_u8, B, T, save=False, er=None):
        """`er`: a workspace whose first block's ln_1 output tan_embed_fwd has already written (the fused front-end)."""
        ready = er is not None
        if er is None:
            er = self._take_ws("video_temporal_encoder", self.num_encoder_layers, B, T, x0.dtype, x0.device)
        self._encoder_fwd(er, x0, vmask_u8, "ln_video_post_enc", save, xn1_ready=ready)
        return er

    def _run_joint_stack(self, x0, text_t, vmask_u8, tmask_u8, B, T, N, save=False, pre=None):
        """`pre` = (er, xj, keypad) from the fused front-end: the joint input, its key-padding mask and the first ln_1 are done."""
        cd, dev = (x0 if pre is None else pre[1]).dtype, (x0 if pre is None else pre[1]).device
        L = T + N
        if pre is not None:
            er, xj, keypad = pre
            self._encoder_fwd(er, xj, keypad, "ln_joint_post_enc", save, xn1_ready=True)
            er.xj, er.keypad = xj, keypad
            return er
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

    # ------------------------------------------------------------------ the fused front-end (bf16): ONE launch
    def _pos_ln_full(self, which):
        """ln_position_init over the WHOLE position table, (rows [P, C] f32, mean [P], rstd [P]): depends on the parameters only, so it
        is rebuilt once per optimizer step next to the other weight images (side stream, `_Flat.refresh_images_async`) and a step's
        random offset is a row offset into it (tan_model.py:162-167)."""
        f = self._flat
        cache = self.__dict__.setdefault("_pos_ln_cache", {})
        ent = cache.get(which)
        if ent is None or ent["flat"] != f.flat.data_ptr():
            table = self._pos_table(which)
            P, dev = table.shape[0], f.flat.device
            ent = cache[which] = {"flat": f.flat.data_ptr(), "epoch": -1, "out": torch.empty(P, WIDTH, device=dev),
                                  "mean": torch.empty(P, device=dev), "rstd": torch.empty(P, device=dev)}
            if self._pos_ln_refresh not in f.image_hooks:
                f.image_hooks.append(self._pos_ln_refresh)
        if ent["epoch"] != f.shadow_epoch:
            ops.layernorm_fwd(self._pos_table(which).contiguous(), self._f("ln_position_init.weight"), self._f("ln_position_init.bias"),
                              ent["out"], ent["mean"], ent["rstd"])
            ent["epoch"] = f.shadow_epoch
        return ent

    def _pos_ln_refresh(self):
        for which in list(self.__dict__.get("_pos_ln_cache", {})):
            self._pos_ln_full(which)

    def _embed_fused_ok(self, video, lang, itp):
        return (self.compute_dtype == torch.bfloat16 and self.panel_kernels and not itp and video.shape[-1] % 128 == 0
                and lang.shape[-1] % 128 == 0 and video.shape[-1] <= 2048 and lang.shape[-1] <= 2048      # (tan_embed_fwd: K <= 2048 for both)
                and os.environ.get("TAN_EMBED_FUSED", "1") != "0")

    @staticmethod
    def _feature_operand(x):
        """a caller's feature tensor as tan_embed_fwd reads it: f32 or bf16, contiguous, no copy when it already is"""
        x = x.detach()
        if not x.is_cuda:
            raise _lib.TanHipError("TemporalAligner needs device tensors: the HIP path has no CPU fallback")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        return x.contiguous()

    def _embed_fused(self, video, lang, vmask_u8, tmask_u8, p_v, p_t, p_j, save):
        """Both modalities' input embeddings, the joint stack's input / key-padding mask and the first ln_1 of both stacks in ONE launch
        (tan_embed_fwd).  Returns what `_run_forward` needs: workspaces, inputs and the saved records of the (unfused) backward."""
        f = self._flat
        B, T, Dv = video.shape
        N, Dt = lang.shape[1], lang.shape[2]
        L, R, Mp = T + N, B * T, B * N
        cd, dev = self.compute_dtype, video.device
        video, lang = self._feature_operand(video), self._feature_operand(lang)
        alt = bool(self.__dict__.get("_ws_alternate"))
        ev = self._take_ws("video_temporal_encoder", self.num_encoder_layers, B, T, cd, dev, alternate=alt)
        ej = self._take_ws("joint_temporal_encoder", self.num_decoder_layers, B, L, cd, dev, alternate=alt)
        em = self._take_emb(B, T, N, Dv, Dt, cd, dev)
        pv = self._pos_ln_full("temporal_pos_embed")
        pt = self._pos_ln_full("text_temporal_pos_embed") if self.use_text_pos_enc else None
        f.join_images()
        wp = f.shadow_p
        keypad = em.keypad if (vmask_u8 is not None or tmask_u8 is not None) else None
        D = (_lib.EmbedDesc * 2)()

        def fill(d, a, a16, rows, K, Tn, wname, ln, proj, mean, rstd):
            d.a, d.a_dtype, d.rows, d.K, d.T, d.C = a.data_ptr(), (_lib.TAN_F32 if a.dtype == torch.float32 else _lib.TAN_BF16), rows, K, Tn, WIDTH
            d.pw = f.ptr(wp, wname)
            d.ln_g, d.ln_b = self._f(ln + ".weight").data_ptr(), self._f(ln + ".bias").data_ptr()
            d.a_bf16 = a16.data_ptr() if (save and a.dtype == torch.float32) else None
            d.proj, d.mean, d.rstd = (proj.data_ptr(), mean.data_ptr(), rstd.data_ptr()) if save else (None, None, None)

        def ln1(d, k, er, prefix):
            base = f"{prefix}.resblocks.0."
            d.ln1_g[k], d.ln1_b[k] = self._f(base + "ln_1.weight").data_ptr(), self._f(base + "ln_1.bias").data_ptr()
            d.xn1[k], d.mean1[k], d.rstd1[k] = er.bufs[0].xn1, er.bufs[0].mean1, er.bufs[0].rstd1

        dv, dt_ = D[0], D[1]
        fill(dv, video, em["video_c"], R, Dv, T, "video_pre_proj.weight", "ln_video_init", em["proj_v"], em["mean_v"], em["rstd_v"])
        dv.out[0], dv.out_grp_rows[0], dv.out_off[0] = em["x0"].data_ptr(), T, 0
        dv.pos[0] = pv["out"].data_ptr() + 4 * WIDTH * p_v
        ln1(dv, 0, ev, "video_temporal_encoder")
        dv.out[1], dv.out_grp_rows[1], dv.out_off[1] = em["xj"].data_ptr(), L, 0
        dv.pos[1] = pv["out"].data_ptr() + 4 * WIDTH * p_j
        ln1(dv, 1, ej, "joint_temporal_encoder")
        fill(dt_, lang, em["lang_c"], Mp, Dt, N, "text_pre_proj.weight", "ln_text_init", em["proj_t"], em["mean_t"], em["rstd_t"])
        dt_.out[0], dt_.out_grp_rows[0], dt_.out_off[0] = em["lang_raw"].data_ptr(), N, 0       # the dual path's text features
        dt_.out[1], dt_.out_grp_rows[1], dt_.out_off[1] = em["xj"].data_ptr(), L, T             # the joint stack's text rows
        if pt is not None:
            dt_.pos[1] = pt["out"].data_ptr() + 4 * WIDTH * p_t
        ln1(dt_, 1, ej, "joint_temporal_encoder")
        if keypad is not None:
            dv.pad_src = vmask_u8.data_ptr() if vmask_u8 is not None else None
            dv.pad_dst, dv.pad_grp_rows, dv.pad_off = keypad.data_ptr(), L, 0
            dt_.pad_src = tmask_u8.data_ptr() if tmask_u8 is not None else None
            dt_.pad_dst, dt_.pad_grp_rows, dt_.pad_off = keypad.data_ptr(), L, T
        _lib.check(_lib.lib().tan_embed_fwd(D, 2, ops._stream()), "tan_embed_fwd")

        def pos_saved(which, ent, n, start):
            return {"which": which, "n": n, "start": start, "interp": None, "pos": self._pos_table(which)[start:start + n],
                    "mean": ent["mean"][start:start + n], "rstd": ent["rstd"][start:start + n]}
        sv_video = {"proj": em["proj_v"], "mean": em["mean_v"], "rstd": em["rstd_v"], "pos": pos_saved("temporal_pos_embed", pv, T, p_v),
                    "video_c": (em["video_c"] if video.dtype == torch.float32 else video).view(B, T, Dv)}
        sv_video_j = {"pos": pos_saved("temporal_pos_embed", pv, T, p_j), "repos_of": sv_video} if p_j != p_v else None
        lang_c = (em["lang_c"] if lang.dtype == torch.float32 else lang).view(B, N, Dt)
        sv_text = {"proj": em["proj_t"], "mean": em["mean_t"], "rstd": em["rstd_t"], "pos": None, "lang_c": lang_c}
        sv_text_t = None
        if pt is not None:      # the joint stack's text rows carry the position term: their gradient is the text embedding's second use
            sv_text_t = {"proj": em["proj_t"], "mean": em["mean_t"], "rstd": em["rstd_t"], "lang_c": lang_c,
                         "pos": pos_saved("text_temporal_pos_embed", pt, N, p_t)}
        return {"ev": ev, "ej": ej, "em": em, "x0": em["x0"], "xj": em["xj"], "keypad": keypad, "lang_raw": em["lang_raw"],
                "sv_video": sv_video, "sv_video_j": sv_video_j, "sv_text": sv_text, "sv_text_t": sv_text_t}

    def _run_forward(self, video, lang, vmask_u8, tmask_u8, opts, keep=True):  # noqa: C901
        """HIP forward of TemporalAligner.forward (tan_model.py:100-149).  Returns the run record used by backward."""
        self._ensure_flat()
        B, T, _ = video.shape
        N = lang.shape[1]
        cd, dev = self.compute_dtype, video.device
        Se, Sd, Cw = self.num_encoder_layers, self.num_decoder_layers, WIDTH
        itp = opts.get("interpolate_from")
        save = bool(opts.get("needs_grad", True))      # False: torch.no_grad() (EMA target, evaluation) -- nothing kept for backward
        # reference RNG order: visual, [text-with-time], joint
        p_v = self._draw(T, itp)
        p_t = self._draw(N, itp) if self.use_text_pos_enc else 0
        p_j = self._draw(T, itp)
        fe = None
        if self._embed_fused_ok(video, lang, itp):
            # ONE launch: cast + both pre-projections + their LayerNorms + the position terms of both offsets + the joint stack's
            # input and key-padding mask + the first ln_1 of both stacks (tan_embed.hip)
            fe = self._embed_fused(video, lang, vmask_u8, tmask_u8, p_v, p_t, p_j, save)
            x0, x0j, lang_raw, lang_t = fe["x0"], None, fe["lang_raw"], None
            sv_video, sv_video_j, sv_text, sv_text_t = fe["sv_video"], fe["sv_video_j"], fe["sv_text"], fe["sv_text_t"]
        else:
            video_c, lang_c = self._prep_inputs(video, lang)
            x0, sv_video = self._video_embed(video_c, p_v, itp, keep)
            if p_j != p_v:  # random_pos_start=1 draws independent offsets for the dual and joint paths: same projection and
                #             LayerNorm input, another slice of the position table (one GEMM, not two; see _video_embed_bwd_pair)
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

        def video_side():
            ev_ = self._run_video_stack(x0, vmask_u8, B, T, save, er=fe["ev"] if fe else None)
            ops.l2norm_fwd_multi([ev_.stage(s) for s in range(Se)], vn_d, inv["vd"], R, Cw)
            ops.l2norm_fwd(lang_raw, tn_d, inv["td"], Mp, Cw)
            return ev_

        def joint_side():
            ej_ = self._run_joint_stack(x0j, lang_t, vmask_u8, tmask_u8, B, T, N, save,
                                        pre=(fe["ej"], fe["xj"], fe["keypad"]) if fe else None)
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
               "vn_d": vn_d, "vn_j": vn_j, "tn_d": tn_d, "tn_j": tn_j, "inv": inv, "vmask": vmask_u8, "tmask": tmask_u8,
               "em": fe["em"] if fe else None}
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
            self._release_ws(run.pop("em"))
        return run

    # ------------------------------------------------------------------ forward + backward as two independent chains (no autograd)
    def _chains_ok(self, video, lang, itp=None, allow_head=False):
        """allow_head: the caller runs the alignability head itself on the joint chain (stage 2: `Trainer._forward_backward_chains2`)"""
        return (self.compute_dtype == torch.bfloat16 and self._embed_fused_ok(video, lang, itp)
                and (allow_head or not self.use_alignability_head) and self._side_stream(video.device) is not None)

    def _ema_stack_diag(self, which, fe, vmask_u8, tmask_u8, B, T, N):
        """Stage 2, on the EMA target model: the no-grad forward of ONE stack from the fused front-end's outputs `fe`, reduced to all that
        self-labelling reads of it (train/loss.py:88-179: `torch.diagonal(ema_logits, dim1=0, dim2=3)[..., -1 stage]`): the last stage's
        same-video cosines [B, T, N] f32.  Runs on the current stream: the two-chain step issues the video stack's on the main chain and
        the joint stack's on the side chain, each in front of the online stack of the same kind."""
        cd, dev, Cw = self.compute_dtype, fe["x0"].device, WIDTH
        R, Mp, L = B * T, B * N, T + N
        vn = torch.empty(1, R, Cw, dtype=cd, device=dev)
        tn = torch.empty(Mp, Cw, dtype=cd, device=dev)
        inv = torch.empty(R + Mp, device=dev)
        if which == "video":
            er = self._run_video_stack(fe["x0"], vmask_u8, B, T, False, er=fe["ev"])
            ops.l2norm_fwd_multi([er.stage(self.num_encoder_layers - 1)], vn, inv[:R], R, Cw)
            ops.l2norm_fwd(fe["lang_raw"], tn, inv[R:], Mp, Cw)
        else:
            er = self._run_joint_stack(None, None, vmask_u8, tmask_u8, B, T, N, False, pre=(fe["ej"], fe["xj"], fe["keypad"]))
            last = [er.stage(self.num_decoder_layers - 1)]
            ops.l2norm_fwd_multi(last, vn, inv[:R], R, Cw, T, L, 0)
            ops.l2norm_fwd_multi(last, tn.view(1, Mp, Cw), inv[R:], Mp, Cw, N, L, T)
        out = torch.empty(B, T, N, device=dev)
        ops.gemm(vn[0], tn, out, M=T, N=N, K=Cw, batch=B, sA=T * Cw, sB=N * Cw, sC=T * N)
        self._release_ws(er)
        return out

    def _run_chains(self, video, lang, vmask_u8, tmask_u8, family, after_video_bwd=None, after_joint_bwd=None, pipe=None, mid=None, need_d_lang=False,
                    pre=None):
        """Forward AND backward of the aligner under a loss that separates into a dual and a joint term (stage 1: train/loss.py:359-373,
        loss = (loss_dual + loss_joint) / 2 with batch-independent weights) as TWO chains that never wait for each other:
            main stream:  video stack forward -> unit features -> family("dual") -> their backward -> video stack backward
            side stream:  joint stack forward -> unit features -> family("joint") -> their backward -> joint stack backward
        joined only in front of the embeddings' backward.  Under autograd (`_run_forward` / get_loss / `_run_backward`) every backward
        kernel waits for the LAST forward kernel: the video stack's backward could not start before the joint stack's forward, the
        joint similarity sweep and its ~12 small launches were through -- 0.6 ms in which the chip runs one stack's kernels or less.
        `family(which, x_video, v_grp, x_text, t_grp, d_video, d_text) -> (v_terms, t_terms)` runs a family's L2 normalisation,
        similarity + NCE forward and backward from the stack's stage outputs to its stage gradients on the current stream
        (`loss.nce_family_stages`; the upstream gradients of its terms depend on masks only).  Parameter gradients land in the flat
        gradient buffer as in `_run_backward`.  Returns (v_d, t_d, v_j, t_j).
        `pipe` (a dict, `Trainer.step` with TAN_STEP_PIPELINE): steps are pipelined across their boundary.  In: the events the previous
        step left in `_Flat.pending` -- "dw_v" / "dw_j" (a stack's last weight-gradient launches: they read the activation workspaces this
        step is about to overwrite: the workspaces alternate between two sets instead of waiting), "video" / "joint" (the optimizer launch
        of a stack's matrices behind them: that stack's forward waits for it, and only for it) -- and "zero" (the gradient fill: the first
        backward kernel of each chain waits for it).  Out: the same events of THIS step, and the main stream is NOT joined with the
        streams that carry them: the embeddings and the video stack of the next step run under the joint stack's last weight gradients
        and optimizer launch.
        `pre` ({"video": fn, "joint": fn}, stage 2): called on each chain's stream / host thread in front of the online stack's forward,
        behind the wait for that stack's optimizer launch of the previous step (the EMA target's stack of the same kind, whose weights
        that launch also wrote).  `family` may then synchronise the two chains itself (it is called once per chain, on the chain's own
        host thread): stage 2's upstream gradients depend on both families' forward results."""
        self._ensure_flat()
        self._bind_grads()
        f = self._flat
        prev = f.pending if pipe is not None else {}
        B, T, _ = video.shape
        N = lang.shape[1]
        cd, dev = self.compute_dtype, video.device
        Se, Sd, Cw = self.num_encoder_layers, self.num_decoder_layers, WIDTH
        R, Mp, L = B * T, B * N, T + N
        p_v = self._draw(T, None)
        p_t = self._draw(N, None) if self.use_text_pos_enc else 0
        p_j = self._draw(T, None)
        f.sync_shadow_t()
        f.sync_shadow_tp()
        # (pipelined: the stacks' activation workspaces alternate between two sets -- the previous step's last weight-gradient launches
        #  still read theirs; each stack waits for its own optimizer launch, which is behind those on the same stream)
        self._ws_alternate = pipe is not None
        try:
            fe = self._embed_fused(video, lang, vmask_u8, tmask_u8, p_v, p_t, p_j, True)
        finally:
            self._ws_alternate = False
        em = fe["em"]
        zero_ev = pipe.get("zero") if pipe is not None else None
        dst_v = torch.empty(Se, R, Cw, dtype=cd, device=dev)
        d_lang_raw = torch.empty(Mp, Cw, dtype=cd, device=dev)
        d_x0 = torch.empty(R, Cw, dtype=cd, device=dev)
        main, side = torch.cuda.current_stream(), self._side_stream(dev)
        # `serialize_streams` (bench.py's `roofline.isolated`): the same schedule with both chains and the weight-gradient tails on the
        # CURRENT stream -- every kernel of the step runs alone on the chip, in the order the two host threads happen to issue them
        serial = bool(getattr(self, "serialize_streams", False))
        if serial:
            side = main
        side.wait_stream(main)
        # The weight-gradient launches of the LAST blocks of each stack's backward feed only the optimizer: they run on an otherwise idle
        # role stream next to the stack's remaining dX kernels and the embeddings' backward (tan_encoder_desc.dw_tail).  Joint chain (the
        # longer one): blocks 1 and 0; video chain: block 0.  Measured (ms per step, ABBA x2 on one box): none 4.58, (1, 1) 4.50 / 4.44,
        # (2, 1) 4.40, (3, 1) 4.40, (2, 2) 4.45, (6, 1) 4.47 -- earlier than the last ~0.4 ms of the chain there are no idle CUs to give.
        tail_j, tail_v = 2, 1
        from .workspace import SPLIT_PANELS
        if B * L <= 64 * SPLIT_PANELS:
            # small batches (the split-hidden launches' range): the chip is mostly idle and the chains are latency-bound -- EVERY block's
            # weight-gradient launch leaves the chain (B = 16: 6 x 40 us per chain)
            tail_j, tail_v = Sd, Se
        aux_j = main if serial else _lib.role_stream(dev, "loss")
        aux_v = main if serial else _lib.role_stream(dev, "opt")
        # data parallel with gradient buckets: a layer's event must mean "every gradient of the layer is final" on the stack's own stream,
        # so the weight gradients stay on the chains; the bucket all-reduces are issued by the hook, from THIS thread, video first
        hook = self._grad_ready_hook
        dw_j, dw_v = (None, None) if hook is not None else (aux_j, aux_v)
        # (Measured in round 5 and not kept: the optimizer launch of a stack's UPPER layers -- whose weight gradients ride the chain -- on a
        #  third stream as soon as the lowest of them is differentiated.  The step's boundary shrinks by 0.07 ms and the backward grows by
        #  as much: 4.290 vs 4.271 ms, ABBA x2 of 60 steps.  HBM-bound launches next to the stacks' kernels cost what they save.)

        import threading
        joint_terms, joint_ready = [], threading.Event()      # the joint family's terms as soon as its launches are enqueued (for `mid`)
        self._joint_terms = (joint_terms, joint_ready)

        def joint_chain():
            try:
                dst_j = torch.empty(Sd, B * L, Cw, dtype=cd, device=dev)
                d_xj = torch.empty(B * L, Cw, dtype=cd, device=dev)
                if prev.get("joint") is not None:      # the joint stack's weights of this step
                    torch.cuda.current_stream().wait_event(prev["joint"])
                if pre is not None:
                    pre["joint"]()
                ej = self._run_joint_stack(None, None, vmask_u8, tmask_u8, B, T, N, True, pre=(fe["ej"], fe["xj"], fe["keypad"]))
                stages = [ej.stage(s) for s in range(Sd)]
                dj = [dst_j[s] for s in range(Sd)]
                if zero_ev is not None:            # (the gradient fill: over long before; a family may add to parameter gradients itself)
                    torch.cuda.current_stream().wait_event(zero_ev)
                # frame rows b*L + t and sentence rows b*L + T + k of the SAME stage buffers (tan_model.py:207-209), and of their gradients
                v_j, t_j = family("joint", stages, (L, 0), stages, (L, T), dj, dj)
                joint_terms.append((v_j, t_j, torch.cuda.current_stream().record_event()))
            finally:
                joint_ready.set()          # (also on failure: `mid` must not wait out its timeout for terms that will never come)
            self._encoder_bwd(ej, ej.xj, ej.keypad, "ln_joint_post_enc", dj, d_xj, dw_stream=dw_j, dw_tail=tail_j)
            return ej, v_j, t_j, d_xj, (dst_j,)
        fut = self._on_side(side, joint_chain)
        try:
            if prev.get("video") is not None:
                main.wait_event(prev["video"])
            if pre is not None:
                pre["video"]()
            ev = self._run_video_stack(fe["x0"], vmask_u8, B, T, True, er=fe["ev"])
            dv = [dst_v[s] for s in range(Se)]
            if zero_ev is not None:
                main.wait_event(zero_ev)
            v_d, t_d = family("dual", [ev.stage(s) for s in range(Se)], (T, 0), [fe["lang_raw"]], (N, 0), dv, [d_lang_raw])
        except BaseException as main_exc:
            # (a family that synchronises the chains fails on BOTH threads when one of them does: report the root cause, not the
            #  "other chain did not arrive" that follows from it; either way the helper thread is through before the error leaves)
            exc = fut.exception()
            if exc is not None and getattr(main_exc, "tan_consequence", False) and not getattr(exc, "tan_consequence", False):
                raise exc
            raise
        self._dual_terms = (v_d, t_d)
        self._encoder_bwd(ev, fe["x0"], vmask_u8, "ln_video_post_enc", dv, d_x0, dw_stream=dw_v, dw_tail=tail_v)
        if hook is not None:
            hook("video", self._layer_events(ev.prefix, ev.layers))
        out_ev = {"dw_v": aux_v.record_event() if dw_v is not None else None}
        if after_video_bwd is not None:          # every gradient of the video stack's blocks is final (enqueued) here: its all-reduce (data
            aux_v.wait_stream(main)                # parallel) and the optimizer launch of its matrices go BEHIND the block-0 weight gradients on
            with torch.cuda.stream(aux_v):         # their stream: this one is free for the embeddings' backward as soon as the joint chain is through
                after_video_bwd()
        out_ev["video"] = aux_v.record_event()
        if mid is not None:                      # (main stream, behind the video stack's backward: e.g. the loss's masked means, which
            try:                                 #  would otherwise sit between the embeddings' backward and the optimizer launch)
                mid()
            except BaseException:
                if fut.done() and fut.exception() is not None:      # the joint chain failed first: that is the error to report
                    raise fut.exception()
                raise
        ej, v_j, t_j, d_xj, keep = fut.result()
        if hook is not None:
            with torch.cuda.stream(side):
                hook("joint", self._layer_events(ej.prefix, ej.layers))
        out_ev["dw_j"] = aux_j.record_event() if dw_j is not None else None
        if after_joint_bwd is not None:          # (issued from this thread like everything that may be a collective: same order on every rank)
            aux_j.wait_stream(side)
            with torch.cuda.stream(aux_j):
                after_joint_bwd()
        main.wait_stream(side)
        for t in (d_xj,) + keep:                 # allocated under the side stream, read (or freed) under this one
            t.record_stream(main)
        run = {"em": em, "B": B, "T": T, "N": N, "sv_video": fe["sv_video"], "sv_video_j": fe["sv_video_j"], "sv_text": fe["sv_text"],
               "sv_text_t": fe["sv_text_t"]}
        self._chain_d_lang = self._embed_bwd_fused(run, d_x0, d_xj, d_lang_raw, need_d_lang)      # [B, N, Dt] f32 or None
        out_ev["joint"] = aux_j.record_event()
        if pipe is not None:                       # the next step waits for each of them where it needs it (`_Flat.pending`)
            pipe["out"] = out_ev
        else:
            main.wait_stream(aux_v)                # the stacks' last weight gradients (and the optimizer launches behind them)
            main.wait_stream(aux_j)
        self._release_ws(ev)
        self._release_ws(ej)
        self._release_ws(em)
        return v_d, t_d, v_j, t_j

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
        d_lang_raw = torch.empty(Mp, Cw, dtype=cd, device=dev)      # (its first writer overwrites: no fill launch on the backward chain)
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
                ops.head_bwd(g_ad.contiguous().view(Mp).float(), run["lang_raw"], w, d_lang_raw, gw, gb, Mp, Cw, accumulate_dx=have_lang_raw)
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
        d_x0 = torch.empty(R, Cw, dtype=cd, device=dev)             # written in full by the video stack's backward, or by the first rows_copy
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
            # the last blocks' weight-gradient launches on idle role streams (see _run_chains); not under a DDP bucket hook, whose layer
            # events mean "every gradient of the layer is final" on the stack's own stream
            tail = self._grad_ready_hook is None and dev.type == "cuda"
            aux_j = _lib.role_stream(dev, "loss") if tail else None
            aux_v = _lib.role_stream(dev, "opt") if tail else None
            fut = self._on_side(side, lambda: self._encoder_bwd(ej, ej.xj, ej.keypad, "ln_joint_post_enc", dst_j, d_xj, dw_stream=aux_j,
                                                                dw_tail=2))
            self._encoder_bwd(ev, run["x0"], run["vmask"], "ln_video_post_enc", dst_v, d_x0, dw_stream=aux_v, dw_tail=1)
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
            tail_streams = [t for t in (aux_j, aux_v) if t is not None]
        else:
            tail_streams = []
            if any_j:
                self._encoder_bwd(ej, ej.xj, ej.keypad, "ln_joint_post_enc", dst_j, d_xj)
                if self._grad_ready_hook is not None:    # joint-stack gradients are final: DDP starts reducing them now
                    self._grad_ready_hook("joint", self._layer_events(ej.prefix, ej.layers))
            if any_v:
                self._encoder_bwd(ev, run["x0"], run["vmask"], "ln_video_post_enc", dst_v, d_x0)
                if self._grad_ready_hook is not None:
                    self._grad_ready_hook("video", self._layer_events(ev.prefix, ev.layers))
        if run.get("em") is not None:          # the fused front-end ran (TAN_EMBED_FUSED): its backward is fused too
            d_lang = self._embed_bwd_fused(run, d_x0 if any_v else None, d_xj, d_lang_raw if have_lang_raw else None, need_d_lang)
            for t in tail_streams:
                main.wait_stream(t)
            self._release_ws(ev)
            self._release_ws(ej)
            self._release_ws(run.get("em"))
            return d_lang
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
        aux = self._side_stream(dev)
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
        for t in tail_streams:
            cur.wait_stream(t)
        self._release_ws(ev)
        self._release_ws(ej)
        self._release_ws(run.get("em"))
        return d_lang
