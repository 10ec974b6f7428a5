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
from .engine import _AlignerEngine, _AlignerFn
from .flat_params import _ALIGN, _Flat, _vp  # noqa: F401  (_vp: re-exported for train.py)
from .tfm_model import TemporalEncoder, _LayerNormParams, _LinearParams, get_position_embedding_sine
from .workspace import HEADS, WIDTH


def _local_bert():
    """HuggingFace BertModel('bert-base-uncased') from local files only (tan_model.py:38), or None."""
    import os
    try:
        from transformers import BertModel
        return BertModel.from_pretrained(os.environ.get("TAN_BERT_PATH", "bert-base-uncased"), local_files_only=True)
    except Exception:          # noqa: BLE001  (no local weights / no transformers: the embeddings come from the caller)
        return None


class TemporalAligner(_AlignerEngine, nn.Module):
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
        elif language_model == "bert":
            # tan_model.py:37-38,41,49: 768-d sentence embeddings into text_pre_proj.  forward() never calls the language model
            # (train/main.py:47-79 embeds the text first) and the BERT encoder is outside this path (f1 is the
            # Word2Vec model): `self.bert` is HuggingFace's BertModel when a local copy of the weights is available
            # (TAN_BERT_PATH or the HF cache; no download is attempted), else None -- the caller then passes [B, N, 768] embeddings
            self.bert = _local_bert()
        else:
            raise NotImplementedError(f"language_model={language_model!r}: 'word2vec' (tan_model.py:39-40), 'bert' (:37-38) or None")
        text_embed_dim = {"bert": 768}.get(language_model, 512)           # tan_model.py:41

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
        self.serialize_streams = False    # measurement: the two-chain step with every stream collapsed onto the current one
        self._side = None
        self._issuer = None               # helper thread issuing the side-stream stack (see _on_side)
        self._lp_cache = {}
        self.transposed_dx = True         # dX GEMMs read W^T copies (K-contiguous); their packed images feed the row-panel backward
        self.panel_kernels = os.environ.get("TAN_PANEL", "1") != "0"           # row-panel fused kernels (packed weight images)
        self._grad_ready_hook = None      # callable(tag, layer_events) fired inside backward once a stack's backward is enqueued
        # load_state_dict copies into the parameter tensors, whose version counters are not the flat buffer's: the bf16
        # shadow (and the W^T copies built from it) must be rebuilt from the f32 masters on the next forward
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_shadow())

    def state_dict(self, *args, **kwargs):
        """nn.Module.state_dict behind the work a pipelined training step may have left on its role streams (`_Flat.pending`)."""
        f = self.__dict__.get("_flat")
        if f is not None:
            f.drain()
        return super().state_dict(*args, **kwargs)

    def _drain_pending(self):
        """The current stream waits for what a pipelined `Trainer.step` left running on its role streams (the stacks' last weight
        gradients, the optimizer launches of their matrices, the weight images): every accessor through which code OUTSIDE the step can
        reach the parameter tensors calls this first.  A no-op inside the step and when nothing is pending."""
        f = self.__dict__.get("_flat")
        if f is not None:
            f.drain()

    def named_parameters(self, *args, **kwargs):
        """(`parameters()` goes through here too.)  Reads / writes of `p.data` on the current stream after a pipelined step are ordered
        behind that step's optimizer launches."""
        self._drain_pending()
        return super().named_parameters(*args, **kwargs)

    def invalidate_shadow(self):
        """Call after writing parameters in place other than through the optimizer kernel / load_state_dict (e.g.
        `p.data.copy_(...)`): the next forward re-casts the bf16 shadow weights from the f32 masters."""
        self._drain_pending()
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
        f.drain()             # (work a pipelined training step left on its role streams; a no-op inside that step and otherwise)
        f.sync_shadow()
        if self.panel_kernels:
            f.sync_shadow_p()
        return f

    def _ensure_flat_nosync(self):
        """The flat buffers without touching the derived images (the caller refreshes them itself)."""
        f = self._flat
        if not f.bound():
            return self._ensure_flat()
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

    # ------------------------------------------------------------------ public surface (tan_model.py:100-312)
    @staticmethod
    def _mask_u8(m):
        if m is None:
            return None
        if m.dtype == torch.bool and m.is_contiguous():
            return m.view(torch.uint8)          # same bytes (0 / 1): no launch
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
        # (called before the children's: an optimizer launch a pipelined step left in flight must not overwrite what is copied in)
        self._drain_pending()
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

    def named_parameters(self, *args, **kwargs):
        """nn.Module walks `_parameters` of the submodules itself: the twins' own accessors are not on that path, so the work a pipelined
        step left on its role streams is waited for here."""
        for m in (self.online, self.target):
            m._drain_pending()
        return super().named_parameters(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        for m in (self.online, self.target):
            m._drain_pending()
        return super().state_dict(*args, **kwargs)

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
