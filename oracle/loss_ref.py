"""CPU restatement of train/loss.py:get_loss (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Written from the algorithm, not from the reference text: windows are built by index
arithmetic instead of flip/cat/unfold, diagonal blocks are taken with einsum, and every
intermediate the parity tests need is returned in `aux`.  Reference quirks that change
results are reproduced and called out where they occur.
"""
from __future__ import annotations

import types

import torch
import torch.nn.functional as F

TEMPERATURE = 0.07
FILL = -6e4


def circulant(x, dim):
    """All cyclic right-shifts of x along `dim`, new axis last: out[..., i, j] = x[..., (j-i) mod S]
    (loss.py:16-23; docstring known-answer [0,1,2] -> [[0,1,2],[2,0,1],[1,2,0]])."""
    S = x.shape[dim]
    x = x.movedim(dim, -1)
    j = torch.arange(S)
    idx = (j[None, :] - j[:, None]) % S                       # [i, j]
    out = x[..., idx]                                         # [..., i, j]
    return out.movedim(-2, dim) if dim not in (-1, x.dim() - 1) else out


def mask_from_time(start_list, end_list, num_timestamp, num_text):
    """get_mask_from_time -- loss.py:26-41.  Ragged python lists -> ([B,N,T] bool, start [B,N'], end [B,N']);
    padded starts are T+100 and padded ends -100 so padded texts get an empty mask."""
    B = len(start_list)
    n_max = max(len(s) for s in start_list)
    start = torch.full((B, n_max), float(num_timestamp) + 100.0)
    end = torch.full((B, n_max), -100.0)
    for b, (s, e) in enumerate(zip(start_list, end_list)):
        start[b, :len(s)] = torch.tensor(s, dtype=torch.float32)
        end[b, :len(e)] = torch.tensor(e, dtype=torch.float32)
    t = torch.arange(num_timestamp)[None, None, :].expand(B, num_text, -1)
    mask = (start[:, :, None] <= t) & (t < end[:, :, None])
    return mask, start, end


def text_pos(start_list, end_list):
    """get_text_pos -- loss.py:44-52: zero-padded [B,N,2]."""
    B = len(start_list)
    n_max = max(len(s) for s in start_list)
    out = torch.zeros(B, n_max, 2)
    for b, (s, e) in enumerate(zip(start_list, end_list)):
        out[b, :len(s), 0] = torch.tensor(s, dtype=torch.float32)
        out[b, :len(e), 1] = torch.tensor(e, dtype=torch.float32)
    return out


def _diag_blocks(x):
    """x [B,S,T,B,N] -> same-video blocks [B,S,T,N] (loss.py:91-95,147-151)."""
    return torch.einsum("bstbn->bstn", x)


def window_bank(durations, T):
    """Normalised sliding windows, loss.py:112-131.  durations [B,N] (float, 0 for padded texts).
    bank[b,n,i,:] averages over [i, i+d) when that fits in [0,T); frames 0 and T-1 are then
    removed from every window ("never choose temp-index 0 / -1") and the rest renormalised."""
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    d = durations[:, :, None, None]
    member = (j >= i) & (j < i + d) & (i + d <= T)            # [B,N,T,T]
    member = member & (j != 0) & (j != T - 1)
    cnt = member.sum(-1, keepdim=True).float().clamp(min=1e-3)
    return member.float() / cnt


def self_label(diag_logits, video_pad, text_pad, bank):
    """One branch (joint or dual) of the self-labelling, loss.py:96-143 / 152-179.
    diag_logits [B,S,T,N] already divided by the temperature.  Returns dict with the masked
    logits, chosen window start (argmax, first index on ties), window weights, mean window logit."""
    z = diag_logits.clone()
    z = z.masked_fill(video_pad[:, None, :, None], FILL)
    z = z.masked_fill(text_pad[:, None, None, :], FILL)
    prob = torch.softmax(torch.softmax(z, dim=-1) / TEMPERATURE, dim=-2)       # over N, then over T
    prob_last = prob[:, -1]                                                    # [B,T,N]
    logit_last = z[:, -1]
    scan = (prob_last.permute(0, 2, 1)[:, :, None, :] * bank).sum(-1)          # [B,N,T]
    max_prob, max_pos = scan.max(-1)
    w = torch.gather(bank, 2, max_pos[:, :, None, None].expand(-1, -1, 1, bank.shape[-1])).squeeze(2)
    max_logit = (logit_last.permute(0, 2, 1) * w).sum(-1)                      # [B,N]
    return {"masked": z, "prob_last": prob_last, "scan": scan, "max_prob": max_prob,
            "max_pos": max_pos, "window": w, "max_logit": max_logit, "tgt": (w > 0)}   # tgt [B,N,T]


def _block_diag(x_btn, B):
    """[B,T,N] -> [B,T,B,N] with x on the same-video blocks, zero elsewhere."""
    eye = torch.eye(B, dtype=x_btn.dtype)
    return x_btn[:, :, None, :] * eye[:, None, :, None]


def nce(scaled_logits, tgt_cols, keep_cols):
    """Symmetric multi-positive NCE over all stages, loss.py:240-275.
    scaled_logits [B,S,T,B,N]; tgt_cols [B*T, M] in {0,1}; keep_cols [B,N] bool (non-padded texts)."""
    B, S, T = scaled_logits.shape[:3]
    x = scaled_logits[:, :, :, keep_cols].permute(1, 0, 2, 3).reshape(S, B * T, -1)
    xp = x.masked_fill(~tgt_cols.bool()[None], FILL)
    v = torch.logsumexp(x, -1) - torch.logsumexp(xp, -1)                       # [S, B*T]
    t = torch.logsumexp(x, -2) - torch.logsumexp(xp, -2)                       # [S, M]
    return v, t


def get_loss(input_data, video_seq, text_embed, video_padding_mask, text_padding_mask, logits, args,
             abs_text_pos=None, decisions=None):
    """get_loss -- loss.py:55-422.  Returns (loss_dict, aux).
    `decisions` (tests only, never set by the golden pinning): the DISCRETE results of the no-grad sections taken from the caller
    instead of computed here -- "tgt" [B,T,N] {0,1} (the de-duplicated agreement target of loss.py:88-229, same-video blocks),
    "th_mask" [M] bool (loss.py:286) and "lab" [M] in {0,1,2} (loss.py:309-328) over the non-padded sentences.  A bf16 forward flips
    a few of these decisions near their thresholds; with them pinned the loss is a smooth function of the logits and its gradient can
    be compared with a norm-relative bound (the decisions themselves are compared bit for bit in fp32 mode)."""
    cotrain = args.model == "cotrain"
    B, T, _ = video_seq.shape
    N = text_embed.shape[1]
    vpad = video_padding_mask.bool()
    tpad = text_padding_mask.bool()
    keep = ~tpad
    scale = (1.0 / TEMPERATURE) if args.sim == "cos" else 1.0
    # the reference divides (loss.py:65-70), creating fresh tensors it later mutates in place
    ld = logits["logits_dual"] / TEMPERATURE if args.sim == "cos" else logits["logits_dual"] * 1.0
    lj = logits["logits_joint"] / TEMPERATURE if args.sim == "cos" else logits["logits_joint"] * 1.0
    aux, out = {}, {}

    tgt_raw, _, _ = mask_from_time(input_data["start"], input_data["end"], T, N)     # [B,N,T] bool
    binary_tgt = _block_diag(tgt_raw.permute(0, 2, 1).float(), B)                    # [B,T,B,N]

    fixed = decisions or {}
    if args.learn_agreement and "tgt" in fixed:
        assert cotrain, "pinned targets: the 'init' in-place leak depends on the self-labelling path"
        tgt_full = _block_diag(fixed["tgt"].float(), B)
        out["iou-threshold"] = torch.tensor(0.5)
    elif args.learn_agreement:
        with torch.no_grad():
            if cotrain:
                src_j = logits["ema-logits_joint"] / TEMPERATURE if args.sim == "cos" else logits["ema-logits_joint"]
                src_d = logits["ema-logits_dual"] / TEMPERATURE if args.sim == "cos" else logits["ema-logits_dual"]
            else:
                src_j, src_d = lj, ld
            dur = tgt_raw.sum(-1).float().clamp(min=1.0).masked_fill(tpad, 0.0)      # loss.py:113-115
            bank = window_bank(dur, T)
            J = self_label(_diag_blocks(src_j), vpad, tpad, bank)
            D = self_label(_diag_blocks(src_d), vpad, tpad, bank)
            jt, dt = J["tgt"], D["tgt"]                                               # [B,N,T] bool
            inter = (jt & dt).sum(-1).float()
            union = (jt | dt).sum(-1).float()
            iou = inter / union.clamp(min=1e-5)                                       # [B,N]
            conf_d = D["max_logit"] >= torch.quantile(D["max_logit"][keep].float(), 0.3)
            conf_j = J["max_logit"] >= torch.quantile(J["max_logit"][keep].float(), 0.3)
            conf_iou = iou >= 0.5
            conf = conf_d & conf_j & conf_iou
            kind = args.temporal_agreement_type
            if kind == "i":
                agree = (jt & dt) & conf[:, :, None]
            elif kind == "u":
                agree = (jt | dt) & conf[:, :, None]
            elif kind == "keep":
                agree = torch.where(conf_iou[:, :, None], jt | dt, tgt_raw)
            elif kind == "keep-joint":
                agree = torch.where(conf_iou[:, :, None], jt, tgt_raw)
            else:
                raise ValueError(kind)
            agree = agree.float()                                                     # [B,N,T]
            # exclusion (loss.py:216-226): per (video, t) keep only the first text; text 0 keeps its own
            # row; texts left with no positive at all get the YouTube target back.
            first = agree.argmax(1)                                                   # [B,T], 0 if none
            dedup = torch.zeros_like(agree)
            dedup.scatter_(1, first[:, None, :], 1.0)
            dedup[:, 0, :] = agree[:, 0, :]
            lost = dedup.sum(-1) == 0                                                 # [B,N]
            dedup[lost] = tgt_raw.float()[lost]
            tgt_full = _block_diag(dedup.permute(0, 2, 1), B)                         # [B,T,B,N]
            out["confidence-ratio"] = conf[keep].float().mean()
            out["iou-threshold"] = torch.tensor(0.5)
            aux.update(max_position_joint=J["max_pos"], max_position_dual=D["max_pos"],
                       max_logits_joint=J["max_logit"], max_logits_dual=D["max_logit"],
                       iou=iou, confidence_mask=conf, agreement_self_tgt=tgt_full,
                       prob_scan_joint=J["scan"], prob_scan_dual=D["scan"])
        if not cotrain:
            # QUIRK (loss.py:96-101,152-157): masked_fill_ runs in place on a *view* of the scaled
            # online logits, so for model='init' the -6e4 fills leak into the NCE below (same-video
            # blocks only, where the frame or the text is padding).
            leak = _block_diag((vpad[:, :, None] | tpad[:, None, :]).float(), B).bool()[:, None]
            lj = torch.where(leak, torch.tensor(FILL), lj)
            ld = torch.where(leak, torch.tensor(FILL), ld)
    else:
        tgt_full = binary_tgt

    tgt_cols = tgt_full[:, :, keep].reshape(B * T, -1)                                # [B*T, M]
    rows_pos = tgt_cols.sum(-1) > 0
    cols_pos = tgt_cols.sum(-2) > 0
    aux["tgt_cols"] = tgt_cols

    v_d, t_d = nce(ld, tgt_cols, keep)
    v_j, t_j = nce(lj, tgt_cols, keep)
    loss_dual = (v_d[:, rows_pos].mean() + t_d[:, cols_pos].mean()) / 2
    loss_joint = (v_j[:, rows_pos].mean() + t_j[:, cols_pos].mean()) / 2
    out["loss-dual"] = loss_dual.detach()
    out["loss-joint"] = loss_joint.detach()

    if args.loss_threshold > 0 or args.use_alignability_head:
        with torch.no_grad():
            # per-text max over time of the last-stage same-video logits (online model), loss.py:280-283
            md = _diag_blocks(ld)[:, -1].permute(1, 0, 2)[:, keep].max(0).values        # [M]
            mj = _diag_blocks(lj)[:, -1].permute(1, 0, 2)[:, keep].max(0).values
            zd = (md - md.mean()) / md.std()
            zj = (mj - mj.mean()) / mj.std()
            metric = -(zd + zj)
            th_mask = metric <= torch.quantile(metric.float(), args.loss_threshold, -1, keepdim=True)
            if "th_mask" in fixed:
                th_mask = fixed["th_mask"].bool()
            tgt_th = tgt_cols.clone()
            tgt_th[:, ~th_mask] = 0
            rows_pos_th = tgt_th.sum(-1) > 0
            aux.update(t_th_mask=th_mask, max_logits_dual_per_text=md, max_logits_joint_per_text=mj)
        if args.loss_threshold > 0:
            out["loss-dual-all"] = loss_dual.detach()
            out["loss-joint-all"] = loss_joint.detach()
            # QUIRK (loss.py:296,301): the [M]-long mask indexes the already cols_pos-filtered tensor,
            # i.e. the reference assumes every non-padded text has a positive.
            t_d_sel, t_j_sel = t_d[:, cols_pos], t_j[:, cols_pos]
            assert t_d_sel.shape[1] == th_mask.shape[0], "reference would raise IndexError here"
            loss_dual_th = (v_d[:, rows_pos_th].mean() + t_d_sel[:, th_mask].mean()) / 2
            loss_joint_th = (v_j[:, rows_pos_th].mean() + t_j_sel[:, th_mask].mean()) / 2
            out["loss-dual"] = loss_dual_th.detach()
            out["loss-joint"] = loss_joint_th.detach()
        if args.use_alignability_head:
            with torch.no_grad():
                lab = torch.full_like(metric, 2.0)                                      # 2 = ignore
                med_d = torch.quantile(md.float(), 0.5, keepdim=True)
                med_j = torch.quantile(mj.float(), 0.5, keepdim=True)
                lab = lab.masked_fill((md > med_d) & (mj > med_j), 1.0)
                lab = lab.masked_fill((md < med_d) & (mj < med_j), 0.0)
                if abs_text_pos is not None:
                    centre = abs_text_pos[keep, :].mean(-1)
                    lab = lab.masked_fill((centre < 0.2) | (centre > 0.8), 0.0)
                if "lab" in fixed:
                    lab = fixed["lab"].float()
                aux["t_align_th_mask"] = lab
            a_dual = logits["dual_logits_alignability"][..., 0][keep][cols_pos]
            a_joint = logits["joint_logits_alignability"][:, 2, :, 0][keep][cols_pos]   # stage index 2 hard-coded
            sel = lab != 2
            y = lab[sel]
            pw = torch.ones_like(y) * (1.0 / y.mean() - 1.0)
            bce_joint = F.binary_cross_entropy_with_logits(a_joint[sel], y, pos_weight=pw)
            bce_dual = F.binary_cross_entropy_with_logits(a_dual[sel], y, pos_weight=pw)   # computed, unused (loss.py:350)
            out["loss-joint-bce"] = bce_joint.detach()
            out["alignability_top1"] = ((a_joint[sel] > 0).float() == y).detach().float().mean()

    nce_w = 0 if args.optim_policy == "bce" else 1
    if args.loss_threshold > 0:
        out["loss-total"] = ((loss_dual + loss_joint) / 2).detach()
        loss = (loss_dual_th + loss_joint_th) / 2
    else:
        loss = (loss_dual + loss_joint) / 2
    if args.use_alignability_head:
        loss = loss * nce_w + bce_joint
    out["loss"] = loss
    return out, aux


def default_args(**kw):
    """The subset of train/config.py:6-53 that get_loss reads, with the reference's defaults."""
    a = dict(model="init", sim="cos", learn_agreement=0, temporal_agreement_type="keep",
             loss_threshold=0.0, use_alignability_head=0, optim_policy="default")
    a.update(kw)
    if a["model"] == "cotrain":                      # train/main.py:361-363
        a["learn_agreement"] = 1
        a["use_alignability_head"] = 1
    return types.SimpleNamespace(**a)
