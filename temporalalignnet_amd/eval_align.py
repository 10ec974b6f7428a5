"""HTM-Align zero-shot alignment evaluation: counterpart of eval/eval_zeroshot_align.py:test_alignment_htm (97-252)
and of the `get_text_visual_sim` closure in train/main.py:171-189.

64-s windows with stride 16 over each video; the active sentences of a window are chosen from the NON-alignable
sentences' ASR timestamps only (no ground-truth leak, :149-167); last-stage joint and dual similarities are stitched
and averaged (:197-205), zeros -> -6e4, softmax over time, arg-max inside [floor(start), ceil(end)] -> R@1; the joint
alignability head (stage index 2, :186) -> ROC-AUC.  Model calls run on the HIP path; the per-video bookkeeping is
host-side index arithmetic on [K, vlen] tensors.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def roc_auc_score(y_true, y_score) -> float:
    """ROC-AUC by the rank statistic with average ranks for ties (= sklearn.metrics.roc_auc_score, :248)."""
    y = np.asarray(y_true).astype(bool)
    s = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    ranks = np.empty(len(s))
    bounds = np.flatnonzero(np.r_[True, ss[1:] != ss[:-1], True])
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ranks[order[lo:hi]] = 0.5 * (lo + hi - 1) + 1.0
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def make_sim_fn(model, embed_text, use_alignability_head=True):
    """The closure of train/main.py:171-189: `embed_text(list[str]) -> [K, 512]` stands for tokenizer + language model."""

    @torch.no_grad()
    def get_text_visual_sim(video_embed, text_str, interpolate_from=None, abs_text_pos=None):
        text_embed = embed_text(text_str)[None]
        out = {"sim": model.get_text_visual_sim_joint(video_embed, text_embed, interpolate_from).transpose(-1, -2) / 0.07,
               "dual-sim": model.get_text_visual_sim_dual(video_embed, text_embed, interpolate_from).transpose(-1, -2) / 0.07}
        if use_alignability_head:
            out.update(model.get_alignability(video_embed, text_embed, interpolate_from, abs_text_pos))
        return out

    return get_text_visual_sim


def make_batched_sim_fn(model, embed_text, use_alignability_head=True, max_windows=256):
    """Batched counterpart of make_sim_fn for `test_alignment_htm(batched_sim=...)`: all windows of a video go through
    `model.eval_windows` in chunks of `max_windows` (one pass of each stack per chunk instead of four B=1 passes per window).
    `run(video [1,vlen,Dv], text list[str], windows [(s0, e0, bool mask [K])], seq_len)` returns, per window, the dict
    `get_text_visual_sim` would have returned for it."""

    @torch.no_grad()
    def run(video, text_str, windows, seq_len):
        dev = video.device
        emb = embed_text(list(text_str))                                  # [K, 512]: sentences are embedded independently
        out = []
        for c0 in range(0, len(windows), max_windows):
            chunk = windows[c0:c0 + max_windows]
            W, kmax = len(chunk), max(int(m.sum()) for _, _, m in chunk)
            vid = torch.zeros(W, seq_len, video.shape[-1], device=dev, dtype=video.dtype)
            vmask = torch.ones(W, seq_len, dtype=torch.bool, device=dev)
            txt = torch.zeros(W, kmax, emb.shape[-1], device=dev, dtype=emb.dtype)
            tmask = torch.ones(W, kmax, dtype=torch.bool, device=dev)
            for w, (s0, e0, m) in enumerate(chunk):
                k = int(m.sum())
                vid[w, :e0 - s0] = video[0, s0:e0]
                vmask[w, :e0 - s0] = False
                txt[w, :k] = emb[torch.from_numpy(m).to(dev)]
                tmask[w, :k] = False
            r = model.eval_windows(vid, txt, vmask, tmask)
            sim_j = r["sim"].transpose(-1, -2) / 0.07                     # [W,S,K,T]
            sim_d = r["dual-sim"].transpose(-1, -2) / 0.07
            for w, (s0, e0, m) in enumerate(chunk):
                k, t = int(m.sum()), e0 - s0
                d = {"sim": sim_j[w:w + 1, :, :k, :t], "dual-sim": sim_d[w:w + 1, :, :k, :t]}
                if use_alignability_head:
                    d["alignability-dual"] = r["alignability-dual"][w:w + 1, :k]
                    d["alignability-joint"] = r["alignability-joint"][w:w + 1, :, :k]
                out.append(d)
        return out

    return run


@torch.no_grad()
def test_alignment_htm(get_text_visual_sim, videos, device="cuda", seq_len=64, use_alignability_head=True,
                       method="overlap-seq", return_per_video=False, batched_sim=None):
    """`videos`: iterable of {'video' [vlen, Dv], 'start' [K], 'end' [K], 'aligned' [K] 0/1, 'str' list[str]}
    (htm_align.json schema, htm_align/readme.md:11-20).  Returns {'Recall', 'AUC'}.  `batched_sim` (make_batched_sim_fn): the
    windows of a video are evaluated together instead of one model call per window (same results, see the GPU tests)."""
    recall, scores, tgts, per_video = [], [], [], []
    for item in videos:
        video = torch.as_tensor(item["video"]).float().to(device)[None]
        text = list(item["str"])
        aligned = np.asarray(item["aligned"]).astype(bool)
        start = np.asarray(item["start"], dtype=np.float64)
        end = np.asarray(item["end"], dtype=np.float64)
        K, vlen = len(text), video.shape[1]
        abs_pos = torch.stack((torch.as_tensor(item["start"]), torch.as_tensor(item["end"])), -1).div(vlen).to(device)
        if method == "overlap-seq":
            steps = np.arange(0, vlen - seq_len // 2, seq_len // 4)
            mid = (start + end) / 2
            acc_j = torch.zeros(K, vlen, device=device)
            acc_d = torch.zeros(K, vlen, device=device)
            cnt = torch.zeros(K, vlen, device=device)
            a_d, a_j, tcnt = (torch.zeros(K, device=device) for _ in range(3))
            na_idx, na_mid = np.arange(K)[~aligned], mid[~aligned]
            windows = []
            for i, s0 in enumerate(steps):
                inside = (s0 - seq_len <= na_mid) & (na_mid <= s0 + 2 * seq_len)
                act = na_idx[inside]
                if len(act) == 0:
                    continue
                left, right = act.min(), act.max()
                if i <= 3:
                    left = 0
                elif i >= len(steps) - 4:
                    right = vlen
                m = np.zeros(K, bool)
                m[left:right + 1] = True
                if not m.any():
                    continue
                windows.append((int(s0), int(min(vlen, s0 + seq_len)), m))
            if batched_sim is not None:
                results = batched_sim(video, text, windows, seq_len)
            else:
                results = (get_text_visual_sim(video[:, s0:e0], [t for t, k in zip(text, m) if k],
                                               abs_text_pos=abs_pos[torch.from_numpy(m).to(device)][None])
                           for s0, e0, m in windows)
            for (s0, e0, m), r in zip(windows, results):
                mt = torch.from_numpy(m).to(device)
                if use_alignability_head:
                    a_d[mt] += r["alignability-dual"][0, :, 0]
                    a_j[mt] += r["alignability-joint"][0, 2, :, 0]
                else:
                    a_d[mt] += r["dual-sim"][0, -1].max(-1).values
                    a_j[mt] += r["sim"][0, -1].max(-1).values
                tcnt[mt] += 1
                acc_j[mt, s0:e0] += r["sim"][0, -1]
                acc_d[mt, s0:e0] += r["dual-sim"][0, -1]
                cnt[mt, s0:e0] += 1
            eps = torch.tensor(1e-5, device=device)
            acc_j, acc_d = acc_j / torch.maximum(cnt, eps), acc_d / torch.maximum(cnt, eps)
            a_j = a_j / torch.maximum(tcnt, eps)
            sim = (acc_j + acc_d) / 2
        elif method == "global":
            r = get_text_visual_sim(video, text, interpolate_from=seq_len)
            sim = r["sim"][0, -1].clone()
            a_j = r["alignability-joint"][0, -1, :, 0] if use_alignability_head else r["sim"][0, -1].max(-1).values
        else:
            raise ValueError(method)
        sim = sim.masked_fill(sim == 0, -6e4)
        prob = sim.softmax(-1)
        score = a_j if use_alignability_head else sim.max(-1)[0]
        scores.append(score.cpu().numpy().copy())
        tgts.append(aligned.astype(np.int64))
        am = prob[torch.from_numpy(aligned).to(device)].argmax(-1).cpu()
        for k, (s, e) in enumerate(zip(start[aligned], end[aligned])):
            recall.append(math.floor(s) <= int(am[k]) <= math.ceil(e))
        per_video.append({"sim": sim.cpu(), "argmax": am, "score": score.cpu()})
    metric = {"Recall": float(np.mean(recall)), "AUC": roc_auc_score(np.concatenate(tgts), np.concatenate(scores))}
    return (metric, per_video) if return_per_video else metric
