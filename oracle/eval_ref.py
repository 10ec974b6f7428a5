"""CPU restatement of eval/eval_zeroshot_align.py:test_alignment_htm (TEST INFRASTRUCTURE).

Takes an iterable of per-video dicts {'video' [vlen,Dv], 'start' [K], 'end' [K], 'aligned' [K] 0/1,
'str' list[str]} and the callback `get_text_visual_sim(video[1,Tw,Dv], list[str], abs_text_pos=)`
returning {'sim' [1,S,K,Tw], 'dual-sim' [1,S,K,Tw], ('alignability-dual' [1,K,1],
'alignability-joint' [1,S,K,1])} -- the closure of train/main.py:171-189.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def roc_auc(y_true, score):
    """Rank-based ROC-AUC with average ranks for ties (what sklearn.metrics.roc_auc_score computes)."""
    y = np.asarray(y_true).astype(bool)
    s = np.asarray(score, dtype=np.float64)
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s), dtype=np.float64)
    sorted_s = s[order]
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and sorted_s[j + 1] == sorted_s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    n_pos, n_neg = y.sum(), (~y).sum()
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


@torch.no_grad()
def test_alignment(videos, get_text_visual_sim, seq_len=64, use_alignability_head=True, method="overlap-seq"):
    """eval_zeroshot_align.py:97-252."""
    recall, all_score, all_tgt, per_video = [], [], [], []
    for item in videos:
        video = torch.as_tensor(item["video"])[None]
        text = list(item["str"])
        aligned = np.asarray(item["aligned"]).astype(bool)
        start = np.asarray(item["start"], dtype=np.float64)
        end = np.asarray(item["end"], dtype=np.float64)
        K, vlen = len(text), video.shape[1]
        abs_pos = torch.stack((torch.as_tensor(item["start"]), torch.as_tensor(item["end"])), -1).div(vlen)
        if method == "overlap-seq":
            steps = np.arange(0, vlen - seq_len // 2, seq_len // 4)                 # :129
            mid = (start + end) / 2                                                  # :134
            acc_j = torch.zeros(K, vlen); acc_d = torch.zeros(K, vlen); cnt = torch.zeros(K, vlen)
            a_d = torch.zeros(K); a_j = torch.zeros(K); tcnt = torch.zeros(K)
            na_idx = np.arange(K)[~aligned]
            na_mid = mid[~aligned]
            for i, s0 in enumerate(steps):
                inside = (s0 - seq_len <= na_mid) & (na_mid <= s0 + 2 * seq_len)     # :151-153
                act = na_idx[inside]
                if len(act) == 0:
                    continue
                left, right = act.min(), act.max()
                if i <= 3:                                                           # :163-166 edge rule
                    left = 0
                elif i >= len(steps) - 4:
                    right = vlen
                m = np.zeros(K, bool)
                m[left:right + 1] = True
                if m.sum() == 0:
                    continue
                mt = torch.from_numpy(m)
                e0 = min(vlen, s0 + seq_len)
                r = get_text_visual_sim(video[:, s0:e0], [t for t, k in zip(text, m) if k],
                                        abs_text_pos=abs_pos[mt][None])
                if use_alignability_head:
                    a_d[mt] += r["alignability-dual"][0, :, 0]
                    a_j[mt] += r["alignability-joint"][0, 2, :, 0]                   # :186 stage index 2
                else:
                    a_d[mt] += r["dual-sim"][0, -1].max(-1).values
                    a_j[mt] += r["sim"][0, -1].max(-1).values
                tcnt[mt] += 1
                acc_j[mt, s0:e0] += r["sim"][0, -1]
                acc_d[mt, s0:e0] += r["dual-sim"][0, -1]
                cnt[mt, s0:e0] += 1
            eps = torch.tensor(1e-5)
            acc_j = acc_j / torch.maximum(cnt, eps)
            acc_d = acc_d / torch.maximum(cnt, eps)
            a_d = a_d / torch.maximum(tcnt, eps)
            a_j = a_j / torch.maximum(tcnt, eps)
            sim = (acc_j + acc_d) / 2
        else:                                                                        # 'global' :207-216
            r = get_text_visual_sim(video, text, interpolate_from=seq_len)
            sim = r["sim"][0, -1].clone()
            if use_alignability_head:
                a_j = r["alignability-joint"][0, -1, :, 0]
            else:
                a_j = r["sim"][0, -1].max(-1).values
        sim = sim.masked_fill(sim == 0, -6e4)                                        # :221
        prob = sim.softmax(-1)
        score = a_j if use_alignability_head else sim.max(-1)[0]
        all_score.append(score.numpy().copy())
        all_tgt.append(aligned.astype(np.int64))
        am = prob[torch.from_numpy(aligned)].argmax(-1)
        for k, (s, e) in enumerate(zip(start[aligned], end[aligned])):
            recall.append(math.floor(s) <= int(am[k]) <= math.ceil(e))              # :234-237
        per_video.append({"sim": sim, "argmax": am, "score": score})
    y = np.concatenate(all_tgt); sc = np.concatenate(all_score)
    return {"Recall": float(np.mean(recall)), "AUC": roc_auc(y, sc)}, per_video
