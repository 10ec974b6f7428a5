"""CPU restatement of eval/eval_zeroshot_retrieval.py (TEST INFRASTRUCTURE -- only tests/ may import this).

Written from the algorithm, independently of temporalalignnet_amd/eval_retrieval.py: recall / median rank by counting how
many videos beat the paired one (`metrics`, :13-27), the ten evaluation windows per annotated segment (`windows`, :104-131) and the
feature pooling + three similarity variants of test_retrieval_yc2 (:157-256) on top of oracle/tan_ref.py.  Pinned by golden
G12, which is the output of the reference's own test_retrieval_yc2 on the synthetic fixture (tests/golden/make_goldens.py).
"""
from __future__ import annotations

import numpy as np
import torch

from . import tan_ref


def metrics(sim):
    """:13-27.  rank of the paired video in each text row = number of videos scoring strictly higher; the reference marks EVERY
    position of the sorted row that equals the paired score, so a tie at the pair's score contributes all tied positions."""
    sim = np.asarray(sim, dtype=np.float64) if np.asarray(sim).dtype == np.float64 else np.asarray(sim)
    hits = []
    for i in range(sim.shape[0]):
        row, d = sim[i], sim[i, i]
        higher = int((row > d).sum())
        ties = int((row == d).sum())
        hits.extend(range(higher, higher + ties))
    hits = np.asarray(hits)
    return {"R1": float((hits == 0).mean()), "R5": float((hits < 5).mean()), "R10": float((hits < 10).mean()),
            "MR": float(np.median(hits) + 1)}


def windows(vlen, start, end, num_clips=10):
    """seq_len == -1 branch of _get_video_feature (:104-131)."""
    dur = int(np.floor(end - start))
    win = min(max(2 * dur, 32), 256)
    slack = abs(win - dur)
    offs = [int(np.floor(0.25 * slack + (0.5 * slack) * k / (num_clips - 1))) for k in range(num_clips)]
    out = []
    for o in offs:
        first = start - o if win >= dur else start + o
        frames = [min(max(first + j, 0), vlen - 1) for j in range(win)]
        out.append((frames, (o, o + dur) if win >= dur else (0, win)))
    return out


@torch.no_grad()
def retrieval(p, clips, text_embed, E, seq_len=64):
    """:157-256 with the oracle model (tan_ref.visual_feature / textual_feature)."""
    V, T = [], []
    for item in clips:
        feat = torch.as_tensor(item["feature"])
        per_window = []
        for frames, (lo, hi) in windows(feat.shape[0], item["start"], item["end"]):
            video = feat[torch.as_tensor(frames)][None]
            Tw = video.shape[1]
            f = tan_ref.visual_feature(video, torch.zeros(1, Tw, dtype=torch.bool), p, E, 0, seq_len if Tw >= seq_len else None)
            f = f[0, -1, lo:hi]
            per_window.append(f / f.norm(dim=-1, keepdim=True))
        v = torch.stack(per_window, 0).mean(0).mean(0)
        t = tan_ref.textual_feature(text_embed(item["str"])[None, None], p)[0, 0]
        V.append((v / v.norm()).numpy())
        T.append((t / t.norm()).numpy())
    V, T = np.stack(V), np.stack(T)
    sim = T @ V.T
    out = dict(metrics(sim))
    Vc, Tc = V - V.mean(0), T - T.mean(0)
    for tag, m in (("C", metrics(Tc @ Vc.T)), ("S", metrics((Tc / Tc.std(0)) @ (Vc / Vc.std(0)).T))):
        out.update({f"{tag}-{k}": v for k, v in m.items()})
    return out, sim
