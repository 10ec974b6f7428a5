"""YouCook2 zero-shot text->video retrieval: counterpart of eval/eval_zeroshot_retrieval.py:
`compute_metrics` (:13-27), the window selection of `YouCook2_Feature._get_video_feature` (:104-148) and `test_retrieval_yc2`
(:157-256).  The model is only reached through `get_visual_feature(video, mask, interpolate_from=)` and
`get_textual_feature(lang_embed)` (tan_model.py:152,231) -- HIP path; everything else here is host-side index arithmetic on
per-clip features and a [n_clips, n_clips] similarity matrix.
"""
from __future__ import annotations

import numpy as np
import torch


def compute_metrics(x) -> dict:
    """Recall@{1,5,10} and median rank of the diagonal of a text x video similarity matrix (:13-27, after MIL-NCE's metrics.py):
    every position where the descending-sorted row equals the diagonal entry counts (ties give several hits per row)."""
    x = np.asarray(x)
    sx = np.sort(-x, axis=1)
    d = np.diag(-x)[:, np.newaxis]
    ind = np.where(sx - d == 0)[1]
    return {"R1": np.array(float(np.sum(ind == 0)) / len(ind)), "R5": np.array(float(np.sum(ind < 5)) / len(ind)),
            "R10": np.array(float(np.sum(ind < 10)) / len(ind)), "MR": np.array(np.median(ind) + 1)}


def clip_windows(vlen: int, start, end, num_clips: int = 10, seq_len: int = -1):
    """Frame indices [num_clips, window] of the windows evaluated for one annotated segment, and the segment's position inside
    every window (start_idx, end_idx), exactly as `_get_video_feature` picks them (:104-148).
    seq_len == -1 (what test_retrieval_yc2 uses): window = clip(2 * floor(end - start), 32, 256) frames; if it is at least as long
    as the segment the windows START `lead` frames before it with lead spread over [25 %, 75 %] of the slack, otherwise they
    start inside the segment with the lag spread the same way.  Indices are clipped to the video."""
    if seq_len == -1:
        duration = np.floor(end - start).astype(int)
        win = int(np.clip(duration * 2, a_min=32, a_max=256))
        if win >= duration:
            lead = np.floor(np.linspace(0.25 * (win - duration), 0.75 * (win - duration), num_clips)).astype(int)
            first, s_idx, e_idx = start - lead, lead, lead + duration
        else:
            lag = np.floor(np.linspace(0.25 * (duration - win), 0.75 * (duration - win), num_clips)).astype(int)
            first, s_idx, e_idx = start + lag, np.zeros_like(lag), np.zeros_like(lag) + win
    else:
        win = int(seq_len)
        first = np.floor(np.linspace(0, end - start - seq_len - 1, num_clips)).astype(int) + start
        s_idx = e_idx = None
    idx = np.clip(np.expand_dims(first, 1) + np.arange(win).astype(int)[None], a_min=0, a_max=vlen - 1)
    return idx, s_idx, e_idx


@torch.no_grad()
def test_retrieval(clips, get_visual_feature, get_text_feature, embed_text, *, sim: str = "cos", seq_len: int = 64,
                   num_clips: int = 10, device="cuda", return_sim: bool = False):
    """test_retrieval_yc2 (:157-256).  `clips`: iterable of {'feature' [vlen, D] float array (per-second features of the clip's
    video), 'start', 'end' (segment, seconds), 'str'}; `embed_text(list[str]) -> [n, 512]` stands for tokenizer + language model.
    Per clip: ten windows through `get_visual_feature` (position table interpolated from `seq_len` when the window is at least that
    long, :180-184), last stage, the segment's frames of every window, L2-normalised, averaged over time and windows,
    normalised again (:197-214); text through `get_text_feature`, normalised.  Metrics on text x video, then on centred and on
    standardised features (:233-256)."""
    vis, txt = [], []
    for item in clips:
        feat = torch.as_tensor(item["feature"])
        idx, s_idx, e_idx = clip_windows(feat.shape[0], item["start"], item["end"], num_clips, -1)
        video = feat[torch.as_tensor(idx)].to(device)                                   # [num_clips, window, D]
        v = get_visual_feature(video, torch.zeros(video.shape[:2], device=device, dtype=torch.bool),
                               interpolate_from=seq_len if video.shape[1] >= seq_len else None)
        if v.dim() == 4:
            v = v[:, -1]                                                                # last deep-supervision stage
        v = torch.stack([v[i, int(s_idx[i]):int(e_idx[i])] for i in range(v.shape[0])], 0).float()
        if sim == "cos":
            v = v / v.norm(dim=-1, keepdim=True)
        v = v.mean(0).mean(0, keepdim=True)
        t = get_text_feature(embed_text([item["str"]]).to(device)).float()
        if sim == "cos":
            v = v / v.norm(dim=-1, keepdim=True)
            t = t / t.norm(dim=-1, keepdim=True)
        vis.append(v.cpu())
        txt.append(t.reshape(1, -1).cpu())
    V, T = torch.cat(vis, 0).numpy(), torch.cat(txt, 0).numpy()
    s = np.dot(T, V.T)
    metrics = compute_metrics(s)
    Vc, Tc = V - V.mean(0, keepdims=True), T - T.mean(0, keepdims=True)
    mc = compute_metrics(np.dot(Tc, Vc.T))
    ms = compute_metrics(np.dot(Tc / Tc.std(0, keepdims=True), (Vc / Vc.std(0, keepdims=True)).T))
    for tag, m in (("C", mc), ("S", ms)):
        for k in ("R1", "R5", "R10", "MR"):
            metrics[f"{tag}-{k}"] = m[k]
    return (metrics, s) if return_sim else metrics
