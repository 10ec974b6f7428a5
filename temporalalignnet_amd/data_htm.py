"""HTM feature data path: the reference's data/loader_htm.py `HTM_FeatureLoader`
(window sampling + `collate_fn` + `pad_sequence_by_last`) and the batch preparation of train/main.py:48-79, feeding the
MI355X train step.

On-disk formats (data/readme.md:22-33, loader_htm.py:136-143,173-176) -- nothing here invents a format:
  features   `{feature_dir}/{vid}.mp4.npy` (fallback `{vid}.webm.npy`): [vlen, 1024] float32/float16, one S3D feature per second
  ASR        `sentencified_htm_370k.json`: {vid: {"text": [str], "start": [float s], "end": [float s]}}
  lengths    `htm_vlen.csv`: rows `vid,vlen` (no header)
  hold-out   `htm_holdout_vid.txt`: one vid per line

What is different from the reference, and why:
  * paths are constructor arguments (the reference hard-codes a cluster path, loader_htm.py:70, and reads the json/csv from
    its own directory);
  * the per-sample pandas DataFrame (loader_htm.py:176-177) is replaced by list arithmetic with the same results --
    INCLUDING the reference's label-vs-position quirk: rows with `end >= vlen` are filtered out but keep their original index
    labels, the window start is drawn from the LABELS (`cap_df.index[...]`, :191-192) and then used as a POSITION
    (`cap_df.iloc[start_idx]`, :193,204).  Captions are time-ordered, so only trailing rows are ever dropped and the two agree;
  * `DevicePrefetcher` stages batches through pinned host memory and a side HIP stream so that the next batch's PCIe copy
    overlaps the current step (the reference: 8 DataLoader workers + a BackgroundGenerator thread, utils/data_utils.py:9-44,
    then blocking `.to(device)` inside the step, main.py:47-53).  At B=1024/step the features are 268 MB fp32 per step.

The numpy global RNG is consumed exactly like the reference (`np.random.choice` once per sample that has a valid window,
loader_htm.py:191), so under the same `np.random.seed` the same windows are drawn (pinned by tests/golden/g9_htm_loader.npz).
"""
from __future__ import annotations

import json
import os
import queue
import threading

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

from .loss import get_mask_from_time, get_text_pos
from .word2vec_model import Word2VecTokenizer


def pad_sequence_by_last(sequences):
    """Stack ragged [n_b, ...] tensors to [B, max n_b, ...], padding each with copies of its LAST row (loader_htm.py:13-23)."""
    n_max = max(s.shape[0] for s in sequences)
    out = sequences[0].new_zeros((len(sequences), n_max) + tuple(sequences[0].shape[1:]))
    for i, s in enumerate(sequences):
        out[i, :s.shape[0]] = s
        out[i, s.shape[0]:] = s[-1]
    return out


def pad_sequence_to_size(sequences, size, batch_first=True, padding_value=0):
    """pad_sequence, but at least `size` long (loader_htm.py:26-38)."""
    dummy = torch.zeros([size] + list(sequences[0].shape[1:]), device=sequences[0].device)
    padded = pad_sequence([dummy] + list(sequences), batch_first=batch_first, padding_value=padding_value)
    return padded[1:] if batch_first else padded[:, 1:]


def read_vlen_csv(path):
    out = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                vid, n = line.rsplit(",", 1)
                out[vid] = int(float(n))
    return out


class HTMFeatureDataset(torch.utils.data.Dataset):
    """`HTM_FeatureLoader` (loader_htm.py:62-258).  `tokenizer=None` gives the reference's dummy `{'input_ids': [0]}`."""

    def __init__(self, feature_dir, asr_json, vlen_csv, holdout_txt=None, tokenizer=None, mode="train", duration=64,
                 text_tag="htm-370k"):
        self.video_feature_path = feature_dir
        self.text_tag, self.mode, self.duration = text_tag, mode, duration
        self.tokenizer = tokenizer if tokenizer else (lambda x, **kw: {"input_ids": [0]})
        with open(asr_json) as f:
            self.vid_to_asr_dict = json.load(f)
        holdout = set()
        if holdout_txt:
            with open(holdout_txt) as f:
                holdout = {line.strip() for line in f}
        vlen = read_vlen_csv(vlen_csv)
        # loader_htm.py:91-108: drop the hold-out set, keep 64 < vlen < 1000 (as MIL-NCE), sort; the first 5 % (<= 1000) is val
        vids = [v for v in self.vid_to_asr_dict if v not in holdout]
        vids = sorted(v for v in vids if v in vlen and 64 < vlen[v] < 1000)
        num_val = min(int(len(vids) * 0.05), 1000)
        if mode == "train":
            self.video_info = vids[num_val:]
        elif mode in ("val", "test"):
            self.video_info = vids[:num_val]
        else:
            raise ValueError(mode)

    def __len__(self):
        return len(self.video_info)

    # ---- loader_htm.py:130-170
    def __getitem__(self, index):
        vid = self.video_info[index]
        path = os.path.join(self.video_feature_path, f"{vid}.mp4.npy")
        if not os.path.exists(path):
            path = os.path.join(self.video_feature_path, f"{vid}.webm.npy")
        feature = torch.from_numpy(np.load(path))
        vlen = feature.shape[0]
        caps, (start_ts, end_ts) = self._get_text(vid, vlen)
        video = feature[start_ts:end_ts].float()
        if isinstance(self.tokenizer, Word2VecTokenizer):
            token = torch.stack(caps["token"], 0)
        else:
            token = pad_sequence_to_size(caps["token"], size=32, batch_first=True, padding_value=0)
        out = {"video": video, "padding_mask": torch.zeros(video.shape[0]).long(), "vid": vid, "text": caps["text"],
               "start": caps["start"], "end": caps["end"], "token": token,
               "abs_text_start": (np.array(caps["start"]).astype(np.float32) + start_ts) / vlen,
               "abs_text_end": (np.array(caps["end"]).astype(np.float32) + start_ts) / vlen}
        if self.mode in ("val", "test"):
            out.update(cut_start=start_ts, cut_end=end_ts)
        return out

    # ---- loader_htm.py:173-244
    def _get_text(self, vid, vlen):
        d = self.vid_to_asr_dict[vid]
        texts, starts, ends = d["text"], d["start"], d["end"]
        keep = [i for i in range(len(ends)) if ends[i] < vlen]           # surviving rows, by ORIGINAL label
        dur = self.duration
        no_caption = not keep
        start_idx = start_ts = end_ts = None
        if not no_caption:
            last = ends[keep[-1]]
            if sum(1 for i in keep if starts[i] < last - dur - 1) == 0:
                no_caption = True
            else:
                labels = np.array([i for i in keep if starts[i] < last - dur])
                start_idx = int(np.random.choice(labels))                # a LABEL, used below as a POSITION (see module doc)
                start_ts = int(round(starts[keep[start_idx]]))
                end_ts = start_ts + dur
        sentences, tokens, out_s, out_e = [], [], [], []
        if not no_caption:
            w2v = isinstance(self.tokenizer, Word2VecTokenizer)
            for pos in range(start_idx, len(keep)):
                i = keep[pos]
                s, e = round(starts[i]), round(ends[i])
                text = str(texts[i]).replace("\n", " ").strip()
                words = text.split()
                if len(words) > 256:
                    text = " ".join(words[:256])
                if s > end_ts or e - s < 1:
                    break
                if e > end_ts:
                    e = end_ts
                token = self.tokenizer(text, max_length=32, truncation=True)["input_ids"]
                trim_s, trim_e = max(s - start_ts, 0), min(e - start_ts, dur)
                if trim_e == trim_s:
                    break
                if w2v and sum(token) == 0:                               # every word is out of vocabulary
                    break
                sentences.append(text); tokens.append(torch.tensor(token)); out_s.append(trim_s); out_e.append(trim_e)
        if not sentences or no_caption:                                   # unlucky sampling (loader_htm.py:232-241)
            text = "[UNK]"
            tokens.append(torch.tensor(self.tokenizer(text)["input_ids"]))
            sentences.append(text); out_s.append(0); out_e.append(dur)
            if no_caption:
                start_ts, end_ts = 0, dur
        return {"text": sentences, "start": out_s, "end": out_e, "token": tokens}, (start_ts, end_ts)

    # ---- loader_htm.py:111-128
    @staticmethod
    def collate_fn(batch):
        out = {"video": pad_sequence_by_last([b["video"] for b in batch]),
               "padding_mask": pad_sequence([b["padding_mask"] for b in batch], batch_first=True, padding_value=1.0)}
        for k in ("text", "start", "end", "vid", "token"):
            out[k] = [b[k] for b in batch]
        for k in ("cut_start", "cut_end", "abs_text_start", "abs_text_end"):
            if k in batch[0]:
                out[k] = [b[k] for b in batch]
        return out


def _seed_worker(worker_id):
    # every DataLoader worker gets its own numpy stream derived from the epoch seed the driver set (main.py:507)
    np.random.seed((torch.initial_seed() + worker_id) % (1 << 32))


def make_loader(dataset, batch_size, num_workers=8, shuffle=True, drop_last=True, sampler=None):
    """torch DataLoader over an HTMFeatureDataset with the reference's collate; pass a DistributedSampler for N>1 ranks
    (end2end/main_nce.py:230 idiom -- each rank draws its own B_local videos)."""
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                       num_workers=num_workers, collate_fn=dataset.collate_fn, drop_last=drop_last,
                                       worker_init_fn=_seed_worker if num_workers else None,
                                       persistent_workers=num_workers > 0, pin_memory=False)


def prepare_batch(collated: dict, device="cuda", pin=True) -> dict:
    """train/main.py:47-79 up to the language model: video / padding mask / token ids to the device (through pinned memory,
    asynchronously on the CURRENT stream), the [B,N,T] timestamp mask and the absolute text positions.  The sentence
    embeddings themselves are produced inside the step from `token` (Trainer.forward_backward -> embed_sentences)."""
    def h2d(t):
        if pin and t.device.type == "cpu" and torch.cuda.is_available():
            t = t.pin_memory()
        return t.to(device, non_blocking=True)

    out = {k: collated[k] for k in ("text", "start", "end", "vid") if k in collated}
    out["video"] = h2d(collated["video"])
    out["padding_mask"] = h2d(collated["padding_mask"]).bool()
    n_per = [int(t.shape[0]) for t in collated["token"]]
    flat = h2d(torch.cat(list(collated["token"]), 0))
    out["token"] = list(torch.split(flat, n_per, dim=0))
    T, N = out["video"].shape[1], max(n_per)
    out["_tgt_raw"], _, _ = get_mask_from_time(collated["start"], collated["end"], T, N, device=device)
    if "abs_text_start" in collated:
        out["abs_text_pos"] = get_text_pos([np.asarray(a) for a in collated["abs_text_start"]],
                                           [np.asarray(a) for a in collated["abs_text_end"]], device=device)
    out["n_text"] = int(sum(n_per))
    return out


class DevicePrefetcher:
    """Iterates a loader of collated batches and yields device-resident, step-ready batches `depth` ahead: a background
    thread pulls from the loader (so worker hand-off and collation never sit on the training thread) and issues the H2D
    copies on a private HIP stream; the consumer's stream waits on the copy event only."""

    def __init__(self, loader, device="cuda", depth=2):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.loader, self.device, self.depth = loader, dev, depth

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        q: queue.Queue = queue.Queue(maxsize=self.depth)
        use_cuda = self.device.type == "cuda"
        stream = torch.cuda.Stream(self.device) if use_cuda else None
        stop = threading.Event()

        def producer():
            try:
                if use_cuda:
                    torch.cuda.set_device(self.device)
                for collated in self.loader:
                    if stop.is_set():
                        return
                    if use_cuda:
                        with torch.cuda.stream(stream):
                            b = prepare_batch(collated, self.device)
                            ev = torch.cuda.Event()
                            ev.record(stream)
                    else:
                        b, ev = prepare_batch(collated, self.device, pin=False), None
                    q.put((b, ev))
                q.put(None)
            except BaseException as e:                 # surface loader errors on the training thread
                q.put(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                b, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    for v in b.values():                # the tensors were allocated on the copy stream
                        for t in (v if isinstance(v, list) else [v]):
                            if torch.is_tensor(t) and t.is_cuda:
                                t.record_stream(torch.cuda.current_stream(self.device))
                yield b
        finally:
            stop.set()
            while th.is_alive():                        # unblock a producer stuck on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)
