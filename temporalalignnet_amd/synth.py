"""Deterministic synthetic HTM-shaped data and weights, independent of torch's RNG.

Counter-based generator (splitmix64 hash of (stream, index) -> uniform -> Box-Muller)
so that the golden-vector script (which imports the reference in the build container),
the CPU oracle, the HIP tests and bench.py all see bit-identical inputs/weights from a
seed alone -- only *outputs* need to be stored as fixtures.

Batch schema mirrors the reference collate function (data/loader_htm.py:112-129,159-168)
and the tensors train/main.py:48-79 derives from it:
  video          [B,T,D_v] float32      (loader_htm.py:151, windows are always full)
  padding_mask   [B,T]     bool (all False for full windows)
  start/end      list[list[int]]        window-relative seconds, end exclusive
  text_embed     [B,N,512] float32      padded by repeating the last sentence
                                        (pad_sequence_by_last, loader_htm.py:13-23)
  text_padding_mask [B,N]  float32 0/1  (main.py:62-65)
  abs_text_pos   [B,N,2]   float32      (get_text_pos, loss.py:44-52)
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _stream_key(seed: int, name: str) -> np.uint64:
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for ch in name.encode():
            h = ((h ^ np.uint64(ch)) * np.uint64(1099511628211)) & _M64
        h = h ^ _splitmix64(np.array([seed], dtype=np.uint64))[0]
    return h


def uniform(seed: int, name: str, n: int) -> np.ndarray:
    """n float64 uniforms in (0,1), a pure function of (seed, name, index)."""
    key = _stream_key(seed, name)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key) & _M64
    bits = _splitmix64(idx) >> np.uint64(11)            # 53 bits
    return (bits.astype(np.float64) + 0.5) / float(1 << 53)


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u1 = uniform(seed, name + "/u1", m)
    u2 = uniform(seed, name + "/u2", m)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return (z * std + mean).astype(np.float32).reshape(shape)


def randint(seed: int, name: str, lo: int, hi: int, n: int) -> np.ndarray:
    """n ints uniform in [lo, hi] inclusive."""
    u = uniform(seed, name, n)
    return (lo + np.floor(u * (hi - lo + 1))).astype(np.int64).clip(lo, hi)


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
BLOCK_KEYS = (
    "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
    "ln_1.weight", "ln_1.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
    "mlp.c_proj.weight", "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias",
)


def param_shapes(num_encoder_layers: int, num_decoder_layers: int, use_alignability_head: bool,
                 d_video: int = 1024, d_text: int = 512, width: int = 512) -> dict:
    """state_dict key -> shape for one TemporalAligner (tan_model.py:43-72), in the
    reference's registration order; language model (`bert.*`) excluded."""
    C = width
    shapes = {}
    def block(prefix):
        shapes[f"{prefix}.attn.in_proj_weight"] = (3 * C, C)
        shapes[f"{prefix}.attn.in_proj_bias"] = (3 * C,)
        shapes[f"{prefix}.attn.out_proj.weight"] = (C, C)
        shapes[f"{prefix}.attn.out_proj.bias"] = (C,)
        shapes[f"{prefix}.ln_1.weight"] = (C,)
        shapes[f"{prefix}.ln_1.bias"] = (C,)
        shapes[f"{prefix}.mlp.c_fc.weight"] = (4 * C, C)
        shapes[f"{prefix}.mlp.c_fc.bias"] = (4 * C,)
        shapes[f"{prefix}.mlp.c_proj.weight"] = (C, 4 * C)
        shapes[f"{prefix}.mlp.c_proj.bias"] = (C,)
        shapes[f"{prefix}.ln_2.weight"] = (C,)
        shapes[f"{prefix}.ln_2.bias"] = (C,)
    shapes["temporal_pos_embed"] = (1024, C)
    shapes["text_temporal_pos_embed"] = (1024, C)
    for i in range(num_encoder_layers):
        block(f"video_temporal_encoder.resblocks.{i}")
    for i in range(num_decoder_layers):
        block(f"joint_temporal_encoder.resblocks.{i}")
    shapes["video_pre_proj.weight"] = (C, d_video)
    shapes["text_pre_proj.weight"] = (C, d_text)
    for ln in ("ln_text_init", "ln_video_init", "ln_position_init", "ln_video_post_enc", "ln_joint_post_enc"):
        shapes[f"{ln}.weight"] = (C,)
        shapes[f"{ln}.bias"] = (C,)
    shapes["mlp.weight"] = (C, C)
    shapes["mlp.bias"] = (C,)
    if use_alignability_head:
        shapes["binary_head.weight"] = (1, C)
        shapes["binary_head.bias"] = (1,)
    return shapes


def make_params(seed: int, num_encoder_layers: int, num_decoder_layers: int,
                use_alignability_head: bool = False, randomize_affine: bool = True,
                d_video: int = 1024, d_text: int = 512, width: int = 512) -> dict:
    """Deterministic parameter set with the reference's init *scales*
    (tan_model.py:76-97) but, when randomize_affine, LayerNorm gains/biases and Linear
    biases perturbed away from 1/0 so that parity tests exercise them (defaults hide bugs).
    Returns {state_dict key: float32 ndarray}."""
    C = width
    layers = max(num_decoder_layers, 1)
    proj_std = (C ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = C ** -0.5
    fc_std = (2 * C) ** -0.5
    out = {}
    for k, shp in param_shapes(num_encoder_layers, num_decoder_layers, use_alignability_head,
                               d_video, d_text, width).items():
        if k.endswith("in_proj_weight"):
            v = normal(seed, k, shp, attn_std)
        elif k.endswith("out_proj.weight") or k.endswith("c_proj.weight"):
            v = normal(seed, k, shp, proj_std)
        elif k.endswith("c_fc.weight"):
            v = normal(seed, k, shp, fc_std)
        elif ".ln_" in k or k.startswith("ln_"):
            if k.endswith("weight"):
                v = normal(seed, k, shp, 0.1, 1.0) if randomize_affine else np.ones(shp, np.float32)
            else:
                v = normal(seed, k, shp, 0.05) if randomize_affine else np.zeros(shp, np.float32)
        elif k.endswith("bias"):
            v = normal(seed, k, shp, 0.02) if randomize_affine else np.zeros(shp, np.float32)
        else:  # pre-proj, pos-embeds, mlp.weight, binary_head.weight: N(0, 0.01) (tan_model.py:59,66,71,77-83)
            v = normal(seed, k, shp, 0.01)
        out[k] = v
    return out


# --------------------------------------------------------------------------------------
# batches
# --------------------------------------------------------------------------------------
def make_batch(seed: int, B: int, T: int, n_min: int = 4, n_max: int = 16, d_video: int = 1024,
               d_text: int = 512, video_pad_tail: int = 0, nonneg_video: bool = True,
               fixed_n: int | None = None) -> dict:
    """One synthetic HTM-shaped batch (numpy). `video_pad_tail` > 0 marks the last frames of
    odd-indexed videos as padding to exercise key_padding_mask (real HTM windows are full)."""
    if fixed_n is not None:
        n_per = np.full(B, fixed_n, dtype=np.int64)
    else:
        n_per = randint(seed, "n_per", n_min, n_max, B)
    N = int(n_per.max())
    video = normal(seed, "video", (B, T, d_video))
    if nonneg_video:
        video = np.abs(video) * 0.3          # S3D MIL-NCE features are post-ReLU pooled
    sent = normal(seed, "sent", (B, N, d_text))
    pad = np.zeros((B, N), np.float32)
    start, end = [], []
    for b in range(B):
        n = int(n_per[b])
        s = np.sort(randint(seed, f"start{b}", 0, max(T - 2, 0), n))
        d = randint(seed, f"dur{b}", 1, 8, n)
        e = np.minimum(s + d, T)
        start.append([int(x) for x in s])
        end.append([int(x) for x in e])
        sent[b, n:] = sent[b, n - 1]
        pad[b, n:] = 1.0
    vmask = np.zeros((B, T), bool)
    if video_pad_tail > 0:
        vmask[1::2, T - video_pad_tail:] = True
    vlen = randint(seed, "vlen", 4 * T, 16 * T, B).astype(np.float64)
    off = uniform(seed, "off", B) * (vlen - T)
    abs_pos = np.zeros((B, N, 2), np.float32)
    for b in range(B):
        n = int(n_per[b])
        abs_pos[b, :n, 0] = (np.array(start[b]) + off[b]) / vlen[b]
        abs_pos[b, :n, 1] = (np.array(end[b]) + off[b]) / vlen[b]
    text = [[f"s{b}_{i}" for i in range(int(n_per[b]))] for b in range(B)]
    return {
        "video": video.astype(np.float32), "padding_mask": vmask, "start": start, "end": end,
        "text": text, "text_embed": sent.astype(np.float32), "text_padding_mask": pad,
        "abs_text_pos": abs_pos, "n_per": n_per,
    }


def align_videos(seed=18, n_videos=3):
    """HTM-Align-shaped fake annotations: per video vlen in [180,260], K in [24,34] sentences in
    temporal order, ~35% alignable (htm_align/readme.md:11-20 schema: [aligned, start, end, text])."""
    vids = []
    for i in range(n_videos):
        vlen = int(randint(seed, f"vlen{i}", 180, 260, 1)[0])
        K = int(randint(seed, f"K{i}", 24, 34, 1)[0])
        mids = np.sort(uniform(seed, f"mid{i}", K) * (vlen - 10) + 5)
        dur = randint(seed, f"dur{i}", 2, 9, K)
        start = np.clip(mids - dur / 2, 0, vlen - 1)
        end = np.clip(mids + dur / 2, 1, vlen)
        aligned = (uniform(seed, f"al{i}", K) < 0.35).astype(np.int64)
        aligned[K // 2] = 0; aligned[0] = 1        # both classes present
        vids.append({
            "vid": f"synth{i}", "video": np.abs(normal(seed, f"video{i}", (vlen, 1024))) * 0.3,
            "start": np.round(start, 2).astype(np.float32), "end": np.round(end, 2).astype(np.float32),
            "aligned": aligned, "str": [f"v{i}_sentence_{k}" for k in range(K)],
            "emb": normal(seed, f"emb{i}", (K, 512)),
        })
    return vids


# --------------------------------------------------------------------------------------
# sentence embedder (row f1)
# --------------------------------------------------------------------------------------
def w2v_params(seed: int, V: int) -> dict:
    """Word2VecModel parameters (model/word2vec_model.py:76-82): word table, fc1 300->2048, fc2 2048->512."""
    return {"word_embd.weight": normal(seed, "word_embd", (V, 300), 0.3),
            "fc1.weight": normal(seed, "fc1.w", (2048, 300), 300 ** -0.5), "fc1.bias": normal(seed, "fc1.b", (2048,), 0.05),
            "fc2.weight": normal(seed, "fc2.w", (512, 2048), 2048 ** -0.5), "fc2.bias": normal(seed, "fc2.b", (512,), 0.05)}


def w2v_tokens(seed: int, M: int, V: int, W: int = 32):
    """[M, W] token ids (0 = padding / unknown) with ragged lengths, one all-zero sentence, and the attention mask."""
    ids = randint(seed, "ids", 1, V - 1, M * W).reshape(M, W)
    lens = randint(seed, "len", 1, W, M)
    for m in range(M):
        ids[m, lens[m]:] = 0
    if M > 2:
        ids[2, :] = 0                     # "all stop words" sentence (word2vec_model.py:92-93)
        ids[1, 3] = 0                     # an unknown word in the middle
    return ids.astype(np.int64), (ids != 0).astype(np.uint8)


def w2v_vocab(n: int):
    return np.array([f"w{i}" for i in range(n)] + ["don't", "stir", "the", "eggs"])


def w2v_sentences():
    return ["Stir the eggs, don't stop!", "w3 W7 unknownword w11", "", "w1 " * 12, "the THE the's eggs_w2"]


# ---------------------------------------------------------------------------------------------------------------------
# HTM-370K-shaped ON-DISK fixture (row f2: the feature loader).  Formats follow data/readme.md:22-33 and
# data/loader_htm.py:136-143,173-176: `{vid}.mp4.npy` (or `.webm.npy`) = [vlen, 1024] float32 S3D features, one per second;
# sentencified ASR = {vid: {"text": [...], "start": [...], "end": [...]}} with float second timestamps; htm_vlen.csv rows
# "vid,vlen"; htm_holdout_vid.txt one vid per line.
def htm_fixture(seed: int = 50):
    """-> dict(vlen={vid: int}, asr={vid: {...}}, holdout=[vid], webm={vid}) covering the loader's edge cases."""
    vocab = list(w2v_vocab(40))
    vlen, asr, webm = {}, {}, set()

    def sentence(k, n_words):
        ids = randint(seed, f"words{k}", 0, len(vocab) - 1, n_words)
        return " ".join(vocab[i] for i in ids)

    specs = [("vidA0001", 240), ("vidB0002", 180), ("vidC0003", 330), ("vidD0004", 150), ("vidE0005", 90),
             ("vidF0006", 400), ("vidG0007", 64), ("vidH0008", 1200), ("vidI0009", 200), ("vidJ0010", 130)]
    for vi, (vid, n) in enumerate(specs):
        vlen[vid] = n
        u = uniform(seed, f"times{vi}", 4000)
        t, k, texts, starts, ends = 2.0 + 3.0 * u[0], 1, [], [], []
        while t < n + 20:                       # some sentences run past the end of the video (filtered: end < vlen)
            dur = 0.4 + 8.0 * u[k]
            s = round(t * 2) / 2 if k % 3 == 0 else round(t, 2)       # x.5 timestamps exercise round-half-to-even
            e = s + (round(dur * 2) / 2 if k % 4 == 0 else round(dur, 2))
            n_words = 3 + int(u[k + 1] * 9)
            txt = sentence(vi * 1000 + k, n_words)
            if k % 7 == 3:
                txt = txt.replace(" ", "\n", 1)
            if vid == "vidC0003" and k == 9:
                txt = "zzz qqq unknownword"            # every word out of vocabulary: the window's text loop stops here
            if vid == "vidF0006" and k == 5:
                txt = sentence(777, 300)                # > 256 words: truncated
            texts.append(txt); starts.append(float(s)); ends.append(float(e))
            t = s + (0.5 + 6.0 * u[k + 2])
            k += 3
        if vid == "vidE0005":                   # captions stop early: no window start qualifies -> '[UNK]' sample
            texts, starts, ends = texts[:3], starts[:3], ends[:3]
        if vid == "vidI0009":
            webm.add(vid)
        asr[vid] = {"text": texts, "start": starts, "end": ends}
    return {"vlen": vlen, "asr": asr, "holdout": ["vidD0004"], "webm": webm}


def htm_features(vid: str, vlen: int, d_video: int = 1024) -> np.ndarray:
    """[vlen, d_video] float32 features of a fixture video: a pure function of the vid string."""
    return np.abs(normal(sum(vid.encode()), "feat/" + vid, (vlen, d_video), 0.3))


def write_htm_fixture(root: str, fx: dict):
    """Materialise the fixture under `root` in the reference's on-disk formats; returns the paths."""
    import json
    import os
    feat = os.path.join(root, "features")
    os.makedirs(feat, exist_ok=True)
    for vid, n in fx["vlen"].items():
        np.save(os.path.join(feat, f"{vid}.{'webm' if vid in fx['webm'] else 'mp4'}.npy"), htm_features(vid, n))
    paths = {"features": feat, "asr": os.path.join(root, "sentencified_htm_370k.json"),
             "vlen": os.path.join(root, "htm_vlen.csv"), "holdout": os.path.join(root, "htm_holdout_vid.txt")}
    with open(paths["asr"], "w") as f:
        json.dump(fx["asr"], f)
    with open(paths["vlen"], "w") as f:
        f.writelines(f"{v},{n}\n" for v, n in fx["vlen"].items())
    with open(paths["holdout"], "w") as f:
        f.writelines(v + "\n" for v in fx["holdout"])
    return paths


def yc2_fixture(seed: int = 60, d_video: int = 1024):
    """Synthetic YouCook2-shaped retrieval fixture (eval/eval_zeroshot_retrieval.py:29-148): a few videos with per-second
    features [vlen, 1024] and annotated segments {sentence, segment [start, end]} -- short segments (window longer than the
    segment, :112-115), a >256-s one (windows inside the segment, :116-119) and one touching the end of its video (index
    clipping, :125).  Returns {'videos': {vid: vlen}, 'clips': [{'vid', 'sentence', 'segment'}]}."""
    vlens = {"ycA": 180, "ycB": 320, "ycC": 700, "ycD": 96}
    segs = {"ycA": [(10, 28), (40, 95), (120, 178)], "ycB": [(5, 17), (60, 200), (250, 318)], "ycC": [(30, 340), (400, 470), (600, 699)],
            "ycD": [(0, 20), (30, 90)]}
    words = w2v_vocab(40)
    clips = []
    for vid in sorted(vlens):
        for j, (s, e) in enumerate(segs[vid]):
            idx = randint(seed, f"yc2.words.{vid}.{j}", 0, len(words) - 1, 5)
            clips.append({"vid": vid, "sentence": " ".join(words[i] for i in idx), "segment": [int(s), int(e)]})
    return {"videos": vlens, "clips": clips}


def yc2_features(vid: str, vlen: int, d_video: int = 1024, seed: int = 61) -> np.ndarray:
    """Per-second features of a fixture video: a smooth video-specific drift plus noise, so that segments are distinguishable."""
    base = normal(seed, f"yc2.base.{vid}", (1, d_video), std=0.5)
    drift = normal(seed, f"yc2.drift.{vid}", (1, d_video), std=0.5)
    t = (np.arange(vlen, dtype=np.float32) / vlen)[:, None]
    return np.abs(base + t * drift + normal(seed, f"yc2.noise.{vid}", (vlen, d_video), std=0.3)).astype(np.float32)


def yc2_text_embedding(sentence: str, seed: int = 62) -> np.ndarray:
    """Deterministic stand-in for the language model's pooler_output of one sentence: [512]."""
    return normal(seed, "yc2.text." + sentence, (512,))
