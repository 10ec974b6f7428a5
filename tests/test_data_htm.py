"""Row f2: HTM feature data path -- window sampling / text trimming / collation against goldens produced by the REFERENCE's
HTM_FeatureLoader (tests/golden/make_goldens.py g9) on the synthetic on-disk fixture, plus the split logic, the prefetcher and
an end-to-end train step fed from disk."""
import numpy as np
import pytest
import torch

from temporalalignnet_amd import data_htm, synth
from temporalalignnet_amd.word2vec_model import Word2VecTokenizer


@pytest.fixture(scope="module")
def fixture_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("htm")
    fx = synth.htm_fixture()
    return fx, synth.write_htm_fixture(str(root), fx)


def _dataset(paths, mode, tok):
    ds = data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], None, tokenizer=tok, mode="train")
    ds.mode = mode
    return ds


def test_matches_reference_loader(golden, fixture_dir):
    g = golden("g9_htm_loader")
    fx, paths = fixture_dir
    tok = Word2VecTokenizer(max_words=32, vocab=synth.w2v_vocab(40))
    for mode, use_tok in (("train", True), ("val", False)):
        ds = _dataset(paths, mode, tok if use_tok else None)
        ds.video_info = sorted(v for v in fx["vlen"] if v not in ("vidG0007", "vidH0008"))
        for seed in (0, 1, 2):
            np.random.seed(seed)
            b = ds.collate_fn([ds[i] for i in range(len(ds))])
            tag = f"{mode}/s{seed}"
            np.testing.assert_array_equal(np.array([len(t) for t in b["text"]]), g[f"{tag}/n"])
            np.testing.assert_array_equal(np.concatenate([np.asarray(x, dtype=np.int64) for x in b["start"]]), g[f"{tag}/start"])
            np.testing.assert_array_equal(np.concatenate([np.asarray(x, dtype=np.int64) for x in b["end"]]), g[f"{tag}/end"])
            tokens = torch.cat([t.reshape(len(tx), -1).long() for t, tx in zip(b["token"], b["text"])], 0).numpy()
            np.testing.assert_array_equal(tokens, g[f"{tag}/token"])
            np.testing.assert_array_equal(np.array(["\x1f".join(t) for t in b["text"]]), g[f"{tag}/text"])
            np.testing.assert_array_equal(b["padding_mask"].numpy(), g[f"{tag}/padding_mask"])
            np.testing.assert_array_equal(b["video"][:, 0, :4].numpy(), g[f"{tag}/video_first"])
            np.testing.assert_array_equal(b["video"][:, -1, :4].numpy(), g[f"{tag}/video_last"])
            np.testing.assert_allclose(b["video"].double().sum((1, 2)).numpy(), g[f"{tag}/video_sum"], rtol=1e-12)
            np.testing.assert_array_equal(np.concatenate(b["abs_text_start"]), g[f"{tag}/abs_start"])
            np.testing.assert_array_equal(np.concatenate(b["abs_text_end"]), g[f"{tag}/abs_end"])
            if mode == "val":
                np.testing.assert_array_equal(np.array([b["cut_start"], b["cut_end"]]), g[f"{tag}/cut"])
    # edge cases the fixture was built to hit
    assert "[UNK]" in "".join(g["train/s0/text"])              # vidE0005: captions stop early -> fallback sample
    assert (g["train/s0/n"] >= 1).all() and g["train/s0/n"].max() > 8


def test_split_rules(fixture_dir):
    """loader_htm.py:91-108: hold-out removed, 64 < vlen < 1000, sorted, first min(5 %, 1000) videos are the val split."""
    fx, paths = fixture_dir
    tr = data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], paths["holdout"], mode="train")
    assert tr.video_info == ["vidA0001", "vidB0002", "vidC0003", "vidE0005", "vidF0006", "vidI0009", "vidJ0010"]
    assert len(data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], paths["holdout"], mode="val")) == 0
    with pytest.raises(ValueError):
        data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], mode="bogus")


def test_pad_helpers():
    a, b = torch.arange(6.).view(3, 2), torch.arange(2.).view(1, 2) + 10
    out = data_htm.pad_sequence_by_last([a, b])
    assert out.shape == (2, 3, 2) and torch.equal(out[1], torch.tensor([[10., 11.]] * 3)) and torch.equal(out[0], a)
    p = data_htm.pad_sequence_to_size([torch.ones(2, 3), torch.ones(5, 3)], size=4)
    assert p.shape == (2, 5, 3) and p[0, 2:].abs().sum() == 0
    q = data_htm.pad_sequence_to_size([torch.ones(2, 3)], size=4)
    assert q.shape == (1, 4, 3)


def test_loader_and_prefetcher_cpu(fixture_dir):
    fx, paths = fixture_dir
    tok = Word2VecTokenizer(max_words=32, vocab=synth.w2v_vocab(40))
    ds = data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], paths["holdout"], tokenizer=tok, mode="train")
    np.random.seed(3)
    loader = data_htm.make_loader(ds, batch_size=3, num_workers=0, shuffle=False, drop_last=True)
    got = list(data_htm.DevicePrefetcher(loader, device="cpu", depth=2))
    assert len(got) == 2
    for b in got:
        B, T = b["video"].shape[:2]
        N = max(t.shape[0] for t in b["token"])
        assert (B, T) == (3, 64) and b["video"].dtype == torch.float32 and b["padding_mask"].dtype == torch.bool
        assert b["_tgt_raw"].shape == (B, N, T) and b["abs_text_pos"].shape == (B, N, 2)
        for i in range(B):                      # the timestamp mask is start <= t < end of the collated lists
            for k, (s, e) in enumerate(zip(b["start"][i], b["end"][i])):
                assert b["_tgt_raw"][i, k].nonzero().flatten().tolist() == list(range(s, e))
    # an exception inside the loader thread surfaces on the consumer
    class Boom:
        def __iter__(self):
            raise RuntimeError("boom")
        def __len__(self):
            return 1
    with pytest.raises(RuntimeError, match="boom"):
        list(data_htm.DevicePrefetcher(Boom(), device="cpu"))


@pytest.mark.gpu
def test_train_steps_from_disk(fixture_dir):
    """disk -> dataset -> loader -> pinned prefetch -> Word2Vec embedder -> TemporalAligner step, all on the HIP path."""
    from temporalalignnet_amd.train import Trainer, build_model, default_args
    from temporalalignnet_amd.word2vec_model import Word2VecModel
    fx, paths = fixture_dir
    vocab = synth.w2v_vocab(40)
    tok = Word2VecTokenizer(max_words=32, vocab=vocab)
    ds = data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], paths["holdout"], tokenizer=tok, mode="train")
    args = default_args(model="init", num_encoder_layers=2, num_decoder_layers=2)
    torch.manual_seed(0)
    model = build_model(args, compute_dtype="bf16", language_model=None)
    model.bert = Word2VecModel(num_embeddings=len(vocab) + 1, compute_dtype="bf16")     # small synthetic dictionary
    model = model.cuda()
    tr = Trainer(model, args, iter_per_epoch=10, warmup=1)
    tr.iteration = 5
    np.random.seed(0)
    loader = data_htm.make_loader(ds, batch_size=3, num_workers=0, shuffle=False)
    losses = []
    for epoch in range(3):
        for b in data_htm.DevicePrefetcher(loader, device="cuda"):
            assert b["video"].is_cuda and b["token"][0].is_cuda
            losses.append(float(tr.step(b)["loss"].item()))
    assert len(losses) == 6 and all(np.isfinite(losses))


def test_loader_with_worker_processes(fixture_dir):
    """DataLoader workers (the reference trains with 8): the dataset pickles, every worker seeds its own numpy stream, and an
    epoch visits every video exactly once."""
    fx, paths = fixture_dir
    tok = Word2VecTokenizer(max_words=32, vocab=synth.w2v_vocab(40))
    ds = data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], paths["holdout"], tokenizer=tok, mode="train")
    torch.manual_seed(0)
    loader = data_htm.make_loader(ds, batch_size=2, num_workers=2, shuffle=True, drop_last=False)
    seen = []
    for epoch in range(2):
        vids = [v for b in loader for v in b["vid"]]
        assert sorted(vids) == sorted(ds.video_info)
        seen.append(vids)
    del loader
