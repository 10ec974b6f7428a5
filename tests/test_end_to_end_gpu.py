"""End to end on the GPU: features on disk -> HTMFeatureDataset -> DataLoader workers -> pinned prefetch -> Word2Vec embedder
-> TemporalAligner / TwinTemporalAligner step (both stages) -> checkpoint -> resume -> HTM-Align style evaluation."""
import numpy as np
import pytest
import torch

from temporalalignnet_amd import checkpoint as ck
from temporalalignnet_amd import data_htm, synth
from temporalalignnet_amd.word2vec_model import Word2VecModel, Word2VecTokenizer

pytestmark = pytest.mark.gpu


def _with_lm(model, vocab, twin):
    V = len(vocab) + 1
    if twin:
        model.online.bert, model.target.bert = Word2VecModel(num_embeddings=V, compute_dtype="bf16"), Word2VecModel(num_embeddings=V, compute_dtype="bf16")
        model.bert = model.online.bert
        model._copy_param()                     # target <- online, target frozen (what the constructor does, tan_model.py:322)
    else:
        model.bert = Word2VecModel(num_embeddings=V, compute_dtype="bf16")
    return model.cuda()


def test_two_stage_training_from_disk_with_checkpoints(tmp_path):
    from temporalalignnet_amd.eval_align import make_sim_fn, test_alignment_htm
    from temporalalignnet_amd.train import Trainer, build_model, default_args, embed_sentences
    fx = synth.htm_fixture()
    paths = synth.write_htm_fixture(str(tmp_path / "htm"), fx)
    vocab = synth.w2v_vocab(40)
    tok = Word2VecTokenizer(max_words=32, vocab=vocab)
    ds = data_htm.HTMFeatureDataset(paths["features"], paths["asr"], paths["vlen"], paths["holdout"], tokenizer=tok, mode="train")
    loader = data_htm.make_loader(ds, batch_size=3, num_workers=2, shuffle=True)
    # ---- stage 1
    a1 = default_args(model="init", num_encoder_layers=3, num_decoder_layers=3)
    torch.manual_seed(0)
    m1 = _with_lm(build_model(a1, compute_dtype="bf16", language_model=None), vocab, twin=False)
    t1 = Trainer(m1, a1, iter_per_epoch=2, warmup=1)
    l1 = []
    for epoch in range(3):
        np.random.seed(epoch)
        for b in data_htm.DevicePrefetcher(loader, device="cuda"):
            l1.append(float(t1.step(b)["loss"].detach()))
    assert len(l1) == 6 and all(np.isfinite(l1))
    p1 = str(tmp_path / "ckpt" / "epoch2.pth.tar")
    ck.save_checkpoint(ck.make_state(t1, epoch=2, best_acc=min(l1)), is_best=1, filename=p1)
    # ---- stage 2 from the stage-1 checkpoint
    a2 = default_args(model="cotrain", num_encoder_layers=3, num_decoder_layers=3, loss_threshold=0.5)
    m2 = _with_lm(build_model(a2, compute_dtype="bf16", language_model=None), vocab, twin=True)
    missing, unexpected = ck.load_pretrain(m2, p1)
    assert unexpected == [] and all("binary_head" in k for k in missing)
    t2 = Trainer(m2, a2, iter_per_epoch=2, warmup=1)
    l2 = []
    for epoch in range(2):
        np.random.seed(10 + epoch)
        for b in data_htm.DevicePrefetcher(loader, device="cuda"):
            out = t2.step(b)
            l2.append(float(out["loss"].detach()))
            assert {"loss-joint-bce", "confidence-ratio", "loss-dual-all"} <= set(out)
    assert all(np.isfinite(l2))
    # the EMA stream follows the online stream, language model included
    d = max((po.detach() - pt.detach()).abs().max().item() for po, pt in zip(m2.online.parameters(), m2.target.parameters()))
    assert 0 < d < 0.05
    p2 = str(tmp_path / "ckpt" / "epoch3.pth.tar")
    ck.save_checkpoint(ck.make_state(t2, epoch=3, best_acc=1.0), filename=p2, keep_all=True)
    m3 = _with_lm(build_model(a2, compute_dtype="bf16", language_model=None), vocab, twin=True)
    t3 = Trainer(m3, a2, iter_per_epoch=2, warmup=1)
    info = ck.load_for_resume(t3, p2)
    assert info["missing"] == [] and info["unexpected"] == [] and t3.iteration == t2.iteration
    for (n, a), (_, b) in zip(m2.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a, b), n
    # ---- evaluation harness on HTM-Align shaped videos with the trained model
    vids = synth.align_videos(n_videos=2)

    def embed_text(sentences):
        t = tok(sentences, return_tensors="pt")
        return m3.lang_model(t["input_ids"].cuda(), t["attention_mask"].cuda())["pooler_output"].float()
    metrics = test_alignment_htm(make_sim_fn(m3, embed_text), vids, device="cuda")
    assert 0.0 <= metrics["Recall"] <= 1.0 and 0.0 <= metrics["AUC"] <= 1.0
