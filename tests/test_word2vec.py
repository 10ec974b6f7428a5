"""Row f1 -- sentence embedder (model/word2vec_model.py): oracle and tokenizer vs reference-generated golden G8 on CPU,
HIP Word2VecModel vs golden / oracle autograd on the GPU."""
import numpy as np
import pytest
import torch

from oracle import w2v_ref
from temporalalignnet_amd import synth

V = 500


def _params():
    return {k: torch.from_numpy(v) for k, v in synth.w2v_params(31, V).items()}


def _fp(g):
    g = g.detach().double().flatten().cpu()
    idx = torch.linspace(0, g.numel() - 1, 16).long()
    return np.concatenate([[g.sum().item(), g.norm().item()], g[idx].numpy()])


def test_oracle_matches_reference_golden(golden):
    g = golden("g8_word2vec")
    p = {k: v.requires_grad_(k != "word_embd.weight") for k, v in _params().items()}
    ids, mask = synth.w2v_tokens(32, 9, V)
    out = w2v_ref.forward(p, torch.from_numpy(ids), torch.from_numpy(mask))
    np.testing.assert_allclose(out["pooler_output"].detach().numpy(), g["pooler_output"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["last_hidden_state"].detach().numpy(), g["last_hidden_state"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(w2v_ref.forward(p, torch.from_numpy(ids))["pooler_output"].detach().numpy(), g["pooler_nomask"],
                               rtol=1e-5, atol=1e-6)
    w = torch.from_numpy(synth.normal(33, "w", (9, 512)))
    (out["pooler_output"] * w).sum().backward()
    for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        np.testing.assert_allclose(_fp(p[k].grad), g["grad/" + k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_tokenizers_match_reference_golden(golden):
    from temporalalignnet_amd.word2vec_model import Word2VecTokenizer
    g = golden("g8_word2vec")
    vocab, sents = synth.w2v_vocab(40), synth.w2v_sentences()
    tok = Word2VecTokenizer(max_words=8, vocab=vocab)
    t = tok(sents, return_tensors="pt")
    assert (t["input_ids"].numpy() == g["tok_ids"]).all() and (t["attention_mask"].numpy() == g["tok_mask"]).all()
    ids, mask = w2v_ref.tokenize(sents, {w: i + 1 for i, w in enumerate(vocab)}, 8)
    assert (ids.numpy() == g["tok_ids"]).all() and (mask.numpy() == g["tok_mask"]).all()
    assert tok("Stir the eggs")["input_ids"][:3] == [42, 43, 44]
    assert tok.tokenize("Don't stop") == ["don't", "stop"]
    with pytest.raises(FileNotFoundError):
        Word2VecTokenizer()                      # MIL-NCE dictionary not shipped: same failure mode as the reference


def _hip_model(dtype):
    from temporalalignnet_amd.word2vec_model import Word2VecModel
    m = Word2VecModel(num_embeddings=V, compute_dtype=dtype)
    m.load_state_dict(_params())
    return m.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 4e-2)])
def test_hip_word2vec_forward_backward(golden, dtype, tol):
    g = golden("g8_word2vec")
    m = _hip_model(dtype)
    ids, mask = synth.w2v_tokens(32, 9, V)
    out = m(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda())
    assert np.abs(out["pooler_output"].detach().cpu().numpy() - g["pooler_output"]).max() < tol * 4
    assert np.abs(out["last_hidden_state"].cpu().numpy() - g["last_hidden_state"]).max() < tol * 4
    assert np.abs(m(torch.from_numpy(ids).cuda())["pooler_output"].detach().cpu().numpy() - g["pooler_nomask"]).max() < tol * 4
    w = torch.from_numpy(synth.normal(33, "w", (9, 512))).cuda()
    (out["pooler_output"] * w).sum().backward()
    named = dict(m.named_parameters())
    assert named["word_embd.weight"].grad is None
    for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        got, want = _fp(named[k].grad), g["grad/" + k]
        scale = abs(want[1])
        assert abs(got[1] - want[1]) < 50 * tol * scale, (k, got[1], want[1])
        assert np.abs(got[2:] - want[2:]).max() < 50 * tol * scale / 10 + 1e-6, k


@pytest.mark.gpu
def test_aligner_with_language_model_trains_end_to_end():
    """train/main.py:58-65 path: tokens -> lang_model -> pooler_output -> pad_sequence_by_last -> aligner; LM gets gradients."""
    from temporalalignnet_amd.tan_model import TemporalAligner
    from temporalalignnet_amd.word2vec_model import Word2VecModel
    torch.manual_seed(0)
    m = TemporalAligner(1, 1, language_model=None)
    m.bert = Word2VecModel(num_embeddings=V)
    m.cuda()
    ids, mask = synth.w2v_tokens(34, 10, V)
    te = m.lang_model(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda())["pooler_output"]
    text = te.view(2, 5, 512)
    b = synth.make_batch(35, B=2, T=16, fixed_n=5)
    out = m(torch.from_numpy(b["video"]).cuda(), text, torch.zeros(2, 16, dtype=torch.bool, device="cuda"),
            torch.zeros(2, 5, dtype=torch.bool, device="cuda"), None)
    out["logits_joint"].square().sum().backward()
    assert m.bert.fc1.weight.grad is not None and m.bert.fc1.weight.grad.abs().sum().item() > 0
    assert m.bert.fc2.bias.grad.abs().sum().item() > 0


@pytest.mark.gpu
def test_trainer_updates_language_model_from_tokens():
    """train() with token inputs (main.py:55-65,112-122): fc1/fc2 of the sentence embedder are optimised with the aligner and
    tracked by the EMA twin; the frozen word table is not."""
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    from temporalalignnet_amd.word2vec_model import Word2VecModel
    torch.manual_seed(1)
    args = default_args(model="cotrain", num_encoder_layers=1, num_decoder_layers=3, lr=1e-3, loss_threshold=0.5)
    model = build_model(args)
    model.online.bert, model.target.bert = Word2VecModel(num_embeddings=V), Word2VecModel(num_embeddings=V)
    model.bert = model.online.bert
    model._copy_param()
    model.cuda()
    b_np = synth.make_batch(41, B=6, T=16, n_min=2, n_max=5)
    b = to_device_batch(b_np)
    ids, _ = synth.w2v_tokens(42, int(b_np["n_per"].sum()), V)
    b["token"] = [t.cuda() for t in torch.split(torch.from_numpy(ids), [int(n) for n in b_np["n_per"]])]
    before = {n: p.detach().clone() for n, p in model.online.bert.named_parameters()}
    tgt_before = model.target.bert.fc1.weight.detach().clone()
    tr = Trainer(model, args)
    losses = [tr.step(b)["loss"].item() for _ in range(3)]
    assert all(np.isfinite(losses))
    after = dict(model.online.bert.named_parameters())
    assert torch.equal(after["word_embd.weight"], before["word_embd.weight"])
    for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        assert (after[n] - before[n]).abs().max().item() > 1e-5, n
    assert (model.target.bert.fc1.weight - tgt_before).abs().max().item() > 0        # EMA follows
    assert (model.target.bert.fc1.weight - after["fc1.weight"]).abs().max().item() > 0


@pytest.mark.gpu
def test_token_step_on_two_chains_matches_the_autograd_step(monkeypatch):
    """Stage 1 starting from token ids (train/main.py:55-65) in bf16: the two-chain step hands the sentence embeddings' gradient -- the
    embeddings' backward produces it -- back to the language model under autograd; against forward -> get_loss -> loss.backward()
    (TAN_STEP_CHAINS=0): loss, every aligner gradient and the gradients of fc1 / fc2 of the sentence embedder."""
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    from temporalalignnet_amd.word2vec_model import Word2VecModel
    args = default_args(model="init", num_encoder_layers=2, num_decoder_layers=2, lr=1e-3)
    b_np = synth.make_batch(51, B=8, T=64, n_min=4, n_max=16)
    ids, _ = synth.w2v_tokens(52, int(b_np["n_per"].sum()), V)
    outs = {}
    for tag, env in (("autograd", "0"), ("chains", "1")):
        monkeypatch.setenv("TAN_STEP_CHAINS", env)
        torch.manual_seed(3)
        model = build_model(args, compute_dtype="bf16", random_pos_start=0)
        model.bert = Word2VecModel(num_embeddings=V)
        model.cuda()
        b = to_device_batch(b_np)
        b["token"] = [t.cuda() for t in torch.split(torch.from_numpy(ids), [int(n) for n in b_np["n_per"]])]
        tr = Trainer(model, args)
        eb = tr._embed_tokens(b)
        assert eb["text_embed"].requires_grad and tr._chains_eligible(tr._pad_sentence_slots(eb), tr.fused_loss) == (env == "1")
        tr.zero_grad()
        ld = tr.forward_backward(b)
        torch.cuda.synchronize()
        outs[tag] = (float(ld["loss"]), tr.online.flat_grad().clone(),
                     {n: p.grad.detach().clone() for n, p in model.bert.named_parameters() if p.grad is not None})
        losses = [float(tr.step(b)["loss"]) for _ in range(3)]           # ... and the whole step trains (pipelined boundary included)
        assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    (l0, g0, lm0), (l1, g1, lm1) = outs["autograd"], outs["chains"]
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0))
    assert (g1 - g0).norm() <= 2e-3 * g0.norm()
    assert set(lm0) == set(lm1) and {"fc1.weight", "fc2.weight"} <= set(lm1)
    for n in lm0:
        assert (lm1[n] - lm0[n]).norm() <= 2e-2 * lm0[n].norm() + 1e-7, (n, float((lm1[n] - lm0[n]).norm()), float(lm0[n].norm()))
