"""Row f4: checkpoint I/O in the reference's formats (train/main.py:407-484,511-523; utils/utils.py:38-57)."""
import os

import numpy as np
import pytest
import torch

from temporalalignnet_amd import checkpoint as ck
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
from temporalalignnet_amd.word2vec_model import Word2VecModel


def _init_model(E=1, D=1, V=50, seed=0):
    torch.manual_seed(seed)
    m = build_model(default_args(model="init", num_encoder_layers=E, num_decoder_layers=D), language_model=None)
    m.bert = Word2VecModel(num_embeddings=V)
    return m


def test_save_checkpoint_rotation(tmp_path):
    d = str(tmp_path / "ckpt")
    for ep in range(8):
        ck.save_checkpoint({"epoch": ep, "x": torch.ones(1)}, is_best=1, gap=1, filename=os.path.join(d, f"epoch{ep}.pth.tar"))
    files = sorted(os.listdir(d))
    assert [f for f in files if f.startswith("epoch")] == ["epoch7.pth.tar"]             # previous epochs removed
    best = sorted(f for f in files if f.startswith("model_best"))
    assert best == [f"model_best_epoch{e}.pth.tar" for e in range(3, 8)]                 # 5 newest bests kept
    ck.save_checkpoint({"epoch": 8}, filename=os.path.join(d, "epoch8.pth.tar"), keep_all=True)
    assert {"epoch7.pth.tar", "epoch8.pth.tar"} <= set(os.listdir(d))


def test_load_released_style_checkpoint(tmp_path):
    """`lang_model.` spelling (main.py:467-469) and a DataParallel `module.` prefix both load strictly."""
    src = _init_model(seed=1)
    sd = {"module." + k.replace("bert.", "lang_model."): v.clone() for k, v in src.state_dict().items()}
    path = str(tmp_path / "epoch3.pth.tar")
    torch.save({"epoch": 3, "state_dict": sd, "best_acc": 1.0, "optimizer": {}, "iteration": 7}, path)
    dst = _init_model(seed=2)
    epoch, missing, unexpected = ck.load_for_test(dst, path)
    assert (epoch, missing, unexpected) == (3, [], [])
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # a checkpoint with a foreign key falls back to the reported non-strict load, like the reference
    sd["extra.weight"] = torch.zeros(1)
    torch.save({"epoch": 4, "state_dict": sd}, path)
    _, missing, unexpected = ck.load_for_test(dst, path)
    assert missing == [] and unexpected == ["extra.weight"]


def test_pretrain_into_twin(tmp_path):
    """main.py:458-484: stage-1 tensors go to both streams, then _copy_param freezes the target."""
    src = _init_model(E=3, D=3, seed=3)
    path = str(tmp_path / "init_epoch9.pth.tar")
    torch.save({"epoch": 9, "state_dict": {k.replace("bert.", "lang_model."): v for k, v in src.state_dict().items()}}, path)
    torch.manual_seed(4)
    tw = build_model(default_args(model="cotrain", num_encoder_layers=3, num_decoder_layers=3), language_model=None)
    tw.online.bert, tw.target.bert = Word2VecModel(num_embeddings=50), Word2VecModel(num_embeddings=50)
    tw.bert = tw.online.bert
    missing, unexpected = ck.load_pretrain(tw, path)
    assert unexpected == [] and all("binary_head" in k for k in missing)        # the head is new in stage 2
    ref = src.state_dict()
    for k, v in tw.state_dict().items():
        base = k.split(".", 1)[1] if k.startswith(("online.", "target.")) else k
        if "binary_head" not in k:
            assert torch.equal(v, ref[base]), k
    assert not any(p.requires_grad for p in tw.target.parameters())
    assert ck.expand_for_cotrain({"lang_model.fc1.bias": 1, "x": 2}).keys() == {"target.lang_model.fc1.bias", "target.x",
                                                                                "online.lang_model.fc1.bias", "online.x",
                                                                                "lang_model.fc1.bias"}


@pytest.mark.gpu
def test_resume_is_exact_and_optimizer_state_is_torch_compatible(tmp_path):
    args = default_args(model="init", num_encoder_layers=2, num_decoder_layers=2)
    batch = to_device_batch(synth.make_batch(5, B=4, T=16, n_min=2, n_max=5))

    def fresh(seed):
        torch.manual_seed(seed)
        m = build_model(args, compute_dtype="fp32").cuda()
        m.random_pos_start = 0
        t = Trainer(m, args, iter_per_epoch=100, warmup=2)
        return m, t

    m1, t1 = fresh(0)
    for _ in range(3):
        t1.step(batch)
    path = str(tmp_path / "epoch0.pth.tar")
    ck.save_checkpoint(ck.make_state(t1, epoch=0, best_acc=2.5), filename=path)
    t1._resume_bump = 1        # what the reference's resume does to the first batch (main.py:499): one schedule position ahead
    l4 = float(t1.step(batch)["loss"].detach())
    p4 = t1.online.flat_parameters().clone()

    m2, t2 = fresh(1)                                   # different init: everything must come from the file
    info = ck.load_for_resume(t2, path)
    assert info["start_epoch"] == 1 and info["best_acc"] == 2.5 and info["missing"] == [] and t2.iteration == 3
    assert t2.batches_seen == 3 and torch.load(path, weights_only=False)["iteration"] == 4     # args.iteration = 1 + batches
    l4b = float(t2.step(batch)["loss"].detach())
    assert abs(l4b - l4) <= 1e-6 * abs(l4)              # f32 atomics in the gradient reductions: not bitwise reproducible
    torch.testing.assert_close(t2.online.flat_parameters(), p4, rtol=1e-5, atol=5e-6)   # Adam normalises near-zero gradients: noise of a few % of one lr-sized update

    # the saved optimizer entry is a valid torch.optim.AdamW state for the reference's parameter groups (main.py:329-356)
    state = torch.load(path, weights_only=False)
    nd, wd = ck._groups(m2, "default")
    opt = torch.optim.AdamW([{"params": [p for _, p in nd], "weight_decay": 0.0}, {"params": [p for _, p in wd], "weight_decay": args.wd}],
                            lr=args.lr)
    opt.load_state_dict(state["optimizer"])
    names = [n for n, _ in nd] + [n for n, _ in wd]
    with_state = {names[i] for i in state["optimizer"]["state"]}
    assert "mlp.weight" not in with_state and "text_temporal_pos_embed" not in with_state      # never receive gradients
    assert "video_pre_proj.weight" in with_state and "ln_video_init.bias" in with_state
    assert all(float(s["step"]) == 3.0 for s in state["optimizer"]["state"].values())
    # torch's own AdamW, continued from that state on the same gradient, lands on the same parameters as our fused kernel
    m3, t3 = fresh(2)
    ck.load_for_resume(t3, path)
    t3.zero_grad()
    t3.forward_backward(batch)
    nd3, wd3 = ck._groups(m3, "default")
    clones = [p.detach().clone().requires_grad_(True) for _, p in nd3 + wd3]
    for c, (_, p) in zip(clones, nd3 + wd3):
        c.grad = None if p.grad is None else p.grad.detach().clone()
    opt3 = torch.optim.AdamW([{"params": clones[:len(nd3)], "weight_decay": 0.0}, {"params": clones[len(nd3):], "weight_decay": args.wd}],
                             lr=t3.args.lr, betas=t3.betas, eps=t3.eps)
    opt3.load_state_dict(state["optimizer"])
    t3.iteration += 0
    lr_next = None
    t3.optimizer_step()
    lr_next = t3.current_lr()
    for g in opt3.param_groups:
        g["lr"] = lr_next
    opt3.step()
    for c, (n, p) in zip(clones, nd3 + wd3):
        torch.testing.assert_close(c.detach(), p.detach(), rtol=2e-6, atol=2e-7, msg=n)


def test_word_table_is_numbered_like_the_reference_optimizer():
    """the reference's word-embedding table is a plain nn.Embedding (requires_grad=True, s3dg.py:197) used under
    no_grad, so optim_policy (main.py:336-343) lists it in the with-decay group, stateless; this build freezes it and must
    still number it.  Group sizes for the twin E1D3 model: [28, 19] in the reference."""
    torch.manual_seed(0)
    tw = build_model(default_args(model="cotrain", num_encoder_layers=1, num_decoder_layers=3), language_model=None)
    tw.online.bert, tw.target.bert = Word2VecModel(num_embeddings=50), Word2VecModel(num_embeddings=50)
    tw.bert = tw.online.bert
    tw._copy_param()
    nd, wd = ck._groups(tw, "default")
    # the reference's rule applied to a replica whose word table is trainable, as in the reference
    replica = [(n, p.requires_grad or n.endswith("online.bert.word_embd.weight")) for n, p in tw.named_parameters()]
    ref_nd = [n for n, rg in replica if rg and any(t in n for t in ck.NO_DECAY_TOKENS)]
    ref_wd = [n for n, rg in replica if rg and not any(t in n for t in ck.NO_DECAY_TOKENS)]
    assert [n for n, _ in nd] == ref_nd and [n for n, _ in wd] == ref_wd
    names_wd = [n for n, _ in wd]
    assert "online.bert.word_embd.weight" in names_wd and not any(n.startswith("target.") for n in names_wd)
    i = names_wd.index("online.bert.word_embd.weight")
    assert names_wd[i + 1] == "online.bert.fc1.weight"          # sits between the aligner's parameters and fc1, as in the reference
    single = _init_model(E=1, D=1)
    nd1, wd1 = ck._groups(single, "default")
    assert "bert.word_embd.weight" in [n for n, _ in wd1]


@pytest.mark.gpu
def test_optimizer_state_round_trips_with_the_word2vec_language_model(tmp_path):
    """The saved AdamW state loads into a torch.optim.AdamW built by the reference's grouping rule (word table included, without
    state), and back into a fresh Trainer; resume with backprop_freq = 2 restores the batch counter and the Adam step separately."""
    args = default_args(model="init", num_encoder_layers=1, num_decoder_layers=1, backprop_freq=2)
    b_np = synth.make_batch(5, B=4, T=16, n_min=2, n_max=5)
    batch = to_device_batch(b_np)
    ids, _ = synth.w2v_tokens(6, int(b_np["n_per"].sum()), 50)
    batch["token"] = [t.cuda() for t in torch.split(torch.from_numpy(ids), [int(n) for n in b_np["n_per"]])]

    def fresh(seed):
        torch.manual_seed(seed)
        m = build_model(args, compute_dtype="fp32", language_model=None)
        m.bert = Word2VecModel(num_embeddings=50)
        m = m.cuda()
        m.random_pos_start = 0
        return m, Trainer(m, args, iter_per_epoch=100, warmup=2)

    m1, t1 = fresh(0)
    for idx in range(5):                   # batches 0, 2, 4 step the optimizer (idx % 2 == 0, main.py:113)
        t1.train_iteration(batch, idx)
    assert (t1.batches_seen, t1.iteration) == (5, 3)
    path = str(tmp_path / "epoch0.pth.tar")
    ck.save_checkpoint(ck.make_state(t1, epoch=0, best_acc=1.0), filename=path)
    state = torch.load(path, weights_only=False)
    assert state["iteration"] == 6
    nd, wd = ck._groups(m1, "default")
    assert [len(nd), len(wd)] == [len(g["params"]) for g in state["optimizer"]["param_groups"]]
    clones = [torch.nn.Parameter(p.detach().clone()) for _, p in nd + wd]
    opt = torch.optim.AdamW([{"params": clones[:len(nd)], "weight_decay": 0.0}, {"params": clones[len(nd):], "weight_decay": args.wd}],
                            lr=args.lr)
    opt.load_state_dict(state["optimizer"])                    # torch accepts it: same group sizes and numbering
    names = [n for n, _ in nd + wd]
    k = names.index("bert.word_embd.weight")
    assert k not in state["optimizer"]["state"] and names.index("bert.fc1.weight") in state["optimizer"]["state"]
    assert all(float(s["step"]) == 3.0 for s in state["optimizer"]["state"].values())
    m2, t2 = fresh(1)
    ck.load_for_resume(t2, path)
    assert (t2.batches_seen, t2.iteration, t2._resume_bump) == (5, 3, 1)
    lm1, lm2 = t1._state["lm"], t2._state["lm"]
    for n in lm1:
        assert torch.equal(lm1[n][0], lm2[n][0]) and torch.equal(lm1[n][1], lm2[n][1])
    from temporalalignnet_amd.train import lr_multiplier
    assert t2.current_lr() == pytest.approx(args.lr * lr_multiplier(6, 100, args.epochs, 2))   # lambda(saved iteration), main.py:499
    t2.train_iteration(batch, 5)
    assert (t2.batches_seen, t2._resume_bump) == (6, 0)
