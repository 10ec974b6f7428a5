"""Host-side logic of the product path that needs no GPU: synthetic data, parameter grouping, schedules, sharding,
state_dict surface and the small loss.py helpers."""
import numpy as np
import pytest
import torch

from oracle import loss_ref, train_ref
from temporalalignnet_amd import dist, synth
from temporalalignnet_amd.loss import circulant, get_mask_from_time, get_text_pos
from temporalalignnet_amd.tan_model import TemporalAligner, TwinTemporalAligner
from temporalalignnet_amd.train import default_args, lr_multiplier


def test_synth_is_a_pure_function_of_the_seed():
    a, b = synth.make_batch(5, B=3, T=16, n_min=2, n_max=5), synth.make_batch(5, B=3, T=16, n_min=2, n_max=5)
    assert np.array_equal(a["video"], b["video"]) and a["start"] == b["start"]
    c = synth.make_batch(6, B=3, T=16, n_min=2, n_max=5)
    assert not np.array_equal(a["video"], c["video"])
    # padded sentences repeat the last real one (pad_sequence_by_last, data/loader_htm.py:13-23)
    for i, n in enumerate(a["n_per"]):
        assert (a["text_embed"][i, n:] == a["text_embed"][i, n - 1]).all()
        assert a["text_padding_mask"][i, :n].sum() == 0 and a["text_padding_mask"][i, n:].all()
        assert all(0 <= s < e <= 16 for s, e in zip(a["start"][i], a["end"][i]))
    x = synth.normal(0, "x", (200000,))
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1) < 0.01


def test_state_dict_surface_matches_reference_key_set():
    m = TemporalAligner(2, 3, use_alignability_head=1, language_model=None)
    assert set(m.state_dict()) == set(synth.param_shapes(2, 3, True))
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == synth.param_shapes(2, 3, True)[k], k
    tw = TwinTemporalAligner(0.999, num_encoder_layers=1, num_decoder_layers=3, use_alignability_head=1, language_model=None)
    assert len(tw.state_dict()) == 132
    assert not any(p.requires_grad for p in tw.target.parameters())
    assert tw.target.random_pos_start == 0
    # interface-drift superset: aliases exist
    assert tw.get_text_visual_sim == tw.online.get_text_visual_sim_joint or callable(tw.get_text_visual_sim)
    assert m.lang_model is m.bert


def test_language_model_checkpoint_spelling_is_remapped():
    m = TemporalAligner(1, 1, language_model="word2vec")
    sd = m.state_dict()
    renamed = {("lang_model." + k[5:] if k.startswith("bert.") else k): v for k, v in sd.items()}
    assert any(k.startswith("lang_model.") for k in renamed)
    missing, unexpected = m.load_state_dict(renamed, strict=True)
    assert not missing and not unexpected


def test_param_modes_follow_the_reference_substring_rule():
    m = TemporalAligner(1, 1, use_alignability_head=1, language_model=None)
    f = m._flat
    for prefix in ("", "online."):
        mode = m.param_modes(prefix)
        for n in f.names:
            o, k, _ = f.off[n]
            want = 2 if n in ("mlp.weight", "mlp.bias", "text_temporal_pos_embed") else (1 if train_ref.decay_flag(prefix + n) else 0)
            assert (mode[o:o + k] == want).all(), (prefix, n)
    # the quirk: top-level ln_* decays under 'init' but not under 'cotrain'
    o, k, _ = f.off["ln_text_init.weight"]
    assert m.param_modes("")[o] == 1 and m.param_modes("online.")[o] == 0


def test_lr_schedule_warmup_then_cosine():
    assert lr_multiplier(0, 100, 10) == 0.0
    assert lr_multiplier(500, 100, 10) == 0.5
    assert lr_multiplier(1000, 200, 10) == pytest.approx(1.0)
    assert lr_multiplier(2000, 200, 10) == pytest.approx(0.0, abs=1e-12)
    assert default_args(model="cotrain").learn_agreement == 1 and default_args(model="cotrain").use_alignability_head == 1


def test_shard_range_covers_every_video_once():
    for n in (0, 1, 7, 128, 1000):
        for w in (1, 2, 3, 8):
            ranges = [dist.shard_range(n, w, r) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_loss_helpers_on_cpu():
    assert circulant(torch.tensor([0, 1, 2]), 0).tolist() == [[0, 1, 2], [2, 0, 1], [1, 2, 0]]
    b = synth.make_batch(3, B=5, T=16, n_min=2, n_max=6)
    N = b["text_embed"].shape[1]
    m, s, e = get_mask_from_time(b["start"], b["end"], 16, N, device="cpu")
    mr, sr, er = loss_ref.mask_from_time(b["start"], b["end"], 16, N)
    assert torch.equal(m, mr) and torch.equal(s, sr) and torch.equal(e, er)
    assert torch.equal(get_text_pos(b["start"], b["end"], device="cpu"), loss_ref.text_pos(b["start"], b["end"]))


def test_eval_auc_matches_sklearn():
    from sklearn import metrics
    from temporalalignnet_amd.eval_align import roc_auc_score
    rng = np.random.RandomState(1)
    y = rng.randint(0, 2, 300)
    s = np.round(rng.randn(300), 1)
    assert roc_auc_score(y, s) == pytest.approx(metrics.roc_auc_score(y, s), abs=1e-12)


def test_sine_position_table_matches_reference(golden):
    """pos_enc='sine': the fixed table of model/tfm_model.py:get_position_embedding_sine (golden G10)."""
    import numpy as np
    from temporalalignnet_amd.tan_model import get_position_embedding_sine
    g = golden("g10_sine_pos")
    t = get_position_embedding_sine(512, 1024).double()
    assert t.shape == (1024, 512)
    np.testing.assert_allclose(t[:6, :10].numpy(), g["corner"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(t[-3:, -6:].numpy(), g["tail"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(t.sum(1).numpy(), g["row_sum"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(t.sum(0).numpy(), g["col_sum"], rtol=0, atol=1e-5)


def test_ddp_gradient_buckets_tile_each_stack_on_cpu():
    """Host logic of the overlapped gradient reduction (Trainer._ddp_buckets): per-layer ranges of the flat gradient grouped
    into buckets, last layers first, disjoint and covering the stack -- no GPU involved."""
    from temporalalignnet_amd.train import Trainer, build_model, default_args
    args = default_args(model="init", num_encoder_layers=6, num_decoder_layers=6)
    model = build_model(args, compute_dtype="bf16")
    for bucket_layers in (1, 2, 4, 6, 7):
        tr = Trainer(model, args, ddp_bucket_layers=bucket_layers)
        covered = 0
        for tag, prefix in (("video", "video_temporal_encoder."), ("joint", "joint_temporal_encoder.")):
            b = tr._ddp_buckets(tag, 6)
            assert len(b) == -(-6 // bucket_layers)
            assert [x[2] for x in b] == sorted((x[2] for x in b), reverse=True) and b[-1][2] == 0
            lo_s, hi_s = tr.online.flat_range(prefix)
            spans = sorted((lo, hi) for lo, hi, _ in b)
            assert spans[0][0] == lo_s and spans[-1][1] == hi_s
            assert all(x[1] == y[0] for x, y in zip(spans, spans[1:]))
            per_layer = 3 * 512 * 512 + 3 * 512 + 512 * 512 + 512 + 2 * 4 * 512 * 512 + 4 * 512 + 512 + 4 * 512   # 12 tensors
            assert hi_s - lo_s >= 6 * per_layer
            covered += hi_s - lo_s
        assert 0.9 < covered / tr.online._flat.total < 0.95              # the two stacks: 92 % of the gradient bytes


def test_uncovered_gradient_ranges():
    from temporalalignnet_amd.train import uncovered_ranges
    assert uncovered_ranges([], 10) == [(0, 10)]
    assert uncovered_ranges([(0, 10)], 10) == []
    assert uncovered_ranges([(6, 8), (2, 4)], 10) == [(0, 2), (4, 6), (8, 10)]
    assert uncovered_ranges([(2, 4), (4, 10)], 10) == [(0, 2)]
    with pytest.raises(ValueError):
        uncovered_ranges([(2, 6), (4, 8)], 10)
