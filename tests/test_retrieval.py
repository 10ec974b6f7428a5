"""Row f4 -- YouCook2 zero-shot retrieval harness (eval/eval_zeroshot_retrieval.py:13-27,82-148,157-256) against golden G12, which
is the output of the reference's OWN test_retrieval_yc2 / YouCook2_Feature methods on the synthetic fixture
(tests/golden/make_goldens.py:g12_retrieval).  CPU: the oracle restatement and the host-side pieces of the product;
GPU: the product harness with the HIP model."""
import numpy as np
import pytest
import torch

from temporalalignnet_amd import synth
from temporalalignnet_amd.eval_retrieval import clip_windows, compute_metrics

KEYS = ("R1", "R5", "R10", "MR", "C-R1", "C-R5", "C-R10", "C-MR", "S-R1", "S-R5", "S-R10", "S-MR")


def _clips():
    fx = synth.yc2_fixture()
    feats = {vid: synth.yc2_features(vid, vlen) for vid, vlen in fx["videos"].items()}
    return [{"feature": feats[c["vid"]], "start": c["segment"][0], "end": c["segment"][1], "str": c["sentence"]} for c in fx["clips"]]


def test_compute_metrics_with_ties_matches_reference(golden):
    from oracle import retrieval_ref
    g = golden("g12_compute_metrics")
    got, orc = compute_metrics(g["x"]), retrieval_ref.metrics(g["x"])
    for k in ("R1", "R5", "R10", "MR"):
        assert float(got[k]) == float(g[k]) == orc[k], k


def test_windows_match_the_reference_dataset_class(golden):
    from oracle import retrieval_ref
    g = golden("g12_retrieval_windows")
    for i, c in enumerate(_clips()):
        idx, s_idx, e_idx = clip_windows(c["feature"].shape[0], c["start"], c["end"], 10, -1)
        assert (s_idx == g[f"{i}/start_idx"]).all() and (e_idx == g[f"{i}/end_idx"]).all(), i
        video = torch.from_numpy(c["feature"])[torch.as_tensor(idx)]
        np.testing.assert_allclose(video.double().sum(-1).numpy(), g[f"{i}/video_checksum"], rtol=1e-12)
        for w, (frames, (lo, hi)) in enumerate(retrieval_ref.windows(c["feature"].shape[0], c["start"], c["end"])):
            assert frames == idx[w].tolist() and (lo, hi) == (int(s_idx[w]), int(e_idx[w])), (i, w)


def test_oracle_retrieval_matches_reference_golden(golden):
    from oracle import retrieval_ref
    g = golden("g12_retrieval")
    p = {k: torch.from_numpy(v) for k, v in synth.make_params(113, 2, 1, False).items()}
    out, sim = retrieval_ref.retrieval(p, _clips(), lambda s: torch.from_numpy(synth.yc2_text_embedding(s)), E=2, seq_len=64)
    np.testing.assert_allclose(sim, g["sim"], rtol=1e-4, atol=2e-6)
    for k in KEYS:
        assert out[k] == pytest.approx(float(g[k]), abs=1e-12), k


@pytest.mark.gpu
def test_hip_retrieval_matches_reference_golden(golden):
    from temporalalignnet_amd.eval_retrieval import test_retrieval
    from temporalalignnet_amd.tan_model import TemporalAligner
    g = golden("g12_retrieval")
    m = TemporalAligner(num_encoder_layers=2, num_decoder_layers=1, use_alignability_head=0, language_model=None, random_pos_start=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(113, 2, 1, False).items()})
    m.cuda()
    embed = lambda strs: torch.stack([torch.from_numpy(synth.yc2_text_embedding(s)) for s in strs])
    metrics, sim = test_retrieval(_clips(), m.get_visual_feature, m.get_textual_feature, embed, seq_len=64, return_sim=True)
    np.testing.assert_allclose(sim, g["sim"], rtol=1e-3, atol=2e-5)
    for k in KEYS:
        assert float(metrics[k]) == pytest.approx(float(g[k]), abs=1e-12), k
