"""tan_adamw_step_images (csrc/tan_optim.hip): AdamW that writes the bf16 weight images itself, against the launches it replaces --
tan_adamw_step, then tan_transpose_batch and 2 x tan_pack_weights (train/main.py:112-122's optimizer.step + _momentum_update on the
images the kernels read).  Parameters, moments, EMA twin and every image must come out BIT-identical."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stage", [1, 2])
def test_adamw_images_is_bit_identical_to_adamw_plus_rebuilds(stage):
    from temporalalignnet_amd import synth
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    torch.manual_seed(0)
    args = default_args(model="init" if stage == 1 else "cotrain", num_encoder_layers=3, num_decoder_layers=3, lr=1e-3, wd=1e-2,
                        loss_threshold=0.0 if stage == 1 else 0.5)
    model = build_model(args, compute_dtype="bf16").cuda()
    if stage == 2:
        model._copy_param()
    tr = Trainer(model, args)
    batch = to_device_batch(synth.make_batch(3, B=8, T=32, n_min=3, n_max=8))
    os.environ["TAN_OPT_IMAGES"] = "0"
    try:
        tr.step(batch)                                    # real gradients, non-zero moments
        tr.step(batch)
        torch.cuda.synchronize()
        f, st = tr._ensure_state()
        ema = model.target._ensure_flat() if stage == 2 else None
        snap = {k: t.clone() for k, t in (("p", f.flat), ("m", st["m"]), ("v", st["v"]))}
        if ema is not None:
            snap["e"] = ema.flat.clone()
        it = tr.iteration

        def run(images):
            os.environ["TAN_OPT_IMAGES"] = "1" if images else "0"
            f.flat.copy_(snap["p"]); st["m"].copy_(snap["m"]); st["v"].copy_(snap["v"])
            if ema is not None:
                ema.flat.copy_(snap["e"])
            tr.iteration = it
            tr.optimizer_step(grad_scale=0.5)
            if not images:                               # the lazily rebuilt images of the old path
                f.sync_shadow_p(); f.sync_shadow_t(); f.sync_shadow_tp()
                if ema is not None:
                    ema.sync_shadow_p()
            torch.cuda.synchronize()
            out = {"p": f.flat, "m": st["m"], "v": st["v"], "shadow": f.shadow, "packed": f.shadow_p, "t": f.shadow_t, "tpacked": f.shadow_tp}
            if ema is not None:
                out.update(e=ema.flat, e_shadow=ema.shadow, e_packed=ema.shadow_p)
            return {k: v.clone() for k, v in out.items()}
        want = run(False)
        got = run(True)
        _, _, n_ent, _, ranges = f.image_table()
        assert n_ent == 3 * 2 * 4 + 2
        for k in want:
            a, b = got[k], want[k]
            if a.dtype == torch.bfloat16:
                a, b = a.view(torch.int16), b.view(torch.int16)
            if k in ("t", "tpacked"):
                # (the pre-projections have no W^T image in the old path: compare the encoder matrices only)
                for lo, hi in ranges[:-2]:
                    assert torch.equal(a[lo:hi], b[lo:hi]), k
            elif k in ("packed", "e_packed"):
                for lo, hi in ranges:
                    assert torch.equal(a[lo:hi], b[lo:hi]), k
            else:
                assert torch.equal(a, b), k
        assert torch.equal(got["shadow"].view(torch.int16), got["p"].bfloat16().view(torch.int16))
        # and a training step that uses the in-optimizer images end to end stays finite and close to the old path's loss
        os.environ["TAN_OPT_IMAGES"] = "1"
        l1 = tr.step(batch)["loss"].item()
        l2 = tr.step(batch)["loss"].item()
        assert l1 == l1 and l2 == l2 and l2 < l1 + 0.5
    finally:
        os.environ.pop("TAN_OPT_IMAGES", None)
