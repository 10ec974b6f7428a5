"""GPU time of the segments of one train step (events on the main stream): zero_grad | forward | loss | backward | optimizer."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import synth
from temporalalignnet_amd.loss import get_loss
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
import sys
STAGE2 = "--stage2" in sys.argv
args = default_args(model="cotrain" if STAGE2 else "init", loss_threshold=0.5 if STAGE2 else 0.0)
model = build_model(args, compute_dtype="bf16").cuda()
if not STAGE2:
    model.random_pos_start = 1
else:
    model._copy_param()
tr = Trainer(model, args, iter_per_epoch=2890, warmup=1000); tr.iteration = 1000
b = to_device_batch(synth.make_batch(888, B=128, T=64, n_min=4, n_max=16))
for _ in range(5): tr.step(b)
names = ["zero_grad", "forward", "ema_forward", "loss", "backward", "optimizer"]
acc = [0.0] * 6; n = 20
for _ in range(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    torch.cuda.synchronize()
    ev[0].record(); tr.zero_grad()
    ev[1].record()
    logits = model(b["video"], b["text_embed"], video_padding_mask=b["padding_mask"], lang_padding_mask=b["text_padding_mask"].bool(),
                   text_timestamp=b.get("_tgt_raw"), abs_text_pos=b.get("abs_text_pos"), fused=True)
    logits["_fused"].n_text_valid = b["n_text"]
    ev[2].record()
    if STAGE2:
        ema = model.forward_from_ema(b["video"], b["text_embed"], video_padding_mask=b["padding_mask"], lang_padding_mask=b["text_padding_mask"].bool(),
                                     text_timestamp=b.get("_tgt_raw"), abs_text_pos=b.get("abs_text_pos"), fused=True)
        logits = {**logits, **{f"ema-{k}": v for k, v in ema.items()}}
    ev[3].record()
    ld = get_loss(b, b["video"], b["text_embed"], b["padding_mask"], b["text_padding_mask"], logits, args, b.get("abs_text_pos"))
    ev[4].record()
    ld["loss"].backward()
    ev[5].record()
    tr.optimizer_step()
    ev[6].record(); torch.cuda.synchronize()
    for i in range(6): acc[i] += ev[i].elapsed_time(ev[i + 1])
print("  ".join(f"{k} {v / n:.2f} ms" for k, v in zip(names, acc)), " total", round(sum(acc) / n, 2))
