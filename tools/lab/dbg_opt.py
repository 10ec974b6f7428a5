import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from temporalalignnet_amd import synth
from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
args = default_args(model="init", num_encoder_layers=3, num_decoder_layers=3, lr=1e-3, wd=1e-2)
model = build_model(args, compute_dtype="bf16").cuda()
tr = Trainer(model, args)
batch = to_device_batch(synth.make_batch(3, B=8, T=32, n_min=3, n_max=8))
os.environ["TAN_OPT_IMAGES"] = "0"
tr.step(batch); tr.step(batch); torch.cuda.synchronize()
f, st = tr._ensure_state()
snap = {k: t.clone() for k, t in (("p", f.flat), ("m", st["m"]), ("v", st["v"]))}
it = tr.iteration
def run(images):
    os.environ["TAN_OPT_IMAGES"] = "1" if images else "0"
    f.flat.copy_(snap["p"]); st["m"].copy_(snap["m"]); st["v"].copy_(snap["v"])
    tr.iteration = it
    tr.optimizer_step(grad_scale=0.5)
    torch.cuda.synchronize()
    return f.flat.clone(), st["m"].clone(), st["v"].clone()
a = run(False); b = run(True)
for nm, x, y in zip("pmv", a, b):
    d = (x != y)
    print(nm, int(d.sum()), "mismatches; max abs", float((x - y).abs().max()))
    if d.any():
        idx = d.nonzero().flatten()
        print(" first", idx[:10].tolist(), "last", idx[-3:].tolist())
        for n in f.names:
            o, k, shp = f.off[n]
            c = int(d[o:o + k].sum())
            if c: print("  ", n, shp, c, "of", k)
