"""Stand-alone time of the similarity statistics sweep (tan_simnce_fwd / tan_simnce_fwd_keep) at the headline shape: what keeping the
exponentials costs the forward.  Run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration."""
import sys, torch
from temporalalignnet_amd import _lib, loss as L
S, B, T, N, shared = 6, 128, 64, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator(device="cpu").manual_seed(1)
R, Mp, Cw = B * T, B * N, 512
vn = torch.nn.functional.normalize(torch.randn(S, R, Cw, generator=g), dim=-1).cuda().bfloat16()
tn = torch.nn.functional.normalize(torch.randn(1 if shared else S, Mp, Cw, generator=g), dim=-1).cuda().bfloat16()
tgt = (torch.rand(B, T, N, generator=g) < 0.15).float().cuda()
tpad = torch.zeros(B, N, dtype=torch.bool)
for b in range(B):
    tpad[b, max(1, 4 + (b * 7) % 13):] = True
col_invalid = tpad.view(-1).to(torch.uint8).cuda()
prep = L.compaction_prep(col_invalid, int((~tpad).sum()))


class Ctx(L._ManualCtx):
    pass


for keep in (True, False):
    ctx = Ctx()
    ctx.needs_input_grad = (keep, keep)
    ts = []
    for it in range(8):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L._FusedNCEFn.forward(ctx, vn, tn, tgt, col_invalid, None, B, T, N, prep)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("keep " if keep else "stats", "forward ms:", " ".join(f"{t:.3f}" for t in ts[2:]))
