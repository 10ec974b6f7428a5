"""Stand-alone time of the loss backward at the headline shape: element-wise d-logits pass + two GEMMs against the one-pass d-logits +
d_vn kernel + one GEMM; prints ms per call (run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations)."""
import sys, torch
from temporalalignnet_amd import loss as L
sys.path.insert(0, "tools/lab")
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
for fused in (0, 1):
    L._FUSED_DVN = fused >= 1
    ts = []
    for it in range(8):
        ctx = L._ManualCtx()
        v_terms, t_terms = L._FusedNCEFn.forward(ctx, vn, tn, tgt, col_invalid, None, B, T, N, prep)
        gv = torch.ones_like(v_terms); gt = torch.ones_like(t_terms)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L._FusedNCEFn.backward(ctx, gv, gt)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(("base ", "dvn  ")[fused], "Mc", prep[0].shape[0], "backward ms:", " ".join(f"{t:.3f}" for t in ts[2:]))
