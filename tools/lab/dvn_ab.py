"""A/B of the one-pass d-logits + d_vn kernel against the element-wise pass + GEMM (numerics and stand-alone time)."""
import sys, torch
from temporalalignnet_amd import _lib, loss as L

def run(S, B, T, N, shared, compact, fused, g_seed=0, reps=1, time_it=False):
    g = torch.Generator(device="cpu").manual_seed(4321 + S + B)
    R, Mp, Cw = B * T, B * N, 512
    vn = torch.nn.functional.normalize(torch.randn(S, R, Cw, generator=g), dim=-1).cuda().bfloat16()
    tn = torch.nn.functional.normalize(torch.randn(1 if shared else S, Mp, Cw, generator=g), dim=-1).cuda().bfloat16()
    tgt = (torch.rand(B, T, N, generator=g) < 0.15).float().cuda()
    tpad = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        tpad[b, max(1, N - (b % N)):] = True
    col_invalid = tpad.view(-1).to(torch.uint8).cuda()
    prep = L.compaction_prep(col_invalid, int((~tpad).sum())) if compact else None
    L._FUSED_DVN = fused
    gv = gt = None
    outs = []
    for _ in range(reps):
        ctx = L._ManualCtx()
        v_terms, t_terms = L._FusedNCEFn.forward(ctx, vn, tn, tgt, col_invalid, None, B, T, N, prep)
        if gv is None:
            gv = torch.randn(v_terms.shape, generator=g).cuda(); gt = torch.randn(t_terms.shape, generator=g).cuda()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d_vn, d_tn = L._FusedNCEFn.backward(ctx, gv, gt)[:2]
        e1.record(); torch.cuda.synchronize()
        outs.append((d_vn.float(), d_tn.float(), e0.elapsed_time(e1)))
    return outs

if __name__ == "__main__":
    cfg = [int(x) for x in sys.argv[1:7]] if len(sys.argv) > 6 else [2, 40, 64, 10, 0, 1]
    S, B, T, N, shared, compact = cfg
    a = run(S, B, T, N, bool(shared), bool(compact), False, reps=4)
    b = run(S, B, T, N, bool(shared), bool(compact), True, reps=4)
    print("baseline self-consistent dt:", torch.equal(a[0][1], a[1][1]), " dv:", torch.equal(a[0][0], a[1][0]))
    print("fused    self-consistent dt:", torch.equal(b[0][1], b[1][1]), " dv:", torch.equal(b[0][0], b[1][0]))
    dv0, dt0 = b[0][:2]; dv1, dt1 = a[0][:2]
    print("dt equal:", torch.equal(dt0, dt1), "max|d|", (dt0 - dt1).abs().max().item(), "rel", ((dt0 - dt1).norm() / dt1.norm()).item())
    print("dv rel:", ((dv0 - dv1).norm() / dv1.norm()).item(), "max", (dv0 - dv1).abs().max().item(), dv1.abs().max().item())
    bad = (dt0 != dt1).nonzero()
    print("dt mismatches:", bad.shape[0], bad[:8].tolist())
    print("backward ms baseline:", [round(x[2], 3) for x in a], " fused:", [round(x[2], 3) for x in b])
