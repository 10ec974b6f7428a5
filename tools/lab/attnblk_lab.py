"""Lab: the fused attention-branch kernel (tan_attnblk_fwd) vs the three launches it replaces -- timing, COLD (six layers' worth of
distinct buffers cycled, as in a stack).  Tool only.  usage: python tools/lab/attnblk_lab.py [B] [L]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch

from temporalalignnet_amd import _lib, ops
from test_attnblk_gpu import pack

bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NL = 6


def make(L):
    R = B * L
    ts = []
    for i in range(NL):
        torch.manual_seed(i)
        t = {"xn1": torch.randn(R, 512, device="cuda").to(bf), "x_in": torch.randn(R, 512, device="cuda").to(bf),
             "w_in": (torch.randn(1536, 512, device="cuda") * 512 ** -0.5).to(bf),
             "w_out": (torch.randn(512, 512, device="cuda") * 512 ** -0.5).to(bf),
             "b_in": torch.randn(1536, device="cuda") * 0.1, "b_out": torch.randn(512, device="cuda") * 0.1,
             "qkv": torch.empty(R, 1536, device="cuda", dtype=bf), "o": torch.empty(R, 512, device="cuda", dtype=bf),
             "lse": torch.empty(B, 8, L, device="cuda"), "x_mid": torch.empty(R, 512, device="cuda", dtype=bf)}
        t["pw_qkv"], t["pw_out"] = pack([(t["w_in"], 384, 32), (t["w_out"], 512, 16)])
        ts.append(t)
    return ts


def fused(t, L, save=True):
    d = _lib.AttnBlkDesc()
    d.B, d.L, d.C, d.H = B, L, 512, 8
    d.xn1, d.x_in, d.key_padding_mask = t["xn1"].data_ptr(), t["x_in"].data_ptr(), None
    d.pw_qkv, d.pw_out, d.b_qkv, d.b_out = t["pw_qkv"].data_ptr(), t["pw_out"].data_ptr(), t["b_in"].data_ptr(), t["b_out"].data_ptr()
    if save:
        d.qkv, d.attn_o, d.lse = t["qkv"].data_ptr(), t["o"].data_ptr(), t["lse"].data_ptr()
    d.x_mid = t["x_mid"].data_ptr()
    _lib.check(_lib.lib().tan_attnblk_fwd(C.byref(d), ops._stream()), "tan_attnblk_fwd")


def unfused(t, L):
    R = B * L
    ops.gemm(t["xn1"], t["w_in"], t["qkv"], M=R, N=1536, K=512, bias=t["b_in"])
    ops.attn_fwd(t["qkv"], None, t["o"], t["lse"], B, L, 8)
    ops.gemm(t["o"], t["w_out"], t["x_mid"], M=R, N=512, K=512, bias=t["b_out"], residual=t["x_in"])


def timeit(fn, ts, reps=20):
    for t in ts:
        fn(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for t in ts:
            fn(t)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(ts))


for L in ([int(sys.argv[2])] if len(sys.argv) > 2 else [64, 80]):
    ts = make(L)
    fl = 2.0 * B * L * 512 * 2048 + 4.0 * B * L * L * 512
    tu = timeit(lambda t: unfused(t, L), ts)
    tf = timeit(lambda t: fused(t, L), ts)
    tn = timeit(lambda t: fused(t, L, False), ts)
    print(f"B={B} L={L}: unfused {tu:.1f} us ({fl / tu * 1e-6:.0f} TF/s) | fused {tf:.1f} us ({fl / tf * 1e-6:.0f} TF/s) | "
          f"fused, nothing saved {tn:.1f} us")


# ---- backward: tan_attnblk_bwd vs out_proj dX GEMM + tan_attn_bwd_bias
def bwd_fused(t, L):
    d = _lib.AttnBlkBwdDesc()
    d.B, d.L, d.C, d.H = B, L, 512, 8
    d.dx2, d.qkv, d.lse, d.key_padding_mask = t["dx2"].data_ptr(), t["qkv"].data_ptr(), t["lse"].data_ptr(), None
    d.pwt_out, d.dqkv, d.g_b_qkv = t["pwt_out"].data_ptr(), t["dqkv"].data_ptr(), t["g"].data_ptr()
    _lib.check(_lib.lib().tan_attnblk_bwd(C.byref(d), ops._stream()), "tan_attnblk_bwd")


def bwd_unfused(t, L):
    R = B * L
    ops.gemm(t["dx2"], t["w_out"], t["d_o"], M=R, N=512, K=512, a_kc=True, b_kc=True, ldb=512)      # (the W^T-copy form: K-contiguous)
    ops.attn_bwd(t["qkv"], None, t["o"], t["lse"], t["d_o"], t["dqkv"], B, L, 8, g_b_qkv=t["g"])


for L in ([int(sys.argv[2])] if len(sys.argv) > 2 else [64, 80]):
    ts = make(L)
    for t in ts:
        unfused(t, L)
        R = B * L
        t["dx2"] = (torch.randn(R, 512, device="cuda") * 0.05).to(bf)
        t["d_o"], t["dqkv"], t["g"] = torch.empty(R, 512, device="cuda", dtype=bf), torch.empty(R, 1536, device="cuda", dtype=bf), torch.zeros(1536, device="cuda")
        (t["pwt_out"],) = pack([(t["w_out"].T.contiguous(), 512, 16)])
    tu = timeit(lambda t: bwd_unfused(t, L), ts)
    tf = timeit(lambda t: bwd_fused(t, L), ts)
    print(f"backward B={B} L={L}: out_proj dX + attention backward {tu:.1f} us | fused {tf:.1f} us")

# ---- phase clocks of the backward kernel (workgroup 0)
for L in ([int(sys.argv[2])] if len(sys.argv) > 2 else [64, 80]):
    ts = make(L)
    t = ts[0]
    unfused(t, L)
    R = B * L
    t["dx2"] = (torch.randn(R, 512, device="cuda") * 0.05).to(bf)
    t["dqkv"], t["g"] = torch.empty(R, 1536, device="cuda", dtype=bf), torch.zeros(1536, device="cuda")
    (t["pwt_out"],) = pack([(t["w_out"].T.contiguous(), 512, 16)])
    dbg = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
    _lib.lib().tan_attnblk_lab_set_dbg(C.c_void_p(dbg.data_ptr()))
    for _ in range(3):
        bwd_fused(t, L)
    torch.cuda.synchronize()
    _lib.lib().tan_attnblk_lab_set_dbg(None)
    d = dbg.view(8, 64).cpu().numpy()
    names = ["start", "prologue", "GEMM-a loop", "d_o + qkv images"] + [f"hp{h} {n}" for h in range(4) for n in ("phase A", "barrier", "phase B", "park+colsum", "copy/refill")] + ["end"]
    for w in (0, 5, 7):
        row = d[w]
        print(f"L={L} wave {w}: " + ", ".join(f"{names[i]} {int(row[i] - row[i - 1])}" for i in range(1, len(names))) + f" | total {int(row[len(names) - 1] - row[0])}")
