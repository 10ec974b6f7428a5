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
