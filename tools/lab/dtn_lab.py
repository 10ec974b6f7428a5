"""Lab: the text-feature gradient of the NCE backward, d_tn[s] = dl[s]^T vn[s]  ([Mc x R] x [R x 512] per stage), as the tiled GEMM
it is today vs the 256 x 256-tile weight-gradient kernel (tan_linear_wgrad_group).  Tool only."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import _lib, ops
L = _lib.lib()
S, R, Mc, Cw = 6, 8192, int(os.environ.get("MC", 1280)), 512
bf = torch.bfloat16
dl = (torch.randn(S, R, Mc, device="cuda") * 0.01).to(bf)
vn = torch.randn(S, R, Cw, device="cuda").to(bf)


def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


d_run = torch.empty(S, Mc, Cw, dtype=bf, device="cuda")
def joint_now():
    ops.gemm(dl, vn, d_run, M=Mc, N=Cw, K=R, a_kc=False, b_kc=False, lda=Mc, ldb=Cw, batch=S, sA=R * Mc, sB=R * Cw, sC=Mc * Cw)
acc = torch.zeros(Mc, Cw, device="cuda")
def dual_now():
    acc.zero_()
    ops.gemm(dl, vn, acc, M=Mc, N=Cw, K=S * R, a_kc=False, b_kc=False, lda=Mc, ldb=Cw, accumulate=True, split_k=8)
accs = torch.zeros(S, Mc, Cw, device="cuda")
def group(items, M):
    n = len(items)
    dy = (C.c_void_p * n)(*[i[0] for i in items]); x = (C.c_void_p * n)(*[i[1] for i in items]); gw = (C.c_void_p * n)(*[i[2] for i in items])
    N = (C.c_int * n)(*[Mc] * n); K = (C.c_int * n)(*[Cw] * n)
    _lib.check(L.tan_linear_wgrad_group(n, dy, x, gw, N, K, C.c_long(M), None, C.c_long(0), _lib.TAN_BF16, ops._stream()), "wgrad_group")
def joint_dw():
    accs.zero_()
    for s0 in (0, 3):
        group([(dl[s].data_ptr(), vn[s].data_ptr(), accs[s].data_ptr()) for s in range(s0, s0 + 3)], R)
def dual_dw():
    acc.zero_()
    group([(dl.data_ptr(), vn.data_ptr(), acc.data_ptr())], S * R)
fl = 2.0 * S * R * Mc * Cw
for name, f in (("joint now", joint_now), ("joint dw256 (2 launches of 3 + fill)", joint_dw), ("dual now (fill + split-K 8)", dual_now), ("dual dw256 (+ fill)", dual_dw)):
    us = t(f)
    print(f"Mc={Mc} {name:40s} {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s")
ref = torch.einsum("srm,src->smc", dl.float(), vn.float())
joint_now(); torch.cuda.synchronize(); print("joint now err", (d_run.float() - ref).abs().max().item(), ref.abs().max().item())
joint_dw(); torch.cuda.synchronize(); print("joint dw err ", (accs - ref).abs().max().item())
dual_dw(); torch.cuda.synchronize(); print("dual dw err  ", (acc - ref.sum(0)).abs().max().item(), ref.sum(0).abs().max().item())
