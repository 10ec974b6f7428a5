"""tan_mlp80_fwd (one video per workgroup) against tan_mlp_fwd (64-row panels): cold-buffer timing at B = 128, L = 64 / 80, with and
without the side outputs.  usage: python tools/lab/mlp80_lab.py [B]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from temporalalignnet_amd import _lib, ops
from test_panel_gpu import pack

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bf = torch.bfloat16
torch.manual_seed(0)
wfc = (torch.randn(2048, 512, device="cuda") * 1024 ** -0.5).to(bf)
wpj = (torch.randn(512, 2048, device="cuda") * 0.03).to(bf)
bfc, bpj = torch.randn(2048, device="cuda") * 0.1, torch.randn(512, device="cuda") * 0.1
g2, b2 = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
pw_fc16, pw_pj16 = pack([(wfc, 128, 32), (wpj, 512, 32)])
pw_fc, pw_pj = pack([wfc, wpj])
NSET = 6          # rotate buffer sets so that every launch sees cold activations


def make(R):
    sets = []
    for _ in range(NSET):
        t = {"x": (torch.randn(R, 512, device="cuda") * 1.5).to(bf)}
        t |= {k: torch.empty(R, 512, device="cuda", dtype=bf) for k in ("xn2", "xout", "xn1")}
        t |= {k: torch.empty(R, 2048, device="cuda", dtype=bf) for k in ("hpre", "hact")}
        t |= {k: torch.empty(R, device="cuda") for k in ("m2", "r2", "m1", "r1")}
        sets.append(t)
    return sets


def launch(t, R, L, k80, save):
    d = _lib.MlpDesc()
    d.rows, d.C, d.FF = R, 512, 2048
    d.x_mid, d.ln_g, d.ln_b = t["x"].data_ptr(), g2.data_ptr(), b2.data_ptr()
    d.pw_fc, d.pw_proj = (pw_fc16.data_ptr(), pw_pj16.data_ptr()) if k80 else (pw_fc.data_ptr(), pw_pj.data_ptr())
    d.b_fc, d.b_proj, d.x_out = bfc.data_ptr(), bpj.data_ptr(), t["xout"].data_ptr()
    if save:
        d.xn2, d.mean2, d.rstd2, d.h_pre, d.h_act = (t[k].data_ptr() for k in ("xn2", "m2", "r2", "hpre", "hact"))
    d.nln_g, d.nln_b, d.xn_next, d.nmean, d.nrstd = g2.data_ptr(), b2.data_ptr(), t["xn1"].data_ptr(), t["m1"].data_ptr(), t["r1"].data_ptr()
    d.eps = 1e-5
    if k80:
        _lib.check(_lib.lib().tan_mlp80_fwd(C.byref(d), L, ops._stream()), "tan_mlp80_fwd")
    else:
        _lib.check(_lib.lib().tan_mlp_fwd(C.byref(d), ops._stream()), "tan_mlp_fwd")


def timeit(fn, sets, reps=30):
    for t in sets:
        fn(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(sets[i % NSET])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for L in (64, 80):
    R = B * L
    sets = make(R)
    fl = 2.0 * R * 512 * 2048 * 2
    for save in (True, False):
        t64 = timeit(lambda t: launch(t, R, L, False, save), sets)
        t80 = timeit(lambda t: launch(t, R, L, True, save), sets)
        print(f"B={B} L={L} save={save}: 64-row panels ({R // 64} workgroups) {t64:.1f} us ({fl / t64 * 1e-6:.0f} TF/s) | "
              f"one video per workgroup ({B}) {t80:.1f} us ({fl / t80 * 1e-6:.0f} TF/s)")
