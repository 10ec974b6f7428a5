"""Grouped weight gradient of one block (tan_linear_wgrad_group): correctness against torch f32 and time per launch, cold-ish
(fresh operands each repetition out of a ring) -- run with TAN_DW256 = 0 / n / -1 to compare the tile variants.  Tool only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import torch
from temporalalignnet_amd import ops, _lib
R = int(os.environ.get("ROWS", 8192)); reps = int(os.environ.get("REPS", 20)); RING = 4
L = _lib.lib()
shapes = [(1536, 512), (512, 512), (2048, 512), (512, 2048)]          # (N out, K in): in_proj, out_proj, c_fc, c_proj
torch.manual_seed(0)
sets = []
for _ in range(RING):
    dys = [torch.randn(R, n, device="cuda").bfloat16() * 0.5 for n, k in shapes]
    xs = [torch.randn(R, k, device="cuda").bfloat16() for n, k in shapes]
    sets.append((dys, xs))
gws = [torch.zeros(n, k, device="cuda") for n, k in shapes]
ws = torch.empty(8 * sum(n * k for n, k in shapes), device="cuda")
arr_p = C.c_void_p * 4; arr_i = C.c_int * 4
def call(dys, xs):
    rc = L.tan_linear_wgrad_group(4, arr_p(*[d.data_ptr() for d in dys]), arr_p(*[x.data_ptr() for x in xs]),
                                  arr_p(*[g.data_ptr() for g in gws]), arr_i(*[n for n, k in shapes]), arr_i(*[k for n, k in shapes]),
                                  C.c_long(R), C.c_void_p(ws.data_ptr()), C.c_long(ws.numel()), _lib.TAN_BF16, ops._stream())
    assert rc == 0, rc
dys, xs = sets[0]
call(dys, xs); call(dys, xs)
torch.cuda.synchronize()
for g, d, x in zip(gws, dys, xs):
    want = 2 * (d.float().t() @ x.float())
    err = (g - want).abs().max().item() / want.abs().max().item()
    print("rel err", f"{err:.2e}", "OK" if err < 2e-5 else "BAD")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): call(*sets[0])
torch.cuda.synchronize()
e0.record()
for i in range(reps): call(*sets[i % RING])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
fl = sum(2.0 * R * n * k for n, k in shapes)
print(f"TAN_DW256={os.environ.get('TAN_DW256', '0')} rows={R}: {us:7.1f} us per group  {fl / us / 1e6:6.0f} TF/s")
