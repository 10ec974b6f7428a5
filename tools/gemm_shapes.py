"""Runs every Linear GEMM shape of one encoder layer (rows = B*T) fwd / dX / dW through libtan_hip, `reps` times each, in a
fixed order -- for rocprofv3 --pmc passes (per-shape HBM traffic) and timing.  Tool only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import ops, _lib
import ctypes as C
R = int(os.environ.get("ROWS", 8192)); reps = int(os.environ.get("REPS", 5))
L = _lib.lib()
def ev():
    return torch.cuda.Event(enable_timing=True)
for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    x = torch.randn(R, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    y = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(R, N, device="cuda").bfloat16(); dx = torch.empty(R, K, device="cuda", dtype=torch.bfloat16)
    gw = torch.zeros(N, K, device="cuda")
    ws = torch.empty(32 * N * K, device="cuda")
    fl = 2.0 * R * N * K
    def dw():
        rc = L.tan_linear_wgrad(C.c_void_p(dy.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(gw.data_ptr()),
                                    C.c_long(R), N, K, C.c_void_p(ws.data_ptr()), C.c_long(ws.numel()), ops._dt(dy), ops._stream())
        assert rc == 0
    fns = {"fwd": lambda: ops.gemm(x, w, y, M=R, N=N, K=K),
           "dX": lambda: ops.gemm(dy, w, dx, M=R, N=K, K=N, a_kc=True, b_kc=False, ldb=K),
           "dW": dw}
    for name, fn in fns.items():
        fn(); torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name:3s} N={N:4d} K={K:4d}: {us:6.1f} us {fl/us/1e6:6.0f} TF/s", flush=True)
