"""Yardstick: libtan_hip GEMM vs torch.matmul (hipBLASLt/rocBLAS) on the train step's shapes.  Tool only (not product)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import ops
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
R = 8192
for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    x = torch.randn(R, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    y = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(R, N, device="cuda").bfloat16(); dx = torch.empty(R, K, device="cuda", dtype=torch.bfloat16)
    gw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * R * N * K
    a = t(lambda: ops.gemm(x, w, y, M=R, N=N, K=K)); b = t(lambda: torch.matmul(x, w.t(), out=y))
    print(f"fwd  [{R}x{K}]x[{N}x{K}]^T: tan {a:6.1f} us {fl/a/1e6:6.0f} TF | torch {b:6.1f} us {fl/b/1e6:6.0f} TF")
    a = t(lambda: ops.gemm(dy, w, dx, M=R, N=K, K=N, a_kc=True, b_kc=False, ldb=K)); b = t(lambda: torch.matmul(dy, w, out=dx))
    print(f"dX   [{R}x{N}]x[{N}x{K}]  : tan {a:6.1f} us {fl/a/1e6:6.0f} TF | torch {b:6.1f} us {fl/b/1e6:6.0f} TF")
    gwb = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    a = t(lambda: ops.gemm(dy, x, gw, M=N, N=K, K=R, a_kc=False, b_kc=False, lda=N, ldb=K, accumulate=True, split_k=4))
    b = t(lambda: torch.matmul(dy.t(), x, out=gwb))
    print(f"dW   [{R}x{N}]^T x[{R}x{K}]: tan(split4 atomics) {a:6.1f} us {fl/a/1e6:6.0f} TF | torch {b:6.1f} us {fl/b/1e6:6.0f} TF")
