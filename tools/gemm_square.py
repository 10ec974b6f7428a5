import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 2048, 512), (8192, 2048, 2048), (8192, 2048, 8192), (49152, 2048, 512)]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    wt = w.t().contiguous()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    a = t(lambda: ops.gemm(x, w, y, M=M, N=N, K=K)); b = t(lambda: torch.matmul(x, w.t(), out=y))
    c = t(lambda: ops.gemm(x, wt, y, M=M, N=N, K=K, a_kc=True, b_kc=False, ldb=N))
    print(f"[{M}x{K}]x[{N}x{K}]^T: tan KC/KC {a:7.1f} us {fl/a/1e6:6.0f} TF | tan KC/KS {c:7.1f} us {fl/c/1e6:6.0f} TF | torch {b:7.1f} us {fl/b/1e6:6.0f} TF", flush=True)
