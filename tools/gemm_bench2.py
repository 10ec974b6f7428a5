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
for (M, N, K, split) in [(2048, 512, 8192, 8), (1536, 512, 8192, 8), (512, 512, 8192, 16), (512, 2048, 8192, 8)]:
    dY = torch.randn(K, M, device="cuda").bfloat16(); X = torch.randn(K, N, device="cuda").bfloat16()
    acc = torch.zeros(M, N, device="cuda")
    part = torch.empty(split, M, N, device="cuda")
    kc = K // split
    us_a = t(lambda: ops.gemm(dY, X, acc, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M, ldb=N, accumulate=True, split_k=split))
    us_p = t(lambda: ops.gemm(dY, X, part, M=M, N=N, K=kc, a_kc=False, b_kc=False, lda=M, ldb=N, batch=split, sA=kc * M, sB=kc * N, sC=M * N))
    us_r = t(lambda: part.sum(0))
    print(f"dW {M}x{N} K={K} split={split}: atomics {us_a:6.1f} us ({2*M*N*K/us_a/1e6:6.1f} TF) | partials {us_p:6.1f} us + torch reduce {us_r:5.1f} us")
