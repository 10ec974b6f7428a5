"""dW-shaped GEMM (both operands K-strided, f32 split-K output) with padded leading dimensions: does the power-of-two row
pitch of dY / X (4 KiB / 1 KiB) cost anything at the L2 / fabric?  Tool only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import ops
R = int(os.environ.get("ROWS", 8192)); reps = int(os.environ.get("REPS", 50))
for (N, K) in [(2048, 512), (512, 2048), (1536, 512), (512, 512)]:
    for pa, pb in ((0, 0), (64, 0), (0, 64), (64, 64), (8, 8)):
        dy = torch.randn(R, N + pa, device="cuda").bfloat16(); x = torch.randn(R, K + pb, device="cuda").bfloat16()
        split = max(1, min(16, 256 // ((N // 128) * (K // 128))))
        parts = torch.zeros(split, N, K, device="cuda")
        def fn():
            # dW[N,K] partials = dY^T X, one f32 plane per K-slice (batch = split, no atomics)
            ops.gemm(dy, x, parts, M=N, N=K, K=R // split, a_kc=False, b_kc=False, lda=N + pa, ldb=K + pb, batch=split,
                     sA=(R // split) * (N + pa), sB=(R // split) * (K + pb), sC=N * K)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"N={N:4d} K={K:4d} split={split:2d} pad=({pa:2d},{pb:2d}): {us:6.1f} us {2.0*R*N*K/us/1e6:6.0f} TF/s", flush=True)
