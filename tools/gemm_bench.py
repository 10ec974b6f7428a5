#!/usr/bin/env python
"""Micro-benchmark of tan_gemm on the shapes of the E6D6 B=128 training step (per-shape TFLOP/s, isolated launches)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import ops

SHAPES = [  # name, M, N, K, a_kc, b_kc, accumulate/split
    ("qkv fwd   ", 8192, 1536, 512, True, True, 0), ("out fwd   ", 8192, 512, 512, True, True, 0),
    ("fc fwd    ", 8192, 2048, 512, True, True, 0), ("proj fwd  ", 8192, 512, 2048, True, True, 0),
    ("joint qkv ", 10240, 1536, 512, True, True, 0),
    ("dX proj   ", 8192, 2048, 512, True, False, 0), ("dX fc     ", 8192, 512, 2048, True, False, 0),
    ("dX qkv    ", 8192, 512, 1536, True, False, 0),
    ("dW fc     ", 2048, 512, 8192, False, False, 8), ("dW qkv    ", 1536, 512, 8192, False, False, 8),
    ("dW out    ", 512, 512, 8192, False, False, 16), ("dW proj   ", 512, 2048, 8192, False, False, 8),
    ("sim fwd   ", 8192, 2048, 512, True, True, 0), ("sim dv    ", 8192, 512, 2048, True, False, 0),
    ("sim dt    ", 2048, 512, 8192, False, False, 8),
]

def main(dtype=torch.bfloat16):
    for name, M, N, K, akc, bkc, split in SHAPES:
        A = torch.randn((M, K) if akc else (K, M), device="cuda").to(dtype)
        B = torch.randn((N, K) if bkc else (K, N), device="cuda").to(dtype)
        C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if split else dtype)
        kw = dict(M=M, N=N, K=K, a_kc=akc, b_kc=bkc)
        if split:
            kw.update(accumulate=True, split_k=split)
        for _ in range(3):
            ops.gemm(A, B, C, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.gemm(A, B, C, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"{name} M={M:6d} N={N:5d} K={K:6d} {'KC' if akc else 'KS'}/{'KC' if bkc else 'KS'} split={split:2d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")

if __name__ == "__main__":
    main()
