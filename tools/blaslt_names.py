import torch
R = 8192
for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    x = torch.randn(R, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    dy = torch.randn(R, N, device="cuda").bfloat16()
    for _ in range(3):
        torch.matmul(x, w.t()); torch.matmul(dy, w); torch.matmul(dy.t(), x)
torch.cuda.synchronize()
