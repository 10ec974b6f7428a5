"""Stand-alone time of the attention core at one shape (bf16): fwd, bwd.  env B, L; TAN_ATTN_MID=0 for the streamed kernels."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from temporalalignnet_amd import ops
B, L, H, C = int(os.environ.get("B", 32)), int(os.environ.get("L", 272)), 8, 512
g = torch.Generator(device="cuda").manual_seed(1)
qkv = (torch.randn(B * L, 3 * C, device="cuda", generator=g) * 1.5).bfloat16()
d_o = torch.randn(B * L, C, device="cuda", generator=g).bfloat16()
o = torch.empty(B * L, C, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, L, device="cuda")
dqkv = torch.empty_like(qkv); gb = torch.zeros(3 * C, device="cuda")
keypad = torch.zeros(B, L, dtype=torch.uint8, device="cuda"); keypad[:, L - 5:] = 1


def t(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"B={B} L={L} mid={os.environ.get('TAN_ATTN_MID', '1')}: fwd {t(lambda: ops.attn_fwd(qkv, keypad, o, lse, B, L, H)):.1f} us, "
      f"bwd {t(lambda: ops.attn_bwd(qkv, keypad, o, lse, d_o, dqkv, B, L, H, g_b_qkv=gb)):.1f} us")
