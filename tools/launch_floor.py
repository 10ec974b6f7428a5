"""Back-to-back launch cost vs size: T(n) = a + n / rate for the simplest streaming kernels (cast f32->bf16, LayerNorm fwd).  Tool only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import ops
reps = 300
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for n in (1024, 65536, 1 << 20, 1 << 22, 1 << 23, 1 << 24):
    f = torch.randn(n, device="cuda"); h = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.cast(f, h))
    t2 = timeit(lambda: h.copy_(f))
    print(f"cast n={n:9d}: {t:6.2f} us ({6*n/t/1e3:6.0f} GB/s)   torch copy_ {t2:6.2f} us")
g, b = torch.randn(512, device="cuda"), torch.randn(512, device="cuda")
for R in (64, 1024, 4096, 8192, 16384, 65536):
    x = torch.randn(R, 512, device="cuda").bfloat16(); y = torch.empty_like(x)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, y))
    print(f"ln_fwd R={R:6d}: {t:6.2f} us ({4*R*512/t/1e3:6.0f} GB/s)")
