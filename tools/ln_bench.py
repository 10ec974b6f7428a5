"""Stand-alone timing of the row-wise kernels at the step's sizes (bf16, C=512): LayerNorm fwd / bwd(+finalize), L2-norm fwd / bwd.
Prints us per call and the HBM rate on the algorithmic bytes.  Tool only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from temporalalignnet_amd import ops
reps = int(os.environ.get("REPS", 200))
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
C = 512
for R in (8192, 10240):
    x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); res = torch.randn(R, C, device="cuda").bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    mean, rstd = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    dg, db, dc = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    inv = torch.empty(R, device="cuda")
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd)); print(f"R={R} ln_fwd            {t:6.1f} us {2*R*C*2/t/1e3:6.0f} GB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, dres=res, dx_colsum=dc)); print(f"R={R} ln_bwd+finalize   {t:6.1f} us {4*R*C*2/t/1e3:6.0f} GB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dres=res)); print(f"R={R} ln_bwd only       {t:6.1f} us {4*R*C*2/t/1e3:6.0f} GB/s")
    t = timeit(lambda: ops.l2norm_fwd(x, y, inv, R, C)); print(f"R={R} l2n_fwd           {t:6.1f} us {2*R*C*2/t/1e3:6.0f} GB/s")
    t = timeit(lambda: ops.l2norm_bwd(dy, y, inv, dx, R, C)); print(f"R={R} l2n_bwd           {t:6.1f} us {3*R*C*2/t/1e3:6.0f} GB/s")
