"""tan_gemm vs torch fp32/fp64 references on the GPU (through the C ABI)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to("cuda").to(dtype)


def _tol(dtype, K):
    # f32 MFMA is an exact f32 fma chain; bf16 operands are exact products accumulated in f32
    return (1e-5 * max(1.0, K ** 0.5) if dtype == torch.float32 else 2e-2 * max(1.0, (K / 512) ** 0.5))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 136), (64, 520, 512), (33, 20, 24), (256, 1536, 512)])
def test_gemm_layouts(dtype, a_kc, b_kc, M, N, K):
    from temporalalignnet_amd import ops
    A = _mk((M, K) if a_kc else (K, M), dtype, 1)
    B = _mk((N, K) if b_kc else (K, N), dtype, 2)
    Cout = torch.full((M, N), float("nan"), device="cuda", dtype=dtype)
    ops.gemm(A, B, Cout, M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc)
    Af = (A if a_kc else A.t()).double()
    Bf = (B.t() if b_kc else B).double()
    ref = Af @ Bf
    err = (Cout.double() - ref).abs().max().item()
    assert err < _tol(dtype, K) * (ref.abs().max().item() + 1.0), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    from temporalalignnet_amd import ops
    M, N, K = 192, 2048, 512
    X, W = _mk((M, K), dtype, 3), _mk((N, K), dtype, 4) * 0.05
    bias = _mk((N,), torch.float32, 5)
    res = _mk((M, N), dtype, 6)
    ref_pre = X.double() @ W.double().t() + bias.double()
    # bias + residual
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(X, W, out, M=M, N=N, K=K, bias=bias, residual=res)
    tol = 1e-4 if dtype == torch.float32 else 6e-2
    assert (out.double() - (ref_pre + res.double())).abs().max().item() < tol
    # quickgelu with pre-activation side output
    pre = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(X, W, out, M=M, N=N, K=K, bias=bias, act=ops.ACT_QUICKGELU, aux=pre)
    assert (pre.double() - ref_pre).abs().max().item() < tol
    assert (out.double() - ref_pre * torch.sigmoid(1.702 * ref_pre)).abs().max().item() < tol
    # gelu-grad epilogue: out = (X W^T) * gelu'(pre)
    g = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(X, W, g, M=M, N=N, K=K, act=ops.ACT_QUICKGELU_GRAD, aux=pre)
    p = pre.double()
    s = torch.sigmoid(1.702 * p)
    want = (X.double() @ W.double().t()) * (s + 1.702 * p * s * (1 - s))
    assert (g.double() - want).abs().max().item() < tol * 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_splitk_accumulate_and_batch(dtype):
    from temporalalignnet_amd import ops
    # dW-style: contraction over a long, non-multiple-of-tile M
    Mc, N, K = 1000, 96, 72
    dY, X = _mk((Mc, N), dtype, 7), _mk((Mc, K), dtype, 8)
    dW = torch.ones(N, K, device="cuda", dtype=torch.float32)
    ops.gemm(dY, X, dW, M=N, N=K, K=Mc, a_kc=False, b_kc=False, lda=N, ldb=K, accumulate=True, split_k=4)
    ref = dY.double().t() @ X.double() + 1.0
    tol = 1e-3 if dtype == torch.float32 else 0.3
    assert (dW.double() - ref).abs().max().item() < tol
    # batched (similarity-style) with f32 output
    S, R, Mp, Cc = 3, 70, 20, 512
    V, T_ = _mk((S, R, Cc), dtype, 9), _mk((S, Mp, Cc), dtype, 10)
    out = torch.empty(S, R, Mp, device="cuda", dtype=torch.float32)
    ops.gemm(V, T_, out, M=R, N=Mp, K=Cc, batch=S, sA=R * Cc, sB=Mp * Cc, sC=R * Mp)
    ref = torch.einsum("src,smc->srm", V.double(), T_.double())
    assert (out.double() - ref).abs().max().item() < (1e-3 if dtype == torch.float32 else 0.5)


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(8192, 512, 512), (1000, 1536, 512), (520, 2048, 2048), (512, 512, 10240), (136, 40, 64)])
def test_gemm_direct_to_lds_path(a_kc, b_kc, M, N, K):
    """aligned bf16 problems (K % 64 == 0) take the global_load_lds kernel: check every operand orientation, edge tiles,
    split-K accumulation and the fused epilogues against an fp64 reference on asymmetric random data."""
    from temporalalignnet_amd import ops
    dtype = torch.bfloat16
    A = _mk((M, K) if a_kc else (K, M), dtype, 11)
    B = _mk((N, K) if b_kc else (K, N), dtype, 12)
    ref = (A if a_kc else A.t()).double() @ (B.t() if b_kc else B).double()
    tol = 2e-2 * max(1.0, (K / 512) ** 0.5) * (ref.abs().max().item() + 1.0)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=dtype)
    ops.gemm(A, B, out, M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc)
    assert (out.double() - ref).abs().max().item() < tol
    acc = torch.ones(M, N, device="cuda", dtype=torch.float32)
    ops.gemm(A, B, acc, M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, accumulate=True, split_k=2 if K >= 128 else 1)
    assert (acc.double() - ref - 1).abs().max().item() < 2e-3 * (ref.abs().max().item() + 1.0)
    bias = _mk((N,), torch.float32, 13)
    res = _mk((M, N), dtype, 14)
    ops.gemm(A, B, out, M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, bias=bias, residual=res)
    assert (out.double() - (ref + bias.double() + res.double())).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1000, 2048, 512), (300, 520, 128), (64, 20, 24)])
def test_gemm_fused_colsum(dtype, M, N, K):
    """bias-gradient column sums fused into the epilogue (direct-to-LDS path) or appended as a pass (other paths)"""
    from temporalalignnet_amd import ops
    A, B = _mk((M, K), dtype, 21), _mk((N, K), dtype, 22)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    cs = torch.ones(N, device="cuda")
    ops.gemm(A, B, out, M=M, N=N, K=K, colsum=cs)
    ref = A.double() @ B.double().t()
    assert (out.double() - ref).abs().max().item() < _tol(dtype, K) * (ref.abs().max().item() + 1)
    err = (cs.double() - ref.sum(0) - 1).abs().max().item()
    assert err < (1e-5 * M * K ** 0.5 if dtype == torch.float32 else 0.05 * M ** 0.5 + 0.5), err


def test_gemm_random_configurations():
    """Seeded sweep over shapes (incl. ragged M / N, every K-step count parity), layouts, epilogues, batching and split-K: the
    direct-to-LDS kernels, their hand-issued transposing reads, the 4-stage variant and the clamped epilogue all get hit."""
    import numpy as np
    from temporalalignnet_amd import ops
    rs = np.random.RandomState(1234)
    for case in range(48):
        M = int(rs.choice([8, 40, 128, 136, 264, 520, 1000]))
        N = int(rs.choice([8, 16, 72, 128, 256, 520]))
        K = int(rs.choice([64, 128, 192, 320, 512, 1088]))
        a_kc, b_kc = bool(rs.randint(2)), bool(rs.randint(2))
        batch = int(rs.choice([1, 1, 3]))
        mode = rs.choice(["plain", "bias_res", "gelu", "gelu_grad", "relu", "f32_split"])
        A = _mk((batch, M, K) if a_kc else (batch, K, M), torch.bfloat16, 100 + case)
        B = _mk((batch, N, K) if b_kc else (batch, K, N), torch.bfloat16, 200 + case) * 0.1
        Af = (A if a_kc else A.transpose(1, 2)).double()
        Bf = (B.transpose(1, 2) if b_kc else B).double()
        ref = Af @ Bf
        kw = dict(M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, batch=batch, sA=M * K, sB=N * K, sC=M * N)
        tol = 3e-2 * max(1.0, (K / 512) ** 0.5)
        if mode == "f32_split":
            out = torch.zeros(batch, M, N, device="cuda")
            ops.gemm(A, B, out, accumulate=True, split_k=int(rs.choice([1, 2, 4])), **kw)
            want = ref
        else:
            out = torch.full((batch, M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            bias = _mk((N,), torch.float32, 300 + case) if mode != "plain" else None
            pre = ref + (bias.double() if bias is not None else 0.0)
            if mode == "bias_res":
                res = _mk((batch, M, N), torch.bfloat16, 400 + case)
                ops.gemm(A, B, out, bias=bias, residual=res, **kw)
                want = pre + res.double()
            elif mode == "gelu":
                aux = torch.full((batch, M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                ops.gemm(A, B, out, bias=bias, act=ops.ACT_QUICKGELU, aux=aux, **kw)
                want = pre * torch.sigmoid(1.702 * pre)
                assert (aux.double() - pre).abs().max().item() < tol * (pre.abs().max().item() + 1.0), (case, "aux")
            elif mode == "gelu_grad":
                aux = _mk((batch, M, N), torch.bfloat16, 500 + case)
                ops.gemm(A, B, out, bias=bias, act=ops.ACT_QUICKGELU_GRAD, aux=aux, **kw)
                p = aux.double()
                sg = torch.sigmoid(1.702 * p)
                want = pre * (sg + 1.702 * p * sg * (1 - sg))
            elif mode == "relu":
                ops.gemm(A, B, out, bias=bias, act=ops.ACT_RELU, **kw)
                want = pre.clamp(min=0)
            else:
                ops.gemm(A, B, out, **kw)
                want = pre
        err = (out.double() - want).abs().max().item()
        assert err < tol * (want.abs().max().item() + 1.0), (case, M, N, K, a_kc, b_kc, batch, mode, err)


@pytest.mark.parametrize("rows", [8192, 10240, 384, 1024 + 64, 200])
@pytest.mark.parametrize("shapes", [[(1536, 512), (512, 512), (2048, 512), (512, 2048)], [(256, 768)], [(1536, 512), (40, 72)]])
def test_linear_wgrad_group(rows, shapes):
    """tan_linear_wgrad_group: gw_i += dy_i^T x_i for the Linear layers of one block in one call.  rows % 128 == 0 with every
    N, K a multiple of 256 takes the 256 x 256-tile kernel (two K slices adding with atomics, or slices of unequal length when
    rows / 128 is odd); everything else the 128 x 128 grouped kernel or the one-by-one path.  gw starts non-zero: it is +=."""
    import ctypes as C
    from temporalalignnet_amd import ops, _lib
    n = len(shapes)
    dys = [_mk((rows, N), torch.bfloat16, 30 + i) * 0.5 for i, (N, K) in enumerate(shapes)]
    xs = [_mk((rows, K), torch.bfloat16, 40 + i) for i, (N, K) in enumerate(shapes)]
    g0 = [_mk((N, K), torch.float32, 50 + i) for i, (N, K) in enumerate(shapes)]
    gws = [g.clone() for g in g0]
    ws = torch.empty(4 * sum(N * K for N, K in shapes), device="cuda")
    arr_p, arr_i = C.c_void_p * n, C.c_int * n
    rc = _lib.lib().tan_linear_wgrad_group(n, arr_p(*[d.data_ptr() for d in dys]), arr_p(*[x.data_ptr() for x in xs]),
                                           arr_p(*[g.data_ptr() for g in gws]), arr_i(*[N for N, K in shapes]),
                                           arr_i(*[K for N, K in shapes]), C.c_long(rows), C.c_void_p(ws.data_ptr()),
                                           C.c_long(ws.numel()), _lib.TAN_BF16, ops._stream())
    assert rc == 0
    for g, g_init, d, x in zip(gws, g0, dys, xs):
        want = g_init.double() + d.double().t() @ x.double()
        err = (g.double() - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("M,N,K", [(1280, 512, 49152), (512, 512, 24576 + 128), (1280, 512, 49152 + 64)])
def test_gemm_long_contraction_accumulate(M, N, K):
    """C[M,N] (f32) += A[K,M]^T B[K,N], bf16, K = S*B*T rows: the text-feature gradient of the similarity loss (K slices adding
    with atomics into a non-zero C)."""
    from temporalalignnet_amd import ops
    A = _mk((K, M), torch.bfloat16, 60) * 0.25
    B = _mk((K, N), torch.bfloat16, 61) * 0.25
    C0 = _mk((M, N), torch.float32, 62)
    C = C0.clone()
    ops.gemm(A, B, C, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M, ldb=N, accumulate=True, split_k=8)
    ref = C0.double() + A.double().t() @ B.double()
    err = (C.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("K,Ms,N,out,split", [(256, [256], 256, "f32", 1), (512, [300, 256, 40], 512, "f32", 2), (1280, [8192], 512, "bf16", 1),
                                             (384, [520] * 6, 256, "bf16", 1), (2048, [1344] * 2, 512, "f32", 4), (128, [64] * 9, 256, "f32", 1)])
def test_gemm_atb_matches_torch(K, Ms, N, out, split):
    """tan_gemm_atb: C_p = A_p^T B_p on the 256 x 256-tile kernel -- ragged M (the last tile reads past the rows: padded buffer),
    several problems per launch, f32 store / in-place add / K slices through atomics, bf16 store."""
    from temporalalignnet_amd import ops
    torch.manual_seed(K + len(Ms))
    n = len(Ms)
    As, Bs, Cs, refs = [], [], [], []
    for M in Ms:
        lda = (M + 7) // 8 * 8
        buf = torch.zeros(K * lda + 256, device="cuda", dtype=torch.bfloat16)
        A = buf[:K * lda].view(K, lda)
        A.copy_((torch.randn(K, lda, device="cuda") * K ** -0.5).bfloat16())
        B = torch.randn(K, N, device="cuda").bfloat16()
        As.append(A); Bs.append(B)
        refs.append(A[:, :M].double().t() @ B.double())
    ldas = [a.shape[1] for a in As]
    if out == "bf16":
        Cs = [torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) for M in Ms]
        ops.gemm_atb(As, Bs, Cs, lda=ldas, M=Ms, N=N, K=K)
        for c, r in zip(Cs, refs):
            assert torch.isfinite(c.float()).all()
            assert (c.double() - r).abs().max().item() <= 2.0 ** -7 * r.abs().max().item() + 1e-6
    else:
        init = [torch.randn(M, N, device="cuda") for M in Ms]
        Cs = [c.clone() for c in init]
        ops.gemm_atb(As, Bs, Cs, lda=ldas, M=Ms, N=N, K=K, accumulate=True, split=split)
        for c, c0, r in zip(Cs, init, refs):
            assert (c.double() - (c0.double() + r)).abs().max().item() <= 2e-4 * max(1.0, r.abs().max().item())
        if split == 1:
            Cs = [torch.full((M, N), float("nan"), device="cuda") for M in Ms]
            ops.gemm_atb(As, Bs, Cs, lda=ldas, M=Ms, N=N, K=K)
            for c, r in zip(Cs, refs):
                assert (c.double() - r).abs().max().item() <= 2e-4 * max(1.0, r.abs().max().item())
