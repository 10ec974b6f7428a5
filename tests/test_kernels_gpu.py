"""Per-kernel parity of the HIP ops (through the C ABI) against plain PyTorch fp32/fp64 references."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)


def close(a, b, tol):
    err = (a.double() - b.double()).abs().max().item()
    assert err < tol, err


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,period", [(7, 0), (130, 13)])
def test_layernorm_fwd_bwd(dtype, rows, period):
    from temporalalignnet_amd import ops
    C = 512
    x = rnd((rows, C), dtype, 1, 2.0)
    g = 1 + rnd((C,), torch.float32, 2, 0.1)
    b = rnd((C,), torch.float32, 3, 0.1)
    add = rnd((period, C), dtype, 4) if period else None
    y = torch.empty_like(x)
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    ops.layernorm_fwd(x, g, b, y, mean, rstd, add, period)
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    if period:
        ref = ref + add.double().repeat((rows + period - 1) // period, 1)[:rows]
    tol = 1e-5 if dtype == torch.float32 else 5e-2
    close(y, ref, tol)
    dy = rnd((rows, C), dtype, 5)
    dres = rnd((rows, C), dtype, 6)
    ref.backward(dy.double())
    dx = torch.empty_like(x)
    dg = torch.ones(C, device="cuda")
    db = torch.ones(C, device="cuda")
    cs = torch.ones(C, device="cuda")
    ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, dres, dx_colsum=cs)
    close(dx, xr.grad + dres.double(), 1e-4 if dtype == torch.float32 else 8e-2)
    close(cs, (xr.grad + dres.double()).sum(0) + 1, 1e-3 if dtype == torch.float32 else 0.5)
    close(dg, gr.grad + 1, 1e-3 if dtype == torch.float32 else 0.3)
    close(db, br.grad + 1, 1e-3 if dtype == torch.float32 else 0.3)


@pytest.mark.parametrize("dtype", DT)
def test_l2norm_grouped(dtype):
    from temporalalignnet_amd import ops
    B, T, N, C = 3, 5, 4, 512
    L = T + N
    x = rnd((B * L, C), dtype, 7)
    for grp, off in ((T, 0), (N, T)):
        rows = B * grp
        y = torch.empty(rows, C, device="cuda", dtype=dtype)
        inv = torch.empty(rows, device="cuda")
        ops.l2norm_fwd(x, y, inv, rows, C, grp, L, off)
        xs = x.view(B, L, C)[:, off:off + grp].reshape(rows, C).double().requires_grad_(True)
        ref = xs / xs.norm(dim=-1, keepdim=True)
        close(y, ref, 1e-6 if dtype == torch.float32 else 1e-2)
        dy = rnd((rows, C), dtype, 8)
        ref.backward(dy.double())
        dx = torch.zeros(B * L, C, device="cuda", dtype=dtype)
        ops.l2norm_bwd(dy, y, inv, dx, rows, C, grp, L, off)
        got = dx.view(B, L, C)[:, off:off + grp].reshape(rows, C)
        close(got, xs.grad, 1e-5 if dtype == torch.float32 else 2e-2)
        other = torch.ones(L, dtype=torch.bool)
        other[off:off + grp] = False
        assert dx.view(B, L, C)[:, other].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DT)
def test_l2norm_multi_equals_one_launch_per_stage(dtype):
    """tan_l2norm_fwd/bwd_multi: S stage buffers in one launch == S single launches, bit for bit (same kernel, blockIdx.y = stage)."""
    from temporalalignnet_amd import ops
    S, B, T, N, C = 3, 3, 5, 4, 512
    L = T + N
    xs = [rnd((B * L, C), dtype, 20 + s) for s in range(S)]
    for grp, off in ((T, 0), (N, T)):
        rows = B * grp
        y1, y2 = (torch.empty(S, rows, C, device="cuda", dtype=dtype) for _ in range(2))
        i1, i2 = (torch.empty(S * rows, device="cuda") for _ in range(2))
        for s in range(S):
            ops.l2norm_fwd(xs[s], y1[s], i1[s * rows:(s + 1) * rows], rows, C, grp, L, off)
        ops.l2norm_fwd_multi(xs, y2, i2, rows, C, grp, L, off)
        assert torch.equal(y1, y2) and torch.equal(i1, i2)
        dy = rnd((S, rows, C), dtype, 30)
        d1 = [torch.zeros(B * L, C, device="cuda", dtype=dtype) for _ in range(S)]
        d2 = [torch.zeros(B * L, C, device="cuda", dtype=dtype) for _ in range(S)]
        for s in range(S):
            ops.l2norm_bwd(dy[s], y1[s], i1[s * rows:(s + 1) * rows], d1[s], rows, C, grp, L, off)
        ops.l2norm_bwd_multi(dy, y2, i2, d2, rows, C, grp, L, off)
        for a, b in zip(d1, d2):
            assert torch.equal(a, b)
    # more than 8 stages (the C entry point's pointer table): the wrapper goes in groups of 8
    S9 = 9
    xs9 = [xs[s % S] for s in range(S9)]
    y9 = torch.empty(S9, rows, C, device="cuda", dtype=dtype)
    i9 = torch.empty(S9 * rows, device="cuda")
    ops.l2norm_fwd_multi(xs9, y9, i9, rows, C, grp, L, off)
    for s in range(S9):
        assert torch.equal(y9[s], y1[s % S]) and torch.equal(i9[s * rows:(s + 1) * rows], i1[(s % S) * rows:(s % S + 1) * rows])
    d9 = [torch.zeros(B * L, C, device="cuda", dtype=dtype) for _ in range(S9)]
    ops.l2norm_bwd_multi(torch.cat([dy] * 3), y9, i9, d9, rows, C, grp, L, off)
    for s in range(S9):
        assert torch.equal(d9[s], d1[s % S])


@pytest.mark.parametrize("dtype", DT)
def test_small_helpers(dtype):
    from temporalalignnet_amd import ops
    rows, C = 300, 1536
    x = rnd((rows, C), dtype, 9)
    out = torch.ones(C, device="cuda")
    ops.colsum_acc(x, out, rows, C)
    close(out, x.double().sum(0) + 1, 1e-3 if dtype == torch.float32 else 0.5)
    # cat / split with accumulate
    B, T, N, C = 3, 5, 4, 512
    L = T + N
    v, t = rnd((B * T, C), dtype, 10), rnd((B * N, C), dtype, 11)
    j = torch.zeros(B * L, C, device="cuda", dtype=dtype)
    ops.rows_copy(v, j, B, T, C, T, 0, L, 0)
    ops.rows_copy(t, j, B, N, C, N, 0, L, T)
    want = torch.cat([v.view(B, T, C), t.view(B, N, C)], 1)
    assert torch.equal(j.view(B, L, C), want)
    acc = v.clone()
    ops.rows_copy(j, acc, B, T, C, L, 0, T, 0, accumulate=True)
    close(acc, 2 * v.double(), 1e-6 if dtype == torch.float32 else 5e-2)
    gs = torch.empty(T, C, device="cuda", dtype=dtype)
    ops.group_sum(v, gs, B, T, C)
    close(gs, v.double().view(B, T, C).sum(0), 1e-5 if dtype == torch.float32 else 5e-2)
    for G, R, Cc in ((1, 3, 12), (7, 5, 36), (16, 2, 512), (37, 9, 260), (128, 64, 512)):   # every wave / unroll remainder
        xg = rnd((G * R, Cc), dtype, 20 + G)
        gs = torch.empty(R, Cc, device="cuda", dtype=dtype)
        ops.group_sum(xg, gs, G, R, Cc)
        close(gs, xg.double().view(G, R, Cc).sum(0), 1e-4 if dtype == torch.float32 else 0.25)
    if dtype == torch.float32:
        for nparts, n in ((1, 1024), (3, 4096 + 8), (4, 640), (9, 70000), (16, 512 * 2048)):
            parts = rnd((nparts, n), torch.float32, 40 + nparts)
            o = rnd((n,), torch.float32, 60 + nparts)
            want_o = o.double() + parts.double().sum(0)
            ops.reduce_add(parts, o, nparts, n)
            close(o, want_o, 1e-4)
    # cast round trip
    f = rnd((1000 + 3,), torch.float32, 12)
    h = torch.empty(1003, device="cuda", dtype=torch.bfloat16)
    ops.cast(f, h)
    assert torch.equal(h, f.to(torch.bfloat16))
    f2 = torch.empty_like(f)
    ops.cast(h, f2)
    assert torch.equal(f2, h.float())
    # binary head
    rows = 37
    x = rnd((rows, 512), dtype, 13)
    w, b = rnd((512,), torch.float32, 14, 0.1), rnd((1,), torch.float32, 15)
    o = torch.empty(rows, device="cuda")
    ops.head_fwd(x, w, b, o, rows, 512)
    close(o, x.double() @ w.double() + b.double(), 1e-4 if dtype == torch.float32 else 1e-2)
    do = rnd((rows,), torch.float32, 16)
    dx = torch.ones(rows, 512, device="cuda", dtype=dtype)
    dw, db = torch.zeros(512, device="cuda"), torch.zeros(1, device="cuda")
    ops.head_bwd(do, x, w, dx, dw, db, rows, 512, accumulate_dx=True)
    close(dx, 1 + do.double()[:, None] * w.double()[None], 1e-5 if dtype == torch.float32 else 3e-2)
    close(dw, do.double() @ x.double(), 1e-3 if dtype == torch.float32 else 1e-1)
    close(db, do.double().sum(), 1e-4)


def test_interp_linear():
    from temporalalignnet_amd import ops
    src = rnd((64, 512), torch.float32, 17)
    for L_out in (100, 40, 64):
        dst = torch.empty(L_out, 512, device="cuda")
        ops.interp_linear(src, dst, 64, L_out, 512)
        ref = F.interpolate(src.t()[None], size=L_out, mode="linear", align_corners=False)[0].t()
        close(dst, ref, 1e-5)
        dd = rnd((L_out, 512), torch.float32, 18)
        ds = torch.zeros(64, 512, device="cuda")
        ops.interp_linear_bwd(dd, ds, 64, L_out, 512)
        s2 = src.clone().requires_grad_(True)
        F.interpolate(s2.t()[None], size=L_out, mode="linear", align_corners=False)[0].t().backward(dd)
        close(ds, s2.grad, 1e-4)


def _attn_ref(qkv, keypad, B, L, H):
    C = H * 64
    q, k, v = qkv.view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * 0.125) @ k.transpose(-1, -2)
    if keypad is not None:
        s = s.masked_fill(keypad.bool()[:, None, None, :], float("-inf"))
    o = torch.softmax(s, -1) @ v
    return o.transpose(1, 2).reshape(B * L, C)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,L,pad", [(2, 64, False), (3, 80, True), (1, 272, True), (2, 17, True), (2, 128, True), (2, 100, False), (2, 200, True), (1, 400, False),
                                     (2, 129, True), (1, 160, False), (2, 192, True), (1, 224, True), (2, 256, True), (2, 288, True), (1, 289, True)])
def test_attention_fwd_bwd(dtype, B, L, pad):
    from temporalalignnet_amd import ops
    H, C = 8, 512
    qkv = rnd((B * L, 3 * C), dtype, 20, 1.5)
    keypad = None
    if pad:
        keypad = torch.zeros(B, L, dtype=torch.uint8, device="cuda")
        keypad[0, L - L // 5:] = 1
        keypad[-1, 3] = 1
    o = torch.empty(B * L, C, device="cuda", dtype=dtype)
    lse = torch.empty(B, H, L, device="cuda")
    ops.attn_fwd(qkv, keypad, o, lse, B, L, H)
    qr = qkv.double().requires_grad_(True)
    ref = _attn_ref(qr, keypad, B, L, H)
    close(o, ref, 2e-5 if dtype == torch.float32 else 4e-2)
    d_o = rnd((B * L, C), dtype, 21)
    ref.backward(d_o.double())
    dqkv = torch.full_like(qkv, float("nan"))
    ops.attn_bwd(qkv, keypad, o, lse, d_o, dqkv, B, L, H)
    close(dqkv, qr.grad, 1e-4 if dtype == torch.float32 else 1e-1)
    # the entry that also accumulates the in_proj bias gradient: same dqkv, g += its column sums (of the stored values)
    dqkv2 = torch.full_like(qkv, float("nan"))
    g0 = rnd((3 * C,), torch.float32, 22)
    g = g0.clone()
    ops.attn_bwd(qkv, keypad, o, lse, d_o, dqkv2, B, L, H, g_b_qkv=g)
    assert torch.equal(dqkv2, dqkv)
    want = dqkv.double().sum(0)
    close(g - g0, want, 1e-4 * (1.0 + want.abs().max().item()))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("L", [64, 80, 272, 256, 400])
def test_attention_with_every_key_padded_gives_zeros_not_nan(dtype, L):
    """Documented difference: a video whose keys are ALL padded makes nn.MultiheadAttention (tfm_model.py:32)
    return NaN for every query of that video -- softmax over an all -inf row.  The HIP kernels define that case as zeros:
    o = 0, lse = -inf, dqkv = 0, and the other videos of the batch are untouched.  (The reference's loaders never produce such a
    window -- data/loader_htm.py:151 -- so nothing downstream depends on the NaN.)"""
    from temporalalignnet_amd import ops
    B, H, C = 3, 8, 512
    qkv = rnd((B * L, 3 * C), dtype, 40, 1.5)
    keypad = torch.zeros(B, L, dtype=torch.uint8, device="cuda")
    keypad[1] = 1                                      # video 1: nothing to attend to
    o = torch.full((B * L, C), float("nan"), device="cuda", dtype=dtype)
    lse = torch.full((B, H, L), float("nan"), device="cuda")
    ops.attn_fwd(qkv, keypad, o, lse, B, L, H)
    assert torch.isnan(_attn_ref(qkv.double(), keypad, B, L, H).view(B, L, C)[1]).all()          # what torch does
    ov = o.view(B, L, C)
    assert (ov[1] == 0).all() and torch.isinf(lse[1]).all() and (lse[1] < 0).all()
    keep = torch.tensor([0, 2], device="cuda")
    ref = _attn_ref(qkv.double(), keypad, B, L, H).view(B, L, C)
    close(ov[keep], ref[keep], 2e-5 if dtype == torch.float32 else 4e-2)
    d_o = rnd((B * L, C), dtype, 41)
    dqkv = torch.full_like(qkv, float("nan"))
    ops.attn_bwd(qkv, keypad, o, lse, d_o, dqkv, B, L, H)
    dv = dqkv.view(B, L, 3 * C)
    assert (dv[1] == 0).all() and torch.isfinite(dv).all()


def test_transpose_batch():
    """tan_transpose_batch: several bf16 matrices inside one flat buffer -> their transposes at the same offsets."""
    import ctypes as C
    from temporalalignnet_amd import _lib, ops
    shapes = [(1536, 512), (512, 512), (2048, 512), (512, 2048), (72, 40)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += (r * c + 15) // 16 * 16
    src = rnd((total,), torch.bfloat16, 70)
    dst = torch.full_like(src, float("nan"))
    table = torch.tensor([[o, r, c] for o, (r, c) in zip(offs, shapes)], dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().tan_transpose_batch(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(table.data_ptr()),
                                              len(shapes), C.c_long(2048), C.c_long(2048), _lib.TAN_BF16, ops._stream()), "transpose")
    for o, (r, c) in zip(offs, shapes):
        assert torch.equal(dst[o:o + r * c].view(c, r), src[o:o + r * c].view(r, c).t())
