"""Row f3: global-negative NCE across ranks (temporalalignnet_amd/dist_nce.py).

Two ranks are SIMULATED inside one process by driving `BlockNCE`'s phases for both and doing the three collectives by hand
(all-gather = python lists, all-reduce = +, reduce-scatter = sum of the parts addressed to a rank); the result must equal the
single-device fused NCE on the concatenated batch of 2*B videos -- which is the semantics of the reference at that batch size.
A second test runs the real collective path (RCCL, one rank) through the Trainer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit(shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    return (x / x.norm(dim=-1, keepdim=True)).to(torch.bfloat16).cuda()


def _rank_inputs(seed, S, St, B, T, N, C, leak):
    rs = np.random.RandomState(seed)
    n_b = rs.randint(2, N + 1, size=B)
    n_b[0] = N
    ci = torch.tensor(np.concatenate([(np.arange(N) >= n).astype(np.uint8) for n in n_b]), device="cuda")
    tgt = torch.zeros(B, T, N)
    for b in range(B):
        for k in range(n_b[b]):
            s = rs.randint(0, T - 2)
            tgt[b, s:s + rs.randint(1, 4), k] = 1.0
    row_leak = None
    if leak:
        row_leak = torch.zeros(B * T, dtype=torch.uint8)
        row_leak[T - 2:T] = 1
        row_leak[3 * T - 1] = 1
        row_leak = row_leak.cuda()
    return dict(vn=_unit((S, B * T, C), seed + 1), tn=_unit((St, B * N, C), seed + 2), tgt=tgt.cuda().contiguous(), ci=ci,
                row_leak=row_leak)


@pytest.mark.parametrize("shared,leak", [(False, False), (True, False), (False, True)])
def test_two_simulated_ranks_equal_the_concatenated_batch(shared, leak):
    from temporalalignnet_amd.dist_nce import BlockNCE
    from temporalalignnet_amd.loss import _FusedNCEFn
    S, B, T, N, C = 3, 6, 16, 5, 128
    St = 1 if shared else S
    R, Mp = B * T, B * N
    ranks = [_rank_inputs(100 + 10 * r, S, St, B, T, N, C, leak) for r in range(2)]
    # ---- reference: one device, 2B videos
    vn_c = torch.cat([x["vn"] for x in ranks], 1).clone().requires_grad_(True)
    tn_c = torch.cat([x["tn"] for x in ranks], 1).clone().requires_grad_(True)
    tgt_c, ci_c = torch.cat([x["tgt"] for x in ranks], 0), torch.cat([x["ci"] for x in ranks], 0)
    leak_c = torch.cat([x["row_leak"] for x in ranks], 0) if leak else None
    v_ref, t_ref = _FusedNCEFn.apply(vn_c, tn_c, tgt_c, ci_c, leak_c, 2 * B, T, N, None)
    g = torch.Generator().manual_seed(5)
    g_v, g_t = torch.randn(S, 2 * R, generator=g).cuda(), torch.randn(S, 2 * Mp, generator=g).cuda()
    g_t = g_t * (1 - ci_c.float())[None]                       # padded sentences never receive a gradient (masked means)
    d_vn_ref, d_tn_ref = torch.autograd.grad([v_ref, t_ref], [vn_c, tn_c], [g_v, g_t])
    # ---- two simulated ranks
    blks = [BlockNCE(x["vn"], x["tgt"], x["row_leak"], B, T, N, shared_text=shared) for x in ranks]
    tn_all, ci_all = [x["tn"] for x in ranks], [x["ci"] for x in ranks]                # all-gather
    colsum_all = sum(blk.sweep(tn_all, ci_all, r) for r, blk in enumerate(blks))       # all-reduce
    outs = [blk.finish(colsum_all) for blk in blks]
    valid_rows = torch.ones(2 * R, dtype=torch.bool, device="cuda") if not leak else ~leak_c.bool()
    v_sim, t_sim = torch.cat([o[0] for o in outs], 1), torch.cat([o[1] for o in outs], 1)
    ok_cols = ~ci_c.bool()
    torch.testing.assert_close(v_sim[:, valid_rows], v_ref[:, valid_rows], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(v_sim, v_ref, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(t_sim[:, ok_cols], t_ref[:, ok_cols], rtol=2e-5, atol=2e-5)
    g_t_all = torch.stack([g_t[:, :Mp], g_t[:, Mp:]], 0).contiguous()                   # all-gather
    back = [blk.backward(g_v[:, r * R:(r + 1) * R].contiguous(), g_t_all) for r, blk in enumerate(blks)]
    d_vn_sim = torch.cat([b[0] for b in back], 1)
    d_tn_sim = torch.cat([back[0][1][r] + back[1][1][r] for r in range(2)], 1)          # reduce-scatter
    def close(a, b, what):
        a, b = a.float(), b.float()
        assert (a - b).norm() <= 2e-2 * b.norm() + 1e-6, (what, float((a - b).norm()), float(b.norm()))
    close(d_vn_sim, d_vn_ref, "d_vn")
    close(d_tn_sim, d_tn_ref, "d_tn")


@pytest.fixture
def one_rank_rccl():
    import torch.distributed as dist
    if dist.is_initialized():
        yield
        return
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_trainer_with_global_negatives_on_one_rank_equals_local(one_rank_rccl):
    """World size 1: the global batch IS the local batch -- same loss, same gradient, through the real collectives."""
    from temporalalignnet_amd import synth
    from temporalalignnet_amd.train import Trainer, build_model, default_args, to_device_batch
    args = default_args(model="init", num_encoder_layers=2, num_decoder_layers=2)
    b = to_device_batch(synth.make_batch(9, B=8, T=32, n_min=3, n_max=7))
    res = []
    for glob in (False, True):
        torch.manual_seed(0)
        m = build_model(args, compute_dtype="bf16", random_pos_start=0).cuda()
        tr = Trainer(m, args, global_negatives=glob)
        tr.zero_grad()
        ld = tr.forward_backward(b)
        res.append(({k: v.item() for k, v in ld.items()}, tr.online.flat_grad().clone()))
    (l0, g0), (l1, g1) = res
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-4 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    assert (g0 - g1).norm() <= 2e-2 * g0.norm()


@pytest.mark.parametrize("shared", [False, True])
def test_two_simulated_ranks_match_the_oracle_on_the_concatenated_batch(shared):
    """the truth for dist_nce.py is the ORACLE's NCE (oracle/loss_ref.nce = train/loss.py:240-275, pinned by
    goldens G3/G4) on the logits of the concatenated 2*B_local batch -- not another HIP path.  Terms, the normalised loss with
    global counts, and the feature gradients."""
    from oracle import loss_ref
    from temporalalignnet_amd.dist_nce import BlockNCE
    S, B, T, N, C = 3, 6, 16, 5, 128
    St = 1 if shared else S
    R, Mp = B * T, B * N
    ranks = [_rank_inputs(300 + 10 * r, S, St, B, T, N, C, False) for r in range(2)]
    # ---- oracle (CPU fp32) on the concatenated batch: cosine logits [2B,S,T,2B,N] / 0.07, block-diagonal targets
    vn_c = torch.cat([x["vn"] for x in ranks], 1).float().cpu().requires_grad_(True)            # [S, 2R, C]
    tn_c = torch.cat([x["tn"] for x in ranks], 1).float().cpu().requires_grad_(True)            # [St, 2Mp, C]
    B2 = 2 * B
    tn_s = tn_c.expand(S, -1, -1) if shared else tn_c
    logits = torch.einsum("srk,smk->srm", vn_c, tn_s).view(S, B2, T, B2, N).permute(1, 0, 2, 3, 4) / 0.07
    tgt_c = torch.cat([x["tgt"] for x in ranks], 0).cpu()                                       # [2B, T, N]
    keep = ~torch.cat([x["ci"] for x in ranks], 0).bool().cpu().view(B2, N)
    tgt_cols = loss_ref._block_diag(tgt_c, B2)[:, :, keep].reshape(B2 * T, -1)
    v_ref, t_ref = loss_ref.nce(logits, tgt_cols, keep)                                         # [S, 2R], [S, M]
    rows_pos, cols_pos = tgt_cols.sum(-1) > 0, tgt_cols.sum(-2) > 0
    loss_ref_val = (v_ref[:, rows_pos].mean() + t_ref[:, cols_pos].mean()) / 2
    d_vn_ref, d_tn_ref = torch.autograd.grad(loss_ref_val, [vn_c, tn_c])
    # ---- two simulated ranks (collectives by hand, as above)
    blks = [BlockNCE(x["vn"], x["tgt"], None, B, T, N, shared_text=shared) for x in ranks]
    tn_all, ci_all = [x["tn"] for x in ranks], [x["ci"] for x in ranks]
    colsum_all = sum(blk.sweep(tn_all, ci_all, r) for r, blk in enumerate(blks))
    outs = [blk.finish(colsum_all) for blk in blks]
    v_sim, t_sim = torch.cat([o[0] for o in outs], 1).cpu(), torch.cat([o[1] for o in outs], 1).cpu()
    torch.testing.assert_close(v_sim, v_ref.detach(), rtol=2e-3, atol=2e-3)                     # bf16 features, f32 sums
    torch.testing.assert_close(t_sim[:, keep.view(-1)], t_ref.detach(), rtol=2e-3, atol=2e-3)
    # global normalisation: each rank divides by the ALL-rank counts, the global loss is the SUM of the rank losses
    n_rows, n_cols = rows_pos.sum().item(), cols_pos.sum().item()
    cols_pos_pad = torch.zeros(2 * Mp, dtype=torch.bool)
    cols_pos_pad[keep.view(-1)] = cols_pos
    rank_losses, g_v, g_t_blocks = [], [], []
    for r in range(2):
        rp, cp = rows_pos[r * R:(r + 1) * R], cols_pos_pad[r * Mp:(r + 1) * Mp]
        lv = outs[r][0].cpu()[:, rp].sum() / (S * n_rows)
        lt = outs[r][1].cpu()[:, cp].sum() / (S * n_cols)
        rank_losses.append((lv + lt) / 2)
        g_v.append((rp.float() / (2 * S * n_rows)).expand(S, R).contiguous().cuda())
        g_t_blocks.append((cp.float() / (2 * S * n_cols)).expand(S, Mp).contiguous().cuda())
    assert abs(sum(rank_losses).item() - loss_ref_val.item()) <= 2e-3 * abs(loss_ref_val.item())
    g_t_all = torch.stack(g_t_blocks, 0).contiguous()
    back = [blk.backward(g_v[r], g_t_all) for r, blk in enumerate(blks)]
    d_vn_sim = torch.cat([b[0] for b in back], 1).float().cpu()
    d_tn_sim = torch.cat([back[0][1][r] + back[1][1][r] for r in range(2)], 1).float().cpu()

    def close(a, b, what):
        assert (a - b).norm() <= 3e-2 * b.norm() + 1e-7, (what, float((a - b).norm()), float(b.norm()))
    close(d_vn_sim, d_vn_ref, "d_vn")
    close(d_tn_sim, d_tn_ref, "d_tn")
