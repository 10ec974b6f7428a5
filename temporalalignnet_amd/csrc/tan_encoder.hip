// Host-side runner for one TemporalEncoder stack (model/tfm_model.py:41-55): S pre-LN residual attention blocks
// (tfm_model.py:17-38) forward and backward as a fixed sequence of kernel launches on one stream -- one C call per
// stack instead of ~25 Python->C transitions per layer.  No allocation, no synchronisation: every buffer is passed in.
//
// forward, per layer (rows R = B*L):
//   xn1 = LN1(x_in)                      tan_layernorm_fwd
//   qkv = xn1 W_in^T + b_in              tan_gemm
//   o   = attention(qkv, key_padding)    tan_attn_fwd
//   x_mid = x_in + o W_out^T + b_out     tan_gemm (+bias +residual epilogue)
//   xn2 = LN2(x_mid)                     tan_layernorm_fwd
//   h   = quickgelu(xn2 W_fc^T + b_fc)   tan_gemm (activation epilogue, pre-activation kept for backward)
//   x_out = x_mid + h W_proj^T + b_proj  tan_gemm (+bias +residual epilogue)
// deep-supervision stage s < S-1 is layer s+1's xn1 (TemporalEncoder.forward drops the first ln_1 output and appends
// the final residual stream); the last stage is post-LN'ed by the caller-provided ln_*_post_enc (tan_model.py:174,206).
#include "tan_common.h"
#include <cstdlib>

#ifndef TAN_DW_TARGET_WGS
#define TAN_DW_TARGET_WGS 256
#endif

namespace tal {
int gemm_dw_grouped(int nprob, const void* const* dy, const void* const* x, float* const* parts, const int* Ms, const int* Ns,
                    long rows, int split, int accumulate, hipStream_t st);      // tan_gemm_glds.hip
int gemm_dw256_grouped(int nprob, const void* const* dy, const void* const* x, float* const* gw, const int* Ms, const int* Ns,
                       long rows, int split, hipStream_t st);                   // tan_gemm_glds.hip
}

using namespace tal;

namespace {

int linear_fwd(int dt, const void* x, const void* w, const float* b, void* y, long M, int N, int K, int act, void* aux,
               const void* residual, void* st) {
    tan_gemm_desc d{};
    d.dtype = dt; d.out_dtype = dt;
    d.M = (int)M; d.N = N; d.K = K;
    d.a_kc = 1; d.b_kc = 1;
    d.A = x; d.lda = K; d.B = w; d.ldb = K; d.C = y; d.ldc = N;
    d.bias = b; d.residual = residual; d.ldr = N; d.act = act; d.aux = aux; d.ldaux = N;
    d.accumulate = 0; d.split_k = 1; d.alpha = 1.0f; d.batch = 1;
    return tan_gemm(&d, st);
}

// dx[M,K] = dy[M,N] W[N,K]  (optionally * gelu'(aux) and + residual)
int linear_bwd_x(int dt, const void* dy, const void* w, const void* wt, void* dx, long M, int N, int K, int act, void* aux,
                 const void* residual, float* colsum, void* st) {
    tan_gemm_desc d{};
    d.dtype = dt; d.out_dtype = dt;
    d.M = (int)M; d.N = K; d.K = N;
    d.A = dy; d.lda = N; d.C = dx; d.ldc = K;
    if (wt) { d.a_kc = 1; d.b_kc = 1; d.B = wt; d.ldb = N; }     // W^T [K(out), N(contract)]: both operands K-contiguous
    else    { d.a_kc = 1; d.b_kc = 0; d.B = w; d.ldb = K; }      // W   [N(contract), K(out)] read K-strided
    d.residual = residual; d.ldr = K; d.act = act; d.aux = aux; d.ldaux = K;
    d.split_k = 1; d.alpha = 1.0f; d.batch = 1;
    d.colsum = colsum;
    return tan_gemm(&d, st);
}

// gw[N,K] += dy[M,N]^T x[M,K]: the long M contraction is cut into `split` slices.  With a workspace the slices are written
// as plain f32 partial tiles (batched GEMM, 16-byte stores) and folded into the gradient by one streaming kernel -- measured
// 37 us vs 47 us for f32 atomics on the c_fc shape; without a workspace (or for ragged M) the slices accumulate with atomics.
int linear_bwd_w(int dt, const void* dy, const void* x, float* gw, long M, int N, int K, float* ws, long ws_floats, void* st) {
    tan_gemm_desc d{};
    d.dtype = dt; d.out_dtype = TAN_F32;
    d.M = N; d.N = K;
    d.a_kc = 0; d.b_kc = 0;
    d.A = dy; d.lda = N; d.B = x; d.ldb = K; d.ldc = K;
    d.alpha = 1.0f;
    const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
    const long target = TAN_DW_TARGET_WGS;
    long want = (target + tiles - 1) / tiles;   // workgroups per dW GEMM (the other stack co-runs on a 2nd stream)
    const long max_split = (M + 255) / 256;
    if (want > max_split) want = max_split;
    if (want > 32) want = 32;
    if (want < 1) want = 1;
    long split = want;
    while (split > 1 && (M % split != 0 || (M / split) % 64 != 0)) --split;      // equal slices, multiples of the K-step
    if (ws && split > 1 && (long)split * N * K <= ws_floats) {
        const long kc = M / split;
        d.K = (int)kc; d.C = ws; d.accumulate = 0; d.split_k = 1;
        d.batch = (int)split; d.sA = kc * N; d.sB = kc * K; d.sC = (long)N * K;
        int rc = tan_gemm(&d, st);
        if (rc) return rc;
        return tan_reduce_add(ws, gw, (int)split, (long)N * K, st);
    }
    d.K = (int)M; d.C = gw; d.accumulate = 1; d.batch = 1; d.split_k = (int)want;
    return tan_gemm(&d, st);
}

#define CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

// The four weight gradients of one block in ONE grouped launch, issued after the block's dX chain: the 256 x 256-tile kernel
// (gemm_dw256_kernel, DW256_SPLIT K slices adding into the f32 gradient with atomics: 2 beat 3, 4 and 5 inside the step), else the
// 128 x 128 grouped kernel (192 workgroups, each contracting ALL rows straight into the gradient), else -- f32, ragged rows -- the
// four GEMMs one by one.
struct DwItem { const void* dy; const void* x; float* gw; int N, K; };

constexpr int DW256_SPLIT = 2;

int linear_bwd_w_group(int dt, const DwItem* it, int n, long M, float* ws, long ws_floats, void* st, int split = 0) {
    if (const int s256 = dt == TAN_BF16 ? (split > 0 ? split : DW256_SPLIT) : 0) {
        bool ok = M % 128 == 0 && M / 128 >= s256;
        for (int i = 0; i < n; ++i) ok = ok && it[i].N % 256 == 0 && it[i].K % 256 == 0;
        if (ok) {
            const void* dy[4]; const void* x[4]; float* gw[4]; int Ms[4], Ns[4];
            double work = 0;
            for (int i = 0; i < n; ++i) {
                dy[i] = it[i].dy; x[i] = it[i].x; gw[i] = it[i].gw; Ms[i] = it[i].N; Ns[i] = it[i].K;
                work += 2.0 * M * it[i].N * (double)it[i].K;
            }
            const int rec = prof_begin((hipStream_t)st, TAN_PROF_GEMM_BF16 + 3, work);
            const int rc = gemm_dw256_grouped(n, dy, x, gw, Ms, Ns, M, s256, (hipStream_t)st);
            prof_end((hipStream_t)st, rec);
            if (rc != -2) return rc;
        }
    }
    if (dt == TAN_BF16 && M % 64 == 0) {
        const void* dy[4]; const void* x[4]; float* parts[4]; int Ms[4], Ns[4];
        double work = 0;
        for (int i = 0; i < n; ++i) {
            dy[i] = it[i].dy; x[i] = it[i].x; parts[i] = it[i].gw; Ms[i] = it[i].N; Ns[i] = it[i].K;
            work += 2.0 * M * it[i].N * (double)it[i].K;
        }
        const int rec = prof_begin((hipStream_t)st, TAN_PROF_GEMM_BF16 + 3, work);
        const int rc = gemm_dw_grouped(n, dy, x, parts, Ms, Ns, M, 1, 1, (hipStream_t)st);
        prof_end((hipStream_t)st, rec);
        if (rc != -2) return rc;
    }
    for (int i = 0; i < n; ++i) CK(linear_bwd_w(dt, it[i].dy, it[i].x, it[i].gw, M, it[i].N, it[i].K, ws, ws_floats, st));
    return 0;
}

}  // namespace

extern "C" int tan_linear_wgrad_group(int n, const void* const* dy, const void* const* x, float* const* gw, const int* N, const int* K,
                                      long M, float* ws, long ws_floats, int dtype, void* stream) {
    TAN_REQUIRE(n >= 1 && n <= 4 && dy && x && gw && N && K && M > 0);
    DwItem it[4];
    for (int i = 0; i < n; ++i) { TAN_REQUIRE(dy[i] && x[i] && gw[i] && N[i] > 0 && K[i] > 0); it[i] = DwItem{dy[i], x[i], gw[i], N[i], K[i]}; }
    return linear_bwd_w_group(dtype, it, n, M, ws, ws_floats, stream);
}

extern "C" int tan_linear_wgrad(const void* dy, const void* x, float* gw, long M, int N, int K, float* ws, long ws_floats,
                                int dtype, void* stream) {
    TAN_REQUIRE(dy && x && gw && M > 0 && N > 0 && K > 0);
    return linear_bwd_w(dtype, dy, x, gw, M, N, K, ws, ws_floats, stream);
}

// Row-panel path (tan_panel.hip): the MLP half of a block -- LN2, c_fc + QuickGELU, c_proj + residual AND the LayerNorm that
// consumes the block's output (the next block's ln_1, or the stack's post-LN) -- is ONE launch, and so is the attention half where
// tan_attnblk_fwd takes the shape (48 < L <= 80); elsewhere the out-projection rides as the head of the MLP launch.  Backward: the
// MLP launch carries the ln_1 backward of the block above as its prologue, that block's in_proj dX GEMM as its head and the own
// block's out_proj dX GEMM as its tail.  TAN_PANEL=0 (the only switch left here): the unfused launches -- LayerNorm + tiled GEMMs --
// which are also what runs for f32, C != 512 or rows % 64 != 0.  Every alternative that was measured slower or neutral in rounds 2-3
// (the one-launch attention backward, the head-only block-0 launch, the in_proj tail, a separate weight-gradient stream, chunk
// rotation, burst schedules: DESIGN_APPENDIX.md A.9) is gone from the library.
static int panel_enabled() {
    static const int on = [] { const char* e = getenv("TAN_PANEL"); return e ? atoi(e) : 1; }();
    return on;
}
// Stacks of fewer row panels than this run the MLP branch through the split-hidden launches (tan_mlp_fwd_split / tan_mlp_bwd_split:
// eight workgroups per panel); the pieces the whole-panel backward carries as head / prologue / tail -- the in_proj dX GEMM, the ln_1
// backward, the out_proj dX GEMM -- are then the launches they were before they were folded in.  TAN_SPLIT_PANELS=0: never.
static long split_panels() {
    static const long n = [] { const char* e = getenv("TAN_SPLIT_PANELS"); return e ? atol(e) : 48L; }();
    return n;
}
// ... and the BACKWARD's MLP branch only up to this many panels: split, it is five launches (in_proj dX GEMM, ln_1 backward, the split
// launch, its row epilogue, out_proj dX GEMM: 26 + 13 + 70 + 23 + 25 us at B = 32) where the whole-panel launch is one (128 us); at
// B = 16 (16 / 20 panels) 26 + 14 + 40 + 15 + 26 against 133.  min(TAN_SPLIT_BWD_PANELS, TAN_SPLIT_PANELS).
static long split_bwd_panels() {
    static const long n = [] { const char* e = getenv("TAN_SPLIT_BWD_PANELS"); const long v = e ? atol(e) : 24L; return v < split_panels() ? v : split_panels(); }();
    return n;
}

extern "C" int tan_encoder_fwd(const tan_encoder_desc* e, void* st) {
    TAN_REQUIRE(e && e->layers > 0 && e->params && e->bufs && e->x0);
    const int dt = e->dtype, C = e->C, H = e->H;
    const long R = (long)e->B * e->L;
    const void* x_in = e->x0;
    const bool panel_ok = panel_enabled() && dt == TAN_BF16 && C == 512 && R % 64 == 0;
    const bool attn_panel_ok = panel_enabled() && tan_attnblk_supported(e->L, C, H, dt);
    const bool split = panel_ok && e->split_part && R / 64 <= split_panels();
    bool ln1_done = e->xn1_ready != 0;      // the previous block's panel kernel (block 0: tan_embed_fwd) already produced this block's xn1 / mean1 / rstd1
    for (int i = 0; i < e->layers; ++i) {
        const tan_layer_params& p = e->params[i];
        const tan_layer_bufs& b = e->bufs[i];
        if (!ln1_done) CK(tan_layernorm_fwd(x_in, p.ln1_g, p.ln1_b, b.xn1, b.mean1, b.rstd1, nullptr, 0, R, C, 1e-5f, dt, st));
        ln1_done = false;
        bool out_head = false;
        if (attn_panel_ok && p.wp_qkv && p.wp_out) {
            // one launch: in_proj GEMM, the 8 heads' attention and out_proj + bias + residual, one workgroup per video (tan_attnblk.hip)
            tan_attnblk_desc ab{};
            ab.B = e->B; ab.L = e->L; ab.C = C; ab.H = H;
            ab.xn1 = b.xn1; ab.x_in = x_in; ab.key_padding_mask = e->key_padding_mask;
            ab.pw_qkv = p.wp_qkv; ab.pw_out = p.wp_out; ab.b_qkv = p.b_qkv; ab.b_out = p.b_out;
            if (!e->no_save) { ab.qkv = b.qkv; ab.attn_o = b.attn_o; ab.lse = b.lse; }
            ab.x_mid = b.x_mid;
            if (split) CK(tan_attnblk_fwd_split(&ab, e->split_part, st));
            else CK(tan_attnblk_fwd(&ab, st));
        } else {
            CK(linear_fwd(dt, b.xn1, p.w_qkv, p.b_qkv, b.qkv, R, 3 * C, C, TAN_ACT_NONE, nullptr, nullptr, st));
            CK(tan_attn_fwd(b.qkv, e->key_padding_mask, b.attn_o, b.lse, e->B, e->L, H, dt, st));
            // out_proj + bias + residual: the head of the row-panel MLP forward below
            out_head = panel_ok && !split && p.wp_fc && p.wp_proj && p.wp_out;
            if (!out_head) CK(linear_fwd(dt, b.attn_o, p.w_out, p.b_out, b.x_mid, R, C, C, TAN_ACT_NONE, nullptr, x_in, st));
        }
        if (panel_ok && p.wp_fc && p.wp_proj) {
            tan_mlp_desc m{};
            m.rows = R; m.C = C; m.FF = 4 * C;
            if (out_head) { m.attn_o = b.attn_o; m.pw_out = p.wp_out; m.b_out = p.b_out; m.x_in = x_in; }
            m.x_mid = b.x_mid; m.ln_g = p.ln2_g; m.ln_b = p.ln2_b;
            m.pw_fc = p.wp_fc; m.pw_proj = p.wp_proj; m.b_fc = p.b_fc; m.b_proj = p.b_proj;
            m.x_out = b.x_out;
            if (!e->no_save) { m.xn2 = b.xn2; m.mean2 = b.mean2; m.rstd2 = b.rstd2; m.h_pre = b.h_pre; m.h_act = b.h_act; }
            m.eps = 1e-5f;
            if (i + 1 < e->layers) {
                const tan_layer_params& pn = e->params[i + 1];
                const tan_layer_bufs& bn = e->bufs[i + 1];
                m.nln_g = pn.ln1_g; m.nln_b = pn.ln1_b; m.xn_next = bn.xn1; m.nmean = bn.mean1; m.nrstd = bn.rstd1;
                ln1_done = true;
            } else if (e->post_out) {
                m.nln_g = e->post_g; m.nln_b = e->post_b; m.xn_next = e->post_out; m.nmean = e->post_mean; m.nrstd = e->post_rstd;
                ln1_done = true;      // = the post-LN is done
            }
            if (split) CK(tan_mlp_fwd_split(&m, e->split_part, st));
            else CK(tan_mlp_fwd(&m, st));
        } else {
            CK(tan_layernorm_fwd(b.x_mid, p.ln2_g, p.ln2_b, b.xn2, b.mean2, b.rstd2, nullptr, 0, R, C, 1e-5f, dt, st));
            CK(linear_fwd(dt, b.xn2, p.w_fc, p.b_fc, b.h_act, R, 4 * C, C, TAN_ACT_QUICKGELU, b.h_pre, nullptr, st));
            CK(linear_fwd(dt, b.h_act, p.w_proj, p.b_proj, b.x_out, R, C, 4 * C, TAN_ACT_NONE, nullptr, b.x_mid, st));
        }
        x_in = b.x_out;
    }
    if (e->post_out && !ln1_done)
        CK(tan_layernorm_fwd(x_in, e->post_g, e->post_b, e->post_out, e->post_mean, e->post_rstd, nullptr, 0, R, C, 1e-5f, dt, st));
    return 0;
}

// d_stage[s] (may be NULL = no gradient) is the gradient w.r.t. deep-supervision stage s; parameter gradients are
// ACCUMULATED into the g_* pointers of params[] (f32).  d_x0 receives the gradient w.r.t. the stack input.
extern "C" int tan_encoder_bwd(const tan_encoder_desc* e, void* st) {
    TAN_REQUIRE(e && e->layers > 0 && e->params && e->bufs && e->x0 && e->d_stage && e->d_x0);
    TAN_REQUIRE(e->scr_dx && e->scr_dx2 && e->scr_dh && e->scr_dqkv && e->scr_do && e->scr_dxn && e->ln_ws);
    const int dt = e->dtype, C = e->C, H = e->H, S = e->layers;
    const long R = (long)e->B * e->L;
    const size_t esz = dt == TAN_F32 ? 4 : 2;
    const void* x_last = e->bufs[S - 1].x_out;
    // ln_1 backward of block i+1 handed to block i's row-panel MLP backward as its prologue; the stack's post-LayerNorm backward goes
    // to the last block the same way (no residual gradient next to it)
    struct { bool on; int layer; const void *dxn, *x, *res; const float *mean, *rstd, *g; float *gg, *gb, *gcol;
             const void *dqkv, *pwt_in, *dstage; } pend{};
    const bool panel_all = panel_enabled() && dt == TAN_BF16 && C == 512 && R % 64 == 0;
    const bool split = panel_all && e->split_part && R / 64 <= split_bwd_panels();
    // the last `tail` blocks' weight-gradient launches on e->dw_stream (see tan_hip.h); tail > 1: those blocks alternate between two
    // sets of the scratch buffers the launch reads
    int tail = e->dw_stream && e->dw_stream != st ? (e->dw_tail > 0 ? e->dw_tail : 0) : 0;
    if (tail > S) tail = S;
    if (tail > 16) tail = 16;
    const bool two_sets = e->scr2_dx && e->scr2_dx2 && e->scr2_dh && e->scr2_dqkv;
    if (tail > 1 && !two_sets) tail = 1;
    static thread_local hipEvent_t ev_in = nullptr, ev_done[16] = {};
    if (tail > 0 && !ev_in) {
        if (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess) return -3;
        for (int i = 0; i < 16; ++i)
            if (hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming) != hipSuccess) return -3;
    }
    auto set_b = [&](int i) { return tail > 1 && i < tail && (i & 1); };
    // gradient w.r.t. the residual stream leaving the current layer (the set of the block that reads it)
    void* dx = set_b(S - 1) ? e->scr2_dx : e->scr_dx;
    if (e->d_stage[S - 1] && panel_all && !split && e->params[S - 1].wtp_fc && e->params[S - 1].wtp_proj) {
        TAN_REQUIRE(e->post_out);
        pend.on = true; pend.layer = -1; pend.dxn = e->d_stage[S - 1]; pend.x = x_last; pend.res = nullptr;
        pend.mean = e->post_mean; pend.rstd = e->post_rstd; pend.g = e->post_g; pend.gg = e->g_post_g; pend.gb = e->g_post_b;
        pend.gcol = e->params[S - 1].g_b_proj;
    } else if (e->d_stage[S - 1]) {
        TAN_REQUIRE(e->post_out);
        // dx = grad of the last layer's x_out: its column sums are that layer's c_proj bias gradient
        CK(tan_layernorm_bwd(e->d_stage[S - 1], x_last, e->post_g, e->post_mean, e->post_rstd, nullptr, dx, e->g_post_g,
                             e->g_post_b, e->params[S - 1].g_b_proj, e->ln_ws, R, C, dt, st));
    } else {
        hipError_t err = hipMemsetAsync(dx, 0, (size_t)R * C * esz, (hipStream_t)st);
        if (err != hipSuccess) return (int)err;
    }
    // bias gradients that are column sums of a LayerNorm-backward OUTPUT are accumulated inside that kernel:
    //   g_b_proj[i] <- colsum(dx entering layer i)   = output of layer i+1's LN1 backward (or of the post-LN backward)
    //   g_b_out[i]  <- colsum(dx2)                   = output of layer i's LN2 backward
    for (int i = S - 1; i >= 0; --i) {
        const tan_layer_params& p = e->params[i];
        const tan_layer_bufs& b = e->bufs[i];
        const void* x_in = i == 0 ? e->x0 : e->bufs[i - 1].x_out;
        const bool B_ = set_b(i);
        dx = B_ ? e->scr2_dx : e->scr_dx;
        void* const dx2 = B_ ? e->scr2_dx2 : e->scr_dx2;
        void* const scr_dh = B_ ? e->scr2_dh : e->scr_dh;
        void* const scr_dqkv = B_ ? e->scr2_dqkv : e->scr_dqkv;
        if (tail > 1 && i + 2 < tail)      // block i + 2 used this set: its weight-gradient launch must have read it
            if (hipStreamWaitEvent((hipStream_t)st, ev_done[i + 2], 0) != hipSuccess) return -3;
        // ---- MLP branch: x_out = x_mid + c_proj(quickgelu(c_fc(LN2(x_mid))))
        bool do_fused = false;                             // d_o = dx2 W_out already produced by the row-panel MLP backward
        if (panel_all && p.wtp_fc && p.wtp_proj) {
            // one launch: dh = (dx W_proj) o quickgelu'(h_pre), dxn = dh W_fc, LN2 backward + residual -> dx2, four parameter
            // gradients that are column sums (tan_panel.hip)
            tan_mlp_bwd_desc m{};
            m.rows = R; m.C = C; m.FF = 4 * C;
            m.dx = dx; m.h_pre = b.h_pre; m.x_mid = b.x_mid; m.mean2 = b.mean2; m.rstd2 = b.rstd2; m.ln_g = p.ln2_g;
            m.pwt_proj = p.wtp_proj; m.pwt_fc = p.wtp_fc;
            m.dh = scr_dh; m.dx2 = dx2;
            m.g_b_fc = p.g_b_fc; m.g_ln_g = p.g_ln2_g; m.g_ln_b = p.g_ln2_b; m.g_b_out = p.g_b_out;
            if (pend.on) {
                m.ln1_dxn = pend.dxn; m.ln1_x = pend.x; m.ln1_res = pend.res; m.ln1_mean = pend.mean; m.ln1_rstd = pend.rstd;
                m.ln1_g = pend.g; m.g_ln1_g = pend.gg; m.g_ln1_b = pend.gb; m.g_dx_colsum = pend.gcol; m.dx_out = dx;
                m.dqkv = pend.dqkv; m.pwt_in = pend.pwt_in; m.dstage = pend.dstage;      // (the in_proj dX GEMM in front of it, or NULLs)
            }
            // the out-projection's dX GEMM as the tail of the same launch
            do_fused = !split && p.wtp_out != nullptr;
            if (do_fused) { m.pwt_out = p.wtp_out; m.d_o = e->scr_do; }
            if (split) CK(tan_mlp_bwd_split(&m, e->split_part, st));
            else CK(tan_mlp_bwd(&m, st));
            if (pend.on) {          // every gradient of block pend.layer is final now
                pend.on = false;
                if (pend.layer >= 0 && e->layer_done && e->layer_done[pend.layer]) {
                    const hipError_t err = hipEventRecord((hipEvent_t)e->layer_done[pend.layer], (hipStream_t)st);
                    if (err != hipSuccess) return (int)err;
                }
            }
        } else {
            CK(linear_bwd_x(dt, dx, p.w_proj, p.wt_proj, scr_dh, R, C, 4 * C, TAN_ACT_QUICKGELU_GRAD, b.h_pre, nullptr, p.g_b_fc, st));
            CK(linear_bwd_x(dt, scr_dh, p.w_fc, p.wt_fc, e->scr_dxn, R, 4 * C, C, TAN_ACT_NONE, nullptr, nullptr, nullptr, st));
            CK(tan_layernorm_bwd(e->scr_dxn, b.x_mid, p.ln2_g, b.mean2, b.rstd2, dx, dx2, p.g_ln2_g, p.g_ln2_b, p.g_b_out, e->ln_ws, R, C,
                                 dt, st));
        }
        // ---- attention branch: x_mid = x_in + out_proj(attn(LN1(x_in)))
        if (!do_fused) CK(linear_bwd_x(dt, dx2, p.w_out, p.wt_out, e->scr_do, R, C, C, TAN_ACT_NONE, nullptr, nullptr, nullptr, st));
        CK(tan_attn_bwd_bias(b.qkv, e->key_padding_mask, b.attn_o, b.lse, e->scr_do, scr_dqkv, p.g_b_qkv, e->B, e->L, H, dt, st));
        // stage i-1 IS this layer's xn1: its gradient joins here
        const void* dstage = i >= 1 ? e->d_stage[i - 1] : nullptr;
        // block i-1's row-panel MLP backward takes the ln_1 backward as its prologue -- and this dX GEMM in front of it as its head
        const bool ln1_next = i > 0 && panel_all && !split && e->params[i - 1].wtp_fc && e->params[i - 1].wtp_proj;
        const bool in_fused = ln1_next && p.wtp_qkv != nullptr;
        if (!in_fused)
            CK(linear_bwd_x(dt, scr_dqkv, p.w_qkv, p.wt_qkv, e->scr_dxn, R, 3 * C, C, TAN_ACT_NONE, nullptr, dstage, nullptr, st));
        {                   // dx, scr_dh, dx2, scr_dqkv are all still intact here (LN1 backward below overwrites dx)
            const DwItem items[4] = {{scr_dh, b.xn2, p.g_w_fc, 4 * C, C}, {dx, b.h_act, p.g_w_proj, C, 4 * C},
                                     {scr_dqkv, b.xn1, p.g_w_qkv, 3 * C, C}, {dx2, b.attn_o, p.g_w_out, C, C}};
            if (i < tail) {
                // nothing of this stack's backward depends on these weight gradients; their operands stay untouched until the block two
                // below reuses the set (it waits for ev_done), block 0's for good
                if (hipEventRecord(ev_in, (hipStream_t)st) != hipSuccess) return -3;
                if (hipStreamWaitEvent((hipStream_t)e->dw_stream, ev_in, 0) != hipSuccess) return -3;
                // (more K slices for the LAST of these launches, which runs next to a draining chip: 4 slices +0.03 ms per step, 8 slices
                //  +0.08, ABBA x2 of 60 steps, round 5)
                CK(linear_bwd_w_group(dt, items, 4, R, e->dw_ws, e->dw_ws_floats, e->dw_stream));
                if (hipEventRecord(ev_done[i], (hipStream_t)e->dw_stream) != hipSuccess) return -3;
            } else
            CK(linear_bwd_w_group(dt, items, 4, R, e->dw_ws, e->dw_ws_floats, st));
        }
        void* dx_in = i == 0 ? e->d_x0 : (set_b(i - 1) ? e->scr2_dx : e->scr_dx);
        float* next_b_proj = i > 0 ? e->params[i - 1].g_b_proj : nullptr;       // dx_in is layer i-1's x_out gradient
        if (ln1_next) {
            // block i-1's row-panel MLP backward does this LayerNorm backward as its prologue (dx2 and scr_dxn / scr_dqkv stay
            // untouched until that launch: it is the next one that writes them)
            pend.on = true; pend.layer = i; pend.dxn = in_fused ? nullptr : e->scr_dxn; pend.x = x_in; pend.res = dx2;
            pend.dqkv = in_fused ? scr_dqkv : nullptr; pend.pwt_in = in_fused ? p.wtp_qkv : nullptr; pend.dstage = in_fused ? dstage : nullptr;
            pend.mean = b.mean1; pend.rstd = b.rstd1; pend.g = p.ln1_g; pend.gg = p.g_ln1_g; pend.gb = p.g_ln1_b; pend.gcol = next_b_proj;
            continue;
        }
        // (this launch writes the OTHER scratch set's dx -- block i - 1's input; with every block's weight gradients on dw_stream, block
        //  i + 1's launch, which reads that buffer, may still be running: in the whole-panel path the buffer is written an iteration later,
        //  behind the wait at its top)
        if (tail > 1 && i >= 1 && i + 1 < tail)
            if (hipStreamWaitEvent((hipStream_t)st, ev_done[i + 1], 0) != hipSuccess) return -3;
        CK(tan_layernorm_bwd(e->scr_dxn, x_in, p.ln1_g, b.mean1, b.rstd1, dx2, dx_in, p.g_ln1_g, p.g_ln1_b, next_b_proj, e->ln_ws, R,
                             C, dt, st));
        // every gradient of layer i is final here (g_b_proj[i] was written during iteration i+1 / by the post-LN backward)
        if (e->layer_done && e->layer_done[i]) {
            const hipError_t err = hipEventRecord((hipEvent_t)e->layer_done[i], (hipStream_t)st);
            if (err != hipSuccess) return (int)err;
        }
    }
    return 0;
}
