// Row-wise normalisation and small HBM-bound helpers (gfx950).  One 64-lane wave owns one row of C channels
// (C = 512 -> 8 channels per lane as two 16-byte / 8-byte vectors), reductions are wave shuffles, no LDS.
#include "tan_common.h"
#include <cstdlib>

namespace tal {

constexpr int ROWS_PER_BLOCK = 4;  // 256 threads

// ------------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma + beta (+ add[row % add_period])
// reference: nn.LayerNorm(512) at tfm_model.py:22,28,35,37 and tan_model.py:50-54,155,167,174,206
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     const T* __restrict__ add, int add_period, long rows, float eps) {
    constexpr int C = NCH * 256;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * C;
    // every load of the row is issued up front (x, the affine parameters, the optional addend -- read from x itself when there
    // is none, so that no load sits behind a branch): behind the two reductions they were a dependent L2 round trip
    const T* ar = add ? add + (long)(row % add_period) * C : xr;
    const float amul = add ? 1.0f : 0.0f;
    float4 v[NCH], gmm[NCH], bta[NCH], adv[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = ld4(xr + c);
        gmm[i] = *reinterpret_cast<const float4*>(gamma + c);
        bta[i] = *reinterpret_cast<const float4*>(beta + c);
        adv[i] = ld4(ar + c);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
    if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
    }
    T* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 g = gmm[i], b = bta[i], a = adv[i];
        st4(yr + c, make_float4(v[i].x * rstd * g.x + b.x + a.x * amul, v[i].y * rstd * g.y + b.y + a.y * amul,
                                v[i].z * rstd * g.z + b.z + a.z * amul, v[i].w * rstd * g.w + b.w + a.w * amul));
    }
}

// LayerNorm backward.  dx = (dres) + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.
// Each block walks a strided set of rows and keeps per-lane partial d_gamma / d_beta, written to
// ws[block][2][C]; ln_bwd_finalize adds them into the f32 parameter gradients.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ ws, long rows) {
    constexpr int C = NCH * 256;
    __shared__ float red[ROWS_PER_BLOCK][3][C];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 dg[NCH], db[NCH], gm[NCH], ds[NCH];      // ds: column sums of the OUTPUT dx (a fused tan_colsum_acc)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        dg[i] = make_float4(0, 0, 0, 0); db[i] = make_float4(0, 0, 0, 0); ds[i] = make_float4(0, 0, 0, 0);
        gm[i] = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
    }
#pragma unroll 2
    for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + w; row < rows; row += (long)gridDim.x * ROWS_PER_BLOCK) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        float4 xh[NCH], g[NCH], rs[NCH];
        float s1 = 0.f, s2 = 0.f;
        // the residual gradient is loaded UNCONDITIONALLY with the other loads of the row (from x when there is none): a load
        // behind `if (dres)` gets a branch and its own s_waitcnt -- after the reductions it was two dependent round trips
        const T* rsrc = dres ? dres : x;
        const float rmul = dres ? 1.0f : 0.0f;          // arithmetic, not a branch: keeps both chunk loads in the batch
#pragma unroll
        for (int i = 0; i < NCH; ++i) rs[i] = ld4(rsrc + row * C + (i * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (i * 64 + lane) * 4;
            const float4 xv = ld4(x + row * C + c), d = ld4(dy + row * C + c);
            xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
            dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
            db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        }
        const float m1 = wave_sum(s1) * (1.0f / C), m2 = wave_sum(s2) * (1.0f / C);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (i * 64 + lane) * 4;
            float4 o = make_float4(rstd * (g[i].x - m1 - xh[i].x * m2), rstd * (g[i].y - m1 - xh[i].y * m2),
                                   rstd * (g[i].z - m1 - xh[i].z * m2), rstd * (g[i].w - m1 - xh[i].w * m2));
            o.x += rs[i].x * rmul; o.y += rs[i].y * rmul; o.z += rs[i].z * rmul; o.w += rs[i].w * rmul;
            st4(dx + row * C + c, o);
            ds[i].x += o.x; ds[i].y += o.y; ds[i].z += o.z; ds[i].w += o.w;
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(&red[w][0][c]) = dg[i];
        *reinterpret_cast<float4*>(&red[w][1][c]) = db[i];
        *reinterpret_cast<float4*>(&red[w][2][c]) = ds[i];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * C; idx += 256) {
        const int which = idx / C, c = idx % C;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; ++r) s += red[r][which][c];
        ws[((long)blockIdx.x * 3 + which) * C + c] = s;
    }
}

// ---- bf16, C = 512 fast path: a lane owns 8 CONTIGUOUS channels, i.e. one 16-byte access per lane per tensor row (the 8-byte
// accesses of the generic kernels run at 0.54-0.70x the 16-byte rate per byte), RPW rows per wave with every load of all rows
// issued before the first reduction, the affine parameters loaded once per wave.
template <int RPW>
__global__ __launch_bounds__(256) void ln_fwd_bf16x8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                            const bf16_t* __restrict__ add, int add_period, long rows, float eps) {
    constexpr int C = 512;
    const int lane = threadIdx.x & 63, c = lane * 8;
    const long row0 = ((long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const float amul = add ? 1.0f : 0.0f;
    f8 v[RPW], a[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = min(row0 + r, rows - 1);           // clamped: the duplicate row is computed and not stored
        v[r] = ld8(x + row * C + c);
        a[r] = ld8(add ? add + (long)(row % add_period) * C + c : x + row * C + c);
    }
    const f8 g = ld8f(gamma + c), b = ld8f(beta + c);
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[r].v[j];
        mean[r] = wave_sum(s) * (1.0f / C);
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[r].v[j] -= mean[r]; q += v[r].v[j] * v[r].v[j]; }
        rstd[r] = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r;
        if (row < rows) {
            f8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = v[r].v[j] * rstd[r] * g.v[j] + b.v[j] + a[r].v[j] * amul;
            st8(y + row * C + c, o);
            if (lane == 0) {
                if (mean_o) mean_o[row] = mean[r];
                if (rstd_o) rstd_o[row] = rstd[r];
            }
        }
    }
}

// ATOMIC = false: per-block partial table ws[block][3][C] (folded by ln_bwd_finalize); ATOMIC = true: the block adds its column
// sums straight into dgamma / dbeta / dx_colsum (each may be NULL) with f32 atomics -- no second kernel, no workspace.
template <bool ATOMIC, int RPI>
__global__ __launch_bounds__(256) void ln_bwd_bf16x8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i, const bf16_t* __restrict__ dres,
                                                            bf16_t* __restrict__ dx, float* __restrict__ ws, long rows,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dx_colsum) {
    constexpr int C = 512;
    __shared__ float red[ROWS_PER_BLOCK][3][C];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane * 8;
    const f8 gm = ld8f(gamma + c);
    f8 dg, db, ds;
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg.v[j] = 0.f; db.v[j] = 0.f; ds.v[j] = 0.f; }
    const bf16_t* rsrc = dres ? dres : x;                   // unconditional third load (see ln_bwd_kernel)
    const float rmul = dres ? 1.0f : 0.0f;
    const long stride = (long)gridDim.x * ROWS_PER_BLOCK;
    // RPI rows per iteration, all 3*RPI row loads up front (a block is alone on its CU in the atomic mode: the latency has to be
    // covered inside the wave); rows past the end are clamped to the first row of the group and weighted 0
    for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + w; row < rows; row += RPI * stride) {
        f8 xv[RPI], dv[RPI], rv[RPI];
        float mean[RPI], rstd[RPI], wgt[RPI];
        long rr[RPI];
#pragma unroll
        for (int q = 0; q < RPI; ++q) {
            const long rq = row + q * stride;
            wgt[q] = rq < rows ? 1.0f : 0.0f;
            rr[q] = rq < rows ? rq : row;
            xv[q] = ld8(x + rr[q] * C + c); dv[q] = ld8(dy + rr[q] * C + c); rv[q] = ld8(rsrc + rr[q] * C + c);
            mean[q] = mean_i[rr[q]]; rstd[q] = rstd_i[rr[q]];
        }
        float s1[RPI], s2[RPI];
#pragma unroll
        for (int q = 0; q < RPI; ++q) {
            s1[q] = 0.f; s2[q] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xv[q].v[j] = (xv[q].v[j] - mean[q]) * rstd[q];              // xhat
                const float d = dv[q].v[j];
                dv[q].v[j] = d * gm.v[j];                                    // g = dy * gamma
                s1[q] += dv[q].v[j]; s2[q] += dv[q].v[j] * xv[q].v[j];
                dg.v[j] += wgt[q] * (d * xv[q].v[j]);
                db.v[j] += wgt[q] * d;
            }
        }
#pragma unroll
        for (int q = 0; q < RPI; ++q) {
            const float m1 = wave_sum(s1[q]) * (1.0f / C), m2 = wave_sum(s2[q]) * (1.0f / C);
            f8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o.v[j] = rstd[q] * (dv[q].v[j] - m1 - xv[q].v[j] * m2) + rv[q].v[j] * rmul;
                ds.v[j] += wgt[q] * o.v[j];
            }
            if (row + q * stride < rows) st8(dx + rr[q] * C + c, o);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[w][0][c + j] = dg.v[j]; red[w][1][c + j] = db.v[j]; red[w][2][c + j] = ds.v[j]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * C; idx += 256) {
        const int which = idx / C, cc = idx % C;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; ++r) s += red[r][which][cc];
        if constexpr (ATOMIC) {
            float* out = which == 0 ? dgamma : (which == 1 ? dbeta : dx_colsum);
            if (out) unsafeAtomicAdd(out + cc, s);
        } else {
            ws[((long)blockIdx.x * 3 + which) * C + cc] = s;
        }
    }
}

// Folds the per-block partial tables ws[nblk][3][C] into the f32 gradients.  Grid (3C/64, 8): a block owns 64 consecutive
// table entries (256-B coalesced rows) and one eighth of the partial list; 256 threads = 64 columns x 4 list lanes; the
// eight list slices meet in the gradient with f32 atomics (3C*8 of them per call).
constexpr int LN_BWD_MAX_BLOCKS = 1024, LN_FIN_SLICES = 8;
__global__ __launch_bounds__(256) void ln_bwd_finalize(const float* __restrict__ ws, int nblk, int C, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, float* __restrict__ dx_colsum) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + col;          // index into [3][C]
    const int per = (nblk + LN_FIN_SLICES - 1) / LN_FIN_SLICES;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < 3 * C) {
        const float* p = ws + idx;
        int b = b0 + part;
        for (; b + 12 < b1; b += 16) {
            s0 += p[(long)b * 3 * C]; s1 += p[(long)(b + 4) * 3 * C]; s2 += p[(long)(b + 8) * 3 * C]; s3 += p[(long)(b + 12) * 3 * C];
        }
        for (; b < b1; b += 4) s0 += p[(long)b * 3 * C];
    }
    red[part][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && idx < 3 * C) {
        const float s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        const int which = idx / C, c = idx % C;
        float* out = which == 0 ? dgamma : (which == 1 ? dbeta : dx_colsum);
        if (out) unsafeAtomicAdd(out + c, s);
    }
}

// ------------------------------------------------------------------------------------------------------
// L2 normalisation over channels (no epsilon): tan_model.py:116-117,136-137.  Rows may be gathered from
// / scattered to a grouped layout: src row = (r / grp) * src_grp_rows + src_off + r % grp.
// blockIdx.y = stage: the deep-supervision stages are separate buffers (xs.p[stage]); outputs are stage-major contiguous.
struct L2nPtrs { const void* p[8]; };
template <typename T, int NCH>
__global__ __launch_bounds__(256) void l2n_fwd_kernel(L2nPtrs xs, T* __restrict__ y, float* __restrict__ inv_o,
                                                      long rows, int grp, int src_grp_rows, int src_off) {
    constexpr int C = NCH * 256;
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* __restrict__ x = (const T*)xs.p[blockIdx.y];
    y += (long)blockIdx.y * rows * C;
    if (inv_o) inv_o += (long)blockIdx.y * rows;
    const long sr = (r / grp) * src_grp_rows + src_off + r % grp;
    float4 v[NCH];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = ld4(x + sr * C + (i * 64 + lane) * 4);
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float inv = 1.0f / sqrtf(wave_sum(q));
    if (lane == 0 && inv_o) inv_o[r] = inv;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        st4(y + r * C + (i * 64 + lane) * 4, make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv));
}

// dx = (dy - y * <y, dy>) * inv_norm, scattered back to the grouped layout (plain store)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void l2n_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                      const float* __restrict__ inv_i, L2nPtrs dxs, long rows,
                                                      int grp, int dst_grp_rows, int dst_off) {
    constexpr int C = NCH * 256;
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= rows) return;
    T* __restrict__ dx = (T*)dxs.p[blockIdx.y];
    dy += (long)blockIdx.y * rows * C;
    y += (long)blockIdx.y * rows * C;
    inv_i += (long)blockIdx.y * rows;
    const long dr = (r / grp) * dst_grp_rows + dst_off + r % grp;
    float4 d[NCH], yv[NCH];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        d[i] = ld4(dy + r * C + (i * 64 + lane) * 4);
        yv[i] = ld4(y + r * C + (i * 64 + lane) * 4);
        dot += (d[i].x * yv[i].x + d[i].y * yv[i].y) + (d[i].z * yv[i].z + d[i].w * yv[i].w);
    }
    dot = wave_sum(dot);
    const float inv = inv_i[r];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        st4(dx + dr * C + (i * 64 + lane) * 4,
            make_float4((d[i].x - yv[i].x * dot) * inv, (d[i].y - yv[i].y * dot) * inv, (d[i].z - yv[i].z * dot) * inv,
                        (d[i].w - yv[i].w * dot) * inv));
}

// ------------------------------------------------------------------------------------------------------
// column sum (bias gradients): out[c] += sum_r x[r][c].  A 256-thread block covers `rows_per_block` rows: TPR = C/8
// threads span one row with 16-byte loads (8 bf16 / 2x4 f32), the remaining 256/TPR thread groups take alternate rows;
// partial sums meet in LDS and leave as one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int C,
                                                     int rows_per_block, int tpr) {
    __shared__ float red[256 * 8];
    const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, ny = 256 / tpr;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int c = (blockIdx.x * tpr + tx) * 8;
    if (c < C && ty < ny) {
#pragma unroll 4
        for (long r = r0 + ty; r < r1; r += ny) {
            const float4 a = ld4(x + r * C + c), b = ld4(x + r * C + c + 4);
            acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
            acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < tpr * 8; i += 256) {
        const int cx = i / 8, e = i % 8, col = (blockIdx.x * tpr + cx) * 8 + e;
        if (col >= C) continue;
        float s = 0.f;
        for (int y = 0; y < ny; ++y) s += red[(y * tpr + cx) * 8 + e];
        unsafeAtomicAdd(out + col, s);
    }
}

// Batched bf16 matrix transpose: matrix i = src[table[3i] ..] of shape [rows=table[3i+1], cols=table[3i+2]] (row-major) ->
// dst at the SAME element offset, shape [cols, rows].  Grid (max tiles per matrix, matrices); 64x64 tiles through LDS with
// 16-byte global accesses on both sides.  Used once per optimizer step for the K-contiguous copies of the Linear weights that
// the dX GEMMs read (tan_layer_params.wt_*).
__global__ __launch_bounds__(256) void transpose_batch_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                              const long* __restrict__ table) {
    __shared__ bf16_t tile[64][72];
    const long off = table[3 * blockIdx.y];
    const int rows = (int)table[3 * blockIdx.y + 1], cols = (int)table[3 * blockIdx.y + 2];
    const int tc = (cols + 63) / 64, ntile = ((rows + 63) / 64) * tc;
    if ((int)blockIdx.x >= ntile) return;
    const int r0 = (blockIdx.x / tc) * 64, c0 = (blockIdx.x % tc) * 64;
    const bf16_t* S = src + off;
    bf16_t* D = dst + off;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = threadIdx.x + 256 * it, r = v >> 3, c = (v & 7) * 8;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (r0 + r < rows && c0 + c < cols) x = *reinterpret_cast<const uint4*>(S + (long)(r0 + r) * cols + c0 + c);
        *reinterpret_cast<uint4*>(&tile[r][c]) = x;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = threadIdx.x + 256 * it, c = v >> 3, r = (v & 7) * 8;      // output row c0+c, 8 source rows r..r+7
        if (c0 + c >= cols || r0 + r >= rows) continue;
        union { uint4 u; bf16_t h[8]; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = tile[r + e][c];
        *reinterpret_cast<uint4*>(D + (long)(c0 + c) * rows + r0 + r) = o.u;
    }
}

// generic fallback (any C): one thread per column, 64 rows per block
template <typename T>
__global__ __launch_bounds__(256) void colsum_generic_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * 64, r1 = min(rows, r0 + 64);
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += ld_f(x + r * C + c);
    unsafeAtomicAdd(out + c, s);
}

// grouped row copy / add: dst[(g*dgs + doff + r)*C + c] (=|+=) src[(g*sgs + soff + r)*C + c]
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int G, int R, int C,
                                                   long sgs, long soff, long dgs, long doff) {
    const long n4 = (long)G * R * C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int c = (int)(e % C);
        const long gr = e / C;
        const int r = (int)(gr % R);
        const long g = gr / R;
        const T* s = src + ((g * sgs + soff + r) * C + c);
        T* d = dst + ((g * dgs + doff + r) * C + c);
        float4 v = ld4(s);
        if constexpr (ACC) {                       // compile-time: both loads issue together
            const float4 o = ld4(d);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        st4(d, v);
    }
}

// row gather with a map: dst[s][m][:] = map[m] >= 0 ? src[s][map[m]][:] : 0   (the compacted text-feature gradient back in the padded
// [B*N] row order: every row is written, so no fill in front of it)
template <typename T>
__global__ __launch_bounds__(256) void rows_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ map,
                                                          int S, long Msrc, long Mdst, int C) {
    const long n4 = (long)S * Mdst * C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int c = (int)(e % C);
        const long sm = e / C, m = sm % Mdst, st = sm / Mdst;
        const int j = map[m];
        const float4 v = j >= 0 ? ld4(src + ((st * Msrc + j) * C + c)) : make_float4(0.f, 0.f, 0.f, 0.f);
        st4(dst + e, v);
    }
}

// out[r][c] = sum_g x[(g*R + r)][c]   (broadcast-add backward: position embedding gradient)
template <typename T>
__global__ __launch_bounds__(256) void group_sum_kernel(const T* __restrict__ x, T* __restrict__ out, int G, int R, int C) {
    // a lane owns 4 consecutive elements of the [R,C] plane, the 4 waves of a workgroup take the groups g = wave (mod 4);
    // 4 group loads in flight per iteration (a plain g-loop compiles to one load + s_waitcnt vmcnt(0) per group)
    __shared__ float4 part[4][64];
    const long n = (long)R * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = ((long)blockIdx.x * 64 + lane) * 4;
    const long ic = i < n ? i : 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int g = wave;
    for (; g + 12 < G; g += 16) {
        const float4 a = ld4(x + (long)g * n + ic), b = ld4(x + (long)(g + 4) * n + ic);
        const float4 c = ld4(x + (long)(g + 8) * n + ic), d = ld4(x + (long)(g + 12) * n + ic);
        s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
        s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; g < G; g += 4) {
        const float4 a = ld4(x + (long)g * n + ic);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < n) {
        const float4 p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
        st4(out + i, make_float4((s.x + p1.x) + (p2.x + p3.x), (s.y + p1.y) + (p2.y + p3.y), (s.z + p1.z) + (p2.z + p3.z),
                                 (s.w + p1.w) + (p2.w + p3.w)));
    }
}

// out[i] += sum_s parts[s][i]   (split-K partial tiles -> f32 gradient)
__global__ __launch_bounds__(256) void reduce_add_kernel(const float* __restrict__ parts, float* __restrict__ out, int nparts, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        float4 a = *reinterpret_cast<const float4*>(out + i);
        int s = 0;
        for (; s + 4 <= nparts; s += 4) {                      // 4 partial tiles in flight (see group_sum_kernel)
            const float4 p0 = *reinterpret_cast<const float4*>(parts + (long)s * n + i);
            const float4 p1 = *reinterpret_cast<const float4*>(parts + (long)(s + 1) * n + i);
            const float4 p2 = *reinterpret_cast<const float4*>(parts + (long)(s + 2) * n + i);
            const float4 p3 = *reinterpret_cast<const float4*>(parts + (long)(s + 3) * n + i);
            a.x += (p0.x + p1.x) + (p2.x + p3.x); a.y += (p0.y + p1.y) + (p2.y + p3.y);
            a.z += (p0.z + p1.z) + (p2.z + p3.z); a.w += (p0.w + p1.w) + (p2.w + p3.w);
        }
        for (; s < nparts; ++s) {
            const float4 p = *reinterpret_cast<const float4*>(parts + (long)s * n + i);
            a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        *reinterpret_cast<float4*>(out + i) = a;
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) st4(d + i, ld4(s + i));
        else for (long j = i; j < n; ++j) st_f(d + j, ld_f(s + j));
    }
}

// QuickGELU.forward (model/tfm_model.py:11-13) stand-alone; inside a block it is the c_fc GEMM's epilogue
template <typename T>
__global__ __launch_bounds__(256) void quickgelu_kernel(const T* __restrict__ s, T* __restrict__ d, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            float4 v = ld4(s + i);
            v.x = quick_gelu_t<T>(v.x); v.y = quick_gelu_t<T>(v.y); v.z = quick_gelu_t<T>(v.z); v.w = quick_gelu_t<T>(v.w);
            st4(d + i, v);
        } else {
            for (long j = i; j < n; ++j) st_f(d + j, quick_gelu_t<T>(ld_f(s + j)));
        }
    }
}

// binary_head (nn.Linear(512,1), tan_model.py:70,147-148): out[r] = <x[r], w> + b   (f32 out)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, long rows) {
    constexpr int C = NCH * 256;
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 v = ld4(x + r * C + c), ww = *reinterpret_cast<const float4*>(w + c);
        s += (v.x * ww.x + v.y * ww.y) + (v.z * ww.z + v.w * ww.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s + b[0];
}

// dx[r] (=|+=) dout[r] * w ; dw += sum_r dout[r] * x[r] ; db += sum_r dout[r]
template <typename T, int NCH>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ x,
                                                       const float* __restrict__ w, T* __restrict__ dx,
                                                       float* __restrict__ dw, float* __restrict__ db, long rows,
                                                       int accumulate_dx) {
    constexpr int C = NCH * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float4 acc[NCH];
    float accb = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = make_float4(0, 0, 0, 0);
    for (long r = (long)blockIdx.x * ROWS_PER_BLOCK + wv; r < rows; r += (long)gridDim.x * ROWS_PER_BLOCK) {
        const float g = dout[r];
        accb += g;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (i * 64 + lane) * 4;
            const float4 v = ld4(x + r * C + c), ww = *reinterpret_cast<const float4*>(w + c);
            acc[i].x += g * v.x; acc[i].y += g * v.y; acc[i].z += g * v.z; acc[i].w += g * v.w;
            float4 o = make_float4(g * ww.x, g * ww.y, g * ww.z, g * ww.w);
            if (accumulate_dx) {
                const float4 p = ld4(dx + r * C + c);
                o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
            }
            st4(dx + r * C + c, o);
        }
    }
    // the four waves meet in LDS first: ONE atomic per column and workgroup (every wave of 256 workgroups adding to the same 513
    // addresses -- 0.5 M device-scope atomics on 2 KiB -- took 103 us for 2 048 rows, on the stage-2 step's joint chain)
    __shared__ float red[4][C + 4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 4;
        red[wv][c] = acc[i].x; red[wv][c + 1] = acc[i].y; red[wv][c + 2] = acc[i].z; red[wv][c + 3] = acc[i].w;
    }
    if (lane == 0) red[wv][C] = accb;
    __syncthreads();
    for (int c = threadIdx.x; c <= C; c += 256) {
        const float v = red[0][c] + red[1][c] + red[2][c] + red[3][c];
        unsafeAtomicAdd(c < C ? dw + c : db, v);
    }
}

// linear interpolation of a [L_in, C] table to [L_out, C], align_corners=False (F.interpolate, tan_model.py:157-160)
__global__ void interp_kernel(const float* __restrict__ src, float* __restrict__ dst, int L_in, int L_out, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)L_out * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    const float scale = (float)L_in / (float)L_out;
    float pos = ((float)t + 0.5f) * scale - 0.5f;
    pos = fmaxf(pos, 0.0f);
    int i0 = min((int)floorf(pos), L_in - 1);
    const int i1 = min(i0 + 1, L_in - 1);
    const float w1 = pos - (float)i0;
    dst[i] = src[(long)i0 * C + c] * (1.0f - w1) + src[(long)i1 * C + c] * w1;
}

// transpose-add of the interpolation: dsrc[L_in, C] += W^T ddst[L_out, C]
__global__ void interp_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc, int L_in, int L_out, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)L_out * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    const float scale = (float)L_in / (float)L_out;
    float pos = fmaxf(((float)t + 0.5f) * scale - 0.5f, 0.0f);
    int i0 = min((int)floorf(pos), L_in - 1);
    const int i1 = min(i0 + 1, L_in - 1);
    const float w1 = pos - (float)i0;
    unsafeAtomicAdd(dsrc + (long)i0 * C + c, ddst[i] * (1.0f - w1));
    unsafeAtomicAdd(dsrc + (long)i1 * C + c, ddst[i] * w1);
}

#define DISPATCH_NCH(C, ...)                                   \
    switch (C) {                                               \
        case 256: { constexpr int NCH = 1; __VA_ARGS__; break; }  \
        case 512: { constexpr int NCH = 2; __VA_ARGS__; break; }  \
        case 1024: { constexpr int NCH = 4; __VA_ARGS__; break; } \
        default: return TAN_ERR_BAD_ARG;                       \
    }
#define DISPATCH_T(dtype, ...)                                        \
    if (dtype == TAN_F32) { typedef float T; __VA_ARGS__; }           \
    else if (dtype == TAN_BF16) { typedef bf16_t T; __VA_ARGS__; }    \
    else return TAN_ERR_BAD_ARG;

// all given pointers (NULL allowed) are 16-byte aligned
template <typename... P>
static inline bool aligned16(P... p) { return ((... | (uintptr_t)p) & 15) == 0; }

}  // namespace tal

using namespace tal;

extern "C" int tan_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 const void* add, int add_period, long rows, int C, float eps, int dtype, void* stream) {
    TAN_REQUIRE(x && gamma && beta && y && rows > 0);
    if (add) TAN_REQUIRE(add_period > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TAN_BF16 && C == 512 && aligned16(x, y, add, gamma, beta)) {
        constexpr int RPW = 2;
        hipLaunchKernelGGL((ln_fwd_bf16x8_kernel<RPW>), dim3(cdiv(rows, ROWS_PER_BLOCK * RPW)), dim3(256), 0, st, (const bf16_t*)x,
                           gamma, beta, (bf16_t*)y, mean, rstd, (const bf16_t*)add, add_period, rows, eps);
        TAN_LAUNCH_CHECK();
        return 0;
    }
    DISPATCH_T(dtype, DISPATCH_NCH(C, hipLaunchKernelGGL((ln_fwd_kernel<T, NCH>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0,
                                                         st, (const T*)x, gamma, beta, (T*)y, mean, rstd, (const T*)add,
                                                         add_period, rows, eps)));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" long tan_layernorm_bwd_ws_floats(int C) { return (long)LN_BWD_MAX_BLOCKS * 3 * C; }

extern "C" int tan_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, float* dgamma, float* dbeta, float* dx_colsum, float* ws, long rows,
                                 int C, int dtype, void* stream) {
    TAN_REQUIRE(dy && x && gamma && mean && rstd && dx && ws && rows > 0);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)min((long)LN_BWD_MAX_BLOCKS, (long)cdiv(rows, ROWS_PER_BLOCK));
    // bf16, C = 512: 256 blocks (one per CU) add their column sums straight into the gradients with f32 atomics -- 393 k atomics per
    // launch instead of a 6-MB partial table and a second kernel; measured inside the step: 7.12 vs 7.16 ms, and with 512 / 1024
    // blocks the contention on the 1536 addresses costs more than the finalize did (7.28 / 7.5 ms).
    constexpr int atomic_blocks = 256;
    if (dtype == TAN_BF16 && C == 512 && aligned16(dy, x, dres, dx, gamma) && atomic_blocks > 0) {
        const int nb = (int)min((long)atomic_blocks, (long)cdiv(rows, ROWS_PER_BLOCK));
        hipLaunchKernelGGL((ln_bwd_bf16x8_kernel<true, 4>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean,
                           rstd, (const bf16_t*)dres, (bf16_t*)dx, ws, rows, dgamma, dbeta, dx_colsum);
        TAN_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == TAN_BF16 && C == 512 && aligned16(dy, x, dres, dx, gamma)) {
        hipLaunchKernelGGL((ln_bwd_bf16x8_kernel<false, 2>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean,
                           rstd, (const bf16_t*)dres, (bf16_t*)dx, ws, rows, nullptr, nullptr, nullptr);
    } else {
        DISPATCH_T(dtype, DISPATCH_NCH(C, hipLaunchKernelGGL((ln_bwd_kernel<T, NCH>), dim3(nblk), dim3(256), 0, st, (const T*)dy,
                                                             (const T*)x, gamma, mean, rstd, (const T*)dres, (T*)dx, ws, rows)));
    }
    TAN_LAUNCH_CHECK();
    if (dgamma || dbeta || dx_colsum) {
        hipLaunchKernelGGL(ln_bwd_finalize, dim3(cdiv(3 * C, 64), LN_FIN_SLICES), dim3(256), 0, st, ws, nblk, C, dgamma, dbeta, dx_colsum);
        TAN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int tan_l2norm_fwd_multi(const tan_ptr8* xs, void* y, float* inv_norm, int nstage, long rows, int C, int grp,
                                    int src_grp_rows, int src_off, int dtype, void* stream) {
    TAN_REQUIRE(xs && y && nstage >= 1 && nstage <= 8 && rows > 0 && grp > 0);
    L2nPtrs ps{};
    for (int i = 0; i < nstage; ++i) { TAN_REQUIRE(xs->p[i]); ps.p[i] = xs->p[i]; }
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, DISPATCH_NCH(C, hipLaunchKernelGGL((l2n_fwd_kernel<T, NCH>), dim3(cdiv(rows, ROWS_PER_BLOCK), nstage), dim3(256), 0,
                                                         st, ps, (T*)y, inv_norm, rows, grp, src_grp_rows, src_off)));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_l2norm_fwd(const void* x, void* y, float* inv_norm, long rows, int C, int grp, int src_grp_rows,
                              int src_off, int dtype, void* stream) {
    TAN_REQUIRE(x);
    tan_ptr8 xs{};
    xs.p[0] = x;
    return tan_l2norm_fwd_multi(&xs, y, inv_norm, 1, rows, C, grp, src_grp_rows, src_off, dtype, stream);
}

extern "C" int tan_l2norm_bwd_multi(const void* dy, const void* y, const float* inv_norm, const tan_ptr8* dxs, int nstage, long rows,
                                    int C, int grp, int dst_grp_rows, int dst_off, int dtype, void* stream) {
    TAN_REQUIRE(dy && y && inv_norm && dxs && nstage >= 1 && nstage <= 8 && rows > 0 && grp > 0);
    L2nPtrs ps{};
    for (int i = 0; i < nstage; ++i) { TAN_REQUIRE(dxs->p[i]); ps.p[i] = dxs->p[i]; }
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, DISPATCH_NCH(C, hipLaunchKernelGGL((l2n_bwd_kernel<T, NCH>), dim3(cdiv(rows, ROWS_PER_BLOCK), nstage), dim3(256), 0,
                                                         st, (const T*)dy, (const T*)y, inv_norm, ps, rows, grp, dst_grp_rows, dst_off)));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, long rows, int C, int grp,
                              int dst_grp_rows, int dst_off, int dtype, void* stream) {
    TAN_REQUIRE(dx);
    tan_ptr8 ds{};
    ds.p[0] = dx;
    return tan_l2norm_bwd_multi(dy, y, inv_norm, &ds, 1, rows, C, grp, dst_grp_rows, dst_off, dtype, stream);
}

extern "C" int tan_colsum_acc(const void* x, float* out, long rows, int C, int dtype, void* stream) {
    TAN_REQUIRE(x && out && rows > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    if (C % 8 != 0 || ((uintptr_t)x % 16) != 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_generic_kernel<T>), dim3(cdiv(C, 256), cdiv(rows, 64)), dim3(256), 0, st,
                                             (const T*)x, out, rows, C));
        TAN_LAUNCH_CHECK();
        return 0;
    }
    int tpr = C / 8;                      // threads spanning one row
    if (tpr > 256) tpr = 256;
    while (256 % tpr) --tpr;              // must divide the block
    if (C % 512 == 0) tpr = 64;           // one wave = 1-KiB row segments, 4 row lanes: fewer, fatter blocks and 4x fewer atomics
    const int rpb = (C % 512 == 0 ? 16 : 8) * (256 / tpr);      // rows per thread group
    dim3 grid(cdiv(C, tpr * 8), cdiv(rows, rpb));
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, st, (const T*)x, out, rows, C, rpb, tpr));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_rows_copy(const void* src, void* dst, int G, int R, int C, long src_grp_rows, long src_off,
                             long dst_grp_rows, long dst_off, int accumulate, int dtype, void* stream) {
    TAN_REQUIRE(src && dst && G > 0 && R > 0 && C > 0 && C % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    const long n4 = (long)G * R * C / 4;
    const unsigned grid = (unsigned)min((long)4096, (long)cdiv(n4, 256));
    if (accumulate) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((rows_kernel<T, true>), dim3(grid), dim3(256), 0, st, (const T*)src, (T*)dst, G, R, C,
                                             src_grp_rows, src_off, dst_grp_rows, dst_off));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((rows_kernel<T, false>), dim3(grid), dim3(256), 0, st, (const T*)src, (T*)dst, G, R, C,
                                             src_grp_rows, src_off, dst_grp_rows, dst_off));
    }
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_group_sum(const void* x, void* out, int G, int R, int C, int dtype, void* stream) {
    TAN_REQUIRE(x && out && G > 0 && R > 0 && C > 0 && C % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((group_sum_kernel<T>), dim3(cdiv((long)R * C, 256)), dim3(256), 0, st, (const T*)x,
                                         (T*)out, G, R, C));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_transpose_batch(const void* src, void* dst, const long* table, int n, long max_rows, long max_cols, int dtype,
                                   void* stream) {
    TAN_REQUIRE(src && dst && table && n > 0 && max_rows > 0 && max_cols > 0 && dtype == TAN_BF16);
    TAN_REQUIRE(max_rows % 8 == 0 && max_cols % 8 == 0);          // every matrix: rows % 8 == 0 and cols % 8 == 0 (16-byte accesses)
    const unsigned tiles = cdiv(max_rows, 64) * cdiv(max_cols, 64);
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(tiles, n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst,
                       table);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_reduce_add(const float* parts, float* out, int nparts, long n, void* stream) {
    TAN_REQUIRE(parts && out && nparts > 0 && n > 0 && n % 4 == 0);
    const unsigned grid = (unsigned)min((long)2048, (long)cdiv(n, 1024));
    hipLaunchKernelGGL(reduce_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, parts, out, nparts, n);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream) {
    TAN_REQUIRE(src && dst && n > 0);
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)min((long)4096, (long)cdiv(n, 1024));
    if (src_dtype == TAN_F32 && dst_dtype == TAN_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == TAN_BF16 && dst_dtype == TAN_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == TAN_F32 && dst_dtype == TAN_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else if (src_dtype == TAN_BF16 && dst_dtype == TAN_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    else return TAN_ERR_BAD_ARG;
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_quickgelu(const void* x, void* y, long n, int dtype, void* stream) {
    TAN_REQUIRE(x && y && n > 0);
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)min((long)4096, (long)cdiv(n, 1024));
    if (dtype == TAN_F32) hipLaunchKernelGGL((quickgelu_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, n);
    else if (dtype == TAN_BF16) hipLaunchKernelGGL((quickgelu_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n);
    else return TAN_ERR_BAD_ARG;
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_head_fwd(const void* x, const float* w, const float* b, float* out, long rows, int C, int dtype,
                            void* stream) {
    TAN_REQUIRE(x && w && b && out && rows > 0);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, DISPATCH_NCH(C, hipLaunchKernelGGL((head_fwd_kernel<T, NCH>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256),
                                                         0, st, (const T*)x, w, b, out, rows)));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_head_bwd(const float* dout, const void* x, const float* w, void* dx, float* dw, float* db, long rows,
                            int C, int accumulate_dx, int dtype, void* stream) {
    TAN_REQUIRE(dout && x && w && dx && dw && db && rows > 0);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)min((long)256, (long)cdiv(rows, 4 * ROWS_PER_BLOCK));  // (>= 4 rows per wave; one atomic per column and workgroup)
    DISPATCH_T(dtype, DISPATCH_NCH(C, hipLaunchKernelGGL((head_bwd_kernel<T, NCH>), dim3(nblk), dim3(256), 0, st, dout,
                                                         (const T*)x, w, (T*)dx, dw, db, rows, accumulate_dx)));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_interp_linear(const float* src, float* dst, int L_in, int L_out, int C, void* stream) {
    TAN_REQUIRE(src && dst && L_in > 0 && L_out > 0 && C > 0);
    hipLaunchKernelGGL(interp_kernel, dim3(cdiv((long)L_out * C, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, L_in,
                       L_out, C);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_interp_linear_bwd(const float* ddst, float* dsrc, int L_in, int L_out, int C, void* stream) {
    TAN_REQUIRE(ddst && dsrc && L_in > 0 && L_out > 0 && C > 0);
    hipLaunchKernelGGL(interp_bwd_kernel, dim3(cdiv((long)L_out * C, 256)), dim3(256), 0, (hipStream_t)stream, ddst, dsrc,
                       L_in, L_out, C);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_rows_gather(const void* src, void* dst, const int* map, int S, long Msrc, long Mdst, int C, int dtype, void* stream) {
    TAN_REQUIRE(src && dst && map && S > 0 && Msrc > 0 && Mdst > 0 && C > 0 && C % 4 == 0);
    const long n4 = (long)S * Mdst * C / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    if (dtype == TAN_F32) hipLaunchKernelGGL((rows_gather_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, map, S, Msrc, Mdst, C);
    else if (dtype == TAN_BF16) hipLaunchKernelGGL((rows_gather_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, map, S, Msrc, Mdst, C);
    else return TAN_ERR_BAD_ARG;
    TAN_LAUNCH_CHECK();
    return 0;
}
