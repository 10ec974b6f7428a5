// Input embeddings of TemporalAligner in ONE launch (gfx950, bf16 throughput mode): for each modality
//     proj = a W^T                                  video_pre_proj / text_pre_proj      (model/tan_model.py:48-49,155,231)
//     y    = LayerNorm(proj; ln_video_init | ln_text_init)                              (tan_model.py:50-51,155,231)
//     out_d = y + LN_position_init(pos)[p_d + t]    d = 0,1: the dual path's and the joint path's position offsets
//                                                    (tan_model.py:162-167 and again 194-199; 224-228 for the text positions)
//     xn1_d = LayerNorm(out_d; block 0's ln_1 of the stack that consumes out_d)         (model/tfm_model.py:35, first block)
// where a is the f32 (or bf16) feature tensor as the caller hands it over: the cast, the GEMM, both LayerNorms' launches, the
// position add for both offsets, the torch.cat that builds the joint stack's input (tan_model.py:201-203: out_1 is written straight
// into rows b*(T+N) + t of the joint input) and the first block's ln_1 of both stacks were ~14 dependent launches in front of the
// first encoder kernel (0.43-0.55 ms of a 4.7 ms step).
//
// One workgroup (8 waves) owns 32 rows: the rows go HBM -> registers -> bf16 -> a swizzled LDS panel [32][K] (and, for the weight
// gradient of the backward, to HBM as bf16); the weight streams from L2 straight into MFMA A-operand registers through the
// fragment-major image tan_pack_weights builds (TN = 512, TK = 16: a k step is 16 fragments of 1 KiB, wave w owns fragments 2w,
// 2w + 1 = output features 64w .. 64w + 63); swapped orientation D[feature][row] as in tan_panel.h, so a lane ends with 2 x 16
// consecutive features of ONE row.  The bf16 results are parked in LDS (over the input panel) and every wave finishes four rows the way
// ln_fwd_kernel does (one wave per row, 8 features per lane, whole-row 1-KiB stores).
#include "tan_panel.h"

namespace tal {

struct EmbedProb {
    const void* a; int a_f32; long rows; int K; int T;
    const char* pw;
    const float *g, *b;
    bf16_t* a_bf16;
    bf16_t* proj; float *mean, *rstd;
    bf16_t* out[2]; long grp[2], off[2]; const float* pos[2];
    const float *ln1_g[2], *ln1_b[2]; bf16_t* xn1[2]; float *mean1[2], *rstd1[2];
    const unsigned char* pad_src; unsigned char* pad_dst; long pad_grp, pad_off;
    int blk0;
};
struct EmbedArgs { EmbedProb p0, p1; int nprob; float eps; };

constexpr int EMB_ROWS = 32, EMB_PF = 4;

__device__ __forceinline__ char* emb_slot(char* panel, int rowb, int row, int chunk) {
    return panel + row * rowb + ((chunk ^ (row & 15)) << 4);
}

__global__ __launch_bounds__(512, 4) void embed_fwd_kernel(const EmbedArgs A) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // (a block-uniform select between two kernel-argument structs: scalar loads + s_cselect, nothing goes to scratch)
    const EmbedProb P = (A.nprob > 1 && (int)blockIdx.x >= A.p1.blk0) ? A.p1 : A.p0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)((int)blockIdx.x - P.blk0) * EMB_ROWS;
    const int K = P.K, rowb = K * 2, cpr = K >> 3;                  // 16-byte chunks per row
    // ---- phase 1: the 32 input rows -> bf16 LDS panel (+ bf16 copy in HBM)
    for (int id = tid; id < EMB_ROWS * cpr; id += 512) {
        const int r = id / cpr, c = id - r * cpr;
        const long row = row0 + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < P.rows) {
            if (P.a_f32) {
                const f8 x = ld8f(reinterpret_cast<const float*>(P.a) + row * K + c * 8);
                v.x = f2bf2(x.v[0], x.v[1]); v.y = f2bf2(x.v[2], x.v[3]); v.z = f2bf2(x.v[4], x.v[5]); v.w = f2bf2(x.v[6], x.v[7]);
                if (P.a_bf16) *reinterpret_cast<uint4*>(P.a_bf16 + row * K + c * 8) = v;
            } else {
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(P.a) + row * K + c * 8);
            }
        }
        *reinterpret_cast<uint4*>(emb_slot(lds, rowb, r, c)) = v;
    }
    if (P.pad_dst && tid < EMB_ROWS && row0 + tid < P.rows) {          // key-padding flags of these rows into the joint stack's [B, L] mask
        const long row = row0 + tid;
        P.pad_dst[(row / P.T) * P.pad_grp + P.pad_off + row % P.T] = P.pad_src ? P.pad_src[row] : (unsigned char)0;
    }
    // ---- phase 2: D[feature][row] += W[feature][k] X[row][k]; the weight fragments of EMB_PF k steps are in flight
    f32x16 acc[2];
    acc_zero(acc[0]); acc_zero(acc[1]);
    const char* wp = P.pw + (2 * wave) * 1024 + lane * 16;
    const int KS = K >> 4;
    bf16x8 wf[EMB_PF][2];
#pragma unroll
    for (int s = 0; s < EMB_PF; ++s) {
        wf[s][0] = *reinterpret_cast<const bf16x8*>(wp + (long)s * 16384);
        wf[s][1] = *reinterpret_cast<const bf16x8*>(wp + (long)s * 16384 + 1024);
    }
    __syncthreads();
    const int m = lane & 31, hi = lane >> 5;
    const char* xrow = lds + m * rowb;
    for (int ks0 = 0; ks0 < KS; ks0 += EMB_PF) {
#pragma unroll
        for (int s = 0; s < EMB_PF; ++s) {
            const int ks = ks0 + s, c = 2 * ks + hi;
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(xrow + ((c ^ (m & 15)) << 4));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][0], x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][1], x, acc[1], 0, 0, 0);
            const int kn = ks + EMB_PF < KS ? ks + EMB_PF : ks;          // (the last EMB_PF steps reload their own tile: no branch around a load)
            wf[s][0] = *reinterpret_cast<const bf16x8*>(wp + (long)kn * 16384);
            wf[s][1] = *reinterpret_cast<const bf16x8*>(wp + (long)kn * 16384 + 1024);
        }
    }
    __syncthreads();                    // every wave is done reading the input panel: the result panel [32][512] bf16 goes over it
    // ---- phase 3: accumulators -> LDS: lane (m, hi), tile j holds features 64 wave + 32 j + 16 hi + r of row m
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { v0[e] = acc[j][e]; v1[e] = acc[j][8 + e]; }
        const int chunk = (64 * wave + 32 * j + 16 * hi) >> 3;
        *reinterpret_cast<uint4*>(emb_slot(lds, 1024, m, chunk)) = pn_pack8(v0);
        *reinterpret_cast<uint4*>(emb_slot(lds, 1024, m, chunk + 1)) = pn_pack8(v1);
    }
    __syncthreads();
    // ---- phase 4: LayerNorm + position terms + the first blocks' ln_1, one wave per row, lane = features 8 lane .. 8 lane + 7
    const int c8 = lane * 8;
    float gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = P.g[c8 + e]; bt[e] = P.b[c8 + e]; }
    constexpr float invC = 1.0f / 512.0f;
#pragma unroll 1
    for (int i = 0; i < EMB_ROWS / 8; ++i) {
        const int r = wave * (EMB_ROWS / 8) + i;
        const long row = row0 + r;
        if (row >= P.rows) break;
        const uint4 pv = *reinterpret_cast<const uint4*>(emb_slot(lds, 1024, r, lane));
        float x[8];
        pn_unpack8(pv, x);
        if (P.proj) *reinterpret_cast<uint4*>(P.proj + row * 512 + c8) = pv;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[e];
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] -= mean; q += x[e] * x[e]; }
        const float rstd = rsqrtf(wave_sum(q) * invC + A.eps);
        if (lane == 0) {
            if (P.mean) P.mean[row] = mean;
            if (P.rstd) P.rstd[row] = rstd;
        }
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = x[e] * rstd * gm[e] + bt[e];
        const long vid = row / P.T;
        const int t = (int)(row - vid * P.T);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (!P.out[d]) continue;
            float o[8];
            if (P.pos[d]) {
                const f8 pp = ld8f(P.pos[d] + (long)t * 512 + c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = y[e] + pp.v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = y[e];
            }
            const long orow = vid * P.grp[d] + P.off[d] + t;
            const uint4 ov = pn_pack8(o);
            *reinterpret_cast<uint4*>(P.out[d] + orow * 512 + c8) = ov;
            if (P.xn1[d]) {                       // the consuming stack's first ln_1, on the values as stored (bf16)
                float z[8];
                pn_unpack8(ov, z);
                float s1 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s1 += z[e];
                const float m1 = wave_sum(s1) * invC;
                float q1 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { z[e] -= m1; q1 += z[e] * z[e]; }
                const float r1 = rsqrtf(wave_sum(q1) * invC + A.eps);
                if (lane == 0) { P.mean1[d][orow] = m1; P.rstd1[d][orow] = r1; }
                const f8 g1 = ld8f(P.ln1_g[d] + c8), b1 = ld8f(P.ln1_b[d] + c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = z[e] * r1 * g1.v[e] + b1.v[e];
                *reinterpret_cast<uint4*>(P.xn1[d] + orow * 512 + c8) = pn_pack8(z);
            }
        }
    }
}

}  // namespace tal

using namespace tal;

extern "C" int tan_embed_fwd(const tan_embed_desc* d, int nprob, void* stream) {
    TAN_REQUIRE(d && nprob >= 1 && nprob <= 2);
    EmbedArgs A{};
    A.nprob = nprob;
    A.eps = 1e-5f;
    int blk = 0, maxK = 0;
    double work = 0;
    for (int i = 0; i < nprob; ++i) {
        const tan_embed_desc& s = d[i];
        TAN_REQUIRE(s.a && s.pw && s.ln_g && s.ln_b && s.rows > 0 && s.T > 0 && s.C == 512);
        TAN_REQUIRE(s.K >= 128 && s.K % 128 == 0 && s.K <= 2048);      // whole groups of 16 chunks per row (the XOR swizzle)
        EmbedProb& p = i ? A.p1 : A.p0;
        p.a = s.a; p.a_f32 = s.a_dtype == TAN_F32; p.rows = s.rows; p.K = s.K; p.T = s.T;
        p.pw = (const char*)s.pw; p.g = s.ln_g; p.b = s.ln_b;
        p.a_bf16 = (bf16_t*)s.a_bf16; p.proj = (bf16_t*)s.proj; p.mean = s.mean; p.rstd = s.rstd;
        for (int k = 0; k < 2; ++k) {
            p.out[k] = (bf16_t*)s.out[k]; p.grp[k] = s.out_grp_rows[k]; p.off[k] = s.out_off[k]; p.pos[k] = s.pos[k];
            TAN_REQUIRE(!s.out[k] || s.out_grp_rows[k] >= s.T + s.out_off[k]);
            p.ln1_g[k] = s.ln1_g[k]; p.ln1_b[k] = s.ln1_b[k]; p.xn1[k] = (bf16_t*)s.xn1[k]; p.mean1[k] = s.mean1[k]; p.rstd1[k] = s.rstd1[k];
            TAN_REQUIRE(!s.xn1[k] || (s.out[k] && s.ln1_g[k] && s.ln1_b[k] && s.mean1[k] && s.rstd1[k]));
        }
        p.pad_src = s.pad_src; p.pad_dst = s.pad_dst; p.pad_grp = s.pad_grp_rows; p.pad_off = s.pad_off;
        p.blk0 = blk;
        blk += (int)cdiv(s.rows, EMB_ROWS);
        if (s.K > maxK) maxK = s.K;
        work += 2.0 * (double)s.rows * 512.0 * s.K;
    }
    const int lds_bytes = EMB_ROWS * (maxK > 512 ? maxK : 512) * 2;
    static std::atomic<unsigned long long> lds_done{0};
    if (ensure_dyn_lds((const void*)embed_fwd_kernel, 160 * 1024, lds_done) != hipSuccess) return -3;
    const int rec = prof_begin((hipStream_t)stream, TAN_PROF_GEMM_BF16, work);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(blk), dim3(512), lds_bytes, (hipStream_t)stream, A);
    prof_end((hipStream_t)stream, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the same front-end in two launches (+ the two weight-gradient GEMMs the caller issues):
//   embed_bwd_kernel   dy = d_out[0] + d_out[1] (the two uses of the embedding: dual and joint path, read where the stacks'
//                      backward left them -- no row copies), d_proj = LayerNorm-backward(dy; proj, mean, rstd, ln_g) -> bf16 (operand
//                      of the pre-projection's weight gradient), g_ln_g / g_ln_b += column sums, d_pos[d][t] += sum over videos of
//                      d_out[d] (the broadcast position add's backward), f32 atomics
//   pos_ln_bwd_kernel  ln_position_init's backward on the (up to three) used slices of the position tables: table gradient rows and
//                      the LayerNorm's own g / b gradients, f32 atomics (the dual and the joint offset overlap in the table)
// replaces ~25 small launches at the very end of backward (rows_copy x2-3, group_sum x2, cast x2, ln_bwd + finalize x2, rows_copy
// accumulate x3, ln_bwd x2), 130 us in front of the optimizer.  Autograd of tan_model.py:155-167, 187-199, 212-234.
namespace tal {

struct EmbBwdProb {
    long rows; int T, nvid;
    const bf16_t* dout[2]; long grp[2], off[2];
    const bf16_t* proj; const float *mean, *rstd, *g;
    bf16_t* dproj; float *g_g, *g_b; float* dpos[2];
    int blk0, tblocks;
};
struct EmbBwdArgs { EmbBwdProb p0, p1; int nprob; };
constexpr int EB_VG = 8;        // videos per workgroup: a wave owns one position t and its rows of EB_VG videos, all loads issued up front

__global__ __launch_bounds__(512) void embed_bwd_kernel(const EmbBwdArgs A) {
    __shared__ float red[2][8][512];
    const EmbBwdProb P = (A.nprob > 1 && (int)blockIdx.x >= A.p1.blk0) ? A.p1 : A.p0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c8 = lane * 8;
    const int bl = (int)blockIdx.x - P.blk0, tb = bl % P.tblocks, vg = bl / P.tblocks;
    const int t = tb * 8 + wave;
    float gg[8], gb[8], dp0[8], dp1[8], gm[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gg[e] = gb[e] = dp0[e] = dp1[e] = 0.f; gm[e] = P.g[c8 + e]; }
    if (t < P.T) {
        constexpr float invC = 1.0f / 512.0f;
        uint4 r0[EB_VG], r1[EB_VG], rx[EB_VG];
        float mu[EB_VG], rs[EB_VG];
#pragma unroll
        for (int i = 0; i < EB_VG; ++i) {              // one memory round trip for the whole group (rows past the batch: zeros)
            const int v = vg * EB_VG + i;
            const bool ok = v < P.nvid;
            const long row = (long)(ok ? v : 0) * P.T + t;
            r0[i] = r1[i] = make_uint4(0, 0, 0, 0);
            if (ok && P.dout[0]) r0[i] = *reinterpret_cast<const uint4*>(P.dout[0] + ((long)v * P.grp[0] + P.off[0] + t) * 512 + c8);
            if (ok && P.dout[1]) r1[i] = *reinterpret_cast<const uint4*>(P.dout[1] + ((long)v * P.grp[1] + P.off[1] + t) * 512 + c8);
            rx[i] = *reinterpret_cast<const uint4*>(P.proj + row * 512 + c8);
            mu[i] = P.mean[row]; rs[i] = P.rstd[row];
        }
#pragma unroll
        for (int i = 0; i < EB_VG; ++i) {
            const int v = vg * EB_VG + i;
            if (v >= P.nvid) break;
            const long row = (long)v * P.T + t;
            float d0[8], d1[8], x[8];
            pn_unpack8(r0[i], d0); pn_unpack8(r1[i], d1); pn_unpack8(rx[i], x);
            float dy[8], xh[8], g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dy[e] = d0[e] + d1[e];
                xh[e] = (x[e] - mu[i]) * rs[i];
                g[e] = dy[e] * gm[e];
                s1 += g[e]; s2 += g[e] * xh[e];
                dp0[e] += d0[e]; dp1[e] += d1[e];
                gg[e] += dy[e] * xh[e]; gb[e] += dy[e];
            }
            const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs[i] * (g[e] - m1 - xh[e] * m2);
            *reinterpret_cast<uint4*>(P.dproj + row * 512 + c8) = pn_pack8(o);
        }
        // position-row sums of this video group: plain stores into the group's partial plane (tan_pos_ln_bwd adds the planes)
        if (P.dpos[0]) { float* d = P.dpos[0] + ((long)vg * P.T + t) * 512 + c8; st4(d, make_float4(dp0[0], dp0[1], dp0[2], dp0[3])); st4(d + 4, make_float4(dp0[4], dp0[5], dp0[6], dp0[7])); }
        if (P.dpos[1]) { float* d = P.dpos[1] + ((long)vg * P.T + t) * 512 + c8; st4(d, make_float4(dp1[0], dp1[1], dp1[2], dp1[3])); st4(d + 4, make_float4(dp1[4], dp1[5], dp1[6], dp1[7])); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][wave][c8 + e] = gg[e]; red[1][wave][c8 + e] = gb[e]; }
    __syncthreads();
    {
        const int c = threadIdx.x;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { a += red[0][w][c]; b += red[1][w][c]; }
        atomicAdd(P.g_g + c, a);
        atomicAdd(P.g_b + c, b);
    }
}

struct PosBwdUse { const float *dpos, *x, *mean, *rstd; float* g_table; int n, nparts; };
struct PosBwdArgs { PosBwdUse u[3]; int nuse; const float* gamma; float *g_g, *g_b; };
constexpr int PB_ROWS = 4;      // rows per workgroup (one per wave): one atomic per column and workgroup for the LayerNorm's own gradients

__global__ __launch_bounds__(256) void pos_ln_bwd_kernel(const PosBwdArgs A) {
    __shared__ float red[2][4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c8 = lane * 8;
    const f8 gm = ld8f(A.gamma + c8);
    float gg[8], gb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gg[e] = gb[e] = 0.f;
    constexpr float invC = 1.0f / 512.0f;
    for (int i = 0; i < PB_ROWS / 4; ++i) {
        int r = (int)blockIdx.x * PB_ROWS + i * 4 + wave;
        PosBwdUse U = A.u[0];
        if (r >= U.n && A.nuse > 1) { r -= U.n; U = A.u[1]; if (r >= U.n && A.nuse > 2) { r -= U.n; U = A.u[2]; } }
        if (r >= U.n) continue;
        float dy[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dy[e] = 0.f;
        for (int p0 = 0; p0 < U.nparts; p0 += 8) {     // the video groups' partial planes [nparts][n][512], eight loads in flight
            f8 q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = ld8f(U.dpos + ((long)min(p0 + j, U.nparts - 1) * U.n + r) * 512 + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = p0 + j < U.nparts ? 1.0f : 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dy[e] += w * q[j].v[e];
            }
        }
        const f8 x = ld8f(U.x + (long)r * 512 + c8);
        const float mean = U.mean[r], rstd = U.rstd[r];
        float xh[8], g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[e] = (x.v[e] - mean) * rstd; g[e] = dy[e] * gm.v[e]; s1 += g[e]; s2 += g[e] * xh[e]; gg[e] += dy[e] * xh[e]; gb[e] += dy[e]; }
        const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
        if (U.g_table) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(U.g_table + (long)r * 512 + c8 + e, rstd * (g[e] - m1 - xh[e] * m2));
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][wave][c8 + e] = gg[e]; red[1][wave][c8 + e] = gb[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        atomicAdd(A.g_g + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(A.g_b + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

}  // namespace tal

extern "C" int tan_embed_bwd_group(void) { return EB_VG; }

extern "C" int tan_embed_bwd(const tan_embed_bwd_desc* d, int nprob, void* stream) {
    TAN_REQUIRE(d && nprob >= 1 && nprob <= 2);
    EmbBwdArgs A{};
    A.nprob = nprob;
    int blk = 0;
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < nprob; ++i) {
        const tan_embed_bwd_desc& s = d[i];
        TAN_REQUIRE(s.rows > 0 && s.T > 0 && s.rows % s.T == 0 && s.C == 512 && s.proj && s.mean && s.rstd && s.ln_g && s.d_proj && s.g_ln_g && s.g_ln_b);
        TAN_REQUIRE(s.d_out[0] || s.d_out[1]);
        EmbBwdProb& p = i ? A.p1 : A.p0;
        p.rows = s.rows; p.T = s.T; p.nvid = (int)(s.rows / s.T);
        for (int k = 0; k < 2; ++k) {
            p.dout[k] = (const bf16_t*)s.d_out[k]; p.grp[k] = s.d_out_grp_rows[k]; p.off[k] = s.d_out_off[k]; p.dpos[k] = s.d_out[k] ? s.d_pos[k] : nullptr;
        }
        p.proj = (const bf16_t*)s.proj; p.mean = s.mean; p.rstd = s.rstd; p.g = s.ln_g;
        p.dproj = (bf16_t*)s.d_proj; p.g_g = s.g_ln_g; p.g_b = s.g_ln_b;
        p.blk0 = blk; p.tblocks = (s.T + 7) / 8;
        blk += p.tblocks * ((p.nvid + EB_VG - 1) / EB_VG);
    }
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(blk), dim3(512), 0, st, A);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_pos_ln_bwd(const tan_pos_ln_bwd_use* u, int nuse, const float* gamma, float* g_gamma, float* g_beta, int C, void* stream) {
    TAN_REQUIRE(u && nuse >= 1 && nuse <= 3 && gamma && g_gamma && g_beta && C == 512);
    PosBwdArgs A{};
    A.nuse = nuse; A.gamma = gamma; A.g_g = g_gamma; A.g_b = g_beta;
    int rows = 0;
    for (int i = 0; i < nuse; ++i) {
        TAN_REQUIRE(u[i].d_pos && u[i].x && u[i].mean && u[i].rstd && u[i].n > 0);
        TAN_REQUIRE(u[i].nparts >= 1);
        A.u[i] = PosBwdUse{u[i].d_pos, u[i].x, u[i].mean, u[i].rstd, u[i].g_table, u[i].n, u[i].nparts};
        rows += u[i].n;
    }
    hipLaunchKernelGGL(pos_ln_bwd_kernel, dim3(cdiv(rows, PB_ROWS)), dim3(256), 0, (hipStream_t)stream, A);
    TAN_LAUNCH_CHECK();
    return 0;
}
