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
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)embed_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const int rec = prof_begin((hipStream_t)stream, TAN_PROF_GEMM_BF16, work);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(blk), dim3(512), lds_bytes, (hipStream_t)stream, A);
    prof_end((hipStream_t)stream, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}
