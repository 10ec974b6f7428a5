// Row-panel MLP forward with ONE VIDEO PER WORKGROUP (up to 80 rows): the MLP branch of a block (model/tfm_model.py:23-27,37) --
// LN2, c_fc + QuickGELU, c_proj + residual and the LayerNorm that consumes the block's output -- for stacks whose sequences are
// 48 < L <= 80 rows (the shapes tan_attnblk_fwd takes: the joint stack of the headline configuration, L = 64 + 16, and the video
// stack, L = 64).
//
// Why a second kernel next to tan_panel.hip's 64-row panels (round 4): at B = 128 the joint stack is 10240 rows = 160 panels of 64,
// the video stack 128, and 288 workgroups that each own a whole CU (160 KiB LDS) do not fit 256 CUs side by side: whichever of the two
// stacks' MLP launches arrives second waits with 32 workgroups for the first CUs to come free -- the joint stack's MLP launches are
// bimodal in the step's trace (71-83 us alone, 118-130 us behind the video stack's), and the step has a 0.15 ms step between B = 112
// (252 workgroups) and B = 120 (270) on top of its 0.027 ms per video (profiles/r04_batch_sweep.txt).  One workgroup per video makes
// it 128 + 128.  It is also the better shape: the kernels are bound by the L2 -> CU weight stream (~48 B/clk), and 80 rows are 25 % more
// FLOPs per streamed byte than 64.
//
// 80 rows do not fit the 64-row kernel's LDS plan (80 KiB panel + 2 x 40 KiB hidden chunk + staging > 160 KiB), so the hidden dimension
// is walked in 16 chunks of 128 features instead of 8 of 256, and the tiles are v_mfma_f32_16x16x32_bf16 (80 = 5 row blocks of 16, no
// padding work; 16-feature weight fragments, the "frag16" format of tan_pack_weights):
//   c_fc(c):   wave w owns hidden features c*128 + 16 w .. + 15 for all rows; K = 512 in 16 steps of 32                5 MFMAs / step
//   epilogue:  + bias, QuickGELU -> bf16 activation chunk [rows][128] in LDS (double-buffered), pre-activation staged next to it
//   c_proj(c): wave w owns output features 64 w .. + 63 (4 blocks of 16) for all rows; K = 128 in 4 steps of 32        20 MFMAs / step
// one barrier pair per chunk; the weights stream from L2 straight into registers through two rings (c_fc: 8 steps ahead, c_proj: 2)
// that run across the chunks; the side outputs (h_act from the activation buffer, h_pre from its staging panel) leave as 256-byte row
// pieces under the c_proj steps.
// ---- what this file needs outside itself (removed from the library with it) ----
// include/tan_hip.h:   int tan_mlp80_supported(int L, int C, int FF, int dtype);   int tan_mlp80_fwd(const tan_mlp_desc* d, int L, void* stream);
// pack_tiles_kernel (tan_panel.hip), in front of the `e.TN == 384` branch -- the "frag16" tile format (TN = 128 | 512, TK = 32):
//        if (e.TK == 32 && (e.TN == 128 || e.TN == 512)) {
//            for (int s = threadIdx.x; s < slots; s += blockDim.x) {
//                const int lane = s & 63, g = s >> 6;
//                const int row = nb * e.TN + 16 * g + (lane & 15), k = kt * 32 + 8 * (lane >> 4);
//                *reinterpret_cast<uint4*>(d0 + (long)s * 8) = *reinterpret_cast<const uint4*>(src + e.src_off + (long)row * e.K + k);
//            }
//            continue;
//        }
#include "tan_panel.h"

namespace tal {

typedef float f32x4_m __attribute__((ext_vector_type(4)));

struct Mlp80Args {
    const bf16_t* x_mid; const float* ln_g; const float* ln_b;
    const char* pw_fc;          // frag16 image of c_fc.weight [2048][512]   (TN = 128, TK = 32)
    const char* pw_proj;        // frag16 image of c_proj.weight [512][2048] (TN = 512, TK = 32)
    const float* b_fc; const float* b_proj;
    bf16_t* xn2; float* mean2; float* rstd2;
    bf16_t* h_pre; bf16_t* h_act; bf16_t* x_out;
    const float* nln_g; const float* nln_b; bf16_t* xn_next; float* nmean; float* nrstd;
    float eps;
    int L;                      // rows per video (workgroup)
};

constexpr int M80_DF = 8, M80_DP = 4;          // ring depths: c_fc steps in flight / c_proj steps (a whole chunk: requested under the c_fc phase)

__device__ __forceinline__ bf16x8 m80_wfc(const char* pw, int c, int ks, int wave, int lane) {
    return *reinterpret_cast<const bf16x8*>(pw + ((long)(c * 16 + ks) * 8 + wave) * 1024 + lane * 16);
}
struct M80WP { bf16x8 f[4]; };
__device__ __forceinline__ void m80_wproj(M80WP& W, const char* pw, int c, int kk, int wave, int lane) {
    const char* p = pw + ((long)(c * 4 + kk) * 32 + wave * 4) * 1024 + lane * 16;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) W.f[fb] = *reinterpret_cast<const bf16x8*>(p + fb * 1024);
}

template <int NRB>
__global__ __launch_bounds__(512, 2) void mlp80_fwd_kernel(Mlp80Args a) {
    constexpr int XROWS = 16 * NRB, RPW = XROWS / 8;                 // rows per workgroup / per wave in the row-wise phases (8 | 10)
    constexpr int XN_OFF = 0, H_OFF = XROWS * 1024, HB = XROWS * 256, PRE_OFF = H_OFF + 2 * HB, LDS_B = PRE_OFF + HB;
    static_assert(LDS_B <= 163840, "LDS budget");
    __shared__ __attribute__((aligned(1024))) char lds[LDS_B];
    typedef __attribute__((address_space(3))) const bf16x8* lds_frag_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    const long row0 = (long)blockIdx.x * L;
    const bool saving = a.h_pre != nullptr;
    const char* const pfc = a.pw_fc;
    const char* const ppj = a.pw_proj;

    // the weight streams do not depend on the activations: start them first
    bf16x8 WF[M80_DF];
    M80WP WP[M80_DP];
#pragma unroll
    for (int j = 0; j < M80_DF; ++j) WF[j] = m80_wfc(pfc, 0, j, wave, lane);
#pragma unroll
    for (int j = 0; j < M80_DP; ++j) m80_wproj(WP[j], ppj, 0, j, wave, lane);

    // ---- prologue: LN2 of the video's rows, one wave per row, lane = 8 features (tan_norm.hip's arithmetic); rows >= L repeat the
    // last row (finite, never stored)
    {
        const f8 g = ld8f(a.ln_g + lane * 8), b = ld8f(a.ln_b + lane * 8);
        f8 v[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) v[r] = ld8(a.x_mid + (row0 + min(wave * RPW + r, L - 1)) * 512 + lane * 8);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int m = wave * RPW + r;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[r].v[j];
            const float mean = wave_sum(s) * (1.0f / 512);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[r].v[j] -= mean; q += v[r].v[j] * v[r].v[j]; }
            const float rstd = rsqrtf(wave_sum(q) * (1.0f / 512) + a.eps);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[r].v[j] * rstd * g.v[j] + b.v[j];
            const uint4 u = pn_pack8(o);
            if (m < L) {
                if (a.xn2) *reinterpret_cast<uint4*>(a.xn2 + (row0 + m) * 512 + lane * 8) = u;
                if (lane == 0 && a.mean2) { a.mean2[row0 + m] = mean; a.rstd2[row0 + m] = rstd; }
            }
            *reinterpret_cast<uint4*>(pn_panel_slot<1024>(lds + XN_OFF, m, lane)) = u;
        }
    }
    __syncthreads();

    f32x4_m acc_o[4][NRB];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc_o[fb][rb] = f32x4_m{0.f, 0.f, 0.f, 0.f};

    // side outputs: [XROWS][256 B] LDS panels -> 256-byte row pieces of the [rows][2048] tensors, 4 rows per wave-instruction
    constexpr int NCI = (RPW + 3) / 4;

#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
        int ln = lane;                       // (an opaque copy per chunk: keeps lane-derived addresses from being hoisted out of the loop and spilled)
        asm volatile("" : "+v"(ln));
        const int r16 = ln & 15, q = ln >> 4;
        const int hb = c & 1;
        char* const Hc = lds + H_OFF + hb * HB;
        // ---- c_fc(c): acc_h[rb] = bias + xn2[rb] W_fc[chunk]^T, K = 512
        f32x4_m acc_h[NRB];
        {
            const float4 bv = *reinterpret_cast<const float4*>(a.b_fc + c * 128 + 16 * wave + 4 * q);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc_h[rb] = f32x4_m{bv.x, bv.y, bv.z, bv.w};
        }
        const unsigned xbase = (unsigned)(uintptr_t)(lds + XN_OFF + r16 * 1024);
        bf16x8 X[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) X[rb] = *(lds_frag_t)(uintptr_t)(xbase + rb * 16384 + ((q ^ r16) << 4));
        __builtin_amdgcn_sched_barrier(0);
        pn_static_for<0, 16>([&](auto jc) {
            constexpr int KS = decltype(jc)::value;
            bf16x8& W = WF[KS % M80_DF];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                acc_h[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, X[rb], acc_h[rb], 0, 0, 0);
                if constexpr (KS < 15) X[rb] = *(lds_frag_t)(uintptr_t)(xbase + rb * 16384 + ((((KS + 1) * 4 + q) ^ r16) << 4));
            }
            // the ring runs across chunks: step KS + 8 of this chunk, or step KS - 8 of the next one (past the end: reloads the last chunk)
            if constexpr (KS + M80_DF < 16) W = m80_wfc(pfc, c, KS + M80_DF, wave, ln);
            else W = m80_wfc(pfc, min(c + 1, 15), KS + M80_DF - 16, wave, ln);
            // this chunk's c_proj weights (4 steps x 4 KiB per wave): one step's fragments every fourth c_fc step (chunk 0: before the loop)
            if constexpr ((KS & 3) == 1) { if (c > 0) m80_wproj(WP[KS >> 2], ppj, c, KS >> 2, wave, ln); }
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- chunk epilogue: QuickGELU, bf16; activation -> Hc (the c_proj operand and h_act), pre-activation -> its staging panel
        {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                float x[4], g[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = acc_h[rb][e]; g[e] = quick_gelu_fast(x[e]); }
                const int row = rb * 16 + r16, chunk = 2 * wave + (q >> 1);
                const int off = row * 256 + ((chunk ^ r16) << 4) + (q & 1) * 8;
                uint2 ua, up;
                ua.x = f2bf2(g[0], g[1]); ua.y = f2bf2(g[2], g[3]);
                *reinterpret_cast<uint2*>(Hc + off) = ua;
                if (saving) {
                    up.x = f2bf2(x[0], x[1]); up.y = f2bf2(x[2], x[3]);
                    *reinterpret_cast<uint2*>(lds + PRE_OFF + off) = up;
                }
            }
        }
        __syncthreads();                 // the chunk is complete in LDS
        // side outputs of this chunk: LDS -> HBM right away (the staging panel is rewritten by the next chunk's epilogue)
        if (saving) {
#pragma unroll
            for (int i = 0; i < NCI; ++i) {
                const int row = wave * RPW + i * 4 + (ln >> 4), chunk = ln & 15;
                const int off = min(row, XROWS - 1) * 256 + ((chunk ^ (row & 15)) << 4);
                const uint4 va = *reinterpret_cast<const uint4*>(Hc + off), vp = *reinterpret_cast<const uint4*>(lds + PRE_OFF + off);
                if (row < (wave + 1) * RPW && row < L) {
                    const long g = (row0 + row) * 2048 + c * 128 + chunk * 8;
                    *reinterpret_cast<uint4*>(a.h_act + g) = va;
                    *reinterpret_cast<uint4*>(a.h_pre + g) = vp;
                }
            }
        }
        bf16x8 Hf[NRB];
        const unsigned hbase = (unsigned)(uintptr_t)(Hc + r16 * 256);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) Hf[rb] = *(lds_frag_t)(uintptr_t)(hbase + rb * 4096 + ((q ^ r16) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                 // every wave has read its part of the staging panel
        // ---- c_proj(c): acc_o += act[chunk] W_proj[:, chunk]^T, K = 128
        pn_static_for<0, 4>([&](auto jc) {
            constexpr int KK = decltype(jc)::value;
            M80WP& W = WP[KK % M80_DP];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) acc_o[fb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.f[fb], Hf[rb], acc_o[fb][rb], 0, 0, 0);
                if constexpr (KK < 3) Hf[rb] = *(lds_frag_t)(uintptr_t)(hbase + rb * 4096 + ((((KK + 1) * 4 + q) ^ r16) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue: x_out = x_mid + acc_o + b_proj -> bf16 panel in LDS (over the input panel), then row by row: x_out out, the next
    // LayerNorm (the next block's ln_1, or the stack's post-LayerNorm) -> xn_next
    {
        const int r16 = lane & 15, q = lane >> 4;
        uint2 res[4][NRB];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
                res[fb][rb] = *reinterpret_cast<const uint2*>(a.x_mid + (row0 + min(rb * 16 + r16, L - 1)) * 512 + 64 * wave + 16 * fb + 4 * q);
        __syncthreads();                 // every wave is done with the input panel (last c_fc) -- it becomes the output panel
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const float4 bv = *reinterpret_cast<const float4*>(a.b_proj + 64 * wave + 16 * fb + 4 * q);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float r0 = __uint_as_float(res[fb][rb].x << 16), r1 = __uint_as_float(res[fb][rb].x & 0xffff0000u);
                const float r2 = __uint_as_float(res[fb][rb].y << 16), r3 = __uint_as_float(res[fb][rb].y & 0xffff0000u);
                uint2 u;
                u.x = f2bf2(acc_o[fb][rb][0] + bv.x + r0, acc_o[fb][rb][1] + bv.y + r1);
                u.y = f2bf2(acc_o[fb][rb][2] + bv.z + r2, acc_o[fb][rb][3] + bv.w + r3);
                const int row = rb * 16 + r16, chunk = 8 * wave + 2 * fb + (q >> 1);
                *reinterpret_cast<uint2*>(pn_panel_slot<1024>(lds + XN_OFF, row, chunk) + (q & 1) * 8) = u;
            }
        }
        __syncthreads();
        f8 gn, bn;
        if (a.xn_next) { gn = ld8f(a.nln_g + lane * 8); bn = ld8f(a.nln_b + lane * 8); }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int m = wave * RPW + r;
            if (m >= L) break;
            const uint4 u = *reinterpret_cast<const uint4*>(pn_panel_slot<1024>(lds + XN_OFF, m, lane));
            *reinterpret_cast<uint4*>(a.x_out + (row0 + m) * 512 + lane * 8) = u;
            if (a.xn_next) {
                float x[8];
                pn_unpack8(u, x);
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += x[j];
                const float mean = wave_sum(s) * (1.0f / 512);
                float qq = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { x[j] -= mean; qq += x[j] * x[j]; }
                const float rstd = rsqrtf(wave_sum(qq) * (1.0f / 512) + a.eps);
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = x[j] * rstd * gn.v[j] + bn.v[j];
                *reinterpret_cast<uint4*>(a.xn_next + (row0 + m) * 512 + lane * 8) = pn_pack8(o);
                if (lane == 0) { a.nmean[row0 + m] = mean; a.nrstd[row0 + m] = rstd; }
            }
        }
    }
}

}  // namespace tal

using namespace tal;

extern "C" int tan_mlp80_supported(int L, int C, int FF, int dtype) {
    return dtype == TAN_BF16 && C == 512 && FF == 2048 && L > 48 && L <= 80;
}

extern "C" int tan_mlp80_fwd(const tan_mlp_desc* d, int L, void* stream) {
    TAN_REQUIRE(d && d->x_mid && d->ln_g && d->ln_b && d->pw_fc && d->pw_proj && d->b_fc && d->b_proj && d->x_out);
    TAN_REQUIRE((d->h_pre != nullptr) == (d->h_act != nullptr) && (d->mean2 != nullptr) == (d->rstd2 != nullptr));
    TAN_REQUIRE(tan_mlp80_supported(L, d->C, d->FF, TAN_BF16) && d->rows > 0 && d->rows % L == 0 && !d->pw_out);
    TAN_REQUIRE(!d->xn_next || (d->nln_g && d->nln_b && d->nmean && d->nrstd));
    Mlp80Args a;
    a.x_mid = (const bf16_t*)d->x_mid; a.ln_g = d->ln_g; a.ln_b = d->ln_b;
    a.pw_fc = (const char*)d->pw_fc; a.pw_proj = (const char*)d->pw_proj; a.b_fc = d->b_fc; a.b_proj = d->b_proj;
    a.xn2 = (bf16_t*)d->xn2; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.h_pre = (bf16_t*)d->h_pre; a.h_act = (bf16_t*)d->h_act; a.x_out = (bf16_t*)d->x_out;
    a.nln_g = d->nln_g; a.nln_b = d->nln_b; a.xn_next = (bf16_t*)d->xn_next; a.nmean = d->nmean; a.nrstd = d->nrstd;
    a.eps = d->eps; a.L = L;
    const dim3 grid((unsigned)(d->rows / L));
    const int rec = prof_begin((hipStream_t)stream, TAN_PROF_PANEL, 2.0 * d->rows * 512.0 * 2048.0 * 2.0);
    if (L <= 64) hipLaunchKernelGGL((mlp80_fwd_kernel<4>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp80_fwd_kernel<5>), grid, dim3(512), 0, (hipStream_t)stream, a);
    prof_end((hipStream_t)stream, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}
