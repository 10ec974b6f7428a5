// MFMA GEMM for the TAN hot path (gfx950).  One kernel template covers every nn.Linear forward / dX / dW
// product and the similarity einsum (see include/tan_hip.h: tan_gemm).
//
// Block tile 128x128, 256 threads = 4 waves in a 2x2 grid, each wave a 64x64 output tile = 2x2 MFMA 32x32
// accumulators.  K-step 64 (bf16, v_mfma_f32_32x32x16_bf16) or 16 (f32, v_mfma_f32_32x32x2_f32 -- an exact f32
// fma chain, used by the parity mode).  Operands are staged global -> registers -> LDS with the global loads
// of tile t+1 in flight while tile t is multiplied (one LDS buffer, two barriers per K-step).
//
// LDS images
//   bf16, K-contiguous source : [rows][64 + 8]   (144-B rows: ds_read_b128 fragment reads are conflict-free)
//   bf16, K-strided source    : [64 k][128 + 8]  (mirrors memory; fragments gathered with 8 u16 reads)
//   f32 (either source)       : [16 k][128 + 4]  (K-strided image; K-contiguous sources are transposed on write)
#include "tan_mma.h"

namespace tal {

constexpr int BM = 128, BN = 128;

template <typename T> struct GemmCfg;
template <> struct GemmCfg<float> {
    static constexpr int BK = 16, VE = 4;
    static constexpr int LD_KC = 0;            // unused: f32 always keeps the K-strided image
    static constexpr int LD_KS = BM + 4;
    static constexpr bool lds_kc(bool) { return false; }
};
template <> struct GemmCfg<bf16_t> {
    static constexpr int BK = 64, VE = 8;
    static constexpr int LD_KC = BK + 8;
    static constexpr int LD_KS = BM + 8;
    static constexpr bool lds_kc(bool src_kc) { return src_kc; }
};

struct GemmArgs {
    const void* A; const void* B; void* C;
    const float* bias; const void* residual; void* aux;
    long lda, ldb, ldc, ldr, ldaux;
    long sA, sB, sC;
    int M, N, K;
    int act, accumulate, split_k, kchunk;
    int vecA, vecB, fast;
    float alpha;
};

union Vec16 { uint4 u; float f[4]; bf16_t h[8]; };

// Load one 16-byte staging vector of an operand tile.  `outer` indexes M (or N), `k` indexes K.
template <typename T, bool KC>
__device__ __forceinline__ Vec16 load_vec(const T* __restrict__ P, long ld, int outer0, int k0, int v, int OUT, int K, int kend,
                                          bool vec_ok) {
    constexpr int VE = GemmCfg<T>::VE, BK = GemmCfg<T>::BK;
    Vec16 r; r.u = make_uint4(0, 0, 0, 0);
    if (KC) {
        constexpr int VPR = BK / VE;
        const int o = outer0 + v / VPR, k = k0 + (v % VPR) * VE;
        if (o >= OUT) return r;
        const T* p = P + (long)o * ld + k;
        if (vec_ok && k + VE <= kend) { r.u = *reinterpret_cast<const uint4*>(p); return r; }
#pragma unroll
        for (int e = 0; e < VE; ++e) if (k + e < kend) { if (sizeof(T) == 4) r.f[e] = ((const float*)p)[e]; else r.h[e] = ((const bf16_t*)p)[e]; }
    } else {
        constexpr int VPR = BM / VE;
        const int k = k0 + v / VPR, o = outer0 + (v % VPR) * VE;
        if (k >= kend) return r;
        const T* p = P + (long)k * ld + o;
        if (vec_ok && o + VE <= OUT) { r.u = *reinterpret_cast<const uint4*>(p); return r; }
#pragma unroll
        for (int e = 0; e < VE; ++e) if (o + e < OUT) { if (sizeof(T) == 4) r.f[e] = ((const float*)p)[e]; else r.h[e] = ((const bf16_t*)p)[e]; }
    }
    return r;
}

// Branch-free variant for aligned operands (16-byte aligned base / leading dim / batch stride, K % VE == 0 and, for a
// K-strided operand, OUT % VE == 0): the address is clamped in bounds, the load is unconditional and out-of-range
// vectors are zeroed with selects -- so all staging loads of a K-step issue back to back.  (hipcc wraps a conditional
// load in a branch + s_waitcnt vmcnt(0), which serialised the 8 loads per thread of the generic path.)
template <typename T, bool KC>
__device__ __forceinline__ Vec16 load_vec_fast(const T* __restrict__ P, long ld, int outer0, int k0, int v, int OUT, int kend) {
    constexpr int VE = GemmCfg<T>::VE, BK = GemmCfg<T>::BK;
    int o, k;
    if (KC) {
        constexpr int VPR = BK / VE;
        o = outer0 + v / VPR; k = k0 + (v % VPR) * VE;
    } else {
        constexpr int VPR = BM / VE;
        k = k0 + v / VPR; o = outer0 + (v % VPR) * VE;
    }
    const bool ok = (o < OUT) && (k < kend);
    const int oc = min(o, OUT - (KC ? 1 : VE)), kc = min(k, kend - (KC ? VE : 1));
    const T* p = KC ? P + (long)oc * ld + kc : P + (long)kc * ld + oc;
    Vec16 r;
    r.u = *reinterpret_cast<const uint4*>(p);
    r.u.x = ok ? r.u.x : 0u; r.u.y = ok ? r.u.y : 0u; r.u.z = ok ? r.u.z : 0u; r.u.w = ok ? r.u.w : 0u;
    return r;
}

template <typename T, bool KC>
__device__ __forceinline__ void store_vec(T* lds, int v, const Vec16& r) {
    constexpr int VE = GemmCfg<T>::VE, BK = GemmCfg<T>::BK;
    if (sizeof(T) == 4) {
        float* L = (float*)lds;
        constexpr int LD = GemmCfg<float>::LD_KS;
        if (KC) {  // transpose on write: element (o, k) -> L[k*LD + o]
            constexpr int VPR = 16 / 4;
            const int o = v / VPR, k = (v % VPR) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) L[(k + e) * LD + o] = r.f[e];
        } else {
            constexpr int VPR = BM / 4;
            const int k = v / VPR, o = (v % VPR) * 4;
            *reinterpret_cast<uint4*>(L + k * LD + o) = r.u;
        }
    } else {
        bf16_t* L = (bf16_t*)lds;
        if (KC) {
            constexpr int VPR = BK / VE, LD = GemmCfg<bf16_t>::LD_KC;
            const int o = v / VPR, k = (v % VPR) * VE;
            *reinterpret_cast<uint4*>(L + o * LD + k) = r.u;
        } else {
            constexpr int VPR = BM / VE, LD = GemmCfg<bf16_t>::LD_KS;
            const int k = v / VPR, o = (v % VPR) * VE;
            *reinterpret_cast<uint4*>(L + k * LD + o) = r.u;
        }
    }
}

template <typename TC, bool GUARD>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], TC* C, const TC* R, TC* AUX, int m0, int n0,
                                              int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + acc_col(lane);
            if (GUARD && col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (GUARD && row >= g.M) continue;
                float v = acc[i][j][r] * g.alpha + bv;
                if (g.act == TAN_ACT_QUICKGELU) {
                    if (AUX) st_f(AUX + (long)row * g.ldaux + col, v);
                    v = quick_gelu(v);
                } else if (g.act == TAN_ACT_QUICKGELU_GRAD) {
                    v *= quick_gelu_grad(ld_f(AUX + (long)row * g.ldaux + col));
                } else if (g.act == TAN_ACT_RELU) {
                    v = fmaxf(v, 0.0f);
                }
                if (R) v += ld_f(R + (long)row * g.ldr + col);
                TC* cp = C + (long)row * g.ldc + col;
                if (g.accumulate) unsafeAtomicAdd((float*)cp, v);
                else st_f(cp, v);
            }
        }
}

template <typename T, typename TC, bool A_KC, bool B_KC, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    typedef GemmCfg<T> Cfg;
    constexpr int BK = Cfg::BK;
    constexpr int NV = BM * BK * (int)sizeof(T) / 16 / 256;  // staging vectors per thread per operand
    constexpr bool LA_KC = Cfg::lds_kc(A_KC), LB_KC = Cfg::lds_kc(B_KC);
    constexpr int LDA = LA_KC ? Cfg::LD_KC : Cfg::LD_KS;
    constexpr int LDB = LB_KC ? Cfg::LD_KC : Cfg::LD_KS;
    constexpr int A_ELEMS = LA_KC ? BM * LDA : BK * LDA;
    constexpr int B_ELEMS = LB_KC ? BN * LDB : BK * LDB;
    __shared__ __attribute__((aligned(16))) T lds[A_ELEMS + B_ELEMS];
    T* As = lds;
    T* Bs = lds + A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int z = blockIdx.z, batch = z / g.split_k, split = z % g.split_k;
    const T* A = (const T*)g.A + (long)batch * g.sA;
    const T* B = (const T*)g.B + (long)batch * g.sB;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);

    Vec16 ra[NV], rb[NV];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (FAST) {
                ra[i] = load_vec_fast<T, A_KC>(A, g.lda, m0, k0, tid + 256 * i, g.M, kend);
                rb[i] = load_vec_fast<T, B_KC>(B, g.ldb, n0, k0, tid + 256 * i, g.N, kend);
            } else {
                ra[i] = load_vec<T, A_KC>(A, g.lda, m0, k0, tid + 256 * i, g.M, g.K, kend, g.vecA);
                rb[i] = load_vec<T, B_KC>(B, g.ldb, n0, k0, tid + 256 * i, g.N, g.K, kend, g.vecB);
            }
        }
    };
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            store_vec<T, A_KC>(As, tid + 256 * i, ra[i]);
            store_vec<T, B_KC>(Bs, tid + 256 * i, rb[i]);
        }
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK; ks += Mma<T>::KS) {
            typename Mma<T>::frag_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = Mma<T>::template load<LA_KC>(As, LDA, wm * 64 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Mma<T>::template load<LB_KC>(Bs, LDB, wn * 64 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<T>::mma(acc[i][j], a[i], b[j]);
        }
        __syncthreads();
    }

    // epilogue
    TC* C = (TC*)g.C + (long)batch * g.sC;
    const TC* R = g.residual ? (const TC*)g.residual + (long)batch * g.sC : nullptr;
    TC* AUX = g.aux ? (TC*)g.aux + (long)batch * g.sC : nullptr;
    if (m0 + BM <= g.M && n0 + BN <= g.N) gemm_epilogue<TC, false>(g, acc, C, R, AUX, m0, n0, wm, wn, lane);
    else gemm_epilogue<TC, true>(g, acc, C, R, AUX, m0, n0, wm, wn, lane);
}

template <typename T, typename TC>
static int launch_gemm(const tan_gemm_desc* d, const GemmArgs& a, dim3 grid, hipStream_t st) {
    const bool fast = a.fast != 0;
#define TAN_GEMM_LAUNCH(AK, BK_, F) hipLaunchKernelGGL((gemm_kernel<T, TC, AK, BK_, F>), grid, dim3(256), 0, st, a)
    if (d->a_kc && d->b_kc) { if (fast) TAN_GEMM_LAUNCH(true, true, true); else TAN_GEMM_LAUNCH(true, true, false); }
    else if (d->a_kc && !d->b_kc) { if (fast) TAN_GEMM_LAUNCH(true, false, true); else TAN_GEMM_LAUNCH(true, false, false); }
    else if (!d->a_kc && d->b_kc) { if (fast) TAN_GEMM_LAUNCH(false, true, true); else TAN_GEMM_LAUNCH(false, true, false); }
    else { if (fast) TAN_GEMM_LAUNCH(false, false, true); else TAN_GEMM_LAUNCH(false, false, false); }
#undef TAN_GEMM_LAUNCH
    TAN_LAUNCH_CHECK();
    return 0;
}

int gemm_glds_try(const tan_gemm_desc* d, hipStream_t st);   // tan_gemm_glds.hip

}  // namespace tal

using namespace tal;

extern "C" int tan_gemm(const tan_gemm_desc* d, void* stream) {
    TAN_REQUIRE(d && d->A && d->B && d->C);
    TAN_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1 && d->split_k >= 1);
    TAN_REQUIRE(d->dtype == TAN_F32 || d->dtype == TAN_BF16);
    TAN_REQUIRE(d->out_dtype == TAN_F32 || d->out_dtype == TAN_BF16);
    if (d->accumulate) TAN_REQUIRE(d->out_dtype == TAN_F32 && !d->bias && !d->residual && d->act == TAN_ACT_NONE);
    else TAN_REQUIRE(d->split_k == 1);
    if (d->act == TAN_ACT_QUICKGELU_GRAD) TAN_REQUIRE(d->aux != nullptr);
    const int esz = d->dtype == TAN_F32 ? 4 : 2, ve = 16 / esz, bk = d->dtype == TAN_F32 ? 16 : 64;
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias; a.residual = d->residual; a.aux = d->aux;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldr = d->ldr; a.ldaux = d->ldaux;
    a.sA = d->sA; a.sB = d->sB; a.sC = d->sC;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.act = d->act; a.accumulate = d->accumulate; a.split_k = d->split_k; a.alpha = d->alpha;
    a.kchunk = (int)(((long)cdiv(cdiv(d->K, d->split_k), bk)) * bk);
    // 16-byte vector loads need aligned bases, leading dims and batch strides
    auto vec_ok = [&](const void* p, long ld, long bs) {
        return ((uintptr_t)p % 16 == 0) && (ld % ve == 0) && (bs % ve == 0);
    };
    a.vecA = vec_ok(d->A, d->lda, d->sA) && (a.kchunk % ve == 0);
    a.vecB = vec_ok(d->B, d->ldb, d->sB) && (a.kchunk % ve == 0);
    // branch-free staging needs vectors that never straddle an edge
    a.fast = a.vecA && a.vecB && (d->K % ve == 0) && (d->a_kc || d->M % ve == 0) && (d->b_kc || d->N % ve == 0);
    dim3 grid(cdiv(d->N, BN), cdiv(d->M, BM), d->batch * d->split_k);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == TAN_F32) TAN_REQUIRE(d->out_dtype == TAN_F32);
    const int kind = (d->dtype == TAN_BF16 ? TAN_PROF_GEMM_BF16 : TAN_PROF_GEMM_F32) + (d->a_kc ? 0 : 2) + (d->b_kc ? 0 : 1);
    const int rec = prof_begin(st, kind, 2.0 * d->M * d->N * (double)d->K * d->batch);
    if (d->colsum) TAN_REQUIRE(d->batch == 1 && !d->accumulate);
    int rc = gemm_glds_try(d, st);            // aligned bf16: direct-to-LDS kernel (fuses colsum when its epilogue is vectorised)
    if (rc == -3) { prof_end(st, rec); return tan_colsum_acc(d->C, d->colsum, d->M, d->N, d->out_dtype, stream); }
    if (rc != -2) { prof_end(st, rec); return rc; }
    if (d->dtype == TAN_F32) rc = launch_gemm<float, float>(d, a, grid, st);
    else if (d->out_dtype == TAN_F32) rc = launch_gemm<bf16_t, float>(d, a, grid, st);
    else rc = launch_gemm<bf16_t, bf16_t>(d, a, grid, st);
    prof_end(st, rec);
    if (rc == 0 && d->colsum) rc = tan_colsum_acc(d->C, d->colsum, d->M, d->N, d->out_dtype, stream);
    return rc;
}
