// Multi-head self-attention core for the TAN encoders (gfx950): softmax_k(q k^T / sqrt(dh) + key_padding) v,
// head dim 64, no causal / attention mask, dropout 0 -- what nn.MultiheadAttention computes at
// model/tfm_model.py:21,30-32 between its in-proj and out-proj GEMMs -- forward and backward.
//
// Activations: qkv [B*L, 3C] (q | k | v, heads contiguous inside each third), o [B*L, C]; row = b*L + t.
// One 256-thread workgroup (4 waves, 2x2) per (64-query tile, head, video).  All contractions are 32x32 MFMA
// tiles (tan_mma.h); operands are staged in LDS as [row][64 + pad] images and read either K-contiguous
// (ds_read_b128 for bf16) or K-strided through the generic fragment loader.
//
// forward : S = (q/8) k^T for every 64-key tile -> LDS score panel [64][Lpad] f32 -> exact row softmax
//           (4 lanes per row, shuffle reductions) -> O = P v over the key tiles.  lse[b,h,t] saved.
// backward: P is recomputed from lse.  dq kernel: workgroup per query tile, loops key tiles.
//           dk/dv kernel: workgroup per key tile, loops query tiles.  No atomics, deterministic.
#include "tan_mma.h"

namespace tal {

constexpr int DH = 64, TQ = 64;

template <typename T> struct AttnCfg;
template <> struct AttnCfg<float> { static constexpr int LD = 65, VE = 4; };
template <> struct AttnCfg<bf16_t> { static constexpr int LD = 72, VE = 8; };

template <typename T> __device__ __forceinline__ float fast_exp(float x);
template <> __device__ __forceinline__ float fast_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float fast_exp<bf16_t>(float x) { return __expf(x); }

// Load rows [row0, row0+64) x 64 channels of a [*, ld] activation into an LDS tile [64][LD]; rows >= L are zero.
template <typename T>
__device__ __forceinline__ void load_tile(T* lds, const T* __restrict__ g, long ld, int row0, int L, float scale) {
    constexpr int VE = AttnCfg<T>::VE, LD = AttnCfg<T>::LD, VPR = DH / VE, NV = TQ * VPR / 256;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = threadIdx.x + 256 * i, r = v / VPR, c = (v % VPR) * VE;
        union { uint4 u; float f[4]; bf16_t h[8]; } x;
        x.u = make_uint4(0, 0, 0, 0);
        if (row0 + r < L) x.u = *reinterpret_cast<const uint4*>(g + (long)(row0 + r) * ld + c);
        if (sizeof(T) == 4) {
            float* d = (float*)lds + r * LD + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = x.f[e] * scale;
        } else {
            if (scale != 1.0f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x.h[e] = f2bf(bf2f(x.h[e]) * scale);
            }
            *reinterpret_cast<uint4*>((bf16_t*)lds + r * LD + c) = x.u;
        }
    }
}

// one wave: acc(32x32) = sum_{k<64} A[o_a0 + i][k] * B[o_b0 + j][k]   (A_KC / B_KC select the LDS orientation)
template <typename T, bool A_KC, bool B_KC>
__device__ __forceinline__ void mma64(f32x16& acc, const T* A, int lda, int a0, const T* B, int ldb, int b0, int lane) {
#pragma unroll
    for (int ks = 0; ks < 64; ks += Mma<T>::KS) {
        typename Mma<T>::frag_t a = Mma<T>::template load<A_KC>(A, lda, a0, ks, lane);
        typename Mma<T>::frag_t b = Mma<T>::template load<B_KC>(B, ldb, b0, ks, lane);
        Mma<T>::mma(acc, a, b);
    }
}

struct AttnArgs {
    const void* qkv; const unsigned char* keypad; void* o; float* lse;
    const void* d_o; void* dqkv;
    int B, L, H, Lpad;
};

// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    constexpr int LD = AttnCfg<T>::LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LS = a.Lpad + 1, LP = a.Lpad + 8;
    T* Qs = (T*)smem;
    T* KVs = Qs + TQ * LD;
    float* S = (float*)(KVs + TQ * LD);
    bf16_t* Pb = (bf16_t*)(S + TQ * LS);  // bf16 mode only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int q0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const T* base = (const T*)a.qkv + (long)b * L * ld + h * DH;
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;

    load_tile<T>(Qs, base, ld, q0, L, 0.125f);
    const int ntile = a.Lpad / 64;
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        load_tile<T>(KVs, base + C, ld, kt * 64, L, 1.0f);
        __syncthreads();
        f32x16 acc; acc_zero(acc);
        mma64<T, true, true>(acc, Qs, LD, wm * 32, KVs, LD, wn * 32, lane);
        const int col = kt * 64 + wn * 32 + acc_col(lane);
        const bool masked = (col >= L) || (kp && kp[col]);
#pragma unroll
        for (int r = 0; r < 16; ++r) S[(wm * 32 + acc_row(r, lane)) * LS + col] = masked ? -INFINITY : acc[r];
    }
    __syncthreads();
    // exact softmax over the L keys of each query row: 4 lanes per row
    {
        const int r = tid >> 2, part = tid & 3;
        float* srow = S + r * LS;
        float m = -INFINITY;
        for (int j = part; j < a.Lpad; j += 4) m = fmaxf(m, srow[j]);
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        float sum = 0.f;
        const bool dead = (m == -INFINITY);  // every key padded: the reference yields NaN here; we emit zeros
        for (int j = part; j < a.Lpad; j += 4) {
            const float e = dead ? 0.f : fast_exp<T>(srow[j] - m);
            srow[j] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float inv = dead ? 0.f : 1.0f / sum;
        for (int j = part; j < a.Lpad; j += 4) {
            const float p = srow[j] * inv;
            if (sizeof(T) == 4) srow[j] = p;
            else Pb[r * LP + j] = f2bf(p);
        }
        if (part == 0 && q0 + r < L) a.lse[((long)b * a.H + h) * L + q0 + r] = dead ? -INFINITY : m + logf(sum);
    }
    f32x16 acc; acc_zero(acc);
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        load_tile<T>(KVs, base + 2 * C, ld, kt * 64, L, 1.0f);
        __syncthreads();
        if (sizeof(T) == 4) mma64<T, true, false>(acc, (const T*)S + kt * 64, LS, wm * 32, KVs, LD, wn * 32, lane);
        else mma64<T, true, false>(acc, (const T*)Pb + kt * 64, LP, wm * 32, KVs, LD, wn * 32, lane);
    }
    T* o = (T*)a.o + (long)b * L * C + h * DH;
    const int col = wn * 32 + acc_col(lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = q0 + wm * 32 + acc_row(r, lane);
        if (row < L) st_f(o + (long)row * C + col, acc[r]);
    }
}

// ------------------------------------------------------------------------------------------------------
// shared by both backward kernels: row statistics of a 64-query tile -> LDS (lse, delta = rowsum(dO * O))
template <typename T>
__device__ __forceinline__ void load_row_stats(const AttnArgs& a, const T* dOs, int q0, int b, int h, float* lse_s,
                                               float* delta_s) {
    constexpr int LD = AttnCfg<T>::LD;
    const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int C = a.H * DH, L = a.L;
    float s = 0.f;
    if (q0 + r < L) {
        const T* orow = (const T*)a.o + ((long)b * L + q0 + r) * C + h * DH + part * 16;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const float4 ov = ld4(orow + e);
            const T* d = dOs + r * LD + part * 16 + e;
            s += ov.x * ld_f(d) + ov.y * ld_f(d + 1) + ov.z * ld_f(d + 2) + ov.w * ld_f(d + 3);
        }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0) {
        delta_s[r] = s;
        lse_s[r] = (q0 + r < L) ? a.lse[((long)b * a.H + h) * L + q0 + r] : 0.f;
    }
}

// P and dS for the wave's 32x32 sub-tile: p = exp(s - lse_i) (0 where masked / out of range), ds = p * (dp - delta_i)
template <typename T>
__device__ __forceinline__ void p_and_ds(f32x16& s, f32x16& dp, const float* lse_s, const float* delta_s, int row_off,
                                         bool col_masked, int q0, int L, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = row_off + acc_row(r, lane);
        const float l = lse_s[i];
        const bool dead = col_masked || (q0 + i >= L) || (l == -INFINITY);
        const float p = dead ? 0.f : fast_exp<T>(s[r] - l);
        s[r] = p;
        dp[r] = p * (dp[r] - delta_s[i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int LD = AttnCfg<T>::LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = (T*)smem;
    T* dOs = Qs + TQ * LD;
    T* Ks = dOs + TQ * LD;
    T* Vs = Ks + TQ * LD;
    T* dSs = Vs + TQ * LD;
    float* lse_s = (float*)(dSs + TQ * LD);
    float* delta_s = lse_s + TQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int q0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const T* base = (const T*)a.qkv + (long)b * L * ld + h * DH;
    const T* dO = (const T*)a.d_o + (long)b * L * C + h * DH;
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;

    load_tile<T>(Qs, base, ld, q0, L, 0.125f);
    load_tile<T>(dOs, dO, C, q0, L, 1.0f);
    __syncthreads();
    load_row_stats<T>(a, dOs, q0, b, h, lse_s, delta_s);
    f32x16 dq; acc_zero(dq);
    const int ntile = a.Lpad / 64;
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        load_tile<T>(Ks, base + C, ld, kt * 64, L, 1.0f);
        load_tile<T>(Vs, base + 2 * C, ld, kt * 64, L, 1.0f);
        __syncthreads();
        f32x16 s, dp; acc_zero(s); acc_zero(dp);
        mma64<T, true, true>(s, Qs, LD, wm * 32, Ks, LD, wn * 32, lane);
        mma64<T, true, true>(dp, dOs, LD, wm * 32, Vs, LD, wn * 32, lane);
        const int col = kt * 64 + wn * 32 + acc_col(lane);
        p_and_ds<T>(s, dp, lse_s, delta_s, wm * 32, (col >= L) || (kp && kp[col]), q0, L, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) st_f(dSs + (wm * 32 + acc_row(r, lane)) * LD + wn * 32 + acc_col(lane), dp[r]);
        __syncthreads();
        // dq[i][d] += sum_key dS[i][key] * K[key][d]
        mma64<T, true, false>(dq, dSs, LD, wm * 32, Ks, LD, wn * 32, lane);
    }
    T* out = (T*)a.dqkv + (long)b * L * ld + h * DH;
    const int col = wn * 32 + acc_col(lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = q0 + wm * 32 + acc_row(r, lane);
        if (row < L) st_f(out + (long)row * ld + col, dq[r] * 0.125f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int LD = AttnCfg<T>::LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = (T*)smem;
    T* dOs = Qs + TQ * LD;
    T* Ks = dOs + TQ * LD;
    T* Vs = Ks + TQ * LD;
    T* dSs = Vs + TQ * LD;
    T* Ps = dSs + TQ * LD;
    float* lse_s = (float*)(Ps + TQ * LD);
    float* delta_s = lse_s + TQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int k0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const T* base = (const T*)a.qkv + (long)b * L * ld + h * DH;
    const T* dO = (const T*)a.d_o + (long)b * L * C + h * DH;
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;

    load_tile<T>(Ks, base + C, ld, k0, L, 1.0f);
    load_tile<T>(Vs, base + 2 * C, ld, k0, L, 1.0f);
    f32x16 dk, dv; acc_zero(dk); acc_zero(dv);
    const int col = k0 + wn * 32 + acc_col(lane);
    const bool col_masked = (col >= L) || (kp && kp[col]);
    const int ntile = a.Lpad / 64;
    for (int qt = 0; qt < ntile; ++qt) {
        const int q0 = qt * 64;
        __syncthreads();
        load_tile<T>(Qs, base, ld, q0, L, 0.125f);
        load_tile<T>(dOs, dO, C, q0, L, 1.0f);
        __syncthreads();
        load_row_stats<T>(a, dOs, q0, b, h, lse_s, delta_s);
        __syncthreads();
        f32x16 s, dp; acc_zero(s); acc_zero(dp);
        mma64<T, true, true>(s, Qs, LD, wm * 32, Ks, LD, wn * 32, lane);
        mma64<T, true, true>(dp, dOs, LD, wm * 32, Vs, LD, wn * 32, lane);
        p_and_ds<T>(s, dp, lse_s, delta_s, wm * 32, col_masked, q0, L, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = (wm * 32 + acc_row(r, lane)) * LD + wn * 32 + acc_col(lane);
            st_f(Ps + off, s[r]);
            st_f(dSs + off, dp[r]);
        }
        __syncthreads();
        // dv[key][d] += sum_q P[q][key] dO[q][d] ; dk[key][d] += sum_q dS[q][key] (q/8)[q][d]   (both K-strided reads)
        mma64<T, false, false>(dv, Ps, LD, wm * 32, dOs, LD, wn * 32, lane);
        mma64<T, false, false>(dk, dSs, LD, wm * 32, Qs, LD, wn * 32, lane);
    }
    T* out = (T*)a.dqkv + (long)b * L * ld + h * DH;
    const int ocol = wn * 32 + acc_col(lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = k0 + wm * 32 + acc_row(r, lane);
        if (row < L) {
            st_f(out + (long)row * ld + C + ocol, dk[r]);
            st_f(out + (long)row * ld + 2 * C + ocol, dv[r]);
        }
    }
}

template <typename T> static size_t fwd_smem(int Lpad) {
    size_t s = 2 * TQ * AttnCfg<T>::LD * sizeof(T) + (size_t)TQ * (Lpad + 1) * 4;
    if (sizeof(T) == 2) s += (size_t)TQ * (Lpad + 8) * 2;
    return s;
}
template <typename T> static size_t bwd_smem(int ntiles) { return (size_t)ntiles * TQ * AttnCfg<T>::LD * sizeof(T) + 2 * TQ * 4; }

template <typename K> static int set_smem(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return TAN_ERR_BAD_ARG;
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace tal

using namespace tal;

extern "C" int tan_attn_fwd(const void* qkv, const unsigned char* key_padding_mask, void* o, float* lse, int B, int L, int H,
                            int dtype, void* stream) {
    TAN_REQUIRE(qkv && o && lse && B > 0 && L > 0 && H > 0);
    AttnArgs a{};
    a.qkv = qkv; a.keypad = key_padding_mask; a.o = o; a.lse = lse; a.B = B; a.L = L; a.H = H;
    a.Lpad = (L + 63) / 64 * 64;
    dim3 grid(a.Lpad / 64, H, B);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const int rec = prof_begin(st, TAN_PROF_ATTN_FWD, 4.0 * B * H * (double)L * L * DH);
    if (dtype == TAN_F32) {
        size_t sm = fwd_smem<float>(a.Lpad);
        if ((rc = set_smem(attn_fwd_kernel<float>, sm))) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<float>), grid, dim3(256), sm, st, a);
    } else if (dtype == TAN_BF16) {
        size_t sm = fwd_smem<bf16_t>(a.Lpad);
        if ((rc = set_smem(attn_fwd_kernel<bf16_t>, sm))) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, dim3(256), sm, st, a);
    } else return TAN_ERR_BAD_ARG;
    prof_end(st, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_attn_bwd(const void* qkv, const unsigned char* key_padding_mask, const void* o, const float* lse,
                            const void* d_o, void* dqkv, int B, int L, int H, int dtype, void* stream) {
    TAN_REQUIRE(qkv && o && lse && d_o && dqkv && B > 0 && L > 0 && H > 0);
    AttnArgs a{};
    a.qkv = qkv; a.keypad = key_padding_mask; a.o = (void*)o; a.lse = (float*)lse; a.d_o = d_o; a.dqkv = dqkv;
    a.B = B; a.L = L; a.H = H; a.Lpad = (L + 63) / 64 * 64;
    dim3 grid(a.Lpad / 64, H, B);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const int rec = prof_begin(st, TAN_PROF_ATTN_BWD, 14.0 * B * H * (double)L * L * DH);
    if (dtype == TAN_F32) {
        if ((rc = set_smem(attn_bwd_dq_kernel<float>, bwd_smem<float>(5)))) return rc;
        if ((rc = set_smem(attn_bwd_dkv_kernel<float>, bwd_smem<float>(6)))) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<float>), grid, dim3(256), bwd_smem<float>(5), st, a);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<float>), grid, dim3(256), bwd_smem<float>(6), st, a);
    } else if (dtype == TAN_BF16) {
        if ((rc = set_smem(attn_bwd_dq_kernel<bf16_t>, bwd_smem<bf16_t>(5)))) return rc;
        if ((rc = set_smem(attn_bwd_dkv_kernel<bf16_t>, bwd_smem<bf16_t>(6)))) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t>), grid, dim3(256), bwd_smem<bf16_t>(5), st, a);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<bf16_t>), grid, dim3(256), bwd_smem<bf16_t>(6), st, a);
    } else return TAN_ERR_BAD_ARG;
    prof_end(st, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}
