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
#include "tan_attn_img.h"
#include <cstdlib>

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
    float* gbias;          // [3C] f32 or null: += column sums of dqkv (the in_proj bias gradient), short bf16 backward only
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


// ======================================================================================================
// Short-sequence bf16 path (L <= 128: the len=64 video stack and the 64+N joint stack of the headline config).
// One workgroup per (video, head); wave w owns the 32 queries (and, in the backward, the 32 keys) 32w..32w+31, so the
// workgroup has ceil(L/32) waves.  q, k, v (and dO) head slices go HBM -> LDS once with direct-to-LDS loads into
// [row][64] images (128-B rows, 16-B chunk index XOR-swizzled by the row so that both the row-wise ds_read_b128 and the
// transposing ds_read_b64_tr_b16 gathers spread over the banks).  Everything after the single barrier is wave-private:
//   scores are computed TRANSPOSED (S^T = K Q^T), so a lane holds one query column: the softmax reductions are a
//   register loop plus one lane^32 exchange, and the 32x32 accumulator tile is, register for register, the B-operand
//   fragment of the next MFMA (O^T = V^T P^T) once the contraction index is permuted the same way on the A side --
//   which the transposing LDS read does for free.  No score panel in LDS, no workgroup barrier inside the math.
// The backward recomputes P in both orientations instead of exchanging it: phase A (wave owns queries) gives dq and
// delta_i = sum_j P_ij dP_ij (== rowsum(dO*O)), phase B (wave owns keys) gives dk, dv.  Deterministic, no atomics.
// (head-image helpers: tan_attn_img.h)

template <int NKB>
__global__ __launch_bounds__(64 * NKB) void attn_fwd_short_kernel(AttnArgs a) {
    constexpr int LP = 32 * NKB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* Qi = smem; char* Ki = Qi + LP * 128; char* Vi = Ki + LP * 128;
    float* bias = (float*)(Vi + LP * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const bf16_t* base = (const bf16_t*)a.qkv + (long)b * L * ld + h * DH;
    stage_images<NKB>(smem, 0, 3, base, C, ld, L, wave, lane);
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;
    for (int j = tid; j < LP; j += 64 * NKB) bias[j] = (j >= L || (kp && kp[j])) ? -INFINITY : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int q0 = wave * 32;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = img_frag_kc(Qi, q0 + c, 2 * ks + hh);
    f32x16 s[NKB];
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        acc_zero(s[kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Ki, 32 * kb + c, 2 * ks + hh), qf[ks], s[kb], 0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * kb + 8 * g4 + 4 * hh);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = s[kb][4 * g4 + e] * 0.125f + bb[e];
                s[kb][4 * g4 + e] = v;
                m = fmaxf(m, v);
            }
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const bool dead = (m == -INFINITY);       // every key padded: the reference yields NaN here; we emit zeros
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = dead ? 0.f : __expf(s[kb][r] - m);
            s[kb][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = dead ? 0.f : 1.0f / sum;
    if (hh == 0 && q0 + c < L) a.lse[((long)b * a.H + h) * L + q0 + c] = dead ? -INFINITY : m + logf(sum);
    f32x16 o[2];
    acc_zero(o[0]); acc_zero(o[1]);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] *= inv;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8 pf = acc_frag(s[kb], j);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Vi, 32 * kb + 16 * j, 32 * db, lane), pf, o[db], 0, 0, 0);
        }
    }
    // O^T tiles -> the wave's own (now dead) q rows in LDS -> 16-byte row-contiguous global stores
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int row = q0 + c, chunk = 4 * db + g4;
            st4((bf16_t*)(Qi + row * 128 + ((chunk ^ img_swz(row)) << 4) + hh * 8),
                make_float4(o[db][4 * g4], o[db][4 * g4 + 1], o[db][4 * g4 + 2], o[db][4 * g4 + 3]));
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16_t* out = (bf16_t*)a.o + (long)b * L * C + h * DH;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = q0 + it * 8 + (lane >> 3), chunk = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(Qi + row * 128 + ((chunk ^ img_swz(row)) << 4));
        if (row < L) *reinterpret_cast<uint4*>(out + (long)row * C + chunk * 8) = v;
    }
}

// launch bound "2 waves per SIMD": without it hipcc kept the accumulators in 128 AGPRs next to 131-148 VGPRs (259-276 registers
// per lane -> ONE wave per SIMD, one workgroup per CU, four rounds of 1024 workgroups); with it 186 / 214 VGPRs, no spill, two
// workgroups per CU: 7.01 -> 6.89 ms per step.  (Three waves per SIMD at L = 64 needs an 80-byte spill and gains nothing.)
template <int NKB>
__global__ __launch_bounds__(64 * NKB, 2) void attn_bwd_short_kernel(AttnArgs a) {
    constexpr int LP = 32 * NKB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* Qi = smem; char* Ki = Qi + LP * 128; char* Vi = Ki + LP * 128; char* Di = Vi + LP * 128;
    float* bias = (float*)(Di + LP * 128);
    float* lse_s = bias + LP;
    float* delta_s = lse_s + LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const bf16_t* base = (const bf16_t*)a.qkv + (long)b * L * ld + h * DH;
    // STAGGERED START (round 5).  A launch of B * H >= 1024 workgroups is exactly one round of the chip (four per CU) and ran in lock
    // step: every workgroup loads (40-60 KB), then every workgroup multiplies, then every workgroup stores -- HBM idles through the
    // arithmetic and the CUs through the transfers: 21.9 us for 61 MB (L = 64).  Groups of 128 / 256 workgroups (in dispatch order) now start
    // a few microseconds apart (`s_sleep`: the wave leaves the issue slots to the others), so that one group's transfers run under
    // another's arithmetic: L <= 64 four phases of groups of 128, 48 x 64 clk apart, 21.9 -> 17.6 us; L <= 96 two phases of groups of 256,
    // 48 x 64 clk apart, 31.9 -> 28.3 us stand-alone (other spacings / group sizes: DESIGN.md section 6); 4.065 -> 4.046 ms per step
    // with the first parameter set (ABBA x3).
    // Smaller launches do not fill a round and start at once.
#ifndef TAN_ATTN_STAGGER
#define TAN_ATTN_STAGGER 1     // (0: the A/B build of tools/lab/build_variant.sh)
#endif
    if (TAN_ATTN_STAGGER && gridDim.x >= 1024) {
        const int ph = NKB == 2 ? (blockIdx.x >> 7) & 3 : (blockIdx.x >> 8) & 1;
        if (NKB == 2) {
            if (ph > 0) __builtin_amdgcn_s_sleep(48);
            if (ph > 1) __builtin_amdgcn_s_sleep(48);
            if (ph > 2) __builtin_amdgcn_s_sleep(48);
        } else if (NKB == 3 && ph) {
            __builtin_amdgcn_s_sleep(48);
        }
    }
    stage_images<NKB>(smem, 0, 3, base, C, ld, L, wave, lane);
    stage_images<NKB>(smem, 3, 1, (const bf16_t*)a.d_o + (long)b * L * C + h * DH, 0, C, L, wave, lane);
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;
    for (int j = tid; j < LP; j += 64 * NKB) {
        bias[j] = (j >= L || (kp && kp[j])) ? -INFINITY : 0.f;
        float l = INFINITY;                      // rows past L and fully padded rows: p = exp(. - inf) = 0
        if (j < L) { l = a.lse[((long)b * a.H + h) * L + j]; if (l == -INFINITY) l = INFINITY; }
        lse_s[j] = l;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16_t* out = (bf16_t*)a.dqkv + (long)b * L * ld + h * DH;

    f32x16 dq[2];
    {   // ---- phase A: this wave's 32 queries against every key: delta and dq
        const int q0 = wave * 32;
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = img_frag_kc(Qi, q0 + c, 2 * ks + hh); dof[ks] = img_frag_kc(Di, q0 + c, 2 * ks + hh); }
        const float my_lse = lse_s[q0 + c];
        f32x16 p[NKB], dp[NKB];
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            acc_zero(p[kb]); acc_zero(dp[kb]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                p[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Ki, 32 * kb + c, 2 * ks + hh), qf[ks], p[kb], 0, 0, 0);
                dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Vi, 32 * kb + c, 2 * ks + hh), dof[ks], dp[kb], 0, 0, 0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * kb + 8 * g4 + 4 * hh);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(p[kb][4 * g4 + e] * 0.125f + bb[e] - my_lse);
                    p[kb][4 * g4 + e] = pv;
                    delta += pv * dp[kb][4 * g4 + e];
                }
            }
        }
        delta += __shfl_xor(delta, 32, 64);
        if (hh == 0) delta_s[q0 + c] = delta;
        acc_zero(dq[0]); acc_zero(dq[1]);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) p[kb][r] *= (dp[kb][r] - delta);     // dS^T
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 df = acc_frag(p[kb], j);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Ki, 32 * kb + 16 * j, 32 * db, lane), df, dq[db], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    {   // ---- phase B: this wave's 32 keys against every query: dk, dv
        const int k0 = wave * 32;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = img_frag_kc(Ki, k0 + c, 2 * ks + hh); vf[ks] = img_frag_kc(Vi, k0 + c, 2 * ks + hh); }
        const float my_bias = bias[k0 + c];
        f32x16 dk[2], dv[2];
        acc_zero(dk[0]); acc_zero(dk[1]); acc_zero(dv[0]); acc_zero(dv[1]);
#pragma unroll
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 p, dp;
            acc_zero(p); acc_zero(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Qi, 32 * qb + c, 2 * ks + hh), kf[ks], p, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Di, 32 * qb + c, 2 * ks + hh), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 lv = *reinterpret_cast<const float4*>(lse_s + 32 * qb + 8 * g4 + 4 * hh);
                const float4 dl = *reinterpret_cast<const float4*>(delta_s + 32 * qb + 8 * g4 + 4 * hh);
                const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(p[4 * g4 + e] * 0.125f + my_bias - ll[e]);
                    p[4 * g4 + e] = pv;
                    dp[4 * g4 + e] = pv * (dp[4 * g4 + e] - dd[e]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 pf = acc_frag(p, j), df = acc_frag(dp, j);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Di, 32 * qb + 16 * j, 32 * db, lane), pf, dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Qi, 32 * qb + 16 * j, 32 * db, lane), df, dk[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();            // every wave is done with the images: each parks its three result tiles in its own rows
        tiles_to_rows(Qi, k0, c, hh, dq, 0.125f);
        tiles_to_rows(Ki, k0, c, hh, dk, 0.125f);
        tiles_to_rows(Vi, k0, c, hh, dv, 1.0f);
        rows_to_global(Qi, k0, lane, out, ld, 0, L);
        rows_to_global(Ki, k0, lane, out + C, ld, 0, L);
        rows_to_global(Vi, k0, lane, out + 2 * C, ld, 0, L);
        if (a.gbias) {
            // in_proj bias gradient: column sums of the 32 rows this wave parked (the bf16 values tan_colsum_acc would read
            // back from dqkv; rows past L hold exact zeros), one lane per head feature
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const int row = k0 + r, off = row * 128 + (((lane >> 3) ^ img_swz(row)) << 4) + (lane & 7) * 2;
                s0 += bf2f(*reinterpret_cast<const bf16_t*>(Qi + off));
                s1 += bf2f(*reinterpret_cast<const bf16_t*>(Ki + off));
                s2 += bf2f(*reinterpret_cast<const bf16_t*>(Vi + off));
            }
            float* g = a.gbias + h * DH + lane;
            unsafeAtomicAdd(g, s0); unsafeAtomicAdd(g + C, s1); unsafeAtomicAdd(g + 2 * C, s2);
        }
    }
}


// ======================================================================================================
// Long-sequence bf16 path (L > 128, e.g. the len=256 configuration: L = 256 / 272): the same wave-private transposed-score
// scheme, streamed.  A workgroup (4 waves x 32) owns 128 queries (forward, dq) or 128 keys (dk/dv) of one (video, head) and
// walks the other axis in 128-row blocks staged in LDS; the forward keeps a running (max, sum) per query column -- a
// per-LANE scalar in this layout, so the online-softmax rescale is one multiply per accumulator register.
// delta_i = rowsum(dO*O) is recomputed from the staged O rows where it is needed (no scratch buffer in the ABI).
constexpr int LBLK = 128;

__device__ __forceinline__ void long_bias(float* bias, const unsigned char* kp, int k0, int L, int tid) {
    if (tid < LBLK) { const int j = k0 + tid; bias[tid] = (j >= L || (kp && kp[j])) ? -INFINITY : 0.f; }
}

__global__ __launch_bounds__(256) void attn_fwd_long_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * LBLK * 128 + LBLK * 4];
    char* Qi = smem; char* Ki = Qi + LBLK * 128; char* Vi = Ki + LBLK * 128;
    float* bias = (float*)(Vi + LBLK * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y % a.H, b = blockIdx.y / a.H, qt0 = blockIdx.x * LBLK;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const bf16_t* base = (const bf16_t*)a.qkv + (long)b * L * ld + h * DH;
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;
    stage_images<4>(smem, 0, 1, base, 0, ld, L, wave, lane, qt0);
    const int q0 = wave * 32;
    bf16x8 qf[4];
    f32x16 o[2];
    acc_zero(o[0]); acc_zero(o[1]);
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < L; k0 += LBLK) {
        __syncthreads();                                   // every wave is done with the previous K/V block
        stage_images<4>(smem, 1, 2, base + C, C, ld, L, wave, lane, k0);
        long_bias(bias, kp, k0, L, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (k0 == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = img_frag_kc(Qi, q0 + c, 2 * ks + hh);
        }
        f32x16 s[4];
        float bm = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            acc_zero(s[kb]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Ki, 32 * kb + c, 2 * ks + hh), qf[ks], s[kb], 0, 0, 0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * kb + 8 * g4 + 4 * hh);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = s[kb][4 * g4 + e] * 0.125f + bb[e];
                    s[kb][4 * g4 + e] = v;
                    bm = fmaxf(bm, v);
                }
            }
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float m_new = fmaxf(m, bm);
        const bool dead = (m_new == -INFINITY);           // nothing but padded keys so far
        const float alpha = dead ? 1.f : __expf(m - m_new);
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = dead ? 0.f : __expf(s[kb][r] - m_new);
                s[kb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        l = l * alpha + sum;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 pf = acc_frag(s[kb], j);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Vi, 32 * kb + 16 * j, 32 * db, lane), pf, o[db], 0, 0, 0);
            }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int qrow = qt0 + q0 + c;
    if (hh == 0 && qrow < L) a.lse[((long)b * a.H + h) * L + qrow] = l > 0.f ? m + logf(l) : -INFINITY;
    // O^T tiles -> the wave's own (now dead) q rows in LDS -> 16-byte row-contiguous global stores
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int row = q0 + c, chunk = 4 * db + g4;
            st4((bf16_t*)(Qi + row * 128 + ((chunk ^ img_swz(row)) << 4) + hh * 8),
                make_float4(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv, o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv));
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16_t* out = (bf16_t*)a.o + (long)b * L * C + h * DH;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = q0 + it * 8 + (lane >> 3), chunk = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(Qi + row * 128 + ((chunk ^ img_swz(row)) << 4));
        if (qt0 + row < L) *reinterpret_cast<uint4*>(out + (long)(qt0 + row) * C + chunk * 8) = v;
    }
}

// delta[row] = sum_d dO[row][d] * O[row][d] for the 128 staged rows (two threads per row)
__device__ __forceinline__ void long_delta(const char* Di, const char* Oi, float* delta_s, int tid) {
    const int row = tid >> 1, half = tid & 1;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int off = row * 128 + (((4 * half + j) ^ img_swz(row)) << 4);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(Di + off), o = *reinterpret_cast<const bf16x8*>(Oi + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)d[e] * (float)o[e];
    }
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) delta_s[row] = acc;
}

__device__ __forceinline__ void long_lse(float* lse_s, const float* lse_g, int r0, int L, int tid) {
    if (tid < LBLK) {
        float v = INFINITY;                    // rows past L and fully padded rows: p = exp(. - inf) = 0
        if (r0 + tid < L) { v = lse_g[r0 + tid]; if (v == -INFINITY) v = INFINITY; }
        lse_s[tid] = v;
    }
}

// dq: workgroup = 128 queries, loop over key blocks
__global__ __launch_bounds__(256) void attn_bwd_dq_long_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * LBLK * 128 + 3 * LBLK * 4];
    char* Qi = smem; char* Di = Qi + LBLK * 128; char* Ki = Di + LBLK * 128; char* Vi = Ki + LBLK * 128;
    char* Oi = Ki;                                         // O rows are only read for delta, before the first K block lands
    float* bias = (float*)(Vi + LBLK * 128);
    float* lse_s = bias + LBLK;
    float* delta_s = lse_s + LBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y % a.H, b = blockIdx.y / a.H, qt0 = blockIdx.x * LBLK;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const bf16_t* base = (const bf16_t*)a.qkv + (long)b * L * ld + h * DH;
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;
    stage_images<4>(smem, 0, 1, base, 0, ld, L, wave, lane, qt0);
    stage_images<4>(smem, 1, 1, (const bf16_t*)a.d_o + (long)b * L * C + h * DH, 0, C, L, wave, lane, qt0);
    stage_images<4>(smem, 2, 1, (const bf16_t*)a.o + (long)b * L * C + h * DH, 0, C, L, wave, lane, qt0);
    long_lse(lse_s, a.lse + ((long)b * a.H + h) * L, qt0, L, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    long_delta(Di, Oi, delta_s, tid);
    __syncthreads();
    const int q0 = wave * 32;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = img_frag_kc(Qi, q0 + c, 2 * ks + hh); dof[ks] = img_frag_kc(Di, q0 + c, 2 * ks + hh); }
    const float my_lse = lse_s[q0 + c], my_delta = delta_s[q0 + c];
    f32x16 dq[2];
    acc_zero(dq[0]); acc_zero(dq[1]);
    for (int k0 = 0; k0 < L; k0 += LBLK) {
        __syncthreads();
        stage_images<4>(smem, 2, 2, base + C, C, ld, L, wave, lane, k0);
        long_bias(bias, kp, k0, L, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x16 p, dp;
            acc_zero(p); acc_zero(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Ki, 32 * kb + c, 2 * ks + hh), qf[ks], p, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Vi, 32 * kb + c, 2 * ks + hh), dof[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * kb + 8 * g4 + 4 * hh);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(p[4 * g4 + e] * 0.125f + bb[e] - my_lse);
                    p[4 * g4 + e] = pv * (dp[4 * g4 + e] - my_delta);            // dS^T
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 df = acc_frag(p, j);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Ki, 32 * kb + 16 * j, 32 * db, lane), df, dq[db], 0, 0, 0);
            }
        }
    }
    bf16_t* out = (bf16_t*)a.dqkv + (long)b * L * ld + h * DH;
    tiles_to_rows(Qi, q0, c, hh, dq, 0.125f);           // the q rows are wave-private (their fragments live in registers)
    rows_to_global(Qi, q0, lane, out, ld, qt0, L);
}

// dk, dv: workgroup = 128 keys, loop over query blocks
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_long_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * LBLK * 128 + 3 * LBLK * 4];
    char* Ki = smem; char* Vi = Ki + LBLK * 128;           // only until the wave's K / V fragments are in registers
    char* Qi = smem; char* Di = Qi + LBLK * 128; char* Oi = Di + LBLK * 128;
    float* bias = (float*)(Oi + LBLK * 128);
    float* lse_s = bias + LBLK;
    float* delta_s = lse_s + LBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y % a.H, b = blockIdx.y / a.H, kt0 = blockIdx.x * LBLK;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const bf16_t* base = (const bf16_t*)a.qkv + (long)b * L * ld + h * DH;
    const bf16_t* dO = (const bf16_t*)a.d_o + (long)b * L * C + h * DH;
    const bf16_t* Og = (const bf16_t*)a.o + (long)b * L * C + h * DH;
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;
    stage_images<4>(smem, 0, 2, base + C, C, ld, L, wave, lane, kt0);
    long_bias(bias, kp, kt0, L, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int k0 = wave * 32;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf[ks] = img_frag_kc(Ki, k0 + c, 2 * ks + hh); vf[ks] = img_frag_kc(Vi, k0 + c, 2 * ks + hh); }
    const float my_bias = bias[k0 + c];
    f32x16 dk[2], dv[2];
    acc_zero(dk[0]); acc_zero(dk[1]); acc_zero(dv[0]); acc_zero(dv[1]);
    for (int qb0 = 0; qb0 < L; qb0 += LBLK) {
        __syncthreads();
        stage_images<4>(smem, 0, 1, base, 0, ld, L, wave, lane, qb0);
        stage_images<4>(smem, 1, 1, dO, 0, C, L, wave, lane, qb0);
        stage_images<4>(smem, 2, 1, Og, 0, C, L, wave, lane, qb0);
        long_lse(lse_s, a.lse + ((long)b * a.H + h) * L, qb0, L, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        long_delta(Di, Oi, delta_s, tid);
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            f32x16 p, dp;
            acc_zero(p); acc_zero(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Qi, 32 * qb + c, 2 * ks + hh), kf[ks], p, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Di, 32 * qb + c, 2 * ks + hh), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 lv = *reinterpret_cast<const float4*>(lse_s + 32 * qb + 8 * g4 + 4 * hh);
                const float4 dl = *reinterpret_cast<const float4*>(delta_s + 32 * qb + 8 * g4 + 4 * hh);
                const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(p[4 * g4 + e] * 0.125f + my_bias - ll[e]);
                    p[4 * g4 + e] = pv;
                    dp[4 * g4 + e] = pv * (dp[4 * g4 + e] - dd[e]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 pf = acc_frag(p, j), df = acc_frag(dp, j);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Di, 32 * qb + 16 * j, 32 * db, lane), pf, dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Qi, 32 * qb + 16 * j, 32 * db, lane), df, dk[db], 0, 0, 0);
                }
            }
        }
    }
    bf16_t* out = (bf16_t*)a.dqkv + (long)b * L * ld + h * DH;
    __syncthreads();
    tiles_to_rows(Qi, k0, c, hh, dk, 0.125f);
    tiles_to_rows(Di, k0, c, hh, dv, 1.0f);
    rows_to_global(Qi, k0, lane, out + C, ld, kt0, L);
    rows_to_global(Di, k0, lane, out + 2 * C, ld, kt0, L);
}

// ======================================================================================================
// Mid-length bf16 path (128 < L <= 288: the len=256 configuration, L = 256 video / 272 joint).  The short kernels' scheme with the
// whole head resident: one workgroup per (video, head), ceil(L/32) waves (up to nine), q / k / v (/ dO) images staged ONCE
// (110 / 147 KiB of LDS) -- the streamed kernels below re-staged K and V per 128-query block (and padded L = 272 to 3 x 3
// blocks of 128: twice the work).  Forward: online softmax over chunks of three key blocks (the running max / sum is a per-lane
// scalar in the transposed-score layout); backward: delta = rowsum(dO * O) from the O rows in global memory, then the short
// kernel's two phases, each streaming the other axis one 32-row block at a time (no score panel in registers).
template <int NKB>
__global__ __launch_bounds__(64 * NKB) void attn_fwd_mid_kernel(AttnArgs a) {
    constexpr int LP = 32 * NKB, CH = 3;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* Qi = smem; char* Ki = Qi + LP * 128; char* Vi = Ki + LP * 128;
    float* bias = (float*)(Vi + LP * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const bf16_t* base = (const bf16_t*)a.qkv + (long)b * L * ld + h * DH;
    stage_images<NKB>(smem, 0, 3, base, C, ld, L, wave, lane);
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;
    for (int j = tid; j < LP; j += 64 * NKB) bias[j] = (j >= L || (kp && kp[j])) ? -INFINITY : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int q0 = wave * 32;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = img_frag_kc(Qi, q0 + c, 2 * ks + hh);
    f32x16 o[2];
    acc_zero(o[0]); acc_zero(o[1]);
    float m = -INFINITY, l = 0.f;
#pragma unroll
    for (int kc = 0; kc < NKB; kc += CH) {
        f32x16 s[CH];
        float bm = -INFINITY;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (kc + i >= NKB) continue;
            const int kb = kc + i;
            acc_zero(s[i]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Ki, 32 * kb + c, 2 * ks + hh), qf[ks], s[i], 0, 0, 0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * kb + 8 * g4 + 4 * hh);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = s[i][4 * g4 + e] * 0.125f + bb[e];
                    s[i][4 * g4 + e] = v;
                    bm = fmaxf(bm, v);
                }
            }
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float m_new = fmaxf(m, bm);
        const bool dead = (m_new == -INFINITY);           // nothing but padded keys so far
        const float alpha = dead ? 1.f : __expf(m - m_new);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (kc + i >= NKB) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = dead ? 0.f : __expf(s[i][r] - m_new);
                s[i][r] = e;
                sum += e;
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        l = l * alpha + sum;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (kc + i >= NKB) continue;
            const int kb = kc + i;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 pf = acc_frag(s[i], j);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Vi, 32 * kb + 16 * j, 32 * db, lane), pf, o[db], 0, 0, 0);
            }
        }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (hh == 0 && q0 + c < L) a.lse[((long)b * a.H + h) * L + q0 + c] = l > 0.f ? m + logf(l) : -INFINITY;
    tiles_to_rows(Qi, q0, c, hh, o, inv);               // the wave's own (now dead) q rows
    rows_to_global(Qi, q0, lane, (bf16_t*)a.o + (long)b * L * C + h * DH, C, 0, L);
}

#ifndef TAN_MID_UNROLL
#define TAN_MID_UNROLL 1
#endif
template <int NKB>
__global__ __launch_bounds__(64 * NKB) void attn_bwd_mid_kernel(AttnArgs a) {
    constexpr int LP = 32 * NKB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* Qi = smem; char* Ki = Qi + LP * 128; char* Vi = Ki + LP * 128; char* Di = Vi + LP * 128;
    float* bias = (float*)(Di + LP * 128);
    float* lse_s = bias + LP;
    float* delta_s = lse_s + LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int C = a.H * DH, L = a.L;
    const long ld = 3L * C;
    const bf16_t* base = (const bf16_t*)a.qkv + (long)b * L * ld + h * DH;
    const bf16_t* dOg = (const bf16_t*)a.d_o + (long)b * L * C + h * DH;
    const bf16_t* Og = (const bf16_t*)a.o + (long)b * L * C + h * DH;
    stage_images<NKB>(smem, 0, 3, base, C, ld, L, wave, lane);
    stage_images<NKB>(smem, 3, 1, dOg, 0, C, L, wave, lane);
    const unsigned char* kp = a.keypad ? a.keypad + (long)b * L : nullptr;
    for (int j = tid; j < LP; j += 64 * NKB) {
        bias[j] = (j >= L || (kp && kp[j])) ? -INFINITY : 0.f;
        float lv = INFINITY;                     // rows past L and fully padded rows: p = exp(. - inf) = 0
        if (j < L) { lv = a.lse[((long)b * a.H + h) * L + j]; if (lv == -INFINITY) lv = INFINITY; }
        lse_s[j] = lv;
    }
    {   // delta[row] = sum_d dO[row][d] * O[row][d] straight from global memory (two lanes per row; rows past L: 0)
        const int row = tid >> 1, half = tid & 1;
        float acc = 0.f;
        if (row < L) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 d = *reinterpret_cast<const bf16x8*>(dOg + (long)row * C + (4 * half + j) * 8);
                const bf16x8 o = *reinterpret_cast<const bf16x8*>(Og + (long)row * C + (4 * half + j) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += (float)d[e] * (float)o[e];
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        if (half == 0 && row < LP) delta_s[row] = acc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16_t* out = (bf16_t*)a.dqkv + (long)b * L * ld + h * DH;

    f32x16 dq[2];
    {   // ---- phase A: this wave's 32 queries against every key block: dq
        const int q0 = wave * 32;
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = img_frag_kc(Qi, q0 + c, 2 * ks + hh); dof[ks] = img_frag_kc(Di, q0 + c, 2 * ks + hh); }
        const float my_lse = lse_s[q0 + c], my_delta = delta_s[q0 + c];
        acc_zero(dq[0]); acc_zero(dq[1]);
#pragma unroll TAN_MID_UNROLL
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 p, dp;
            acc_zero(p); acc_zero(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Ki, 32 * kb + c, 2 * ks + hh), qf[ks], p, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Vi, 32 * kb + c, 2 * ks + hh), dof[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * kb + 8 * g4 + 4 * hh);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(p[4 * g4 + e] * 0.125f + bb[e] - my_lse);
                    p[4 * g4 + e] = pv * (dp[4 * g4 + e] - my_delta);            // dS^T
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 df = acc_frag(p, j);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Ki, 32 * kb + 16 * j, 32 * db, lane), df, dq[db], 0, 0, 0);
            }
        }
    }
    {   // ---- phase B: this wave's 32 keys against every query block: dk, dv (reads only the images: no barrier since phase A)
        const int k0 = wave * 32;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = img_frag_kc(Ki, k0 + c, 2 * ks + hh); vf[ks] = img_frag_kc(Vi, k0 + c, 2 * ks + hh); }
        const float my_bias = bias[k0 + c];
        f32x16 dk[2], dv[2];
        acc_zero(dk[0]); acc_zero(dk[1]); acc_zero(dv[0]); acc_zero(dv[1]);
#pragma unroll TAN_MID_UNROLL
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 p, dp;
            acc_zero(p); acc_zero(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Qi, 32 * qb + c, 2 * ks + hh), kf[ks], p, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Di, 32 * qb + c, 2 * ks + hh), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 lv = *reinterpret_cast<const float4*>(lse_s + 32 * qb + 8 * g4 + 4 * hh);
                const float4 dl = *reinterpret_cast<const float4*>(delta_s + 32 * qb + 8 * g4 + 4 * hh);
                const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(p[4 * g4 + e] * 0.125f + my_bias - ll[e]);
                    p[4 * g4 + e] = pv;
                    dp[4 * g4 + e] = pv * (dp[4 * g4 + e] - dd[e]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 pf = acc_frag(p, j), df = acc_frag(dp, j);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Di, 32 * qb + 16 * j, 32 * db, lane), pf, dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Qi, 32 * qb + 16 * j, 32 * db, lane), df, dk[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();            // every wave is done with the images: each parks its three result tiles in its own rows
        tiles_to_rows(Qi, k0, c, hh, dq, 0.125f);
        tiles_to_rows(Ki, k0, c, hh, dk, 0.125f);
        tiles_to_rows(Vi, k0, c, hh, dv, 1.0f);
        rows_to_global(Qi, k0, lane, out, ld, 0, L);
        rows_to_global(Ki, k0, lane, out + C, ld, 0, L);
        rows_to_global(Vi, k0, lane, out + 2 * C, ld, 0, L);
        if (a.gbias) {               // in_proj bias gradient: column sums of the 32 rows this wave parked (rows past L hold zeros)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const int row = k0 + r, off = row * 128 + (((lane >> 3) ^ img_swz(row)) << 4) + (lane & 7) * 2;
                s0 += bf2f(*reinterpret_cast<const bf16_t*>(Qi + off));
                s1 += bf2f(*reinterpret_cast<const bf16_t*>(Ki + off));
                s2 += bf2f(*reinterpret_cast<const bf16_t*>(Vi + off));
            }
            float* g = a.gbias + h * DH + lane;
            unsafeAtomicAdd(g, s0); unsafeAtomicAdd(g + C, s1); unsafeAtomicAdd(g + 2 * C, s2);
        }
    }
}

template <int NKB> static int launch_fwd_mid(const AttnArgs& a, hipStream_t st) {
    const size_t sm = 3 * NKB * 32 * 128 + NKB * 32 * 4;
    int rc = set_smem(attn_fwd_mid_kernel<NKB>, sm);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_fwd_mid_kernel<NKB>), dim3(a.B * a.H), dim3(64 * NKB), sm, st, a);
    return 0;
}
template <int NKB> static int launch_bwd_mid(const AttnArgs& a, hipStream_t st) {
    const size_t sm = 4 * NKB * 32 * 128 + 3 * NKB * 32 * 4;
    int rc = set_smem(attn_bwd_mid_kernel<NKB>, sm);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_mid_kernel<NKB>), dim3(a.B * a.H), dim3(64 * NKB), sm, st, a);
    return 0;
}

template <int NKB> static int launch_fwd_short(const AttnArgs& a, hipStream_t st) {
    const size_t sm = 3 * NKB * 32 * 128 + NKB * 32 * 4;
    int rc = set_smem(attn_fwd_short_kernel<NKB>, sm);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_fwd_short_kernel<NKB>), dim3(a.B * a.H), dim3(64 * NKB), sm, st, a);
    return 0;
}
template <int NKB> static int launch_bwd_short(const AttnArgs& a, hipStream_t st) {
    const size_t sm = 4 * NKB * 32 * 128 + 3 * NKB * 32 * 4;
    int rc = set_smem(attn_bwd_short_kernel<NKB>, sm);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_short_kernel<NKB>), dim3(a.B * a.H), dim3(64 * NKB), sm, st, a);
    return 0;
}

}  // namespace tal

using namespace tal;

// bf16 dispatch by length: L <= 128 the whole head in registers / LDS ("short"), 128 < L <= 288 the whole head resident in LDS
// ("mid"), longer: streamed ("long"); f32 runs the tiled kernels
static bool long_path() { return true; }
static bool mid_path(int dtype, int L) { return dtype == TAN_BF16 && L > 128 && L <= 288; }
static bool short_path(int dtype, int L) { return dtype == TAN_BF16 && L <= 128; }

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
    } else if (short_path(dtype, L)) {
        const int nkb = (L + 31) / 32;
        rc = nkb == 1 ? launch_fwd_short<1>(a, st) : nkb == 2 ? launch_fwd_short<2>(a, st)
           : nkb == 3 ? launch_fwd_short<3>(a, st) : launch_fwd_short<4>(a, st);
        if (rc) return rc;
    } else if (mid_path(dtype, L)) {
        const int nkb = (L + 31) / 32;
        rc = nkb == 5 ? launch_fwd_mid<5>(a, st) : nkb == 6 ? launch_fwd_mid<6>(a, st) : nkb == 7 ? launch_fwd_mid<7>(a, st)
           : nkb == 8 ? launch_fwd_mid<8>(a, st) : launch_fwd_mid<9>(a, st);
        if (rc) return rc;
    } else if (dtype == TAN_BF16 && long_path()) {
        hipLaunchKernelGGL(attn_fwd_long_kernel, dim3(cdiv(L, LBLK), B * H), dim3(256), 0, st, a);
    } else if (dtype == TAN_BF16) {
        size_t sm = fwd_smem<bf16_t>(a.Lpad);
        if ((rc = set_smem(attn_fwd_kernel<bf16_t>, sm))) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, dim3(256), sm, st, a);
    } else return TAN_ERR_BAD_ARG;
    prof_end(st, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}

static int attn_bwd_impl(const void* qkv, const unsigned char* key_padding_mask, const void* o, const float* lse,
                         const void* d_o, void* dqkv, float* g_b_qkv, int B, int L, int H, int dtype, void* stream) {
    TAN_REQUIRE(qkv && o && lse && d_o && dqkv && B > 0 && L > 0 && H > 0);
    AttnArgs a{};
    const bool fuse = g_b_qkv && (short_path(dtype, L) || mid_path(dtype, L));
    a.gbias = fuse ? g_b_qkv : nullptr;
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
    } else if (short_path(dtype, L)) {
        const int nkb = (L + 31) / 32;
        rc = nkb == 1 ? launch_bwd_short<1>(a, st) : nkb == 2 ? launch_bwd_short<2>(a, st)
           : nkb == 3 ? launch_bwd_short<3>(a, st) : launch_bwd_short<4>(a, st);
        if (rc) return rc;
    } else if (mid_path(dtype, L)) {
        const int nkb = (L + 31) / 32;
        rc = nkb == 5 ? launch_bwd_mid<5>(a, st) : nkb == 6 ? launch_bwd_mid<6>(a, st) : nkb == 7 ? launch_bwd_mid<7>(a, st)
           : nkb == 8 ? launch_bwd_mid<8>(a, st) : launch_bwd_mid<9>(a, st);
        if (rc) return rc;
    } else if (dtype == TAN_BF16 && long_path()) {
        hipLaunchKernelGGL(attn_bwd_dq_long_kernel, dim3(cdiv(L, LBLK), B * H), dim3(256), 0, st, a);
        hipLaunchKernelGGL(attn_bwd_dkv_long_kernel, dim3(cdiv(L, LBLK), B * H), dim3(256), 0, st, a);
    } else if (dtype == TAN_BF16) {
        if ((rc = set_smem(attn_bwd_dq_kernel<bf16_t>, bwd_smem<bf16_t>(5)))) return rc;
        if ((rc = set_smem(attn_bwd_dkv_kernel<bf16_t>, bwd_smem<bf16_t>(6)))) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t>), grid, dim3(256), bwd_smem<bf16_t>(5), st, a);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<bf16_t>), grid, dim3(256), bwd_smem<bf16_t>(6), st, a);
    } else return TAN_ERR_BAD_ARG;
    prof_end(st, rec);
    TAN_LAUNCH_CHECK();
    if (g_b_qkv && !fuse) return tan_colsum_acc(dqkv, g_b_qkv, (long)B * L, 3 * H * DH, dtype, stream);
    return 0;
}

extern "C" int tan_attn_bwd(const void* qkv, const unsigned char* key_padding_mask, const void* o, const float* lse,
                            const void* d_o, void* dqkv, int B, int L, int H, int dtype, void* stream) {
    return attn_bwd_impl(qkv, key_padding_mask, o, lse, d_o, dqkv, nullptr, B, L, H, dtype, stream);
}

extern "C" int tan_attn_bwd_bias(const void* qkv, const unsigned char* key_padding_mask, const void* o, const float* lse,
                                 const void* d_o, void* dqkv, float* g_b_qkv, int B, int L, int H, int dtype, void* stream) {
    TAN_REQUIRE(g_b_qkv);
    return attn_bwd_impl(qkv, key_padding_mask, o, lse, d_o, dqkv, g_b_qkv, B, L, H, dtype, stream);
}
