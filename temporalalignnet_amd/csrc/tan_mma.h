// 32x32 MFMA tile primitives shared by the GEMM, attention and similarity kernels.
//
// One wave (64 lanes) owns a 32x32 f32 accumulator tile (16 VGPR/AGPR per lane):
//   C/D layout (dtype independent on gfx950): col = lane & 31,
//                                              row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
// Operand fragments for D[i][j] += sum_k A[i][k] * B[k][j]:
//   bf16  v_mfma_f32_32x32x16_bf16 : lane holds 8 consecutive k (k = 8*(lane>>5) .. +7) of row/col (lane&31)
//   f32   v_mfma_f32_32x32x2_f32   : lane holds the single k = lane>>5 of row/col (lane&31)   (exact f32 fma chain)
// Both A and B fragments are indexed [outer = lane&31][k]; an LDS tile is either "K-contiguous"
// (element (outer,k) at base[outer*ld + k]) or "K-strided" (at base[k*ld + outer]).
#pragma once
#include "tan_common.h"

namespace tal {

template <typename T> struct Mma;

template <> struct Mma<float> {
    static constexpr int KS = 2;  // k per instruction
    typedef float frag_t;
    template <bool KC>
    static __device__ __forceinline__ frag_t load(const float* base, int ld, int o0, int k0, int lane) {
        const int o = o0 + (lane & 31), k = k0 + (lane >> 5);
        return KC ? base[o * ld + k] : base[k * ld + o];
    }
    static __device__ __forceinline__ void mma(f32x16& acc, frag_t a, frag_t b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
};

template <> struct Mma<bf16_t> {
    static constexpr int KS = 16;
    typedef bf16x8 frag_t;
    template <bool KC>
    static __device__ __forceinline__ frag_t load(const bf16_t* base, int ld, int o0, int k0, int lane) {
        const int o = o0 + (lane & 31), k = k0 + 8 * (lane >> 5);
        if (KC) {
            // 16-byte aligned: ld % 8 == 0 and k0 % 8 == 0 are guaranteed by the callers
            return *reinterpret_cast<const bf16x8*>(base + o * ld + k);
        } else {
            union { bf16x8 v; bf16_t s[8]; } u;
#pragma unroll
            for (int e = 0; e < 8; ++e) u.s[e] = base[(k + e) * ld + o];
            return u.v;
        }
    }
    static __device__ __forceinline__ void mma(f32x16& acc, frag_t a, frag_t b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

__device__ __forceinline__ void acc_zero(f32x16& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.0f;
}

}  // namespace tal
